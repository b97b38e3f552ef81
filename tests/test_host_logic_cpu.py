"""Host-side logic that needs no GPU: Update/Combine bookkeeping of the aggregate mirror, the record ->
struct conversion, the synthetic generator and the shard partitioning."""
import numpy as np
import pytest
import torch

from conftest import import_pkg


def test_update_skips_null_rows_and_groups_columns():
    pkg = import_pkg()
    agg = pkg.WlsFitAgg()
    agg.update(["b", "a", "b", "a", "c"], [1.0, 2.0, None, 4.0, 5.0],
               [[1.0, 10.0], [2.0, 20.0], [3.0, 30.0], None, [5.0, 50.0]], [1.0, 1.0, 1.0, 1.0, None])
    agg.update(["a"], [6.0], [[6.0, 60.0]], [2.0])
    keys, offs, y, x_cols, w = agg.grouped_columns()
    assert keys.tolist() == ["a", "b", "c"]
    assert offs.tolist() == [0, 2, 3, 3]                  # c exists but accumulated nothing
    assert y.tolist() == [2.0, 6.0, 1.0]
    assert x_cols[0].tolist() == [2.0, 6.0, 1.0] and x_cols[1].tolist() == [20.0, 60.0, 10.0]
    assert w.tolist() == [1.0, 2.0, 1.0]
    assert agg.n_features == 2


def test_update_accepts_arrays_and_masked_arrays():
    pkg = import_pkg()
    agg = pkg.OlsFitAgg()
    y = np.ma.array([1.0, 2.0, 3.0], mask=[False, True, False])
    agg.update(np.array([0, 0, 1]), y, np.array([[1.0], [2.0], [3.0]]))
    _, offs, yy, xc, w = agg.grouped_columns()
    assert offs.tolist() == [0, 1, 2] and yy.tolist() == [1.0, 3.0] and w is None
    with pytest.raises(pkg.InvalidInputException, match="Inconsistent feature count: expected 1, got 3"):
        agg.update([0], [1.0], np.ones((1, 3)))


def test_result_from_records_null_and_inference():
    pkg = import_pkg()
    p = 2
    core = np.array([[1.0, 2.0, 0.5, 0.9, 0.89, 0.1, 10.0, 0.0],
                     [np.nan] * 7 + [6.0],
                     [np.nan] * 7 + [100.0]])
    inf = np.arange(3 * 12, dtype=np.float64).reshape(3, 12)
    res = pkg.result_from_records(["a", "b", "c"], core, inf, p)
    assert res.is_null.tolist() == [False, True, True]
    r = res.row(0)
    assert r["coefficients"] == [1.0, 2.0] and r["intercept"] == 0.5 and r["n_observations"] == 10
    assert r["std_errors"] == [0.0, 1.0] and r["ci_upper"] == [8.0, 9.0] and r["f_pvalue"] == 11.0
    assert res.row(1) is None and res.as_dict()["c"] is None


def test_synthetic_generator_is_counter_based_and_sharded():
    synth = import_pkg("synth")
    offs, y, xc, w = synth.make_grouped(12, 7, 3, weights=True, chunk_groups=5)
    assert offs.tolist() == [7 * i for i in range(13)]
    o2, y2, xc2, w2 = synth.make_grouped(4, 7, 3, group_start=8, weights=True)
    assert torch.equal(y[8 * 7:], y2) and torch.equal(xc[2][8 * 7:], xc2[2]) and torch.equal(w[8 * 7:], w2)
    assert float(xc[0].min()) >= -10 and float(xc[0].max()) <= 10
    assert float(w.min()) >= 0.5 and float(w.max()) <= 1.5
    # a different seed gives different data
    _, y3, _, _ = synth.make_grouped(4, 7, 3, seed=7)
    assert not torch.equal(y[:28], y3)


def test_synthetic_distribution_moments():
    synth = import_pkg("synth")
    _, y, xc, _ = synth.make_grouped(64, 1000, 8)
    x = torch.stack(xc)
    assert abs(float(x.mean())) < 0.05 and abs(float(x.var()) - 400.0 / 12.0) < 0.5
    import oracle
    core, _ = oracle.fit_groups(y.numpy(), [c.numpy() for c in xc], np.arange(65) * 1000, n_threads=4)
    assert abs(float(np.mean(core[:, 8 + 3])) - 2.0) < 0.05            # noise sd = 2.0


def test_shard_ranges_cover_all_groups():
    d = import_pkg("distributed")
    for G in (0, 1, 7, 8, 1000, 1_000_000):
        for W in (1, 2, 3, 4, 8):
            got = []
            for r in range(W):
                lo, hi = d.shard_range(G, r, W)
                assert 0 <= lo <= hi <= G and hi - lo <= d.padded_shard_len(G, W)
                got.extend(range(lo, hi)) if G <= 1000 else None
            if G <= 1000:
                assert got == list(range(G))


def test_window_frame_parsing():
    agg = import_pkg("aggregate")
    assert agg._parse_frame(None, "current row") == (None, 0)
    assert agg._parse_frame(None, "1 preceding") == (None, 1)
    assert agg._parse_frame(("9 preceding", "current row"), "current row") == (9, 0)
    assert agg._parse_frame((7, 3), "current row") == (7, 3)
    assert agg._parse_frame(("unbounded", "2 preceding"), "current row") == (None, 2)
    assert agg._parse_frame(("3 preceding", "2 following"), "current row") == (3, -2)
    assert agg._parse_frame(("1 following", "unbounded following"), "current row") == (-1, None)
    assert agg._parse_frame(("unbounded preceding", "unbounded following"), "current row") == (None, None)
    assert agg._parse_frame(("current row", "current row"), "current row") == (0, 0)
    pkg = import_pkg()
    for bad in (("current row", "3 preceding"), (2, 5), ("2 following", "1 following"), ("x", 0),
                ("unbounded following", 0), (0, "unbounded preceding")):
        with pytest.raises(pkg.InvalidInputException):
            agg._parse_frame(bad, "current row")


def test_vif_agg_null_rules_need_no_gpu():
    """vif_aggregate.cpp:154: NULL unless >= 2 features and >= 3 buffered values; a NaN shortens only its own column
    (:88-93), unequal columns make compute_vif fail -> NULL.  None of these groups reaches the GPU."""
    pkg = import_pkg()
    keys, res = pkg.vif_agg(["a"] * 4 + ["b"] * 2 + ["c"] * 4 + ["d"] * 3,
                            [[1.0], [2.0], [3.0], [4.0],                      # a: one feature
                             [1.0, 2.0], [2.0, 1.0],                          # b: two rows
                             [1.0, 2.0], [2.0, float("nan")], [3.0, 1.0], [4.0, 5.0],   # c: NaN shortens column 2
                             None, None, None])                               # d: only NULL lists
    assert keys.tolist() == ["a", "b", "c", "d"] and res == [None, None, None, None]
    with pytest.raises(pkg.InvalidInputException, match="Inconsistent feature count"):
        pkg.vif_agg([0, 0, 0], [[1.0, 2.0], [1.0], [2.0, 3.0]])
