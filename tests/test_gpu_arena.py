"""GPU test of the DuckDB shim's arena (anofox-statistics_amd/duckdb_shim/agg_arena.hpp through arena_capi.cpp): the
calls the glue in fit_agg_hip.cpp makes — Update vectors of <= 2048 rows from several threads with lazily initialised
states, Combine of the threads' states, Finalize vector by vector — against the oracle's fit of the same rows."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle
from conftest import ROOT, assert_records_match, import_pkg

pytestmark = pytest.mark.gpu

_DP = C.POINTER(C.c_double)


def _lib():
    pkg = import_pkg()           # loads libanofox_stats_hip.so first (RTLD_GLOBAL)
    path = os.path.join(ROOT, "anofox-statistics_amd", "duckdb_shim", "libanofox_arena_capi.so")
    assert os.path.exists(path), "build it with make -C anofox-statistics_amd/duckdb_shim (or __graft_entry__.build())"
    lib = C.CDLL(path)
    abi = import_pkg("_abi")
    lib.arena_create.restype = C.c_void_p
    lib.arena_create.argtypes = [abi.AnofoxHipBatchOptions, C.c_size_t]
    lib.arena_destroy.argtypes = [C.c_void_p]
    lib.arena_update.restype = C.c_int
    lib.arena_update.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_void_p, C.c_char_p]
    lib.arena_combine.restype = C.c_int
    lib.arena_combine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]
    lib.arena_finalize.restype = C.c_int
    lib.arena_finalize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.arena_feature_count.restype = C.c_size_t
    lib.arena_feature_count.argtypes = [C.c_void_p]
    lib.arena_rows.restype = C.c_uint64
    lib.arena_rows.argtypes = [C.c_void_p]
    lib.arena_unrefined.restype = C.c_int64
    lib.arena_unrefined.argtypes = [C.c_void_p]
    lib.arena_retaining.restype = C.c_int
    lib.arena_retaining.argtypes = [C.c_void_p]
    return pkg, lib


@pytest.mark.parametrize("model,flush_rows,p,hc", [("ols", 5000, 6, None), ("wls", 1 << 20, 6, None), ("ridge", 777, 6, None),
                                                   ("ols", 5000, 12, None), ("wls", 5000, 20, None), ("ols", 5000, 4, "hc1")])
def test_threads_update_combine_finalize(model, flush_rows, p, hc):
    """p <= 8 without HC errors: rows stream into O(p^2) moments on the GPU; wider designs and HC errors: the library keeps
    the rows themselves in HBM (log-only state) and fits them at Finalize — same calls from the arena, same results."""
    pkg, lib = _lib()
    rng = np.random.default_rng(len(model) + p)
    G, T = 400, 3
    kw = dict(compute_inference=True)
    if hc:
        kw["hc_type"] = hc
    if model == "ridge":
        kw["alpha"] = 0.5
    arena = lib.arena_create(pkg.RegressionOptions(**kw).batch_options(model), flush_rows)
    assert arena
    # thread t owns a hash table: state k of thread t = key k
    data = []
    for t in range(T):
        n = 30_000
        keys = rng.integers(0, G, n).astype(np.uint32)
        X = rng.uniform(-5, 5, (n, p)) + 3.0 * t
        y = X @ rng.uniform(-2, 2, p) + 0.01 * keys + rng.standard_normal(n)
        w = rng.uniform(0.5, 2.0, n)
        accept = (rng.uniform(size=n) > 0.1).astype(np.uint8)          # NULL y / x / w rows
        X[rng.uniform(size=n) < 0.02, 2] = np.nan                       # NULL list elements
        data.append((keys, y, X, w, accept, np.full(G, -1, dtype=np.int64)))
    errs = []

    def run(t):
        keys, y, X, w, accept, slots = data[t]
        msg = C.create_string_buffer(256)
        for c0 in range(0, len(keys), 2048):                            # STANDARD_VECTOR_SIZE
            sl = slice(c0, c0 + 2048)
            k, yy, xx, ww, aa = (np.ascontiguousarray(v[sl]) for v in (keys, y, X, w, accept))
            rc = lib.arena_update(arena, len(k), k.ctypes.data, slots.ctypes.data, yy.ctypes.data, xx.ctypes.data, p,
                                  ww.ctypes.data if model == "wls" else None, aa.ctypes.data, msg)
            if rc != 0:
                errs.append(msg.value.decode())
                return

    threads = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    assert lib.arena_feature_count(arena) == p
    assert lib.arena_rows(arena) == sum(int(d[4].sum()) for d in data)
    # Combine thread 1 and 2 into thread 0, as the glue does: adopt where the target has no slot, merge otherwise
    tgt = data[0][5]
    msg = C.create_string_buffer(256)
    for t in (1, 2):
        src_slots = data[t][5]
        adopt = (tgt < 0) & (src_slots >= 0)
        tgt[adopt] = src_slots[adopt]
        both = (src_slots >= 0) & ~adopt
        s = np.ascontiguousarray(src_slots[both], dtype=np.uint32)
        d = np.ascontiguousarray(tgt[both], dtype=np.uint32)
        assert lib.arena_combine(arena, s.ctypes.data, d.ctypes.data, len(s), msg) == 0, msg.value
    # Finalize, 2048 states per call
    have = np.nonzero(tgt >= 0)[0]
    core = np.full((G, p + 6), np.nan)
    inf = np.full((G, 5 * p + 2), np.nan)
    isnull = np.ones(G, dtype=np.uint8)
    for c0 in range(0, len(have), 2048):
        idx = have[c0:c0 + 2048]
        sl = np.ascontiguousarray(tgt[idx], dtype=np.uint32)
        oc = np.empty((len(idx), p + 6))
        oi = np.empty((len(idx), 5 * p + 2))
        nn = np.empty(len(idx), dtype=np.uint8)
        assert lib.arena_finalize(arena, len(idx), sl.ctypes.data, oc.ctypes.data, oi.ctypes.data, nn.ctypes.data, msg) == 0, msg.value
        core[idx], inf[idx], isnull[idx] = oc, oi, nn
    # the reference's buffers: thread 0's accepted rows of a key, then thread 1's, then thread 2's
    keys = np.concatenate([d[0][d[4] != 0] for d in data])
    y = np.concatenate([d[1][d[4] != 0] for d in data])
    X = np.concatenate([d[2][d[4] != 0] for d in data])
    w = np.concatenate([d[3][d[4] != 0] for d in data])
    order = np.argsort(keys, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(keys, minlength=G))]).astype(np.int64)
    rcore, rinf = oracle.fit_groups(y[order], [np.ascontiguousarray(X[order, j]) for j in range(p)], offs,
                                    w=(w[order] if model == "wls" else None), model=model, **kw)
    ok = rcore[:, p + 5] == 0
    assert np.array_equal(isnull == 0, ok)
    assert_records_match(core[ok], rcore[ok], p, inf[ok], rinf[ok], what=f"arena {model}")
    lib.arena_destroy(arena)


def test_inconsistent_feature_count_and_empty_query():
    pkg, lib = _lib()
    arena = lib.arena_create(pkg.RegressionOptions().batch_options("ols"), 100)
    slots = np.full(2, -1, dtype=np.int64)
    msg = C.create_string_buffer(256)
    k = np.zeros(3, dtype=np.uint32)
    y = np.arange(3.0)
    assert lib.arena_update(arena, 3, k.ctypes.data, slots.ctypes.data, y.ctypes.data, np.ones((3, 2)).ctypes.data, 2, None, None, msg) == 0
    assert lib.arena_update(arena, 3, k.ctypes.data, slots.ctypes.data, y.ctypes.data, np.ones((3, 3)).ctypes.data, 3, None, None, msg) == -1
    assert msg.value.decode() == "Inconsistent feature count: expected 2, got 3"      # ols_aggregate.cpp:172-175
    lib.arena_destroy(arena)
    # a query whose every row is skipped: states exist, nothing reaches the GPU, every group is NULL
    arena = lib.arena_create(pkg.RegressionOptions().batch_options("ols"), 100)
    slots = np.full(1, -1, dtype=np.int64)
    acc = np.zeros(3, dtype=np.uint8)
    assert lib.arena_update(arena, 3, k.ctypes.data, slots.ctypes.data, y.ctypes.data, np.ones((3, 2)).ctypes.data, 2, None, acc.ctypes.data, msg) == 0
    assert slots[0] == 0
    nn = np.zeros(1, dtype=np.uint8)
    sl = np.zeros(1, dtype=np.uint32)
    oc = np.empty((1, 8))
    assert lib.arena_finalize(arena, 1, sl.ctypes.data, oc.ctypes.data, None, nn.ctypes.data, msg) == 0
    assert nn[0] == 1
    lib.arena_destroy(arena)


@pytest.mark.parametrize("retain", [True, False])
def test_arena_keeps_rows_for_the_groups_moments_cannot_resolve(retain, monkeypatch):
    """Keys whose fit is nearly exact (noise 1e-6 on a signal of order 10): from the moments alone sigma is lost
    (rss = tss - |z|^2 cancels); the arena lets the device state keep the rows (64 GiB by default) and Finalize refits
    those keys through the batch path.  ANOFOX_HIP_RETAIN_BYTES=0 switches that off — and shows what it buys."""
    if not retain:
        monkeypatch.setenv("ANOFOX_HIP_RETAIN_BYTES", "0")
    pkg, lib = _lib()
    rng = np.random.default_rng(77)
    G, p, n = 300, 5, 60_000
    kw = dict(compute_inference=True)
    arena = lib.arena_create(pkg.RegressionOptions(**kw).batch_options("ols"), 5000)
    keys = rng.integers(0, G, n).astype(np.uint32)
    X = rng.uniform(-5, 5, (n, p)) + 2.0
    beta = rng.uniform(-3, 3, (G, p))
    exact = np.arange(G) % 3 == 0
    y = np.einsum("ij,ij->i", beta[keys], X) + 7.0 + np.where(exact[keys], 1e-6, 1.0) * rng.standard_normal(n)
    slots = np.full(G, -1, dtype=np.int64)
    msg = C.create_string_buffer(256)
    for c0 in range(0, n, 2048):
        sl = slice(c0, c0 + 2048)
        k, yy, xx = (np.ascontiguousarray(v[sl]) for v in (keys, y, X))
        assert lib.arena_update(arena, len(k), k.ctypes.data, slots.ctypes.data, yy.ctypes.data, xx.ctypes.data, p, None, None, msg) == 0, msg.value
    assert bool(lib.arena_retaining(arena)) == retain
    sl = np.ascontiguousarray(slots, dtype=np.uint32)
    core = np.empty((G, p + 6))
    inf = np.empty((G, 5 * p + 2))
    nn = np.empty(G, dtype=np.uint8)
    assert lib.arena_finalize(arena, G, sl.ctypes.data, core.ctypes.data, inf.ctypes.data, nn.ctypes.data, msg) == 0, msg.value
    order = np.argsort(keys, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(keys, minlength=G))]).astype(np.int64)
    rcore, rinf = oracle.fit_groups(y[order], [np.ascontiguousarray(X[order, j]) for j in range(p)], offs, model="ols", **kw)
    if retain:
        assert lib.arena_unrefined(arena) == 0
        assert_records_match(core, rcore, p, inf, rinf, what="arena, rows kept")
    else:
        assert lib.arena_unrefined(arena) >= int(exact.sum())
        easy = ~exact
        assert_records_match(core[easy], rcore[easy], p, inf[easy], rinf[easy], what="arena, moments only, easy keys")
        # sigma of the nearly exact keys cannot be had from the moments: those keys come back as SQL NULL, not as numbers
        assert np.all(nn[exact] == 1) and np.all(nn[easy] == 0)
    lib.arena_destroy(arena)


# ---- hash-partitioned ingest: W device states, rows routed by hash64(key) % W (duckdb_shim/sharded_arena.hpp) ----

def _sharded_lib():
    pkg, lib = _lib()
    abi = import_pkg("_abi")
    lib.sharded_create.restype = C.c_void_p
    lib.sharded_create.argtypes = [abi.AnofoxHipBatchOptions, C.c_uint32, C.c_void_p, C.c_size_t]
    lib.sharded_destroy.argtypes = [C.c_void_p]
    lib.sharded_update.restype = C.c_int
    lib.sharded_update.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.sharded_finalize.restype = C.c_int
    lib.sharded_finalize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.sharded_key_count.restype = C.c_size_t
    lib.sharded_key_count.argtypes = [C.c_void_p]
    lib.sharded_rows_of_shard.restype = C.c_uint64
    lib.sharded_rows_of_shard.argtypes = [C.c_void_p, C.c_uint32]
    lib.sharded_keys_of_shard.restype = C.c_size_t
    lib.sharded_keys_of_shard.argtypes = [C.c_void_p, C.c_uint32]
    lib.sharded_shard_of.restype = C.c_uint32
    lib.sharded_shard_of.argtypes = [C.c_uint64, C.c_uint32]
    return pkg, lib


def _run_sharded(pkg, lib, W, model, kw, keys, y, X, w, accept, n_threads=4):
    p = X.shape[1]
    h = lib.sharded_create(pkg.RegressionOptions(**kw).batch_options(model), W, None, 1 << 18)
    assert h
    errs = []

    def feed(t):          # thread t owns the rows of every n_threads-th Update vector (keys shared between threads)
        msg = C.create_string_buffer(256)
        for v, c0 in enumerate(range(0, len(keys), 2048)):
            if v % n_threads != t:
                continue
            sl = slice(c0, c0 + 2048)
            k, yy, xx, ww, aa = (np.ascontiguousarray(a[sl]) for a in (keys, y, X, w, accept))
            if lib.sharded_update(h, len(k), k.ctypes.data, yy.ctypes.data, xx.ctypes.data, p, ww.ctypes.data, aa.ctypes.data, msg) != 0:
                errs.append(msg.value)
    if n_threads == 1:
        feed(0)
    else:
        th = [threading.Thread(target=feed, args=(t,)) for t in range(n_threads)]
        [t.start() for t in th]
        [t.join() for t in th]
    assert not errs, errs
    K = lib.sharded_key_count(h)
    okeys = np.empty(K, dtype=np.uint64)
    core = np.empty((K, p + 6))
    inf = np.empty((K, 5 * p + 2))
    status = np.empty(K, dtype=np.int32)
    msg = C.create_string_buffer(256)
    assert lib.sharded_finalize(h, K, okeys.ctypes.data, core.ctypes.data, inf.ctypes.data, status.ctypes.data, msg) == 0, msg.value
    rows = [lib.sharded_rows_of_shard(h, s) for s in range(W)]
    nkeys = [lib.sharded_keys_of_shard(h, s) for s in range(W)]
    lib.sharded_destroy(h)
    order = np.argsort(okeys, kind="stable")
    return okeys[order], core[order], inf[order], status[order], rows, nkeys


@pytest.mark.parametrize("model,p", [("ols", 8), ("wls", 5), ("ols", 12)])
def test_hash_partitioned_ingest_matches_oracle_and_one_shard(model, p):
    """W = 2 and 4 device states on this box's one GPU (a node has one per GPU): every key's rows meet in one shard, the
    shards' fits are the oracle's, and — the rows of a key arriving in the same order whatever W is — bit-identical to W = 1
    for moment states (p <= 8), identical to rounding for log-only ones."""
    pkg, lib = _sharded_lib()
    dmod = import_pkg("distributed")
    rng = np.random.default_rng(900 + p)
    K, n = 5000, 150_000
    key_values = rng.integers(1, 1 << 63, K, dtype=np.uint64)
    kidx = rng.integers(0, K, n)
    keys = key_values[kidx]
    X = rng.uniform(-5, 5, (n, p)) + 1.0
    y = np.einsum("ij,ij->i", rng.uniform(-3, 3, (K, p))[kidx], X) + 2.0 + rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    accept = (rng.random(n) > 0.05).astype(np.uint8)
    kw = dict(compute_inference=True)
    # single-threaded feeding: arrival order of a key's rows is then the same for every W (bit-for-bit comparison)
    base = _run_sharded(pkg, lib, 1, model, kw, keys, y, X, w, accept, n_threads=1)
    keep = accept.astype(bool)
    ukeys, inv = np.unique(keys, return_inverse=True)
    order = np.nonzero(keep)[0][np.argsort(inv[keep], kind="stable")]
    offs = np.concatenate([[0], np.cumsum(np.bincount(inv[keep], minlength=len(ukeys)))]).astype(np.int64)
    rcore, rinf = oracle.fit_groups(y[order], [np.ascontiguousarray(X[order, j]) for j in range(p)], offs,
                                    w=(w[order] if model == "wls" else None), model=model, **kw)
    assert np.array_equal(base[0], ukeys) and np.array_equal(base[3], rcore[:, p + 5].astype(np.int32))
    fitted = rcore[:, p + 5] == 0
    assert_records_match(base[1][fitted], rcore[fitted], p, base[2][fitted], rinf[fitted], what=f"sharded W=1 {model} p={p}")
    for W in (2, 4):
        got = _run_sharded(pkg, lib, W, model, kw, keys, y, X, w, accept, n_threads=1)
        assert np.array_equal(got[0], ukeys) and np.array_equal(got[3], base[3])
        if p <= 8:      # moment states sum a key's rows in arrival order whatever the shard holds besides: bit for bit
            assert np.array_equal(got[1][fitted], base[1][fitted]) and np.array_equal(got[2][fitted], base[2][fitted])
        else:           # log-only states run the batch kernels, whose summation order follows the group's row offset in
            # the gathered batch (16-byte load alignment): the same fit to rounding
            assert_records_match(got[1][fitted], base[1][fitted], p, got[2][fitted], base[2][fitted],
                                 what=f"sharded W={W} vs W=1 {model} p={p}", coef_rtol=1e-12, diag_rtol=1e-10)
        # the routing is the documented hash: shard sizes are what hash64(key) % W says, and balanced
        owner = dmod.hash_partition(keys, W)
        assert got[4] == [int(np.sum(keep & (owner == s))) for s in range(W)]
        assert got[5] == [int(len(np.unique(keys[owner == s]))) for s in range(W)]
        assert all(lib.sharded_shard_of(int(k), W) == int(o) for k, o in zip(keys[:200], owner[:200]))
        assert max(got[5]) < 1.15 * K / W
    # several feeding threads per shard: same groups, rounding-level differences at most (arrival order across threads varies)
    mt = _run_sharded(pkg, lib, 4, model, kw, keys, y, X, w, accept, n_threads=4)
    assert np.array_equal(mt[0], ukeys) and np.array_equal(mt[3], base[3])
    assert_records_match(mt[1][fitted], rcore[fitted], p, mt[2][fitted], rinf[fitted], what=f"sharded W=4 threads {model} p={p}")
