"""Accuracy of the wide primary solve against the oracle on benchmark-shaped data (run with ANOFOX_SOLVE_TILES=0/1)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("anofox-statistics_amd")
synth = importlib.import_module("anofox-statistics_amd.synth")
import oracle

G, n, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
offs, y, x_cols, w = synth.make_grouped(G, n, p, weights=True, device="cuda:0")
ctx = pkg.Context(0)
for model in ("ols", "wls", "ridge"):
    opts = pkg.RegressionOptions(compute_inference=True).batch_options(model)
    core, inf = ctx.fit_batch_device(offs, y, x_cols, w if model == "wls" else None, opts)
    torch.cuda.synchronize()
    rcore, rinf = oracle.fit_groups(y.cpu().numpy(), [c.cpu().numpy() for c in x_cols], offs.cpu().numpy(),
                                    w=w.cpu().numpy() if model == "wls" else None, model=model, compute_inference=True, n_threads=16)
    core, inf = core.cpu().numpy(), inf.cpu().numpy()
    scale = np.max(np.abs(rcore[:, :p + 1]), axis=1, keepdims=True)
    cerr = np.abs(core[:, :p + 1] - rcore[:, :p + 1]) / np.maximum(np.abs(rcore[:, :p + 1]), 1e-3 * scale)
    derr = np.abs(core[:, p + 1:p + 4] / rcore[:, p + 1:p + 4] - 1.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        ierr = np.abs(inf - rinf) / np.maximum(np.abs(rinf), 1e-300)
    print(f"{model} tiles={os.environ.get('ANOFOX_SOLVE_TILES','1')} G={G} n={n} p={p}: coef {cerr.max():.2e} (intercept {cerr[:, p].max():.2e})  "
          f"r2/adj/sigma {derr.max(axis=0)}  inference max {np.nanmax(ierr):.2e}  se {np.nanmax(ierr[:, :p]):.2e}  status eq {np.array_equal(core[:, p+5], rcore[:, p+5])}"
          f" refined {ctx.last_refine_count() if hasattr(ctx,'last_refine_count') else '?'}")
