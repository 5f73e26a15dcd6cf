#!/usr/bin/env python
"""VERDICT r2 item 7 (warm start): characterise what the repeat-solve test compares.  For the 16 x 32 (dense) case of
tests/test_hip_parity.py::test_repeat_solve_warm_start_through_the_plugin_api, print -- for the library selected by
PYCLLP_HIP_LIB -- the distribution of |iterations(kernel) - iterations(oracle)| over the 300 warm-started LPs, the objective
deviation, and, for the LPs that differ, how far the two runs are apart after k iterations (relative distance of x).
GPU box only.  Usage: python tests/dev/warm_band.py [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import port  # noqa: E402
from pycllp_amd import problems  # noqa: E402
from pycllp_amd.lp import SparseMatrix, StandardLP  # noqa: E402
from pycllp_amd.solvers import solver_registry  # noqa: E402

print("# library:", os.environ.get("PYCLLP_HIP_LIB", "default"))
for seed in [int(a) for a in sys.argv[1:]] or [6, 7, 8, 9]:
    m, n = 16, 32
    rs = np.random.RandomState(3)
    A, b, c = problems.random_dense_arrays(m, n, 300, seed=seed)
    lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
    s = solver_registry["hip_dense_primal_normal"](warm_start=True)
    lp.init(s); lp.solve(s)
    it_cold = s.iters.copy()
    x0, y0, z0 = s.x.copy(), s.y.copy(), s.z.copy()
    lp.b[:] = lp.b * (1.0 + 0.01 * rs.rand(*lp.b.shape))
    lp.c[:, :n] = lp.c[:, :n] * (1.0 + 0.01 * rs.rand(300, n))
    lp.solve(s)
    Ae = lp.A.todense()
    r = port.dense_solve(Ae, lp.b, lp.c, nthreads=8, x0=x0, y0=y0, z0=z0, flags=1)
    full = port.dense_solve(Ae, lp.b, lp.c, nthreads=8)
    diff = np.abs(s.iters.astype(int) - r["iters"])
    err = np.abs(s.primal_obj - full["pobj"]) / np.maximum(1.0, np.abs(full["pobj"]))
    print("seed %d: status0 %d/300  cold median %g  warm median kernel %g oracle %g  |diff| histogram %s  share<=1 %.3f max %d  "
          "objective deviation from the cold oracle %.2e"
          % (seed, int((s.status == 0).sum()), np.median(it_cold), np.median(s.iters), np.median(r["iters"]),
             np.bincount(diff).tolist(), (diff <= 1).mean(), diff.max(), err.max()))
    # trajectories of the LPs that differ most: relative distance of x after k iterations
    worst = np.argsort(-diff)[:3]
    bd, cd = torch.as_tensor(lp.b, device="cuda"), torch.as_tensor(lp.c, device="cuda")
    for k in (1, 2, 3, 5, 8):
        buf = s.solve_device  # (x, z, y of the previous solve are the start: re-seed them for every probe)
        st = s.buffers["set0"]
        st["x"].copy_(torch.as_tensor(x0, device="cuda")); st["z"].copy_(torch.as_tensor(z0, device="cuda")); st["y"].copy_(torch.as_tensor(y0, device="cuda"))
        g = s.solve_device(bd, cd, warm_start=True, max_iter=k); torch.cuda.synchronize()
        rk = port.dense_solve(Ae, lp.b, lp.c, nthreads=8, x0=x0, y0=y0, z0=z0, flags=1, max_iter=k)
        gx = g["x"].cpu().numpy()
        d = np.abs(gx - rk["x"]).max(axis=1) / np.abs(rk["x"]).max(axis=1)
        print("   after %d iteration(s): max over all LPs of |x_kernel - x_oracle| / |x| = %.2e; at the 3 LPs with the largest iteration gap: %s"
              % (k, d.max(), ["%.1e" % v for v in d[worst]]))
