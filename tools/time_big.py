#!/usr/bin/env python
"""Throughput of the large-LP kernel (csrc/ipm_big.hip) on a few shapes beyond m = 128 / n = 512, device-resident solve,
with the reference CPU solver (oracle/_ref, hsd.c) timed on a few LPs of the same batch.  GPU box only.

    python tools/time_big.py > gpurun_out/time_big.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pycllp_amd import problems  # noqa: E402
from pycllp_amd.lp import SparseMatrix, StandardLP  # noqa: E402
from pycllp_amd.solvers import solver_registry  # noqa: E402
from oracle import hsd_ref  # noqa: E402

CASES = [("dense", 200, 200, 4096, None), ("dense", 256, 256, 2048, None), ("dense", 144, 400, 4096, None),
         ("sparse", 256, 512, 8192, 0.02), ("sparse", 256, 1024, 4096, 0.01), ("sparse", 160, 800, 8192, 0.03)]

print("# tools/time_big.py: ipm_big_kernel, 1x MI355X, device-resident solve (median of 3), hsd=False")
for kind, m, n, B, dens in CASES:
    if kind == "dense":
        A, b, c = problems.random_dense_arrays(m, n, B, seed=0)
        name = "hip_dense_primal_normal"
    else:
        A, b, c = problems.random_sparse_arrays(m, n, B, density=dens, seed=0)
        name = "hip_sparse_primal_normal"
    lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
    s = solver_registry[name](hsd=False)
    lp.init(s)
    bd = torch.as_tensor(b, device="cuda")
    cd = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
    s.solve_device(bd, cd); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r = s.solve_device(bd, cd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt = float(np.median(ts))
    st = r["status"].cpu().numpy(); it = r["iters"].cpu().numpy()
    info = s.launch_info()
    cpu = ""
    if hsd_ref.available() and not os.environ.get("PYCLLP_HIP_LIB"):
        k = 4
        t = time.perf_counter(); rr = hsd_ref.solve_standard(A, b[:k], c[:k]); dc = (time.perf_counter() - t) / k
        err = np.abs(r["pobj"][:k].cpu().numpy() - rr["pobj"]) / np.maximum(1.0, np.abs(rr["pobj"]))
        cpu = "  reference hsd.c %.1f ms/LP on one core (%.0f LPs/s), objectives within %.1e of it" % (1e3 * dc, 1.0 / dc, err.max())
    print("%-6s m=%3d n=%4d %s B=%5d: %8.1f ms = %9.0f LPs/s  (%s, %d B LDS, status counts %s, mean iters %.1f)%s"
          % (kind, m, n, ("density %.2f" % dens) if dens else "", B, 1e3 * dt, B / dt, info["variant"], info["lds_bytes"],
             dict(zip(*[a.tolist() for a in np.unique(st, return_counts=True)])), it.mean(), cpu))
    sys.stdout.flush()
