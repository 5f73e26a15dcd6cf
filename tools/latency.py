#!/usr/bin/env python
"""Small-batch latency of the plugin path: `lp.solve(solver)` host numpy -> host numpy, and the
device-resident `solve_device` launch + synchronise, at B = 1, 64, 1 024 -- the reference's "repeat solve" regime
(README.md:5-6: many solves of a small batch with mutated b, c on one initialised solver).  GPU box only.

    python tools/latency.py > gpurun_out/latency.txt
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


def run(name, make, reps=30):
    for B in (1, 64, 1024):
        A, b, c = make(B)
        lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
        for hsd in ("auto", False):
            s = solver_registry[name](hsd=hsd)
            lp.init(s)
            lp.solve(s)
            host = []
            for _ in range(reps):
                t = time.perf_counter(); lp.solve(s); host.append(time.perf_counter() - t)
            bd = torch.as_tensor(lp.b, device="cuda"); cd = torch.as_tensor(lp.c, device="cuda")
            s.solve_device(bd, cd); torch.cuda.synchronize()
            devt = []
            for _ in range(reps):
                t = time.perf_counter(); s.solve_device(bd, cd); torch.cuda.synchronize(); devt.append(time.perf_counter() - t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s.solve_device(bd, cd); e1.record(); torch.cuda.synchronize()
            print("%-26s B=%5d hsd=%-5s lp.solve() host->host median %8.1f us (min %8.1f)   solve_device+sync median %8.1f us   "
                  "kernel (events) %8.1f us   mean iters %.1f"
                  % (name, B, hsd, 1e6 * np.median(host), 1e6 * np.min(host), 1e6 * np.median(devt), 1e3 * e0.elapsed_time(e1),
                     s.iters.mean()))
            sys.stdout.flush()


if __name__ == "__main__":
    print("# tools/latency.py: small-batch latency, 1x MI355X; times are per solve() call of the whole batch")
    run("hip_dense_primal_normal", lambda B: problems.random_dense_arrays(32, 64, B, seed=0))
    run("hip_dense_primal_normal", lambda B: problems.random_dense_arrays(16, 32, B, seed=0))
    run("hip_sparse_primal_normal", lambda B: problems.random_sparse_arrays(128, 256, B, density=0.025, seed=0))
