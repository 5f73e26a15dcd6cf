"""Diagnostic: phase cycle shares of the large-LP kernel (csrc/ipm_big.hip); needs the -DPYCLLP_PROFILE build (tools/build_prof.sh)
selected with PYCLLP_HIP_LIB=proflib/libpycllp_hip_prof.so.  usage: python tools/phase_profile_big.py [dense|sparse]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems, _native
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
NPHASE = 12
names = ["0 A'y, A x, norms, stop tests", "1 d, t, right-hand side, diag(M)", "2 Gram", "3 LDL' (diag block, panel, trailing update)",
         "4 block substitution(s) + A'dy, dx", "5 refinement test (+ passes)", "6 step", "7 load / store LP"]
for kind in (sys.argv[1:] or ["dense", "sparse"]):
    if kind == "dense":
        m, n, B = 200, 200, 2048
        A, b, c = problems.random_dense_arrays(m, n, B, seed=0); name = "hip_dense_primal_normal"
    else:
        m, n, B = 256, 512, 4096
        A, b, c = problems.random_sparse_arrays(m, n, B, density=0.02, seed=0); name = "hip_sparse_primal_normal"
    lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
    s = solver_registry[name](hsd=False); lp.init(s)
    L = _native.lib()
    prof = torch.zeros(2048 * NPHASE, dtype=torch.int64, device="cuda")
    L.pycllp_hip_debug_set_prof.argtypes = [ctypes.c_void_p]
    L.pycllp_hip_debug_set_prof(ctypes.c_void_p(prof.data_ptr()))
    be = torch.as_tensor(b, device="cuda"); ce = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
    buf = s.solve_device(be, ce); torch.cuda.synchronize()
    p = prof.cpu().numpy().reshape(-1, NPHASE).astype(np.float64); p = p[p.sum(1) > 0]
    iters = buf["iters"].cpu().numpy()
    per_it = p.sum(0) / iters.sum()
    print("# ipm_big_kernel, %s m=%d n=%d, %d LPs, %d workgroups, mean iterations %.1f (s_memtime stamps of wave 0, cycles per LP-iteration)"
          % (kind, m, n, B, len(p), iters.mean()))
    for i in range(len(names)):
        print("%-50s %6.1f%%   %9.0f" % (names[i], 100 * p[:, i].sum() / p.sum(), per_it[i]))
    print("total %.0f cycles per LP-iteration per workgroup" % per_it.sum())
