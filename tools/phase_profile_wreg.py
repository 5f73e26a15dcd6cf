"""Diagnostic: phase cycle shares of the register-resident wave kernel (ipm_wreg_kernel) from in-kernel s_memtime stamps.
Needs a -DPYCLLP_PROFILE build: PYCLLP_HIP_LIB=proflib/libpycllp_hip_prof.so python tools/phase_profile_wreg.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems, _native
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
NPHASE = 12
names = ["0 A'y, norms, d, t, A x, tests, A(dt), diag(M)", "1 Gram scatter + block loads", "2 LDL': Schur MFMA of the diagonal block -> tile",
         "3 LDL': original diagonal block from the tables -> tile", "4 LDL': tile -> rows, 16-step pivot chain", "5 LDL': W = L_KK^-1, panel (MFMA)", "6 LDL': trailing update (MFMA)",
         "7 block substitution (solve)", "8 A'dy, dx, A dx, refinement test", "9 step, load/store LP", "10 (of 0) A'y, sigma, gamma, objectives", "11 (of 0) d, t, A x, rho, stop tests"]
m, n, B = 128, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if os.environ.get("DENSE"):      # DENSE=m,n: a dense A of that shape (dense-image variant of the kernel)
    m, n = (int(v) for v in os.environ["DENSE"].split(","))
    A, b, c = problems.random_dense_arrays(m, n, B, seed=0)
else:
    A, b, c = problems.random_sparse_arrays(m, n, B, density=0.025, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
s = solver_registry["hip_sparse_primal_normal"](hsd=bool(int(os.environ.get("HSD", "0")))); lp.init(s)
L = _native.lib()
prof = torch.zeros(1024 * NPHASE, dtype=torch.int64, device="cuda")
L.pycllp_hip_debug_set_prof.argtypes = [ctypes.c_void_p]
L.pycllp_hip_debug_set_prof(ctypes.c_void_p(prof.data_ptr()))
be = torch.as_tensor(b, device="cuda"); ce = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
buf = s.solve_device(be, ce); torch.cuda.synchronize(); prof.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); buf = s.solve_device(be, ce); e1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(-1, NPHASE).astype(np.float64); p = p[p.sum(1) > 0]
iters = buf["iters"].cpu().numpy()
print("kernel %.1f ms (stamped build), %d waves, mean iterations %.2f, %s" % (e0.elapsed_time(e1), len(p), iters.mean(), s.launch_info()))
per_it = p.sum(0) / iters.sum()
for i in range(NPHASE):
    print("%-55s %6.1f%%   %9.0f cycles per LP-iteration" % (names[i], 100 * p[:, i].sum() / p.sum(), per_it[i]))
print("total %.0f cycles per LP-iteration per wave" % per_it.sum())
