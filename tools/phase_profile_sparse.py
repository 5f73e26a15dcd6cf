"""Diagnostic: phase cycle shares of the sparse block kernel (needs the -DPYCLLP_PROFILE build, see phase_profile.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems, _native
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
NPHASE = 12
names = ["0 A'y, sigma, reductions", "1 staging, zero M, A x, Gram assembly, A d t", "2 rhs, beta", "3 LDL': barriers + trailing update (MFMA)", "4 solve (wave 0)",
         "5 A'dy, dx", "6 refinement check (+passes)", "7 tests + step", "8 LDL': diagonal block + panel (wave 0)", "9 load/store LP"]
m, n, B = 128, 256, 4096
A, b, c = problems.random_sparse_arrays(m, n, B, density=0.025, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
s = solver_registry["hip_sparse_primal_normal"](); lp.init(s)
L = _native.lib()
prof = torch.zeros(1024 * NPHASE, dtype=torch.int64, device="cuda")
L.pycllp_hip_debug_set_prof.argtypes = [ctypes.c_void_p]
L.pycllp_hip_debug_set_prof(ctypes.c_void_p(prof.data_ptr()))
be = torch.as_tensor(b, device="cuda"); ce = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
buf = s.solve_device(be, ce); torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(-1, NPHASE).astype(np.float64); p = p[p.sum(1) > 0]
iters = buf["iters"].cpu().numpy()
per_it = p.sum(0) / (iters.sum() + B)
for i in range(min(NPHASE, len(names))):
    print("%-50s %6.1f%%   %9.0f cycles per LP-iteration" % (names[i], 100 * p[:, i].sum() / p.sum(), per_it[i]))
print("total %.0f cycles per LP-iteration per workgroup; %d workgroups" % (per_it.sum(), len(p)))
