"""Diagnostic: per-phase cycle shares of the solve kernel from in-kernel s_memtime stamps.

Needs the -DPYCLLP_PROFILE build (build/libpycllp_hip_prof.so); run as
    PYCLLP_HIP_LIB=build/libpycllp_hip_prof.so python tools/phase_profile.py
Read the SHARES, not the absolute time (the stamps fence the scheduler)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems, _native
from pycllp_amd.lp import EqualityLP, SparseMatrix
from pycllp_amd.solvers.hip import HipDensePrimalNormalSolver

NPHASE = 12
names = ["0 elementwise+reductions", "1 d,t,kbuf", "2 gram(MFMA)+Ax", "3 slab->rows,beta", "4 factor LDL", "5 fwd/back",
         "6 A'dy", "7 A dx + maxe (refine check)", "8 stop tests + step (+store)", "9 load LP"]
LPW = 2  # LPs per wave in the group kernel at m<=32 (4 at m<=16)
m, n, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 64, 65536)))
A, b, c = problems.random_dense_arrays(m, n, B)
Ae, be, ce = problems.equality_arrays(A, b, c)
s = HipDensePrimalNormalSolver()
EqualityLP(SparseMatrix(matrix=Ae), be[:1], ce[:1], 0.0).init(s)
L = _native.lib()
prof = torch.zeros(4096 * NPHASE, dtype=torch.int64, device="cuda")
L.pycllp_hip_debug_set_prof.argtypes = [ctypes.c_void_p]
L.pycllp_hip_debug_set_prof(ctypes.c_void_p(prof.data_ptr()))
bd = torch.as_tensor(be, device="cuda"); cd = torch.as_tensor(ce, device="cuda")
s.solve_device(bd, cd); torch.cuda.synchronize()
prof.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); buf = s.solve_device(bd, cd); e1.record(); torch.cuda.synchronize()
info = s.launch_info()
nw = info["grid"] * info["block"] // 64
p = prof.cpu().numpy()[:nw * NPHASE].reshape(nw, NPHASE).astype(np.float64)
iters = buf["iters"].cpu().numpy()
tot = p.sum(1)
print("kernel %.2f ms (stamped build), %d waves, LP-iterations per wave %.1f" % (e0.elapsed_time(e1), nw, (iters.sum() + B) / nw))
print("(cycles below are per LP-iteration, i.e. wave cycles divided by the LP-iterations the wave served)")
print("cycles per wave: mean %.3g  min %.3g  max %.3g" % (tot.mean(), tot.min(), tot.max()))
per_it = p.sum(0) / (iters.sum() + B)
for i in range(min(NPHASE, len(names))):
    print("%-34s %6.1f%%   %8.0f cycles per LP-iteration" % (names[i], 100 * p[:, i].sum() / p.sum(), per_it[i]))
print("total %.0f cycles per LP-iteration per wave" % per_it.sum())
