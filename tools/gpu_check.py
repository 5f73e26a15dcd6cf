"""Quick GPU sanity run (development aid): parity vs the oracle + a first timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd.solvers.hip import HipDensePrimalNormalSolver
from pycllp_amd import problems
from pycllp_amd.lp import EqualityLP, SparseMatrix
from oracle import port

def run(m, n, B, Bcheck=512):
    A, b, c = problems.random_dense_arrays(m, n, B)
    Ae, be, ce = problems.equality_arrays(A, b, c)
    lp = EqualityLP(SparseMatrix(matrix=Ae), be, ce, 0.0)
    s = HipDensePrimalNormalSolver()
    lp.init(s)
    # newton step parity
    rs = np.random.RandomState(1)
    nb = 64
    x = 0.1 + rs.rand(nb, n + m); z = 0.1 + rs.rand(nb, n + m); y = rs.rand(nb, m)
    dy = s.newton_step(x, z, y, be[:nb], ce[:nb], 1.0)
    ref = np.stack([port.newton_step_known_answer(Ae, x[i], z[i], y[i], be[i], ce[i], 1.0) for i in range(nb)])
    print("newton max rel err", np.abs(dy - ref).max() / np.abs(ref).max(), "nref", s.nrefine.max())
    t = time.time(); st = lp.solve(s); t1 = time.time() - t
    print("launch", s.launch_info())
    print("(%d,%d) B=%d first solve %.3fs status" % (m, n, B, t1), np.bincount(st), "iters", s.iters.mean(), s.iters.min(), s.iters.max())
    r = port.dense_solve(Ae, be[:Bcheck], ce[:Bcheck], nthreads=8)
    ep = np.abs(s.primal_obj[:Bcheck] - r['pobj']) / np.maximum(1, np.abs(r['pobj']))
    ed = np.abs(s.dual_obj[:Bcheck] - r['dobj']) / np.maximum(1, np.abs(r['dobj']))
    print("vs oracle: pobj", ep.max(), "dobj", ed.max(), "x", np.abs(s.x[:Bcheck] - r['x']).max(), "iters equal", (s.iters[:Bcheck] == r['iters']).mean(), "status equal", (st[:Bcheck] == r['status']).all())
    bd = torch.as_tensor(be, device='cuda'); cd = torch.as_tensor(ce, device='cuda')
    for _ in range(2): s.solve_device(bd, cd)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): s.solve_device(bd, cd)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    print("device-resident solve %.2f ms -> %.0f LPs/s" % (dt * 1e3, B / dt))

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(16, 32, 4096)
    run(32, 64, 65536)
