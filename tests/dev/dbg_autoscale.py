import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from oracle import port
rs = np.random.RandomState(5)
for (m,n,dens) in ((60,140,0.08),(128,256,0.025),(30,100,0.05)):
    A,_,_ = problems.random_sparse_arrays(m,n,1,density=dens,seed=3)
    B=24
    b = 1e-3*(0.5+rs.rand(B,m)); c = 1e2*(0.5+rs.rand(B,n))
    lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
    s = solver_registry["hip_sparse_primal_normal"](autoscale=True, hsd=False); lp.init(s); st = lp.solve(s)
    r = port.dense_solve(lp.A.todense(), lp.b, lp.c, nthreads=8, flags=8)
    print(m,n,s.launch_info()["kernel"], "status same", np.array_equal(st,r["status"]), "iters diff", np.abs(s.iters.astype(int)-r["iters"]).max(), "obj err", (np.abs(s.primal_obj-r["pobj"])/np.abs(r["pobj"])).max(), "x err", np.abs(s.x-r["x"]).max()/np.abs(r["x"]).max(), "y err", np.abs(s.y-r["y"]).max()/np.abs(r["y"]).max())
