"""Randomised parity sweep of the ROUND-3 kernels against the oracle (development aid, GPU box): the large-LP kernel
(csrc/ipm_big.hip: dense and sparse, edge sizes m = 129 / 256, n up to 1280), per-problem A on the wave kernel, and the
predictor-corrector option on every kernel that has it -- plain path and HSD.  FUZZ_SEED / FUZZ_N select the stream / count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from pycllp_amd.solvers.hip import autoscale_wanted
from oracle import port

rs = np.random.RandomState(int(os.environ.get("FUZZ_SEED", 1)))
N = int(os.environ.get("FUZZ_N", 10))
rel = lambda a, r: np.abs(a - r) / np.maximum(1.0, np.abs(r))
bad = 0


def check(tag, s, st, r, tol=1e-8, hsd_run=False):
    global bad
    same = np.array_equal(st, r["status"])
    okm = st == 0
    ep = rel(s.primal_obj[okm], r["pobj"][okm]).max() if okm.any() else 0.0
    ed = rel(s.dual_obj[okm], r["dobj"][okm]).max() if okm.any() else 0.0
    dit = np.abs(s.iters.astype(int) - r["iters"]).max()
    flag = "" if (same and ep < tol and ed < tol and dit <= (5 if hsd_run else 2)) else "   <-- MISMATCH"
    bad += bool(flag)
    print("%-78s status %s same=%s obj err %.1e %.1e |d iters| %d%s" % (tag, np.bincount(st, minlength=6).tolist(), same, ep, ed, dit, flag))
    sys.stdout.flush()


def oflags(lp, fl):
    return fl | (8 if autoscale_wanted(lp.b, lp.c) else 0)


# ---- large-LP kernel ----
from fuzz_cases import big_cases
gen = big_cases(int(os.environ.get("FUZZ_SEED", 1)), N)
for case in gen:
    if not isinstance(case, tuple):
        rs = case            # the stream continues below
        break
    m, n, B, dense, hsd, pc, A, b, c = case
    name = "hip_dense_primal_normal" if dense else "hip_sparse_primal_normal"
    lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
    s = solver_registry[name](hsd=hsd, predcorr=pc); lp.init(s)
    try:
        st = lp.solve(s)
    except NotImplementedError as exc:       # the option is refused on kernels that do not implement it (block kernel)
        print("big? m=%d n=%d pc=%d: refused (%s)" % (m, n, pc, str(exc)[-60:])); continue
    r = port.dense_solve(np.asarray(lp.A.todense()), lp.b, lp.c, nthreads=8, flags=oflags(lp, (32 if hsd else 0) | (128 if pc else 0)))
    info = s.launch_info()
    check("big? %s m=%d n=%d B=%d hsd=%d pc=%d [%s/%s]" % ("dense" if dense else "sparse", m, n, B, hsd, pc, info.get("kernel"), info.get("variant")), s, st, r, hsd_run=hsd)

# ---- per-problem A (wave kernel PA variants), plain and HSD ----
for t in range(max(2, N // 2)):
    m = int(rs.randint(2, 129)); n = int(rs.randint(3, 513 - m)); B = int(rs.choice([1, 7, 50]))
    A, b, c = problems.random_sparse_arrays(m, n, B, density=min(1.0, max(float(rs.choice([0.02, 0.05, 0.2])), 3.0 / n)), seed=int(rs.randint(1 << 30)))
    rows, cols, data = problems.per_problem_values(A, B, seed=int(rs.randint(1 << 30)))
    hsd = bool(rs.rand() < 0.5)
    lp = StandardLP(SparseMatrix(rows, cols, data), b, c, 0.0).to_equality_form()
    s = solver_registry["hip_sparse_primal_normal"](hsd=hsd); lp.init(s)
    try:
        st = lp.solve(s)
    except NotImplementedError as exc:       # a structure whose tables fit neither kernel's LDS
        print("perA m=%d n=%d nnz %d: refused (%s)" % (m, n, A.nnz, str(exc)[-70:])); continue
    info = s.launch_info()
    res = dict(status=[], pobj=[], dobj=[], iters=[])
    for k in range(B):
        rk = port.dense_solve(np.asarray(lp.A.todense(k)), lp.b[k:k + 1], lp.c[k:k + 1], flags=oflags(lp, 32 if hsd else 0))
        for key in res: res[key].append(rk[key][0])
    r = {k: np.array(v) for k, v in res.items()}
    check("perA m=%d n=%d B=%d nnz %d hsd=%d [%s/%s]" % (m, n, B, A.nnz, hsd, info["kernel"], info["variant"]), s, st, r, hsd_run=hsd)

# ---- predictor-corrector on the lane-group, wave (tables / dense image) kernels ----
for t in range(max(3, N // 2)):
    kind = ["group", "tables", "image"][t % 3]
    B = int(rs.choice([1, 9, 130]))
    if kind == "group":
        m = int(rs.randint(1, 33)); n = int(rs.randint(1, 129 - m))
        A = rs.rand(m, n); b = 0.5 + rs.rand(B, m); c = 0.5 + rs.rand(B, n); name = "hip_dense_primal_normal"
    elif kind == "tables":
        m = int(rs.randint(2, 129)); n = int(rs.randint(3, 513 - m))
        A, b, c = problems.random_sparse_arrays(m, n, B, density=min(1.0, max(0.03, 3.0 / n)), seed=int(rs.randint(1 << 30))); name = "hip_sparse_primal_normal"
    else:
        m = int(rs.randint(33, 129)); n = int(rs.randint(8, 120))
        A = rs.rand(m, n); b = 0.5 + rs.rand(B, m); c = 0.5 + rs.rand(B, n); name = "hip_dense_primal_normal"
    lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
    s = solver_registry[name](hsd=False, predcorr=True); lp.init(s)
    try:
        st = lp.solve(s)
    except NotImplementedError as exc:
        print("predcorr %s m=%d n=%d: refused (%s)" % (kind, m, n, str(exc)[-60:])); continue
    r = port.dense_solve(np.asarray(lp.A.todense()), lp.b, lp.c, nthreads=8, flags=oflags(lp, 128))
    info = s.launch_info()
    check("predcorr %s m=%d n=%d B=%d [%s/%s]" % (kind, m, n, B, info.get("kernel", "group"), info.get("variant", "")), s, st, r)
print("mismatches:", bad)
