"""Bring-up checks of the register-resident wave kernel (csrc/ipm_wreg.hip) on a GPU: cross-lane primitives, the
stand-alone register LDL' solve, the stand-alone sparse Newton step and the full solve against the oracle."""
import ctypes, sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pycllp_amd import _native, problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from oracle import port

L = _native.lib()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())

def selftest():
    raw = ctypes.CDLL(_native.LIB_PATH)
    out = torch.zeros(1024, dtype=torch.float64, device=dev)
    rc = raw.pycllp_hip_debug_wreg_selftest(P(out), None)
    torch.cuda.synchronize()
    o = out.cpu().numpy(); lane = np.arange(64); v = 1.0 + lane
    exp_quad = np.array([v[(l & 15) + 16 * np.arange(4)].sum() for l in lane])
    exp_row = np.array([v[(l & ~15):(l & ~15) + 16].sum() for l in lane])
    ok = {
        "quad_sum": np.array_equal(o[:64], exp_quad), "row_sum": np.array_equal(o[64:128], exp_row),
        "wsum": np.all(o[128:192] == 2080.0), "row_bcast5": np.array_equal(o[192:256], 1.0 + (lane & ~15) + 5),
        "wmax": np.all(o[512:576] == 64.0)}
    for r in range(4):
        exp = (100.0 * (4 * r + (lane >> 4)) + 1) * ((lane & 15) + 1)
        ok["mfma_r%d" % r] = np.array_equal(o[256 + 64 * r:320 + 64 * r], exp)
    print("selftest rc", rc, ok)
    if not all(ok.values()):
        print("quad", o[:64]); print("row", o[64:128]); print("mfma0", o[256:320])
    return all(ok.values())

def ldl_solve(n, B=64):
    rs = np.random.RandomState(n)
    A = rs.rand(B, n, 3 * n)
    M = np.einsum("bik,bk,bjk->bij", A, rs.rand(B, 3 * n) + 0.01, A)
    rhs = rs.rand(B, n)
    Md, rd = torch.as_tensor(M, device=dev), torch.as_tensor(rhs, device=dev)
    x = torch.zeros((B, n), dtype=torch.float64, device=dev)
    rc = L.pycllp_hip_ldl_solve(n, B, P(Md), P(rd), P(x), 0, 1.0, 0.0, None)
    torch.cuda.synchronize()
    ref = np.linalg.solve(M, rhs[..., None])[..., 0]
    err = np.abs(x.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("ldl_solve n=%d rc=%d rel err %.2e" % (n, rc, err))
    return rc == 0 and err < 1e-9

def newton(m=128, n=256, B=32, density=0.025):
    A, b, c = problems.random_sparse_arrays(m, n, B, density=density, seed=1)
    lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
    s = solver_registry["hip_sparse_primal_normal"](device=dev)
    lp.init(s)
    Ae = np.hstack([A.toarray(), np.eye(m)]); N = n + m
    rs = np.random.RandomState(5)
    x = 0.5 + rs.rand(B, N); z = 0.5 + rs.rand(B, N); y = rs.rand(B, m); ce = np.hstack([c, np.zeros((B, m))])
    t = [torch.as_tensor(np.ascontiguousarray(v), device=dev) for v in (x, z, y, b, ce)]
    dy = torch.zeros((B, m), dtype=torch.float64, device=dev); nref = torch.zeros(B, dtype=torch.int32, device=dev)
    o = _native.default_opts()
    rc = L.pycllp_hip_sparse_newton(s._handle, B, *[P(v) for v in t], 1.0, P(dy), P(nref), ctypes.byref(o), None)
    torch.cuda.synchronize()
    ref = np.stack([port.newton_step_known_answer(Ae, x[i], z[i], y[i], b[i], ce[i], 1.0) for i in range(B)])
    err = np.abs(dy.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("newton (%d,%d) rc=%d rel err %.2e nref max %d" % (m, n, rc, err, int(nref.max())))
    return rc == 0 and err < 1e-7

def full(m=128, n=256, B=256, density=0.025, flags=0, hsd=False):
    A, b, c = problems.random_sparse_arrays(m, n, B, density=density, seed=0)
    lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
    s = solver_registry["hip_sparse_primal_normal"](device=dev, flags=flags, hsd=hsd)
    lp.init(s)
    t0 = time.time(); st = lp.solve(s); dt = time.time() - t0
    Ae = np.hstack([A.toarray(), np.eye(m)]); ce = np.hstack([c, np.zeros((B, m))])
    ref = port.dense_solve(Ae, b, ce, nthreads=16, flags=32 if hsd else 0)
    err = np.abs(s.primal_obj - ref["pobj"]) / np.maximum(1, np.abs(ref["pobj"]))
    print("full hsd=%s flags=%d B=%d: status0 %d/%d, iters equal %s (max diff %d), max obj err %.2e, %.3fs"
          % (hsd, flags, B, int((st == 0).sum()), B, np.array_equal(s.iters, ref["iters"]),
             int(np.abs(s.iters - ref["iters"]).max()), err.max(), dt))
    return (st == 0).all() and err.max() < 1e-9

def timing(B=16384, flags=0, hsd=False, reps=3):
    A, b, c = problems.random_sparse_arrays(128, 256, B, density=0.025, seed=0)
    lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
    s = solver_registry["hip_sparse_primal_normal"](device=dev, flags=flags, hsd=hsd)
    lp.init(s)
    bd = torch.as_tensor(b, device=dev); cd = torch.as_tensor(np.hstack([c, np.zeros((B, 128))]), device=dev)
    for k in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.time()
        r = s.solve_device(bd, cd); torch.cuda.synchronize(); dt = time.time() - t0
        if k: print("timing flags=%d hsd=%s: %.1f ms  %.0f LPs/s, mean iters %.2f, status0 %d" % (
            flags, hsd, dt * 1e3, B / dt, float(r["iters"].float().mean()), int((r["status"] == 0).sum())))

if __name__ == "__main__":
    what = sys.argv[1:] or ["selftest", "ldl", "newton", "full", "timing"]
    ok = True
    if "selftest" in what: ok &= selftest()
    if "ldl" in what:
        for n in (128, 100, 37, 16): ok &= ldl_solve(n)
    if "newton" in what: ok &= newton()
    if "full" in what:
        ok &= full(flags=64, B=64); ok &= full(flags=0, B=256); ok &= full(flags=4, B=32)
        ok &= full(flags=0, B=256, hsd=True); ok &= full(flags=64, B=64, hsd=True)
    if "timing" in what:
        timing(flags=0); timing(flags=0, hsd=True)
    print("BRINGUP", "OK" if ok else "FAILED")
