"""A/B timing of several builds of libpycllp_hip.so in ONE process, interleaved rounds (cdna guide rule 24).
usage: python tools/ab_bench.py build/a.so build/b.so ...   [env AB_M, AB_N, AB_B, AB_ROUNDS, AB_FLAGS]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems, _native

m, n, B = int(os.environ.get("AB_M", 32)), int(os.environ.get("AB_N", 64)), int(os.environ.get("AB_B", 65536))
rounds = int(os.environ.get("AB_ROUNDS", 7))
flags = int(os.environ.get("AB_FLAGS", 0))
A, b, c = problems.random_dense_arrays(m, n, B)
Ae, be, ce = problems.equality_arrays(A, b, c)
N = Ae.shape[1]
dev = "cuda"
Ad = torch.as_tensor(Ae, device=dev); bd = torch.as_tensor(be, device=dev); cd = torch.as_tensor(ce, device=dev)
x = torch.empty((B, N), dtype=torch.float64, device=dev); z = torch.empty_like(x)
y = torch.empty((B, m), dtype=torch.float64, device=dev)
po = torch.empty(B, dtype=torch.float64, device=dev); du = torch.empty_like(po)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty_like(st)
vp = ctypes.c_void_p
libs = []
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.pycllp_hip_dense_init.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, ctypes.POINTER(vp)]
    L.pycllp_hip_dense_solve.argtypes = [vp, ctypes.c_long] + [vp] * 9 + [ctypes.POINTER(_native.Opts), vp]
    L.pycllp_hip_default_opts.argtypes = [ctypes.POINTER(_native.Opts)]
    h = vp()
    assert L.pycllp_hip_dense_init(m, N, vp(Ad.data_ptr()), None, ctypes.byref(h)) == 0
    o = _native.Opts(); L.pycllp_hip_default_opts(ctypes.byref(o)); o.flags = flags
    if os.environ.get("AB_PIVOT_FLOOR"): o.pivot_floor = float(os.environ["AB_PIVOT_FLOOR"])
    if os.environ.get("AB_R"): o.r = float(os.environ["AB_R"])
    libs.append((path, L, h, o))

def run(L, h, o):
    rc = L.pycllp_hip_dense_solve(h, B, vp(bd.data_ptr()), vp(cd.data_ptr()), vp(x.data_ptr()), vp(y.data_ptr()), vp(z.data_ptr()),
                                  vp(po.data_ptr()), vp(du.data_ptr()), vp(st.data_ptr()), vp(it.data_ptr()), ctypes.byref(o), None)
    assert rc == 0

times = {p: [] for p, *_ in libs}
for p, L, h, o in libs:
    run(L, h, o); torch.cuda.synchronize()
    print(p, "status0", int((st == 0).sum()), "mean iters %.3f" % it.double().mean().item(), "pobj sum %.12g" % po.sum().item())
for r in range(rounds):
    for p, L, h, o in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(L, h, o); e1.record(); torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1))
for p in times:
    t = np.array(times[p])
    print("%-40s median %.3f ms  min %.3f ms  -> %.3f M LPs/s" % (p, np.median(t), t.min(), B / np.median(t) / 1e3))
