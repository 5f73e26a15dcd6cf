#!/usr/bin/env python
"""Generate tests/golden/*.npz -- run HERE (this container), never on the GPU box.

Golden OUTPUTS come from the reference's own CPU solver, `pycllp/ipo.py` -> `ipo/hsd.c` (Vanderbei's
homogeneous self-dual IPM), compiled from the reference sources into oracle/_ref/libhsd_ref.so by
oracle/Makefile and called exactly like the reference's wrapper does (oracle/hsd_ref.py).  INPUTS are the
data of the reference's own tests (tests/vanderbei_problems.py, tests/test_simple.py, tests/helpers.py)
and the SURVEY section 8d synthetic generator.  Only data is written: inputs + expected outputs.

The reference's tests use GLPK as ground truth for random problems (tests/helpers.py:63-64,91-95); GLPK is
absent here, so the HSD solver (the parity oracle BASELINE.json names) stands in for it.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hsd_ref, port  # noqa: E402
from pycllp_amd import problems  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def hsd(A, b, c):
    r = hsd_ref.solve_standard(A, b, c)
    return dict(x=r["x"], y=r["y"], pobj=r["pobj"], dobj=r["dobj"], status=r["status"])


def textbook():
    lp29, xopt29 = problems.vanderbei_2_9()
    A, b, c = lp29.A.todense(), lp29.b, lp29.c
    g = hsd(A, b, c)
    lp210, xopt210 = problems.vanderbei_2_10()
    np.savez(os.path.join(OUT, "vanderbei.npz"),
             v29_A=A, v29_b=b, v29_c=c, v29_xopt=xopt29, v29_x=g["x"], v29_y=g["y"], v29_pobj=g["pobj"],
             v29_dobj=g["dobj"], v29_status=g["status"],
             v210_A=lp210.A.todense(), v210_b=lp210.b, v210_c=lp210.c, v210_xopt=xopt210)
    print("vanderbei 2.9: HSD x", g["x"][0], "pobj", g["pobj"][0], "dobj", g["dobj"][0])


def small_problem():
    A, b, c = problems.small_problem_arrays()
    g1 = hsd(A, b[None], c[None])
    Ap, bb, cc = problems.parallel_small_problem_arrays(32)
    g = hsd(Ap, bb, cc)
    np.savez(os.path.join(OUT, "small_problem.npz"), A=A, b=b, c=c, x1=g1["x"], pobj1=g1["pobj"], dobj1=g1["dobj"],
             bb=bb, cc=cc, x=g["x"], y=g["y"], pobj=g["pobj"], dobj=g["dobj"], status=g["status"])
    print("small problem: x", g1["x"][0], "status x32", np.bincount(g["status"]))


def helpers_random(m, n, density=1.0, nproblems=1):
    """Input generator of the reference's tests/helpers.py:35-60 (np.random.seed(0), rows from
    scipy.sparse.rand, b,c ~ U[0,1))."""
    from scipy.sparse import rand
    np.random.seed(0)
    A = np.empty((m, n))
    for i in range(m):
        A[i, :] = rand(1, n, density=max(density, 3. / n)).todense()
    b = np.random.rand(nproblems, m)
    c = np.random.rand(nproblems, n)
    return A, b, c


def random_helpers():
    out = {}
    for (m, n) in ((10, 10), (20, 20)):
        A, b, c = helpers_random(m, n)
        g = hsd(A, b, c)
        k = "r%dx%d_" % (m, n)
        out.update({k + "A": A, k + "b": b, k + "c": c, k + "x": g["x"], k + "y": g["y"], k + "pobj": g["pobj"],
                    k + "dobj": g["dobj"], k + "status": g["status"]})
        print("helpers random (%d,%d): status" % (m, n), g["status"], "pobj", g["pobj"], "gap", g["pobj"] - g["dobj"])
    np.savez(os.path.join(OUT, "random_helpers.npz"), **out)


def baseline_config(m, n, nobj=4096, nfull=256):
    """SURVEY 8d generator; inputs are regenerated from the seed by the tests (a checksum pins them)."""
    A, b, c = problems.random_dense_arrays(m, n, nobj, seed=0)
    g = hsd(A, b, c)
    gap = np.abs(g["pobj"] - g["dobj"]) / np.maximum(1.0, np.abs(g["pobj"]))
    np.savez_compressed(os.path.join(OUT, "config_%dx%d.npz" % (m, n)), m=m, n=n, seed=0, nobj=nobj,
                        input_checksum=np.array([A.sum(), b.sum(), c.sum()]),
                        pobj=g["pobj"], dobj=g["dobj"], status=g["status"].astype(np.int8),
                        x=g["x"][:nfull], y=g["y"][:nfull])
    print("config (%d,%d): %d LPs, status" % (m, n, nobj), np.bincount(g["status"]), "max internal gap", gap.max())


def newton_states():
    """(A, x, z, y, b, c, mu) -> dy by the known-answer formula of the reference's tests/test_ldl.py:196-216,
    on (i) the input recipe of tests/test_ldl.py:226-238 (seed 123456) at kernel-sized shapes and (ii) states
    harvested along an IPM trajectory (late iterations are ill conditioned: x/z spans many decades)."""
    out = {}
    for (m, n, nb) in ((16, 32, 32), (32, 64, 32), (20, 30, 32)):
        np.random.seed(123456)
        A = np.c_[np.random.rand(m, n), np.eye(m)]
        x = np.random.rand(m + n, nb).T.copy(); z = np.random.rand(m + n, nb).T.copy()
        y = np.random.rand(m, nb).T.copy(); b = np.random.rand(m, nb).T.copy()
        c = np.r_[np.random.rand(n, nb), np.zeros((m, nb))].T.copy()
        dy = np.stack([port.newton_step_known_answer(A, x[i], z[i], y[i], b[i], c[i], 1.0) for i in range(nb)])
        k = "t%dx%d_" % (m, n)
        out.update({k + "A": A, k + "x": x, k + "z": z, k + "y": y, k + "b": b, k + "c": c, k + "mu": 1.0, k + "dy": dy})
    # trajectory states of config 3 LP 0..3 at iterations 5, 12, 18, 21 (numpy path following, delta/r of the CL kernel)
    m, n = 32, 64
    A0, b0, c0 = problems.random_dense_arrays(m, n, 4, seed=0)
    A, b, c = problems.equality_arrays(A0, b0, c0)
    N = n + m
    xs, zs, ys, bs, cs, mus, dys = [], [], [], [], [], [], []
    for p in range(4):
        x = np.ones(N); z = np.ones(N); y = np.ones(m)
        for it in range(22):
            gamma = z @ x
            mu = 0.02 * gamma / (N + m)
            dy = port.newton_step_known_answer(A, x, z, y, b[p], c[p], mu)
            if it in (5, 12, 18, 21):
                xs.append(x.copy()); zs.append(z.copy()); ys.append(y.copy()); bs.append(b[p]); cs.append(c[p])
                mus.append(mu); dys.append(dy)
            d = x / z
            dx = (c[p] - A.T @ y + mu / x - A.T @ dy) * d
            dz = (mu - z * dx) / x - z
            th = max(0.0, np.max(-dx / x), np.max(-dz / z))
            th = min(0.9 / th, 1.0)
            x += th * dx; z += th * dz; y += th * dy
    out.update(traj_A=A, traj_x=np.array(xs), traj_z=np.array(zs), traj_y=np.array(ys), traj_b=np.array(bs),
               traj_c=np.array(cs), traj_mu=np.array(mus), traj_dy=np.array(dys))
    print("newton states: cond-ish spread x/z", [float(np.log10((a / b_).max() / (a / b_).min())) for a, b_ in zip(xs[:4], zs[:4])])
    np.savez_compressed(os.path.join(OUT, "newton_states.npz"), **out)


def baseline_sparse(m=128, n=256, density=0.025, nlp=64):
    """BASELINE config 5 shape: shared sparse A (stored in the fixture as CSR), HSD objectives for the first LPs."""
    A, b, c = problems.random_sparse_arrays(m, n, nlp, density=density, seed=0)
    g = hsd(np.asarray(A.todense()), b, c)
    np.savez_compressed(os.path.join(OUT, "config_sparse_%dx%d.npz" % (m, n)), m=m, n=n, density=density, seed=0,
                        A_data=A.data, A_indices=A.indices, A_indptr=A.indptr, b=b, c=c,
                        pobj=g["pobj"], dobj=g["dobj"], status=g["status"].astype(np.int8), x=g["x"][:16], y=g["y"][:16])
    gap = np.abs(g["pobj"] - g["dobj"]) / np.maximum(1.0, np.abs(g["pobj"]))
    print("sparse config (%d,%d,%.3f): nnz %d, status" % (m, n, density, A.nnz), np.bincount(g["status"]), "max gap", gap.max())


def dense_image_config(m=100, n=80, nlp=64):
    """A dense StandardLP beyond the lane-group kernels (the shape of the reference's own kernel test is m = 100,
    tests/test_ldl.py:226-238): SURVEY 8d generator, objectives of the reference solver for the first LPs (inputs are
    regenerated from the seed; a checksum pins them)."""
    A, b, c = problems.random_dense_arrays(m, n, nlp, seed=0)
    g = hsd(A, b, c)
    np.savez_compressed(os.path.join(OUT, "config_dense_%dx%d.npz" % (m, n)), m=m, n=n, seed=0, nobj=nlp,
                        input_checksum=np.array([A.sum(), b.sum(), c.sum()]),
                        pobj=g["pobj"], dobj=g["dobj"], status=g["status"].astype(np.int8), x=g["x"][:8], y=g["y"][:8])
    gap = np.abs(g["pobj"] - g["dobj"]) / np.maximum(1.0, np.abs(g["pobj"]))
    print("dense config (%d,%d): %d LPs, status" % (m, n, nlp), np.bincount(g["status"]), "max internal gap", gap.max())


def per_problem_a_config(m=128, n=256, density=0.025, nlp=32, batch=16384):
    """Per-problem values of A on config 5's structure.  The reference's LP classes refuse such a batch
    (lp.py:335-336) but its CPU solver takes ONE matrix per call: every LP is handed to hsd.c with ITS OWN matrix, which
    pins the extension to the reference LP by LP.  The LPs are the FIRST `nlp` of bench.py's `perA` workload (`batch` LPs:
    problems.random_sparse_arrays(seed 0) draws c after all of b, so the batch size is part of the recipe; values:
    problems.per_problem_values, seed 7); a checksum pins them."""
    import scipy.sparse as sp
    A, b, c = problems.random_sparse_arrays(m, n, batch, density=density, seed=0)
    b, c = b[:nlp], c[:nlp]
    rows, cols, data = problems.per_problem_values(A, nlp, seed=7)
    po, du, st = [], [], []
    for k in range(nlp):
        Ak = sp.csr_matrix((data[k], (rows, cols)), shape=(m, n))
        g = hsd_ref.solve_standard(Ak, b[k:k + 1], c[k:k + 1])
        po.append(g["pobj"][0]); du.append(g["dobj"][0]); st.append(g["status"][0])
    po, du = np.array(po), np.array(du)
    np.savez_compressed(os.path.join(OUT, "config_perA_%dx%d.npz" % (m, n)), m=m, n=n, density=density, seed=0, value_seed=7,
                        nobj=nlp, batch=batch, input_checksum=np.array([data.sum(), b.sum(), c.sum()]),
                        pobj=po, dobj=du, status=np.array(st, dtype=np.int8))
    print("per-problem-A config (%d,%d): %d LPs, status" % (m, n, nlp), np.bincount(st),
          "max gap", (np.abs(po - du) / np.maximum(1.0, np.abs(po))).max())


def status_cases(nshape=8, nlp=32):
    """Mixed-sign random LPs, most of them infeasible or unbounded: inputs, the status the reference's HSD solver
    reports (ipo/hsd.c:156-177: 0 optimal, 2 primal infeasible, 4 dual infeasible) and, as an independent arbiter,
    the verdict of scipy's HiGHS (0 optimal, 2 infeasible, 4 unbounded).  hsd.c decides between 2 and 4 from the sign of
    b'y alone once mu < 1e-12, which mislabels some unbounded LPs as primal infeasible -- HiGHS settles those."""
    from scipy.optimize import linprog
    rs = np.random.RandomState(20241004)
    out = dict(nshape=nshape)
    tot = {}
    for k in range(nshape):
        m = int(rs.randint(3, 33)); n = int(rs.randint(3, 64))
        A = rs.rand(m, n) * 2 - (1.0, 0.3, 0.05)[k % 3]
        A[rs.rand(m, n) < 0.5 * rs.rand()] = 0.0
        b = rs.rand(nlp, m) * 2 - (0.7, 0.2, 0.2)[k % 3]
        c = rs.rand(nlp, n) * 2 - (0.7, 0.3, 0.3)[k % 3]
        g = hsd(A, b, c)
        hi = np.array([{0: 0, 2: 2, 3: 4}.get(linprog(-c[i], A_ub=A, b_ub=b[i], bounds=(0, None), method="highs").status, -1)
                       for i in range(nlp)], dtype=np.int8)
        out.update({"A%d" % k: A, "b%d" % k: b, "c%d" % k: c, "status%d" % k: g["status"].astype(np.int8),
                    "pobj%d" % k: g["pobj"], "dobj%d" % k: g["dobj"], "highs%d" % k: hi})
        for a, h in zip(g["status"], hi):
            tot[(int(a), int(h))] = tot.get((int(a), int(h)), 0) + 1
    np.savez_compressed(os.path.join(OUT, "status_cases.npz"), **out)
    print("status cases (reference hsd.c status, HiGHS verdict): count", tot)


if __name__ == "__main__":
    if not hsd_ref.available():
        sys.exit("oracle/_ref/libhsd_ref.so missing: run `make -C oracle` in the container that has /root/reference")
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:          # only the named fixtures, e.g. `gen_golden.py dense_image_config per_problem_a_config`
        for name in sys.argv[1:]:
            if name == "large_lp_config":
                dense_image_config(200, 200, 16)     # a shape only the large-LP kernel (csrc/ipm_big.hip) covers
            else:
                globals()[name]()
        sys.exit(0)
    textbook()
    small_problem()
    random_helpers()
    newton_states()
    baseline_config(16, 32)
    baseline_config(32, 64)
    baseline_sparse()
    dense_image_config()
    dense_image_config(200, 200, 16)
    per_problem_a_config()
    status_cases()
