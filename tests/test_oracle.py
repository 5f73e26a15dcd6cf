"""CPU tests of the ORACLE (oracle/ipm_dense_ref.c) against the committed golden vectors.

Goldens were produced by the reference's own CPU solver (pycllp/ipo.py -> ipo/hsd.c) -- see
tools/gen_golden.py.  Tolerances: objectives 1e-8 relative (BASELINE.json north_star); x as the reference's
own tests (tests/test_vanderbei.py:42 rtol=atol=1e-6; tests/helpers.py:97-102 rtol=atol=1e-3).
"""
import numpy as np
import pytest

from conftest import golden, rel_err, status_cases, check_certificates
from pycllp_amd import problems

OBJ_TOL = 1e-8


def eq(A, b, c):
    return problems.equality_arrays(np.asarray(A), np.atleast_2d(b), np.atleast_2d(c))


def test_vanderbei_2_9(oracle_port):
    g = golden("vanderbei.npz")
    Ae, b, ce = eq(g["v29_A"], g["v29_b"], g["v29_c"])
    r = oracle_port.dense_solve(Ae, b, ce)
    assert r["status"][0] == 0
    np.testing.assert_allclose(r["x"][0, :3], g["v29_xopt"], rtol=1e-6, atol=1e-6)   # tests/test_vanderbei.py:42
    assert rel_err(r["pobj"], g["v29_pobj"]).max() < OBJ_TOL
    assert rel_err(r["dobj"], g["v29_dobj"]).max() < OBJ_TOL
    assert abs(r["pobj"][0] - 10.5) < 1e-7


def test_vanderbei_2_10(oracle_port):
    g = golden("vanderbei.npz")   # already an EqualityLP (tests/vanderbei_problems.py:22-36)
    r = oracle_port.dense_solve(g["v210_A"], g["v210_b"], g["v210_c"])
    assert r["status"][0] == 0
    np.testing.assert_allclose(r["x"][0], g["v210_xopt"], rtol=1e-6, atol=1e-6)
    assert abs(r["pobj"][0] - 9.0) < 1e-7 and abs(r["dobj"][0] - 9.0) < 1e-7


def test_small_problem_and_32_perturbations(oracle_port):
    g = golden("small_problem.npz")
    Ae, b, ce = eq(g["A"], g["b"], g["c"])
    r = oracle_port.dense_solve(Ae, b, ce)
    assert r["status"][0] == 0
    np.testing.assert_allclose(r["x"][0, :3], (1.00997e-13, 1.22527e-12, 5.18790e+00), rtol=1e-1, atol=1e-1)  # test_simple.py:66-67
    Ae, bb, cc = eq(g["A"], g["bb"], g["cc"])
    r = oracle_port.dense_solve(Ae, bb, cc)
    assert (r["status"] == 0).all() and (g["status"] == 0).all()
    np.testing.assert_allclose(r["x"][:, :3], g["x"], rtol=1e-3, atol=1e-3)          # test_simple.py:92-93
    assert rel_err(r["pobj"], g["pobj"]).max() < OBJ_TOL
    assert rel_err(r["dobj"], g["dobj"]).max() < OBJ_TOL


@pytest.mark.parametrize("shape", ["10x10", "20x20"])
def test_helpers_random(oracle_port, shape):
    g = golden("random_helpers.npz")
    k = "r%s_" % shape
    Ae, b, ce = eq(g[k + "A"], g[k + "b"], g[k + "c"])
    r = oracle_port.dense_solve(Ae, b, ce)
    assert (r["status"] == 0).all()
    n = g[k + "A"].shape[1]
    np.testing.assert_allclose(r["x"][:, :n], g[k + "x"], rtol=1e-3, atol=1e-3)      # tests/helpers.py:97-102
    assert rel_err(r["pobj"], g[k + "pobj"]).max() < OBJ_TOL
    assert rel_err(r["dobj"], g[k + "dobj"]).max() < OBJ_TOL


@pytest.mark.parametrize("m,n", [(16, 32), (32, 64)])
def test_baseline_configs_objective_parity(oracle_port, m, n):
    g = golden("config_%dx%d.npz" % (m, n))
    nobj = int(g["nobj"])
    A, b, c = problems.random_dense_arrays(m, n, nobj, seed=int(g["seed"]))
    np.testing.assert_allclose([A.sum(), b.sum(), c.sum()], g["input_checksum"], rtol=0, atol=0)
    Ae, be, ce = problems.equality_arrays(A, b, c)
    r = oracle_port.dense_solve(Ae, be, ce, nthreads=8)
    assert (r["status"] == 0).all() and (g["status"] == 0).all()
    assert rel_err(r["pobj"], g["pobj"]).max() < OBJ_TOL
    assert rel_err(r["dobj"], g["dobj"]).max() < OBJ_TOL
    nf = g["x"].shape[0]
    np.testing.assert_allclose(r["x"][:nf, :n], g["x"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["y"][:nf], g["y"], rtol=1e-5, atol=1e-6)
    assert r["iters"].max() < 60


@pytest.mark.parametrize("key", ["t16x32_", "t32x64_", "t20x30_"])
def test_newton_step_known_answer(oracle_port, key):
    g = golden("newton_states.npz")
    A = g[key + "A"]
    for i in range(g[key + "x"].shape[0]):
        dy = oracle_port.solve_primal_normal(A, g[key + "x"][i], g[key + "z"][i], g[key + "y"][i], g[key + "b"][i],
                                             g[key + "c"][i], float(g[key + "mu"]), pivot_floor=0.0)
        np.testing.assert_allclose(dy, g[key + "dy"][i], rtol=1e-7, atol=1e-9)       # tests/test_ldl.py:216


# Late-iteration states: t = c - A'y + mu/x carries ~1e-16 absolute rounding that x/z (up to 5e11 at iteration
# 21) amplifies into the right-hand side, so two correct implementations that merely sum A'y in a different
# order differ by ~eps*max(x/z) in dy.  Tolerance per harvested iteration (5, 12, 18, 21), relative to max|dy|.
TRAJ_RTOL = (1e-10, 1e-8, 1e-6, 1e-4)


def test_newton_step_trajectory_states(oracle_port):
    g = golden("newton_states.npz")
    A = g["traj_A"]
    for i in range(g["traj_x"].shape[0]):
        x, z, y, b, c, mu = g["traj_x"][i], g["traj_z"][i], g["traj_y"][i], g["traj_b"][i], g["traj_c"][i], float(g["traj_mu"][i])
        dy = oracle_port.solve_primal_normal(A, x, z, y, b, c, mu)
        ref = g["traj_dy"][i]
        assert np.abs(dy - ref).max() <= TRAJ_RTOL[i % 4] * np.abs(ref).max()


def test_ldl_against_numpy_cholesky(oracle_port):
    rs = np.random.RandomState(3)
    for n in (5, 16, 32, 60):
        X = rs.rand(n, n + 7)
        S = X @ X.T + n * np.eye(n)
        chol = np.linalg.cholesky(S)
        L, D = oracle_port.ldl(S)
        np.testing.assert_allclose(L * np.sqrt(D), chol, rtol=1e-10, atol=1e-12)       # tests/test_ldl.py:111-116
        L2, D2 = oracle_port.ldl(S, modified=True)
        np.testing.assert_allclose(L2 * np.sqrt(D2), chol, rtol=1e-10, atol=1e-12)     # tests/test_ldl.py:53-63
        assert np.allclose(np.diag(L), 1.0)


def test_modified_ldl_guards_indefinite(oracle_port):
    """Nocedal-Wright guard: D stays >= delta and finite on a semi-definite matrix (tests/test_ldl.py:79-87)."""
    rs = np.random.RandomState(4)
    X = rs.rand(12, 5)
    S = X @ X.T                      # rank 5 < 12
    L, D = oracle_port.ldl(S, modified=True, delta=1e-6)
    assert np.isfinite(L).all() and (D >= 1e-6).all()


def test_status_codes_infeasible_unbounded(oracle_port):
    # primal infeasible: x1 + x2 = -1 with x >= 0
    r = oracle_port.dense_solve(np.array([[1.0, 1.0]]), np.array([[-1.0]]), np.array([[1.0, 1.0]]))
    assert r["status"][0] in (2, 3, 4, 5) and r["status"][0] != 0
    # unbounded: max x1 s.t. x1 - x2 = 0
    r = oracle_port.dense_solve(np.array([[1.0, -1.0]]), np.array([[0.0]]), np.array([[1.0, 0.0]]))
    assert r["status"][0] != 0


def test_hsd_embedding_statuses_against_reference_and_highs(oracle_port):
    """flags=32 (homogeneous self-dual embedding) on the mixed-sign fixture: the status must be the true
    one (HiGHS verdict stored in the fixture) for every LP, equal to the reference hsd.c status wherever that one is
    right, objectives of the optimal LPs within 1e-8 of the reference, certificates valid."""
    agree = total = 0
    for A, b, c, ref_status, highs, ref_pobj in status_cases():
        Ae, be, ce = problems.equality_arrays(A, b, c)
        r = oracle_port.dense_solve(Ae, be, ce, nthreads=4, flags=32)
        np.testing.assert_array_equal(r["status"], highs)
        right = ref_status == highs
        np.testing.assert_array_equal(r["status"][right], ref_status[right])
        opt = highs == 0
        assert rel_err(r["pobj"][opt], ref_pobj[opt]).max() < OBJ_TOL if opt.any() else True
        assert r["iters"].max() < 60
        check_certificates(Ae, be, ce, r)
        agree += int(right.sum()); total += len(highs)
    assert agree >= total - 8      # the reference mislabels a handful of unbounded LPs as infeasible (fixture docstring)


@pytest.mark.parametrize("m,n", [(16, 32), (32, 64)])
def test_hsd_embedding_objective_parity_on_baseline_configs(oracle_port, m, n):
    g = golden("config_%dx%d.npz" % (m, n))
    A, b, c = problems.random_dense_arrays(m, n, int(g["nobj"]), seed=0)
    Ae, be, ce = problems.equality_arrays(A, b[:512], c[:512])
    r = oracle_port.dense_solve(Ae, be, ce, nthreads=4, flags=32)
    assert (r["status"] == 0).all()
    assert rel_err(r["pobj"], g["pobj"][:512]).max() < OBJ_TOL
    assert rel_err(r["dobj"], g["dobj"][:512]).max() < OBJ_TOL


def test_sparse_config_objective_parity(oracle_port):
    """BASELINE config 5 shape (m=128, n=256, density 0.025): the oracle on the densified equality form."""
    import scipy.sparse as sp
    g = golden("config_sparse_128x256.npz")
    A = sp.csr_matrix((g["A_data"], g["A_indices"], g["A_indptr"]), shape=(int(g["m"]), int(g["n"])))
    A2, b2, c2 = problems.random_sparse_arrays(int(g["m"]), int(g["n"]), g["b"].shape[0], density=float(g["density"]), seed=0)
    assert (A != A2).nnz == 0 and np.array_equal(b2, g["b"])          # the generator reproduces the fixture
    k = 16
    Ae, be, ce = problems.equality_arrays(np.asarray(A.todense()), g["b"][:k], g["c"][:k])
    r = oracle_port.dense_solve(Ae, be, ce, nthreads=8)
    assert (r["status"] == 0).all() and (g["status"] == 0).all()
    assert rel_err(r["pobj"], g["pobj"][:k]).max() < OBJ_TOL and rel_err(r["dobj"], g["dobj"][:k]).max() < OBJ_TOL
    np.testing.assert_allclose(r["x"][:, :int(g["n"])], g["x"][:k], rtol=1e-5, atol=1e-6)


def test_live_reference_solver_agrees_with_goldens():
    """Where oracle/_ref/libhsd_ref.so is present (it is built from /root/reference in this container and
    travels to the GPU box), the goldens must be reproducible from it."""
    from oracle import hsd_ref
    if not hsd_ref.available():
        pytest.skip("oracle/_ref not built")
    g = golden("config_16x32.npz")
    A, b, c = problems.random_dense_arrays(16, 32, int(g["nobj"]), seed=0)   # the stream depends on B: slice after
    r = hsd_ref.solve_standard(A, b[:64], c[:64])
    np.testing.assert_allclose(r["pobj"], g["pobj"][:64], rtol=1e-12)
    np.testing.assert_allclose(r["dobj"], g["dobj"][:64], rtol=1e-12)
