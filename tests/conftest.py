import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle_port():
    from oracle import port
    port.build()
    return port


def rel_err(a, ref):
    return np.abs(np.asarray(a) - np.asarray(ref)) / np.maximum(1.0, np.abs(np.asarray(ref)))


def status_cases():
    """(A, b, c, reference hsd.c status, HiGHS verdict, reference pobj) per shape of tests/golden/status_cases.npz."""
    g = golden("status_cases.npz")
    return [(g["A%d" % k], g["b%d" % k], g["c%d" % k], g["status%d" % k].astype(np.int32),
             g["highs%d" % k].astype(np.int32), g["pobj%d" % k]) for k in range(int(g["nshape"]))]


def check_certificates(Ae, be, ce, r, tol=1e-6):
    """status 4: x >= 0 is a ray of the feasible set (A x ~ 0) along which c'x > 0; status 2: (y, z >= 0) with
    A'y - z ~ 0 and b'y < 0 -- the Farkas certificates, checked from the returned vectors alone."""
    for i in np.where(r["status"] == 4)[0]:
        x = r["x"][i]; cx = ce[i] @ x
        assert cx > 0 and x.min() >= 0 and np.linalg.norm(Ae @ x) <= tol * cx * (1 + np.linalg.norm(be[i])), i
    for i in np.where(r["status"] == 2)[0]:
        y, z = r["y"][i], r["z"][i]; by = be[i] @ y
        assert by < 0 and z.min() >= 0 and np.linalg.norm(Ae.T @ y - z) <= tol * -by * (1 + np.linalg.norm(ce[i])), i
