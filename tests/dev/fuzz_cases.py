"""Case generator of tests/dev/fuzz_r3.py (large-LP kernel sweep), importable so that a case the sweep found can be replayed by
index in tests/test_hip_parity.py without storing its arrays."""
import numpy as np

from pycllp_amd import problems

EDGE = [(129, 1), (256, 5), (256, 1024), (130, 1150), (255, 257), (1, 600), (17, 1263), (200, 200), (144, 16)]


def big_cases(seed, count):
    """Yields (m, n, B, dense, hsd, pc, A, b, c) exactly as the sweep draws them from RandomState(seed)."""
    rs = np.random.RandomState(seed)
    for t in range(count):
        if t < len(EDGE):
            m, n = EDGE[t]
        else:
            m = int(rs.randint(1, 257)); n = int(rs.randint(1, 1281 - m))
            if m <= 128 and m + n <= 512:
                n = 513 - m + int(rs.randint(0, 700 - (513 - m) + 1)) if 513 - m < 700 else n
        B = int(rs.choice([1, 2, 5]))
        dense = bool(rs.rand() < 0.4) or n < 8
        hsd = bool(rs.rand() < 0.4)
        pc = (not hsd) and bool(rs.rand() < 0.4)
        if dense:
            A = rs.rand(m, n) * (rs.rand(m, n) < rs.choice([1.0, 0.7]))
            A[:, A.sum(0) == 0] = 0.5
            b = 0.5 + rs.rand(B, m); c = 0.5 + rs.rand(B, n)
        else:
            dens = float(rs.choice([0.01, 0.03, 0.1]))
            A, b, c = problems.random_sparse_arrays(m, n, B, density=min(1.0, max(dens, 3.0 / n)), seed=int(rs.randint(1 << 30)))
        yield m, n, B, dense, hsd, pc, A, b, c
    # (the per-problem-A and predictor-corrector parts of the sweep continue on the same stream: see fuzz_r3.py)
    yield rs
