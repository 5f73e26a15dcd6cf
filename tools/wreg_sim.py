"""Lane-level numpy model of the register-resident LDL' / triangular solves of csrc/ipm_wreg.hip.

Design check, not product code: every "register" is a numpy array of 64 lane values and the cross-lane
primitives (v_mfma_f64_16x16x4_f64, DPP row_newbcast, the quad reductions) are modelled with the lane maps the
kernel relies on.  Running it verifies the index algebra of the kernel against numpy.linalg -- on the CPU, where
there is no GPU to try things on.

Layouts (l = lane, q = l >> 4, c = l & 15):
  acc layout of a 16x16 block X : reg r holds X[4r + q][c]                       (MFMA C/D operand)
  A operand of step s           : lane holds A[m = c][k = 4s + q]
  B operand of step s           : lane holds B[k = 4s + q][n = c]
  => the acc registers of X are, unchanged, the B operand of X and the A operand of X' (transpose).
The factor is kept as U = L' in 16x16 blocks U[K][I] (K <= I): acc layout of L[I-rows][K-cols]'.
"""
import numpy as np

LANES = np.arange(64)
Q, C = LANES >> 4, LANES & 15


def mfma(a, b, acc):
    """acc[r][l] += sum_k A[4r+q][k] B[k][c] with A[m][k] = a[lane(c=m, q=k)], B[k][n] = b[lane(c=n, q=k)]."""
    A = np.zeros((16, 4)); Bm = np.zeros((4, 16))
    A[C, Q] = a; Bm[Q, C] = b
    P = A @ Bm
    return [acc[r] + P[4 * r + Q, C] for r in range(4)]


def row_bcast(v, k):
    """DPP row_newbcast:k -- every lane reads lane k of its own 16-lane row."""
    return v[(LANES & ~15) | k]


def quad_sum(v):
    """sum over the 4 quads for each c (xor 16, xor 32): result replicated."""
    v = v + v[LANES ^ 16]
    return v + v[LANES ^ 32]


def row_sum(v):
    """sum over the 16 lanes of each DPP row: result replicated inside the row."""
    out = np.zeros(64)
    for q in range(4):
        out[16 * q:16 * q + 16] = v[16 * q:16 * q + 16].sum()
    return out


def to_acc(X):
    return [X[4 * r + Q, C].copy() for r in range(4)]


def from_acc(regs):
    X = np.zeros((16, 16))
    for r in range(4):
        X[4 * r + Q, C] = regs[r]
    return X


def factor(M, floor=0.0):
    """Blocked LDL' of M (n = 16*MB).  Returns (U blocks with W_K = L_KK^-1 operand regs on the diagonal, rD)."""
    n = M.shape[0]; MB = n // 16
    # U[K][I] raw = M[I-rows][K-cols]' = M[K-rows][I-cols] by symmetry
    U = {(K, I): to_acc(M[16 * K:16 * K + 16, 16 * I:16 * I + 16]) for K in range(MB) for I in range(K, MB)}
    rD_all = np.zeros(n)
    for K in range(MB):
        # ---- diagonal block: acc layout -> tile -> lane = row (every DPP row a redundant copy) ----
        T = from_acc(U[(K, K)])
        Wd = [np.where(k <= C, T[C, k], 0.0) for k in range(16)]      # Wd[k][lane] = T[row=c][col=k], lower triangle
        Ld = [None] * 16
        rDr = [np.zeros(64) for _ in range(4)]
        for j in range(16):
            u = Wd[j]
            piv = row_bcast(u, j)
            aD = np.maximum(np.abs(piv), floor)
            rD = 1.0 / aD
            below = C > j
            li = np.where(below, u * rD, 0.0)
            for k in range(j + 1, 16):
                Wd[k] = Wd[k] - li * row_bcast(u, k)
            Ld[j] = li
            rDr[j >> 2] = np.where(Q == (j & 3), rD, rDr[j >> 2])
            rD_all[16 * K + j] = rD[0]
        # ---- W = L_KK^-1 in A-operand layout: Ws[s][lane] = W[row = c][col = 4s + q] ----
        Ws = [np.where(C == 4 * s + Q, 1.0, 0.0) for s in range(4)]
        for j in range(15):
            for s in range(4):
                if 4 * s <= j:
                    Ws[s] = Ws[s] - Ld[j] * row_bcast(Ws[s], j)
        # ---- panel: Y = W raw (MFMA), U = rD Y ----
        Yn = {}
        for I in range(K + 1, MB):
            acc = [np.zeros(64) for _ in range(4)]
            for s in range(4):
                acc = mfma(Ws[s], U[(K, I)][s], acc)
            Yn[I] = [-acc[r] for r in range(4)]
            U[(K, I)] = [acc[r] * rDr[r] for r in range(4)]
        # ---- trailing update: U[J][I] -= Y_KJ' U_KI ----
        for J in range(K + 1, MB):
            for I in range(J, MB):
                acc = U[(J, I)]
                for s in range(4):
                    acc = mfma(Yn[J][s], U[(K, I)][s], acc)
                U[(J, I)] = acc
        U[(K, K)] = Ws
    return U, rD_all


def solve(U, rD_all, rhs):
    """(L D L')^-1 rhs with the blocks of factor(); vectors pass through a small 'LDS' array as in the kernel."""
    n = rhs.shape[0]; MB = n // 16
    um = rhs.copy()
    tRL = {}
    for I in range(MB):
        p = np.zeros(64)
        for K in range(I):
            for r in range(4):
                p = p + U[(K, I)][r] * tRL[K][r]
        sCL = um[16 * I + C]
        rCLv = sCL - quad_sum(p)
        rr = np.zeros(16); rr[C] = rCLv                        # LDS round trip: CL -> RL
        rRL = [rr[4 * s + Q] for s in range(4)]
        pt = np.zeros(64)
        for s in range(4):
            pt = pt + U[(I, I)][s] * rRL[s]
        tCL = quad_sum(pt)
        um[16 * I + C] = tCL
        tRL[I] = [um[16 * I + 4 * r + Q] for r in range(4)]
    xCL = {}
    for K in range(MB - 1, -1, -1):
        pr = [np.zeros(64) for _ in range(4)]
        for I in range(K + 1, MB):
            for r in range(4):
                pr[r] = pr[r] + U[(K, I)][r] * xCL[I]
        rRL = [um[16 * K + 4 * r + Q] * rD_all[16 * K + 4 * r + Q] - row_sum(pr[r]) for r in range(4)]
        rr = np.zeros(16)
        for r in range(4):
            rr[4 * r + Q] = rRL[r]                              # LDS round trip: RL -> CL
        rCLv = rr[C]
        xRL = [row_sum(U[(K, K)][s] * rCLv) for s in range(4)]  # x[4s+q] = sum_j W[j][4s+q] r[j]
        for s in range(4):
            um[16 * K + 4 * s + Q] = xRL[s]
        xCL[K] = um[16 * K + C]
    return um


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for n in (16, 32, 64, 128):
        A = rs.rand(n, 3 * n)
        M = (A * rs.rand(3 * n)) @ A.T
        U, rD = factor(M)
        # reconstruct L from U (off-diagonal) and check against a reference LDL'
        Lc = np.linalg.cholesky(M); Dref = np.diag(Lc) ** 2; Lref = Lc / np.diag(Lc)
        L = np.eye(n)
        for (K, I), regs in U.items():
            if I > K:
                L[16 * I:16 * I + 16, 16 * K:16 * K + 16] = from_acc(regs).T
        for K in range(n // 16):
            Wk = np.zeros((16, 16))
            for s in range(4):
                Wk[C, 4 * s + Q] = regs_ = U[(K, K)][s]
            L[16 * K:16 * K + 16, 16 * K:16 * K + 16] = np.linalg.inv(Wk)
        rhs = rs.rand(n)
        x = solve(U, rD, rhs)
        print("n=%3d  |1/rD - D| %.1e   |L - Lref| %.1e   solve err %.1e" % (
            n, np.abs(1 / rD - Dref).max() / Dref.max(), np.abs(L - Lref).max(),
            np.abs(x - np.linalg.solve(M, rhs)).max() / np.abs(x).max()))
