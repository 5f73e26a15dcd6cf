// ipm_big.hip -- LPs beyond the register-resident kernels: one LP per WORKGROUP (four wavefronts), 128 < m <= 256 rows
// and/or 512 < n <= 1280 columns (equality form), the m x m normal-equations matrix and its LDL' factor as 16 x 16 blocks
// in LDS when they fit (m <= ~144) and in an L2-resident workspace otherwise.
//
// Why: the reference's hosts take any (m, n) (pycllp/solvers/cl.py:28-83, 127-278; examples/random_problem.py:30-49 is run
// with arbitrary sizes); the lane-group kernels stop at m = 32, the wavefront-per-LP kernel (ipm_wreg.hip) at m = 128 --
// its factor fills the register file of a SIMD.  Beyond that one LP gets a whole compute unit:
//   * M = A diag(x/z) A' and its factor are kept as the upper block triangle of U = L' in 16 x 16 blocks stored in the
//     ACCUMULATOR layout of v_mfma_f64_16x16x4_f64 (element [4r + l/16][l%16] of a block = double r*64 + l): a block
//     is loaded with four coalesced 512-byte reads straight into the registers an MFMA wants -- that layout is at once the
//     B operand of the block and the A operand of its transpose (the algebra of ipm_wreg.hip, the blocks in memory instead
//     of registers), so panel (Y_KI = L_KK^-1 M_KI) and trailing update (M_JI -= Y_KJ' D_K^-1 Y_KI) are MFMAs on loaded
//     blocks with no layout conversion; the blocks of a stage are dealt round-robin to the four waves, three workgroup
//     barriers per block column;
//   * the diagonal block of a stage is factored by wave 0 in "lane = row" form with the fused 64-bit DPP FMA chain of the
//     other kernels (chain_asm.inc), its inverse W_K goes to LDS for the panel and for the triangular solves;
//   * the Gram matrix comes from a term list (sparse A: nnz-proportional, deterministic) or, for a dense A, from the
//     matrix cores: the dense columns as a k-major image [k/4][row][4] in global memory (L2-resident, shared by all
//     workgroups: every operand fetch of a wave is one contiguous 512-byte read), the identity columns of [A | I] as a
//     diagonal update;
//   * N-vectors in registers (thread = column, up to five per thread), m-vectors in LDS, mat-vecs on CSR / CSC copies of A
//     read through L2, block substitution on wave 0.
// Semantics = oracle/ipm_dense_ref.c (ipm_one_path / hsd_one_raw / newton_dy), like every other kernel of this library:
// reference kernels replaced as in ipm_block.inc (pycllp/cl/primal_normal.cl:201-375, pycllp/cl/ldl.cl:314-712).
// The Nocedal-Wright guard (ldl.cl:368) is recorded, not applied, by the blocked factorisation; when it would have bitten
// (rare: collapsing iterates of the embedding) M is re-formed and a column-by-column cold path applies it exactly.
#include "big.h"
#ifndef PYCLLP_WINV_FUSED
#define PYCLLP_WINV_FUSED 1     // see ipm_wreg.hip
#endif

namespace {

#ifndef BIG_PAIRS_PER_TRIP
#define BIG_PAIRS_PER_TRIP 2             // block pairs of the trailing update whose reads are in flight together (per wave)
#endif
constexpr int BT = 256;                  // threads per workgroup
constexpr int BNC_MAX = BIG_MAX_N / BT;  // N-vector registers per thread at the largest n (the kernel is instantiated for 2, 3 and 5)

struct BigTab {
    int m, n, nnz, MB, dense, m_in_lds, lds_bytes;
    int nd, ks, imgR, n_sl;              // dense mode: nd dense columns, ks = k-steps of 4 columns, image rows, identity columns behind them
    const double* csr_val; const int* csr_ptr; const int* csr_col;   // A by rows
    const double* csc_val; const int* csc_ptr; const int* csc_row;   // A by columns
    const double* img;                   // [ks][imgR][4]
    const double* img_rm;                // dense mode: the dense columns row-major [m][nd] (A'u: thread = column, coalesced)
    const double* img_cm;                // ... and column-major [nd][imgR] (A v: thread = row, coalesced)
    int n_ent;                           // term-list mode: entries (i, k), i >= k, of M with their terms a_ij a_kj, column j
    const int* ent_dst; const int* ent_dst2; const int* ent_ptr; const double* term_w; const int* term_col;
};

__host__ __device__ inline int bidx(int K, int I) { return I * (I + 1) / 2 + K; }     // block (K, I), K <= I, of U = L'
// offset of element [kl][il] inside a block
__host__ __device__ inline int boff(int kl, int il) { return (kl >> 2) * 64 + (kl & 3) * 16 + il; }

__device__ __forceinline__ double bsum(double v, double* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double bmax(double v, double* red, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// doubles of LDS the kernel carves (host and device agree through this one function)
__host__ __device__ inline size_t big_lds_doubles(int MB, int n, bool m_in_lds) {
    const size_t MP = 16 * (size_t)MB, NPv = ((size_t)n + 7) & ~(size_t)7;
    return (m_in_lds ? (size_t)MB * (MB + 1) / 2 * 256 : 0) + (size_t)MB * 256 + 2 * NPv + 9 * MP + 272 + 256 + 32;
}

// WGPC: workgroups per CU the instance is compiled for (2: 256 registers per lane; 3: 168, more spills, more overlap).
// BNC: N-vector registers per thread, n <= 256 BNC -- an LP with 400 columns carries two registers per N-vector, not the five
// the size cap needs (a dozen N-vectors are live across the Newton step: 60 fewer registers at BNC = 2).
template <int WGPC, int BNC>
__global__ void __launch_bounds__(BT, WGPC)
ipm_big_kernel(BigTab T, long B, const double* __restrict__ bg, const double* __restrict__ cg, double* __restrict__ xg,
               double* __restrict__ yg, double* __restrict__ zg, double* __restrict__ pobj, double* __restrict__ dobj,
               int* __restrict__ status, int* __restrict__ iters, int* __restrict__ queue, double* ws, double nwt_mu,
               double* __restrict__ nwt_dy, int* __restrict__ nwt_nref, DevOpts o) {
    // nwt_dy != nullptr: stand-alone Newton step (solve_primal_normal, ldl.cl:602-653 / 656-712): x, z, y are INPUTS, mu is
    // nwt_mu, one pass of the Newton machinery runs and dy (+ refinement passes used) is all that is stored
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c16 = lane & 15;
    const int m = T.m, n = T.n, MB = T.MB, MP = 16 * MB, nblk = MB * (MB + 1) / 2;
    const int NPv = (n + 7) & ~7;
    // ---- LDS carve-up (big_lds_doubles) ----
    double* p_ = lds;
    double* Mw = T.m_in_lds ? p_ : ws + (size_t)blockIdx.x * nblk * 256;      // the blocks (see header); generic pointer
    if (T.m_in_lds) p_ += (size_t)nblk * 256;
    double* wl = p_;  p_ += MB * 256;        // W_K = L_KK^-1, row-major 16 x 16
    double* vx = p_;  p_ += NPv;             // staging of an N-vector for the CSR products
    double* vd = p_;  p_ += NPv;             // d = x/z
    double* ys = p_;  p_ += MP;              // y
    double* bs = p_;  p_ += MP;              // b
    double* um = p_;  p_ += MP;              // solve vector in/out
    double* dyv = p_; p_ += MP;              // dy
    double* rdv = p_; p_ += MP;              // 1/D
    double* tdv = p_; p_ += MP;              // D^-1 t of the forward sweep
    double* pv = p_;  p_ += MP;              // HSD: p
    double* flr = p_; p_ += MP;              // HSD: per-column pivot floors
    double* qv = p_;  p_ += MP;              // HSD: right-hand side of q
    double* tile = p_; p_ += 272;            // diagonal block, stride 17
    double* wsA = p_; p_ += 256;             // W_K in the A-operand layout (four doubles per lane of the panel's MFMAs)
    double* red = p_;                        // [32] reduction scratch
    const double* csr_val = T.csr_val; const int* csr_ptr = T.csr_ptr; const int* csr_col = T.csr_col;
    const double* csc_val = T.csc_val; const int* csc_ptr = T.csc_ptr; const int* csc_row = T.csc_row;

    const bool nwt = nwt_dy != nullptr;
    const bool warm = nwt || (o.flags & PYCLLP_FLAG_WARM_START) != 0;
    const bool autoscale = !nwt && (o.flags & PYCLLP_FLAG_AUTOSCALE) != 0;
    const bool hsd = !nwt && (o.flags & PYCLLP_FLAG_HSD) != 0;
    const bool pc = !nwt && !hsd && (o.flags & PYCLLP_FLAG_PREDCORR) != 0;   // Mehrotra's predictor-corrector (oracle ipm_one_pc)
    const double eta = 1.0 - o.delta, einf = 100.0 * o.eps;
    const double nm = (double)(n + m);

    // A'u for this thread's columns, u in LDS.  Dense A: straight from the row-major image -- the threads of a wave read
    // consecutive columns of one row (one coalesced L2 read per row, u_i a broadcast), four rows in flight per trip; the identity
    // columns behind the dense ones pick their own u_i.  Sparse A: the CSC arrays.
    auto At_cols = [&](const double* u, double (&out)[BNC]) {
        if (T.dense) {
            const int nd = T.nd;
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                const int j = tid + BT * k;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                if (j < nd) {
                    const double* col = T.img_rm + j;
                    int i = 0;
                    for (; i + 4 <= m; i += 4) {
                        const double v0 = col[(size_t)i * nd], v1 = col[(size_t)(i + 1) * nd], v2 = col[(size_t)(i + 2) * nd], v3 = col[(size_t)(i + 3) * nd];
                        a0 = fma(v0, u[i], a0); a1 = fma(v1, u[i + 1], a1); a2 = fma(v2, u[i + 2], a2); a3 = fma(v3, u[i + 3], a3);
                    }
                    for (; i < m; i++) a0 = fma(col[(size_t)i * nd], u[i], a0);
                }
                out[k] = (j < nd) ? (a0 + a1) + (a2 + a3) : ((j < n) ? u[j - nd] : 0.0);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < BNC; k++) {
            const int j = tid + BT * k;
            double acc = 0.0;
            if (j < n) for (int p = csc_ptr[j]; p < csc_ptr[j + 1]; p++) acc = fma(csc_val[p], u[csc_row[p]], acc);
            out[k] = acc;
        }
    };
    // (A v)_i for row i = tid (< m), v in LDS; with dg also diag(A diag(d) A')_i (d in vd) from the same pass.  Dense A: the
    // column-major image (consecutive rows = consecutive threads: coalesced), v_j a broadcast, + the row's identity column.
    auto A_row_dg = [&](const double* v, bool dg, double& mdg) {
        double a0 = 0.0, a1 = 0.0, m0 = 0.0, m1 = 0.0;
        if (T.dense) {
            if (tid < m) {
                const int nd = T.nd, R = T.imgR;
                const double* row = T.img_cm + tid;
                int j = 0;
                for (; j + 2 <= nd; j += 2) {
                    const double v0 = row[(size_t)j * R], v1 = row[(size_t)(j + 1) * R];
                    a0 = fma(v0, v[j], a0); a1 = fma(v1, v[j + 1], a1);
                    if (dg) { m0 = fma(v0 * v0, vd[j], m0); m1 = fma(v1 * v1, vd[j + 1], m1); }
                }
                for (; j < nd; j++) { const double v0 = row[(size_t)j * R]; a0 = fma(v0, v[j], a0); if (dg) m0 = fma(v0 * v0, vd[j], m0); }
                if (T.n_sl) { a0 += v[nd + tid]; if (dg) m0 += vd[nd + tid]; }
            }
        } else if (tid < m) {
            for (int p = csr_ptr[tid]; p < csr_ptr[tid + 1]; p++) {
                const double a = csr_val[p]; const int j = csr_col[p];
                a0 = fma(a, v[j], a0);
                if (dg) m0 = fma(a * a, vd[j], m0);
            }
        }
        mdg = m0 + m1;
        return a0 + a1;
    };
    auto A_row = [&](const double* v) { double dm; return A_row_dg(v, false, dm); };

    // ---- M = A diag(d) A' (d in vd) into the blocks ----
    auto gram = [&]() {
        if (T.dense) {
            // Work unit: block row K x up to GA consecutive block columns I0 .. I0 + cnt - 1 (I0 >= K), units dealt round-robin
            // to the waves.  Per k-step ONE scaled operand of row K feeds cnt MFMAs (independent accumulators), and the
            // operands of step s + 1 are requested before the MFMAs of step s are issued: the first version (one block per
            // trip, two global reads per MFMA, nothing in flight) spent 46 % of the dense kernel's time here at 1/9 of the
            // matrix pipe's rate (profiles/r03/phase_shares_large_lp_kernel_v1.txt).
            constexpr int GA = 8;
            const size_t step = (size_t)T.imgR * 4;
            int unit = 0;
            for (int K = 0; K < MB; K++)
                for (int I0 = K; I0 < MB; I0 += GA, unit++) {
                    if ((unit & 3) != wave) continue;
                    const int cnt = (MB - I0 < GA) ? MB - I0 : GA;
                    double4_t acc[GA];
#pragma unroll
                    for (int u = 0; u < GA; u++) acc[u] = (double4_t){0.0, 0.0, 0.0, 0.0};
                    const double* pk = T.img + (size_t)(16 * K + c16) * 4 + q;
                    const double* pi = T.img + (size_t)(16 * I0 + c16) * 4 + q;       // block column I0 + u: + u * 64 doubles
                    double akn = pk[0], ain[GA];
#pragma unroll
                    for (int u = 0; u < GA; u++) ain[u] = (u < cnt) ? pi[u * 64] : 0.0;
                    for (int s4 = 0; s4 < T.ks; s4++) {
                        const double ak = akn * vd[4 * s4 + q];
                        double ai[GA];
#pragma unroll
                        for (int u = 0; u < GA; u++) ai[u] = ain[u];
                        if (s4 + 1 < T.ks) {
                            akn = pk[(size_t)(s4 + 1) * step];
#pragma unroll
                            for (int u = 0; u < GA; u++) if (u < cnt) ain[u] = pi[(size_t)(s4 + 1) * step + u * 64];
                        }
#pragma unroll
                        for (int u = 0; u < GA; u++)
                            if (u < cnt) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ak, ai[u], acc[u], 0, 0, 0);
                    }
                    if (I0 == K) {      // the diagonal block: identity columns of [A | I] and the padded rows (identity rows of M)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int gi = 16 * K + c16;
                            if (4 * r + q == c16) {
                                if (gi < m) { if (T.n_sl) acc[0][r] += vd[T.nd + gi]; }
                                else acc[0][r] = 1.0;
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < GA; u++)
                        if (u < cnt) {
                            double* blk = Mw + (size_t)bidx(K, I0 + u) * 256;
#pragma unroll
                            for (int r = 0; r < 4; r++) blk[r * 64 + lane] = acc[u][r];
                        }
                }
            __syncthreads();
            return;
        }
        for (int i = tid; i < nblk * 256; i += BT) Mw[i] = 0.0;
        __syncthreads();
        for (int e = tid; e < T.n_ent; e += BT) {
            double acc = 0.0;
            for (int t = T.ent_ptr[e]; t < T.ent_ptr[e + 1]; t++) acc = fma(T.term_w[t], vd[T.term_col[t]], acc);
            Mw[T.ent_dst[e]] = acc;
            const int d2 = T.ent_dst2[e];
            if (d2 >= 0) Mw[d2] = acc;
        }
        for (int i = m + tid; i < MP; i += BT) Mw[(size_t)bidx(i >> 4, i >> 4) * 256 + boff(i & 15, i & 15)] = 1.0;
        __syncthreads();
    };

    // ---- blocked LDL' of the blocks in place (see header).  RELF: the pivot floor of column j is flr[j].  Returns whether the
    //      Nocedal-Wright guard would have bitten anywhere (workgroup-uniform). ----
    auto factor = [&](double beta2, double floor_, bool relf) -> bool {
        double ymax = 0.0;       // running max of u^2 / D (pivot chains) and Y^2 / D (panels): the guard bites iff > beta^2
        for (int K = 0; K < MB; K++) {
            if (wave == 0) {
                const double* blk = Mw + (size_t)bidx(K, K) * 256;
#pragma unroll
                for (int r = 0; r < 4; r++) tile[(4 * r + q) * 17 + c16] = blk[r * 64 + lane];
                wave_lds_sync();
                double Wd[16], Ws[4];
                [[maybe_unused]] double Ld[16];
#pragma unroll
                for (int k = 0; k < 16; k++) Wd[k] = tile[c16 * 17 + k];
#pragma unroll
                for (int s = 0; s < 4; s++) Ws[s] = (c16 == 4 * s + q) ? 1.0 : 0.0;
                const double myf = relf ? flr[16 * K + c16] : floor_;
                double rdiag = 1.0, rD;
                {
                    const double piv = bcast64<0>(Wd[0]);
                    rD = fast_rcp(fmax(fabs(piv), row_bcast<0>(myf)));
                }
                static_for<0, 16>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const double u = Wd[j];
                    // lanes below the pivot / the pivot's lane of every DPP row as compile-time EXEC masks (wave_common.h)
                    constexpr unsigned m16 = ((0xFFFFu << (j + 1)) & 0xFFFFu) * 0x10001u, one16 = (1u << j) * 0x10001u;
                    double nli;
                    chain_head_exec<m16, one16>(u, rD, nli, ymax, rdiag);
                    if constexpr (j < 15) {
                        double aDn, rDn;
                        chain_step_pipe_relf<j>(Wd, u, nli, 0.0, myf, aDn, rDn);
                        rD = rDn;
                        if constexpr (PYCLLP_WINV_FUSED) winv_step<j>(Ws, nli); else Ld[j] = nli;    // see ipm_wreg.hip
                    }
                });
                if (q == 0) rdv[16 * K + c16] = rdiag;
                // W = L_KK^-1: Ws[s] = W[row c16][column 4s + q] (the A-operand layout of the panel's MFMAs); Ld holds -L
                if constexpr (!PYCLLP_WINV_FUSED) {
                    static_for<0, 15>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        winv_step<j>(Ws, Ld[j]);
                    });
                }
#pragma unroll
                for (int s = 0; s < 4; s++) { wsA[s * 64 + lane] = Ws[s]; wl[K * 256 + c16 * 16 + 4 * s + q] = Ws[s]; }
            }
            __syncthreads();
            // ---- panel: Y_KI = W_K M_KI = D_K L_IK' (blocks stay unscaled); guard test Y^2 > beta^2 D ----
            double WsL[4], rDr[4];
#pragma unroll
            for (int s = 0; s < 4; s++) { WsL[s] = wsA[s * 64 + lane]; rDr[s] = rdv[16 * K + 4 * s + q]; }
            for (int I = K + 1 + wave; I < MB; I += 4) {
                double* blk = Mw + (size_t)bidx(K, I) * 256;
                double mb[4];
#pragma unroll
                for (int s = 0; s < 4; s++) mb[s] = blk[s * 64 + lane];
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(WsL[s], mb[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) { ymax = fmax(ymax, acc[r] * acc[r] * rDr[r]); blk[r * 64 + lane] = acc[r]; }
            }
            __syncthreads();
            // ---- trailing update: M_JI -= Y_KJ' D_K^-1 Y_KI, K < J <= I (the diagonal blocks included, in full).  The pairs
            //      (J, I) in row-major order are dealt round-robin to the waves; a wave takes BIG_PAIRS_PER_TRIP of its pairs per trip -- all
            //      their block reads in flight before the first MFMA -- because one round trip to a block that lives in the L2
            //      workspace costs more than the 16 MFMAs it feeds ----
            {
                const int nt = MB - K - 1;                        // trailing block rows
                const int npair = nt * (nt + 1) / 2;
                constexpr int PT = BIG_PAIRS_PER_TRIP;
                for (int p0 = wave; p0 < npair; p0 += 4 * PT) {
                    double4_t acc[PT];
                    double yn[PT][4], yb[PT][4];
                    double* dst[PT];
#pragma unroll
                    for (int u = 0; u < PT; u++) {
                        const int pp = p0 + 4 * u;
                        const bool on = pp < npair;
                        // pair index -> (row a, column b) of the upper triangle of an nt x nt array, a <= b, row-major
                        int a = 0, rem = on ? pp : 0;
                        while (rem >= nt - a) { rem -= nt - a; a++; }
                        const int J = K + 1 + a, I = J + rem;
                        const double* yj = Mw + (size_t)bidx(K, J) * 256;
                        const double* yi = Mw + (size_t)bidx(K, I) * 256;
                        dst[u] = on ? Mw + (size_t)bidx(J, I) * 256 : nullptr;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            acc[u][r] = dst[u] ? dst[u][r * 64 + lane] : 0.0;
                            yn[u][r] = -(yj[r * 64 + lane] * rDr[r]);
                            yb[u][r] = yi[r * 64 + lane];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < PT; u++) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(yn[u][s4], yb[u][s4], acc[u], 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < PT; u++)
                        if (dst[u]) {
#pragma unroll
                            for (int r = 0; r < 4; r++) dst[u][r * 64 + lane] = acc[u][r];
                        }
                }
            }
            __syncthreads();
        }
        const double bad = bmax((ymax > beta2) ? 1.0 : 0.0, red, tid);
        return bad > 0.0;
    };

    // ---- cold path: the reference's modified LDL' with the Nocedal-Wright guard APPLIED (ldl.cl:314-378; oracle
    //      modified_ldl_core), column by column on the blocks (re-formed by the caller), the 256 threads over the rows.  Runs only
    //      when factor() reported that the guard would have bitten (seen on collapsing iterates of the embedding); leaves the
    //      same representation factor() does: off-diagonal blocks Y = D L' (raw column entries), 1/D in rdv, W_K = L_KK^-1 in wl.
    auto factor_guarded = [&](double beta2, double floor_, bool relf) {
        auto EL = [&](int i, int k) -> double& {      // element (i, k), k <= i, of the lower triangle = U[k][i]
            return Mw[(size_t)bidx(k >> 4, i >> 4) * 256 + boff(k & 15, i & 15)];
        };
        for (int j = 0; j < MP; j++) {
            double th = 0.0;
            for (int i = j + 1 + tid; i < MP; i += BT) th = fmax(th, fabs(EL(i, j)));
            th = bmax(th, red, tid);
            const double piv = EL(j, j);
            const double aD = fmax(fmax(fabs(piv), relf ? flr[j] : floor_), th * th / beta2);      // ldl.cl:368
            const double rD = 1.0 / aD;
            if (tid == 0) rdv[j] = rD;
            for (int i = j + 1 + tid; i < MP; i += BT) {
                const double li = EL(i, j) * rD;
                for (int k = j + 1; k <= i; k++) EL(i, k) = fma(-li, EL(k, j), EL(i, k));
            }
            __syncthreads();
        }
        // W_K = L_KK^-1 (unit lower triangular, L_ij = u_ij / D_j), column t of block K by thread (K, t)
        for (int w0 = tid; w0 < MB * 16; w0 += BT) {
            const int K = w0 >> 4, t = w0 & 15;
            double wcol[16];
#pragma unroll
            for (int i = 0; i < 16; i++) wcol[i] = (i == t) ? 1.0 : 0.0;
            for (int i = t + 1; i < 16; i++) {
                double acc = 0.0;
                for (int jj = t; jj < i; jj++) acc = fma(EL(16 * K + i, 16 * K + jj) * rdv[16 * K + jj], wcol[jj], acc);
                wcol[i] = -acc;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) wl[K * 256 + i * 16 + t] = wcol[i];
        }
        __syncthreads();
    };

    // ---- um <- (L D L')^-1 um: block substitution on wave 0 (forms and reductions as in ipm_wreg.hip's solve()) ----
    auto solve = [&]() {
        __syncthreads();
        if (wave == 0) {
            for (int I = 0; I < MB; I++) {          // forward, row oriented: t_I = W_I (s_I - sum_{K<I} Y_KI' D_K^-1 t_K)
                double p = 0.0;
                {   // (four blocks per trip: sixteen reads in flight; blocks (0..I-1, I) are contiguous)
                    const double* col = Mw + (size_t)bidx(0, I) * 256;
                    int K = 0;
                    for (; K + 4 <= I; K += 4) {
                        double bv[16];
#pragma unroll
                        for (int u = 0; u < 4; u++)
#pragma unroll
                            for (int r = 0; r < 4; r++) bv[4 * u + r] = col[(size_t)(K + u) * 256 + r * 64 + lane];
#pragma unroll
                        for (int u = 0; u < 4; u++)
#pragma unroll
                            for (int r = 0; r < 4; r++) p = fma(bv[4 * u + r], tdv[16 * (K + u) + 4 * r + q], p);
                    }
                    for (; K < I; K++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) p = fma(col[(size_t)K * 256 + r * 64 + lane], tdv[16 * K + 4 * r + q], p);
                    }
                }
                double rC = um[16 * I + c16];
                if (I > 0) rC -= quad_sum(p);
                double tR[4];
#pragma unroll
                for (int s = 0; s < 4; s++) tR[s] = row_sum(wl[I * 256 + (4 * s + q) * 16 + c16] * rC);
                wave_lds_sync();
                if (c16 == 0) {
#pragma unroll
                    for (int s = 0; s < 4; s++) { um[16 * I + 4 * s + q] = tR[s]; tdv[16 * I + 4 * s + q] = tR[s] * rdv[16 * I + 4 * s + q]; }
                }
                wave_lds_sync();
            }
            for (int K = MB - 1; K >= 0; K--) {     // backward: x_K = W_K' D_K^-1 (t_K - sum_{I>K} Y_KI x_I)
                double pr[4] = {0.0, 0.0, 0.0, 0.0};
                {
                    int I = K + 1;
                    for (; I + 4 <= MB; I += 4) {
                        double bv[16], xC[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const double* blk = Mw + (size_t)bidx(K, I + u) * 256;
                            xC[u] = um[16 * (I + u) + c16];
#pragma unroll
                            for (int r = 0; r < 4; r++) bv[4 * u + r] = blk[r * 64 + lane];
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++)
#pragma unroll
                            for (int r = 0; r < 4; r++) pr[r] = fma(bv[4 * u + r], xC[u], pr[r]);
                    }
                    for (; I < MB; I++) {
                        const double* blk = Mw + (size_t)bidx(K, I) * 256;
                        const double xC = um[16 * I + c16];
#pragma unroll
                        for (int r = 0; r < 4; r++) pr[r] = fma(blk[r * 64 + lane], xC, pr[r]);
                    }
                }
                double px = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double v = um[16 * K + 4 * r + q];
                    if (K < MB - 1) v -= row_sum(pr[r]);
                    px = fma(wl[K * 256 + (4 * r + q) * 16 + c16], v * rdv[16 * K + 4 * r + q], px);
                }
                const double xk = quad_sum(px);
                wave_lds_sync();
                if (q == 0) um[16 * K + c16] = xk;
                wave_lds_sync();
            }
        }
        __syncthreads();
    };

    STAMP_DECL
    for (;;) {
        // ---- next LP from the device-wide queue ----
        __syncthreads();
        if (tid == 0) ((int*)red)[60] = atomicAdd(queue, 1);
        __syncthreads();
        const long lp = ((int*)red)[60];
        if (lp >= B) break;

        double x[BNC], z[BNC], c[BNC];
        bool ok[BNC];
        {
            // (an opaque copy of the thread index: the per-thread 64-bit addresses of this LP's vectors must not be hoisted out of
            // the LP loop as invariants -- forty registers that then sit in scratch for the whole kernel)
            int to = tid;
            asm volatile("" : "+v"(to));
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                const int j = to + BT * k;
                ok[k] = j < n;
                c[k] = ok[k] ? cg[lp * n + j] : 0.0;
                x[k] = (warm && ok[k]) ? xg[lp * n + j] : 1.0;
                z[k] = (warm && ok[k]) ? zg[lp * n + j] : 1.0;
            }
            if (to < MP) {
                bs[to] = (to < m) ? bg[lp * m + to] : 0.0;
                ys[to] = (to < m) ? ((warm && yg) ? yg[lp * m + to] : (hsd ? 0.0 : 1.0)) : 0.0;
            }
        }
        __syncthreads();
        double sb = 1.0, sc = 1.0;
        if (autoscale) {   // solve with b/max|b| and c/max|c| (PYCLLP_FLAG_AUTOSCALE); undone when storing
            double cm = 0.0;
#pragma unroll
            for (int k = 0; k < BNC; k++) cm = fmax(cm, fabs(c[k]));
            sb = bmax((tid < m) ? fabs(bs[tid]) : 0.0, red, tid);
            sc = bmax(cm, red, tid);
            sb = (sb > 0.0) ? sb : 1.0; sc = (sc > 0.0) ? sc : 1.0;
            if (tid < m) { bs[tid] = bs[tid] / sb; if (warm) ys[tid] = ys[tid] / sc; }
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                c[k] = c[k] / sc;
                if (warm) { x[k] = x[k] / sb; z[k] = z[k] / sc; }
            }
            __syncthreads();
        }
        double c2 = 0.0;
#pragma unroll
        for (int k = 0; k < BNC; k++) c2 = fma(c[k], c[k], c2);
        const double nb2 = bsum((tid < m) ? bs[tid] * bs[tid] : 0.0, red, tid);
        const double nc2 = bsum(c2, red, tid);
        const double nbn = sqrt(nb2), ncn = sqrt(nc2);
        const double tol_r = o.eps * (1.0 + nbn), tol_s = o.eps * (1.0 + ncn);
        const double etol = o.refine_tol * (1.0 + nbn);
        double tau = 1.0, kap = 1.0;
        if (hsd && warm) {
            double g0 = 0.0;
#pragma unroll
            for (int k = 0; k < BNC; k++) g0 += ok[k] ? x[k] * z[k] : 0.0;
            kap = bsum(g0, red, tid) / (double)n;
        }
        double normr0 = 1e300, norms0 = 1e300, po = 0.0, du = 0.0;
        int stat = PYCLLP_STATUS_ITERATION_LIMIT, it = 0;

        for (; it < o.max_iter; it++) {
            // ---- residuals, gap, objectives (primal_normal.cl:30-48, 76-94, 245-248; tau-scaled on the embedding) ----
            double v[BNC], sg[BNC];
            At_cols(ys, v);
            double s2 = 0.0, gam = 0.0, pp = 0.0;
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                const int j = tid + BT * k;
                sg[k] = ok[k] ? c[k] * tau - v[k] + z[k] : 0.0;
                s2 = fma(sg[k], sg[k], s2);
                gam += ok[k] ? x[k] * z[k] : 0.0;
                pp += ok[k] ? c[k] * x[k] : 0.0;
                if (j < NPv) vx[j] = ok[k] ? x[k] : 0.0;
            }
            __syncthreads();
            const double Ax = A_row(vx);
            const double rho_i = (tid < m) ? bs[tid] * tau - Ax : 0.0;
            const double dd = (tid < MP) ? bs[tid] * ys[tid] : 0.0;
            const double norms = sqrt(bsum(s2, red, tid));
            gam = bsum(gam, red, tid); po = bsum(pp, red, tid); du = bsum(dd, red, tid);
            const double normr = sqrt(bsum(rho_i * rho_i, red, tid));
            // (predictor-corrector: 0 for the predictor, set from its outcome below)
            double mu = nwt ? nwt_mu : (hsd ? o.delta * (gam + tau * kap) / (double)(n + 1) : (pc ? 0.0 : o.delta * gam / nm));
            const double phi = du - po + kap;
            if (!nwt) {
                if (hsd) {      // oracle hsd_one_raw: optimal, or a primal / dual ray
                    const bool p_ray = po > 0.0 && fma(nbn, tau, normr) <= einf * po;
                    const bool d_ray = du < 0.0 && fma(ncn, tau, norms) <= einf * -du;
                    if (!(isfinite(normr) && isfinite(norms) && isfinite(gam) && isfinite(tau) && isfinite(kap))) { stat = PYCLLP_STATUS_NUMERICAL; break; }
                    if (normr <= tol_r * tau && norms <= tol_s * tau && gam <= o.eps * tau * (tau + fabs(po))) { stat = PYCLLP_STATUS_OPTIMAL; break; }
                    if (p_ray || d_ray) {
                        stat = (p_ray && d_ray) ? ((-du > po) ? PYCLLP_STATUS_PRIMAL_INFEASIBLE : PYCLLP_STATUS_DUAL_INFEASIBLE)
                                                : (p_ray ? PYCLLP_STATUS_DUAL_INFEASIBLE : PYCLLP_STATUS_PRIMAL_INFEASIBLE);
                        break;
                    }
                } else {        // primal_normal.cl:256-269; oracle ipm_one_path
                    if (!(isfinite(normr) && isfinite(norms) && isfinite(gam))) { stat = PYCLLP_STATUS_NUMERICAL; break; }
                    if (normr <= tol_r && norms <= tol_s && gam <= o.eps * (1.0 + fabs(po))) { stat = PYCLLP_STATUS_OPTIMAL; break; }
                    if (normr > 10.0 * normr0 && normr > PYCLLP_GROWTH_FLOOR * tol_r) { stat = PYCLLP_STATUS_PRIMAL_INFEASIBLE; break; }
                    if (norms > 10.0 * norms0 && norms > PYCLLP_GROWTH_FLOOR * tol_s) { stat = PYCLLP_STATUS_DUAL_INFEASIBLE; break; }
                }
            }
            STAMP(0)
            // ---- d = x/z, t (plain path: c - A'y + mu/x; embedding: r1 = mu/x - z + eta sigma) ----
            double d[BNC], t[BNC];
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                d[k] = ok[k] ? x[k] / z[k] : 0.0;
                t[k] = ok[k] ? (hsd ? fma(eta, sg[k], mu / x[k] - z[k]) : c[k] - v[k] + mu / x[k]) : 0.0;
            }
            __syncthreads();      // every A_row read of vx is done
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                const int j = tid + BT * k;
                if (j < NPv) { vx[j] = d[k] * t[k]; vd[j] = d[k]; }
            }
            __syncthreads();
            // ---- right-hand side A(d t) - rho (embedding: A(d r1) - eta rho) and diag(M) from one pass over the row ----
            double mdg = 0.0;
            const double adt = A_row_dg(vx, true, mdg);
            const double beta2 = bmax((tid < m) ? fabs(mdg) : 0.0, red, tid);     // ldl.cl:280-294
            if (hsd) {
                // M p = A(d c) - b first; q's right-hand side waits in qv
                __syncthreads();
#pragma unroll
                for (int k = 0; k < BNC; k++) { const int j = tid + BT * k; if (j < NPv) vx[j] = d[k] * c[k]; }
                __syncthreads();
                const double adc = A_row(vx);
                if (tid < MP) {
                    um[tid] = (tid < m) ? adc - bs[tid] : 0.0;
                    qv[tid] = (tid < m) ? fma(-eta, rho_i, adt) : 0.0;
                    flr[tid] = o.pivot_floor * o.pivot_floor * fabs(mdg);
                }
            } else if (tid < MP) um[tid] = (tid < m) ? adt - rho_i : 0.0;
            __syncthreads();
            STAMP(1)
            gram();
            STAMP(2)
            const bool viol = factor(beta2, hsd ? 0.0 : o.pivot_floor, hsd);
            if (viol || (o.flags & PYCLLP_FLAG_FORCE_GUARD_PATH)) {
                // the Nocedal-Wright guard would have bitten somewhere: re-form M and factor with the guard applied
                gram();
                factor_guarded(beta2, hsd ? 0.0 : o.pivot_floor, hsd);
            }
            STAMP(3)

            double dx[BNC], w2[BNC], cor[BNC];
#pragma unroll
            for (int k = 0; k < BNC; k++) cor[k] = mu;        // (plain path: the complementarity target is mu itself)
            double dtau = 0.0, etol_it = etol, rhot_i = rho_i;
            int nref = 0;
            if (hsd) {
                // p, then q; dy = p dtau + q, dx = u dtau + v with u = d (c - A'p), v = d (r1 - A'q)
                solve();
                if (tid < MP) { pv[tid] = um[tid]; um[tid] = qv[tid]; }
                __syncthreads();      // every thread reads ALL of pv below (round 3, found through a run-to-run difference of 1e-12 in
                                      // the objectives: without this barrier waves 1-3 could read pv before wave 0 had written it)
                double uu[BNC];
                At_cols(pv, w2);
#pragma unroll
                for (int k = 0; k < BNC; k++) uu[k] = ok[k] ? c[k] - w2[k] : 0.0;
                solve();
                At_cols(um, w2);
                double dsum = 0.0, nsum = 0.0;
#pragma unroll
                for (int k = 0; k < BNC; k++) {
                    dx[k] = d[k] * (t[k] - w2[k]);
                    dsum = fma(d[k] * uu[k], uu[k], dsum);
                    nsum = fma(c[k], dx[k], nsum);
                }
                const double bq = bsum((tid < MP) ? bs[tid] * um[tid] : 0.0, red, tid);
                const double den = bsum(dsum, red, tid) + kap / tau;
                const double num = fma(eta, phi, mu / tau - kap) + bq - bsum(nsum, red, tid);
                dtau = num / den;
                if (tid < MP) dyv[tid] = fma(pv[tid], dtau, um[tid]);
                rhot_i = (tid < m) ? fma(bs[tid], dtau, eta * rho_i) : 0.0;       // A dx - b dtau = eta rho
#pragma unroll
                for (int k = 0; k < BNC; k++) dx[k] = fma(d[k] * uu[k], dtau, dx[k]);
                etol_it = o.refine_tol * (1.0 + nbn) * fmax(tau, kap);
            } else {
                if (pc) {
                    // predictor (mu = 0) -> how far it gets -> centering -> the corrector's target and right-hand side
                    solve();
                    At_cols(um, w2);
                    double dxa[BNC], dza[BNC], tha = 0.0;
#pragma unroll
                    for (int k = 0; k < BNC; k++) {
                        dxa[k] = (t[k] - w2[k]) * d[k];
                        dza[k] = ok[k] ? -z[k] - z[k] * dxa[k] / x[k] : 0.0;
                        if (ok[k]) tha = fmax(tha, fmax(-dza[k] / z[k], -dxa[k] / x[k]));
                    }
                    tha = bmax(tha, red, tid);
                    const double theta_a = fmin(1.0 / tha, 1.0);
                    double ga = 0.0;
#pragma unroll
                    for (int k = 0; k < BNC; k++) ga += ok[k] ? fma(theta_a, dxa[k], x[k]) * fma(theta_a, dza[k], z[k]) : 0.0;
                    ga = bsum(ga, red, tid);
                    const double sgm = ga / gam;
                    mu = sgm * sgm * sgm * gam / (double)n;
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < BNC; k++) {
                        const int j = tid + BT * k;
                        cor[k] = ok[k] ? mu - dxa[k] * dza[k] : 0.0;
                        t[k] = ok[k] ? t[k] + cor[k] / x[k] : 0.0;
                        if (j < NPv) vx[j] = d[k] * t[k];
                    }
                    __syncthreads();
                    const double adt2 = A_row(vx);
                    if (tid < MP) um[tid] = (tid < m) ? adt2 - rho_i : 0.0;
                }
                solve();
                if (tid < MP) dyv[tid] = um[tid];
                At_cols(um, w2);
#pragma unroll
                for (int k = 0; k < BNC; k++) dx[k] = (t[k] - w2[k]) * d[k];
            }
            STAMP(4)
            // ---- x-space refinement (oracle newton_dy): e = rho - A dx; M eta = e; dx += d A'eta; dy -= eta ----
            for (;;) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < BNC; k++) { const int j = tid + BT * k; if (j < NPv) vx[j] = ok[k] ? dx[k] : 0.0; }
                __syncthreads();
                const double e_i = (tid < m) ? rhot_i - A_row(vx) : 0.0;
                const double maxe = bmax(fabs(e_i), red, tid);
                if (!(maxe > etol_it) || nref >= o.max_refine) break;
                if (tid < MP) um[tid] = e_i;
                solve();
                if (tid < MP) dyv[tid] -= um[tid];
                At_cols(um, w2);
#pragma unroll
                for (int k = 0; k < BNC; k++) dx[k] = fma(d[k], w2[k], dx[k]);
                nref++;
            }
            STAMP(5)
            const double bad = bmax((tid < MP && !isfinite(dyv[tid])) ? 1.0 : 0.0, red, tid);
            if (nwt) {
                if (tid < m) nwt_dy[lp * m + tid] = dyv[tid];
                if (nwt_nref && tid == 0) nwt_nref[lp] = nref;
                break;
            }
            if (bad > 0.0 || !isfinite(dtau)) { stat = PYCLLP_STATUS_NUMERICAL; break; }
            // ---- step (primal_normal.cl:122-156; embedding: ratio test over x, z, tau, kappa) ----
            const double dkap = hsd ? mu / tau - kap - kap / tau * dtau : 0.0;
            double dz[BNC], th = hsd ? fmax(fmax(-dtau / tau, -dkap / kap), 0.0) : 0.0;
#pragma unroll
            for (int k = 0; k < BNC; k++) {
                dz[k] = ok[k] ? (cor[k] - z[k] * dx[k]) / x[k] - z[k] : 0.0;
                if (ok[k]) th = fmax(th, fmax(-dz[k] / z[k], -dx[k] / x[k]));
            }
            th = bmax(th, red, tid);
            const double theta = fmin(o.r / th, 1.0);
            if (tid < MP) ys[tid] = fma(theta, dyv[tid], ys[tid]);
#pragma unroll
            for (int k = 0; k < BNC; k++) { x[k] = fma(theta, dx[k], x[k]); z[k] = fma(theta, dz[k], z[k]); }
            if (hsd) { tau = fma(theta, dtau, tau); kap = fma(theta, dkap, kap); }
            normr0 = normr; norms0 = norms;
            __syncthreads();
            STAMP(6)
        }
        __syncthreads();
        if (nwt) continue;
        // HSD: optimal (and iteration-limit) points leave the homogeneous scaling (hsd.c:266-273); certificates stay
        const double rt = (hsd && (stat == PYCLLP_STATUS_OPTIMAL || stat == PYCLLP_STATUS_ITERATION_LIMIT)) ? 1.0 / tau : 1.0;
        int to = tid;
        asm volatile("" : "+v"(to));
#pragma unroll
        for (int k = 0; k < BNC; k++) {
            const int j = to + BT * k;
            if (ok[k]) { xg[lp * n + j] = x[k] * rt * sb; if (zg) zg[lp * n + j] = z[k] * rt * sc; }
        }
        if (yg && to < m) yg[lp * m + to] = ys[to] * rt * sc;
        if (tid == 0) {
            if (pobj) pobj[lp] = po * rt * (sb * sc);
            if (dobj) dobj[lp] = du * rt * (sb * sc);
            status[lp] = stat;
            if (iters) iters[lp] = it;
        }
        STAMP(7)
    }
    if (o.prof && tid == 0) { STAMP_FLUSH_BLOCK(o, blockIdx.x) }
}

template <typename T>
size_t put(std::vector<char>& host, const std::vector<T>& v) {
    size_t off = (host.size() + 15) & ~(size_t)15;
    host.resize(off + std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!v.empty()) memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

}  // namespace

struct BigPlan {
    BigTab tab;
    void* dev_blob = nullptr;
    size_t ws_doubles_per_block = 0;
};

int big_plan_create(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds, hipStream_t st,
                    BigPlan** out) {
    if (m > BIG_MAX_M || n > BIG_MAX_N || m < 1 || n < 1) return 1;
    const int MB = (m + 15) / 16, MP = 16 * MB;
    BigPlan* P = new BigPlan();
    BigTab& T = P->tab;
    memset(&T, 0, sizeof(T));
    T.m = m; T.n = n; T.nnz = nnz; T.MB = MB;
    // ---- CSC by counting sort (rows ascending inside a column) ----
    std::vector<int> cptr(n + 1, 0), crow(nnz);
    std::vector<double> cval(nnz);
    for (int e = 0; e < nnz; e++) cptr[col[e] + 1]++;
    for (int j = 0; j < n; j++) cptr[j + 1] += cptr[j];
    {
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int i = 0; i < m; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++) { const int p = fill[col[e]]++; crow[p] = i; cval[p] = val[e]; }
    }
    // ---- identity tail (equality form of a StandardLP)? ----
    bool sl = n > m;
    {
        std::vector<int> tail_cnt(sl ? m : 0, 0);
        for (int i = 0; i < m && sl; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++)
                if (col[e] >= n - m) { if (col[e] != n - m + i || val[e] != 1.0) sl = false; else tail_cnt[i]++; }
        for (int i = 0; i < m && sl; i++) if (tail_cnt[i] != 1) sl = false;
    }
    const int nd = sl ? n - m : n;
    // ---- Gram by term list or on the matrix cores?  terms = sum over columns of len (len + 1) / 2 products per Newton step,
    //      against MP^2 / 2 * nd for the dense product at the 8x higher rate of the matrix pipe over a gather-bound loop ----
    double n_terms = 0.0;
    for (int j = 0; j < n; j++) { const double l = cptr[j + 1] - cptr[j]; n_terms += l * (l + 1) / 2; }
    T.dense = n_terms > 0.125 * (double)MP * MP * nd ? 1 : 0;
    std::vector<double> img, img_rm, img_cm;
    std::vector<int> ent_dst, ent_dst2, ent_ptr, term_col;
    std::vector<double> term_w;
    if (T.dense) {
        T.nd = nd; T.n_sl = n - nd; T.ks = (nd + 3) / 4; T.imgR = MP;
        img.assign((size_t)T.ks * MP * 4, 0.0);
        img_rm.assign((size_t)m * nd, 0.0);
        img_cm.assign((size_t)nd * MP, 0.0);
        for (int i = 0; i < m; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++)
                if (col[e] < nd) {
                    img[((size_t)(col[e] >> 2) * MP + i) * 4 + (col[e] & 3)] = val[e];
                    img_rm[(size_t)i * nd + col[e]] = val[e];
                    img_cm[(size_t)col[e] * MP + i] = val[e];
                }
    } else {
        struct Term { int key, colj; double w; };
        std::vector<Term> terms;
        for (int j = 0; j < n; j++)
            for (int a = cptr[j]; a < cptr[j + 1]; a++)
                for (int b2 = cptr[j]; b2 <= a; b2++) {
                    const int i = crow[a], k = crow[b2];     // rows ascend inside a column: i >= k
                    terms.push_back({i * m + k, j, cval[a] * cval[b2]});
                    if (terms.size() > ((size_t)1 << 24)) { delete P; return 1; }
                }
        std::stable_sort(terms.begin(), terms.end(), [](const Term& x, const Term& y) { return x.key < y.key; });
        term_col.resize(terms.size()); term_w.resize(terms.size());
        for (size_t t = 0; t < terms.size(); t++) {
            if (t == 0 || terms[t].key != terms[t - 1].key) {
                const int i = terms[t].key / m, k = terms[t].key % m;
                const int K = k >> 4, I = i >> 4, kl = k & 15, il = i & 15;
                ent_dst.push_back(bidx(K, I) * 256 + boff(kl, il));
                ent_dst2.push_back((K == I && kl != il) ? bidx(K, K) * 256 + boff(il, kl) : -1);
                ent_ptr.push_back((int)t);
            }
            term_col[t] = terms[t].colj; term_w[t] = terms[t].w;
        }
        ent_ptr.push_back((int)terms.size());
        T.n_ent = (int)ent_dst.size();
    }
    // ---- LDS plan: the blocks in LDS only when two workgroups still fit a CU with them -- measured (round 3,
    //      profiles/r03/large_lp_kernel.txt): a second resident workgroup is worth more than LDS-resident blocks (m = 144 with
    //      the blocks in LDS and one workgroup per CU: 22 k LPs/s; with the blocks in the L2 workspace and two: 35 k, three: 44 k) ----
    T.m_in_lds = big_lds_doubles(MB, n, true) * sizeof(double) * 2 <= (size_t)max_lds ? 1 : 0;
    const size_t lds = big_lds_doubles(MB, n, T.m_in_lds != 0) * sizeof(double);
    if (lds > (size_t)max_lds) { delete P; return 1; }
    T.lds_bytes = (int)lds;
    P->ws_doubles_per_block = T.m_in_lds ? 0 : (size_t)MB * (MB + 1) / 2 * 256;
    // ---- device copies ----
    std::vector<char> host;
    std::vector<double> csr_val(val, val + nnz);
    std::vector<int> csr_ptr(ptr, ptr + m + 1), csr_col(col, col + nnz);
    const size_t a1 = put(host, csr_val), a2 = put(host, csr_ptr), a3 = put(host, csr_col), a4 = put(host, cval),
                 a5 = put(host, cptr), a6 = put(host, crow), a7 = put(host, img), a8 = put(host, ent_dst),
                 a9 = put(host, ent_dst2), a10 = put(host, ent_ptr), a11 = put(host, term_w), a12 = put(host, term_col),
                 a13 = put(host, img_rm), a14 = put(host, img_cm);
    hipError_t e = hipMalloc(&P->dev_blob, host.size());
    if (e == hipSuccess) e = hipMemcpyAsync(P->dev_blob, host.data(), host.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { if (P->dev_blob) (void)hipFree(P->dev_blob); delete P; return 1000 + (int)e; }
    char* db = (char*)P->dev_blob;
    T.csr_val = (const double*)(db + a1); T.csr_ptr = (const int*)(db + a2); T.csr_col = (const int*)(db + a3);
    T.csc_val = (const double*)(db + a4); T.csc_ptr = (const int*)(db + a5); T.csc_row = (const int*)(db + a6);
    T.img = (const double*)(db + a7); T.img_rm = (const double*)(db + a13); T.img_cm = (const double*)(db + a14);
    T.ent_dst = (const int*)(db + a8); T.ent_dst2 = (const int*)(db + a9); T.ent_ptr = (const int*)(db + a10);
    T.term_w = (const double*)(db + a11); T.term_col = (const int*)(db + a12);
    *out = P;
    return 0;
}

void big_plan_free(BigPlan* p) {
    if (!p) return;
    if (p->dev_blob) (void)hipFree(p->dev_blob);
    delete p;
}

int big_lds_bytes(const BigPlan* p) { return p ? p->tab.lds_bytes : 0; }
int big_dense_mode(const BigPlan* p) { return p ? p->tab.dense : 0; }

static hipError_t big_launch(BigPlan* p, long B, const double* b, const double* c, double* x, double* y, double* z, double* pobj,
                             double* dobj, int* status, int* iters, int* qhead, double mu, double* nwt_dy, int* nwt_nref,
                             DevOpts o, int num_cu, hipStream_t st, int* grid_out) {
    long cus = (long)num_cu - o.reserve_cus > 0 ? (long)num_cu - o.reserve_cus : 1;
    // two workgroups per CU where the LDS allows it (256 registers per lane each): the kernel is bound by the latency of its
    // L2 / LDS round trips, and a second workgroup fills them
    // workgroups per CU: the kernel is bound by the latency of its L2 / LDS round trips, co-resident workgroups fill them.
    // Three where the LDS allows (the instance compiled for 168 registers per lane), else two, else one.
    const long per_cu = (3 * (long)p->tab.lds_bytes <= 160 * 1024) ? 3 : ((2 * (long)p->tab.lds_bytes <= 160 * 1024) ? 2 : 1);
    cus *= per_cu;
    const int n_ = p->tab.n;
    auto kern = (n_ <= 2 * BT) ? ((per_cu == 3) ? ipm_big_kernel<3, 2> : ipm_big_kernel<2, 2>)
              : (n_ <= 3 * BT) ? ((per_cu == 3) ? ipm_big_kernel<3, 3> : ipm_big_kernel<2, 3>)
                               : ((per_cu == 3) ? ipm_big_kernel<3, BNC_MAX> : ipm_big_kernel<2, BNC_MAX>);
    long grid = std::min(cus, B);
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = (int)grid;
    hipError_t e = set_dyn_lds((const void*)kern, p->tab.lds_bytes);
    if (e != hipSuccess) return e;
    double* ws = nullptr;
    if (p->ws_doubles_per_block) {
        e = hipMallocAsync((void**)&ws, sizeof(double) * p->ws_doubles_per_block * (size_t)grid, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BT), p->tab.lds_bytes, st, p->tab, B, b, c, x, y, z, pobj, dobj,
                       status, iters, qhead, ws, mu, nwt_dy, nwt_nref, o);
    e = hipGetLastError();
    if (ws) { hipError_t e2 = hipFreeAsync(ws, st); if (e == hipSuccess) e = e2; }
    return e;
}

hipError_t big_launch_solve(BigPlan* p, long B, const double* b, const double* c, double* x, double* y, double* z,
                            double* pobj, double* dobj, int* status, int* iters, int* qhead, DevOpts o, int num_cu,
                            hipStream_t st, int* grid_out) {
    return big_launch(p, B, b, c, x, y, z, pobj, dobj, status, iters, qhead, 0.0, nullptr, nullptr, o, num_cu, st, grid_out);
}

hipError_t big_launch_newton(BigPlan* p, long B, const double* x, const double* z, const double* y, const double* b,
                             const double* c, double mu, double* dy, int* nref, int* qhead, DevOpts o, int num_cu,
                             hipStream_t st) {
    return big_launch(p, B, b, c, (double*)x, (double*)y, (double*)z, nullptr, nullptr, nullptr, nullptr, qhead, mu, dy, nref, o,
                      num_cu, st, nullptr);
}
