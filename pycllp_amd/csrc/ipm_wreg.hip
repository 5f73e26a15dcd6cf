// ipm_wreg.hip -- sparse shared-A path, third generation: ONE LP PER WAVEFRONT, the normal-equations matrix and its
// LDL' factor held in REGISTERS for the whole Newton step, no workgroup barrier anywhere in the solve.
//
// Replaces (as ipm_block.inc does, which stays as the general/guarded path) the reference's sparse twins: host
// ClSparsePrimalNormalSolver (pycllp/solvers/cl.py:127-278) and the kernels sparse_standard_primal_normal
// (pycllp/cl/primal_normal.cl:287-375), sparse_primal_normal_step (:158-198), sparse_*_infeasibility (:50-74, :96-120),
// sparse_AXZAt_ij/_ii (pycllp/cl/ldl.cl:140-196), sparse_primal_normal_rhs_i (:221-257), sparse_factor_primal_normal
// (:381-502), sparse_forward_backward_primal_normal (:540-574), sparse_solve_primal_normal (:656-712).
//
// Why: with M (m = 128: 66 KB packed) in LDS only two LPs fit a CU, so ipm_block_kernel has to spread ONE LP over
// four wavefronts and pays for it in barriers, in waves idling during the serial pivot chain and the triangular solves
// (44 % + 26 % of its run time), and in SIMDs idling.  The register file of a CU is 512 KB -- three times its LDS.  Here
// each of the 4 SIMDs of a CU runs one wavefront with the full 512-register budget that owns one LP:
//   * the factor is kept as U = L' in 16 x 16 blocks U[K][I] (K <= I) in the ACCUMULATOR layout of
//     v_mfma_f64_16x16x4_f64 (register r of lane l holds element [4r + (l >> 4)][l & 15]): 36 blocks x 4 doubles at
//     m = 128 (see below: 28 of them resident).  That layout is, unchanged, the B operand of the block and the A operand of its transpose, so both the
//     panel solve  Y_KI = L_KK^-1 M_KI  and the trailing update  U_JI -= Y_KJ' U_KI  are MFMAs straight on the resident
//     registers -- no operand ever moves;
//   * only the 28 OFF-DIAGONAL blocks live in registers (224 of the 256 accumulator registers).  A diagonal block is
//     formed when its turn comes (left-looking): its Schur update on the matrix cores, plus the original block straight
//     from the tables, into a 2 KB LDS tile; from there it is read in "lane = row" form and factored by a 16-step DPP
//     (row_newbcast) chain -- every 16-lane row of the wave redundantly, so nothing is broadcast across rows -- and its
//     inverse W_K = L_KK^-1 is formed directly in the MFMA A-operand layout (quad q owns columns q, q+4, ...) for the
//     panel; W_K is also what the triangular solves use, from a packed copy in LDS (1 KB per block);
//   * M = A diag(x/z) A' is assembled from entry/term tables built once at init (deterministic, atomic-free),
//     scattered through a 8 KB LDS stage one block column at a time and loaded in the accumulator layout;
//   * A x and A'u use ELL copies of A (by rows / by columns) in LDS; N-vectors live in registers (lane = column),
//     m-vectors in a per-wave LDS area;
//   * the triangular solves are 16-row block steps: 4 FMAs per off-diagonal block, quad/row reductions by
//     v_permlane swaps and DPP.
// The Nocedal-Wright guard (ldl.cl:487) is not applied here: the sweep records whether it WOULD have bitten and such an
// LP (never seen on a positive definite M) is deferred to ipm_block_kernel, which applies it exactly.
// Semantics = oracle/ipm_dense_ref.c (ipm_one_path / hsd_one_raw), like every other kernel of this library.
// tools/wreg_sim.py is a lane-level numpy model of the layouts used below.
#include "wreg.h"

namespace {

// An inline-asm operand of the accumulator register class: with one in the kernel the compiler keeps the AGPR form of
// the MFMAs (C/D -- the resident U blocks -- in a0..a255, A/B read from either file).
#define USE_AGPR_FORM() do { int agpr_hint_; asm volatile("; accumulator file in use" : "=a"(agpr_hint_)); } while (0)

typedef double double2_t __attribute__((ext_vector_type(2)));

// A finished panel block, PARKED in eight accumulator registers.  The values are written there by inline asm with an
// accumulator-class output, which is the one way to tell the register allocator where a long-lived, rarely read value
// belongs: left to itself it keeps the panel results in architectural VGPRs (their next use is an MFMA B operand, which
// may come from either file) and, out of VGPRs, spills them to scratch -- with one wavefront per SIMD every scratch
// reload is a fully exposed memory round trip (first version of this kernel: 345 k cycles per iteration, 42 % of them in
// the pivot chains waiting for reloads).  Reads are plain uses: the compiler copies them out (v_accvgpr_read) itself and
// keeps track of the hazards.
struct PBlk { int h[8]; };
__device__ __forceinline__ void park(PBlk& p, const double4_t& v) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int lo = __double2loint(v[r]), hi = __double2hiint(v[r]);
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(p.h[2 * r]) : "v"(lo));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(p.h[2 * r + 1]) : "v"(hi));
    }
}
__device__ __forceinline__ double unpark(const PBlk& p, int r) { return __hiloint2double(p.h[2 * r + 1], p.h[2 * r]); }

constexpr int HB = 4;            // 16 x 16 blocks per Gram staging chunk (8 KB of LDS)
constexpr int STAGE_D = HB * 256;
constexpr int TILE_OFF = 512, RR_OFF = 800;   // aliases inside the stage (free once the blocks are in registers)
constexpr int WL = 128;          // doubles reserved per packed W block (120 used: strictly lower triangle, row i at i(i-1)/2)
constexpr int MAX_NQ = 8;
constexpr int META_COFF = MAX_NQ, META_SEG = 2 * MAX_NQ, META_DSEG = META_SEG + 24, META_N = META_DSEG + 24;

template <int MB>
struct WGeo {
    static constexpr int MP = 16 * MB;
    static constexpr int MR = (MP + 63) / 64;    // m-vector registers per lane in "lane = row" form
    static constexpr int MPL = 64 * MR;
    static constexpr int NBLK = MB * (MB - 1) / 2;
    // off-diagonal block (K, I), K < I, of U = L'
    __host__ __device__ static constexpr int bix(int K, int I) { return K * MB - K * (K + 1) / 2 + (I - K - 1); }
    // Gram staging chunks: block row K of U holds the blocks I = K+1 .. MB-1, staged HB at a time
    __host__ __device__ static constexpr int nch(int K) { return (MB - 1 - K + HB - 1) / HB; }
    __host__ __device__ static constexpr int chbase(int K) { int s = 0; for (int k = 0; k < K; k++) s += nch(k); return s; }
    static constexpr int NCHUNK = chbase(MB);
    static constexpr int WAVE_D(int NQ) { return STAGE_D + 64 * NQ + 6 * MP + MB * WL; }   // per-wave LDS doubles
};

// Device view of the tables of one constraint matrix (built by wreg_plan_create).
struct WregTab {
    int m, n, nnz;
    int rmax, n_ent, n_term;
    int meta[META_N];     // [0..8) ELL depth of column register q, [META_COFF..) its first ELL slot, [META_SEG..) first Gram
                          // entry of staging chunk i (chunks in (K, ch) order; NCHUNK + 1 used), [META_DSEG..) first Gram
                          // entry of diagonal block K (MB + 1 used) -- copied to LDS
    const double* csr_val; const unsigned short* csr_col; const unsigned short* csr_ptr; const unsigned short* csr_len;
    // A by columns in ELL form over column POSITIONS: the columns are dealt to the (lane, register) positions of the
    // N-vectors sorted by length, so that each register's 64 columns are about equally long (JDS); colmap[pos] = column
    const double* ec_val; const unsigned short* ec_row; const unsigned short* colmap; int ctot;
    const unsigned* e_ptr; const unsigned short* e_dst;   // Gram entries: term range, offset inside the stage / tile
    const double* t_w; const unsigned short* t_col;       // Gram terms: a_ij a_kj and the column j
    int o_csr_val, o_ec_val, o_t_w, o_wave, o_e_ptr, o_meta, o_csr_col, o_csr_ptr, o_csr_len, o_ec_row, o_colmap,
        o_e_dst, o_t_col;                                 // LDS byte offsets
    int wave_doubles, lds_bytes;
};

// ---- cross-lane helpers ----------------------------------------------------------------------------------------
// v[l] + v[l ^ 16], then + the other 32 lanes: the sum over the four 16-lane rows, identical in all of them
__device__ __forceinline__ double quad_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(v); hi = __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
// sum over the 16 lanes of each DPP row (identical inside the row)
__device__ __forceinline__ double row_sum(double v) {
    asm volatile("" : "+v"(v));
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return v;
}
__device__ __forceinline__ double row_max(double v) {
    v = fmax(v, dpp_d<0xB1>(v));
    v = fmax(v, dpp_d<0x4E>(v));
    v = fmax(v, dpp_d<0x141>(v));
    v = fmax(v, dpp_d<0x140>(v));
    return v;
}
// a wave-uniform double, moved to a scalar register pair: the f64 arithmetic that produced it left it in VGPRs, where a
// kernel-lifetime scalar (tolerances, mu, theta, ...) costs two registers of every lane -- or a scratch slot
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// whole-wave reductions: DPP inside the rows, then the four row results through SGPRs (wave-uniform result)
__device__ __forceinline__ double wsum(double v) {
    v = row_sum(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double wmax(double v) {
    v = row_max(v);
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}

// ---- the per-wave machinery ------------------------------------------------------------------------------------
template <int MB, int NQ>
struct WReg {
    using G = WGeo<MB>;
    static constexpr int MP = G::MP, MR = G::MR, MPL = G::MPL, NP = 64 * NQ;

    // Off-diagonal blocks [bix(K, I)], K < I.  Life of a block: gram() parks the original M_KI in P; the trailing update
    // of stage_() 0 takes it out as an MFMA accumulator (U) where it stays through the following stages' updates; panel K
    // turns it into Y_KI = D_K L_IK' and parks that in P for the rest of the Newton step.
    double4_t U[G::NBLK > 0 ? G::NBLK : 1];
    PBlk P[G::NBLK > 0 ? G::NBLK : 1];
    // LDS: shared tables (A by rows and by columns in compact form, Gram entries/terms)
    const double* csr_val; const unsigned short* csr_col; const unsigned short* csr_ptr; const unsigned short* csr_len;
    const double* ec_val; const unsigned short* ec_row; const unsigned short* colmap;
    const unsigned* e_ptr; const unsigned short* e_dst;
    const double* t_w; const unsigned short* t_col;
    const int* meta;
    // LDS: this wave's area, every array at a COMPILE-TIME offset from the one base pointer W0 -- so that the address
    // arithmetic of all of them folds into a handful of lane-dependent bases plus immediate offsets (as separate
    // run-time pointers every (array, index pattern) pair costs a VGPR for the whole kernel)
    double* W0;
    __device__ __forceinline__ double* stage_() const { return W0; }                            // [STAGE_D] Gram staging; aliases: vx = stage_()[0..NP), tile, rr
    __device__ __forceinline__ double* vd_() const { return W0 + STAGE_D; }                     // [NP] d = x/z
    __device__ __forceinline__ double* ys_() const { return W0 + STAGE_D + NP; }                // [MP] y
    __device__ __forceinline__ double* bs_() const { return W0 + STAGE_D + NP + MP; }           // [MP] b
    __device__ __forceinline__ double* um_() const { return W0 + STAGE_D + NP + 2 * MP; }       // [MP] solve vector in/out
    __device__ __forceinline__ double* rdv_() const { return W0 + STAGE_D + NP + 3 * MP; }      // [MP] 1/D
    __device__ __forceinline__ double* flr_() const { return W0 + STAGE_D + NP + 4 * MP; }      // [MP] per-column pivot floors (HSD)
    __device__ __forceinline__ double* adv_() const { return W0 + STAGE_D + NP + 5 * MP; }      // [MP] D (the floored pivots)
    __device__ __forceinline__ double* wl_() const { return W0 + STAGE_D + NP + 6 * MP; }       // [MB][WL] W_K = L_KK^-1, strict lower triangle packed by rows
    int lane, q, c16, m, n, rmax;

    // out_q = (A'u)_j for the column at position lane + 64 q (see colmap), u in LDS.  ELL: slot t of register q sits at
    // (coff_q + t) 64 + lane -- an immediate offset from one lane-dependent base; padded slots hold value 0, row 0.
    __device__ __forceinline__ void At(const double* u, double (&out)[NQ]) const {
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const int cm = __builtin_amdgcn_readfirstlane(meta[qq]);
            const int base = __builtin_amdgcn_readfirstlane(meta[META_COFF + qq]) * 64 + lane;
            double a0 = 0.0, a1 = 0.0;
            int t = 0;
            for (; t + 1 < cm; t += 2) {
                a0 = fma(ec_val[base + 64 * t], u[ec_row[base + 64 * t]], a0);
                a1 = fma(ec_val[base + 64 * t + 64], u[ec_row[base + 64 * t + 64]], a1);
            }
            if (t < cm) a0 = fma(ec_val[base + 64 * t], u[ec_row[base + 64 * t]], a0);
            out[qq] = a0 + a1;
        }
    }
    // (A v)_i for the rows i = lane + 64 r2 of this lane, v staged in LDS; with DIAG also diag(A diag(d) A')_i (d in vd_();
    // padded rows get 1: identity rows of M) from the same pass over the row
    template <bool DIAG>
    __device__ __forceinline__ void Arow(const double* v, double (&out)[MR], double (&md)[MR]) const {
        int ptr[MR], len[MR];
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            ptr[r2] = csr_ptr[lane + 64 * r2]; len[r2] = csr_len[lane + 64 * r2];
            out[r2] = 0.0; md[r2] = 0.0;
        }
#pragma unroll 4
        for (int t = 0; t < rmax; t++) {
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const bool on = t < len[r2];
                const int p = on ? ptr[r2] + t : 0;
                const double a = on ? csr_val[p] : 0.0;
                const int cidx = csr_col[p];
                out[r2] = fma(a, v[cidx], out[r2]);
                if (DIAG) md[r2] = fma(a * a, vd_()[cidx], md[r2]);
            }
        }
        if (DIAG) {
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) md[r2] = (lane + 64 * r2 < m) ? md[r2] : 1.0;
        }
    }

    // dstbuf[e_dst[e]] (+)= value of Gram entry e (sum over its terms of a_ij a_kj d_j) for the entries [e0, e1): FOUR
    // entries per lane per trip with every table read of the trip issued before the first use (one wavefront alone on its
    // SIMD hides no latency by itself); the first term of an entry -- for most entries the only one -- is on that fast
    // path, further terms in a short tail loop
    template <bool ADD>
    __device__ __forceinline__ void scatter_entries(double* dstbuf, int e0, int e1) const {
        const double* vdp = vd_();
        for (int e = e0 + lane; e < e1; e += 256) {
            unsigned p0[4], p1[4];
            int dst[4];
            bool on[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                on[k] = e + 64 * k < e1;
                const int ek = on[k] ? e + 64 * k : e0;
                p0[k] = e_ptr[ek]; p1[k] = e_ptr[ek + 1]; dst[k] = e_dst[ek];
            }
            double wv[4]; int cj[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { wv[k] = t_w[p0[k]]; cj[k] = t_col[p0[k]]; }
            double acc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = wv[k] * vdp[cj[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                for (unsigned p = p0[k] + 1; p < p1[k]; p++) acc[k] = fma(t_w[p], vdp[t_col[p]], acc[k]);
                if (on[k]) { if (ADD) dstbuf[dst[k]] += acc[k]; else dstbuf[dst[k]] = acc[k]; }
            }
        }
    }

    // Off-diagonal blocks of M = A diag(d) A' (d in vd_()) -> U, one staging chunk of <= HB blocks at a time.  The diagonal
    // blocks are NOT kept in registers: factor() rebuilds block K from the tables when its turn comes (diag_from_tables).
    __device__ __forceinline__ void gram() {
        static_for<0, MB>([&](auto Kc) {
            constexpr int K = decltype(Kc)::value;
            static_for<0, G::nch(K)>([&](auto chc) {
                constexpr int ch = decltype(chc)::value;
                constexpr int I0 = K + 1 + HB * ch;
                constexpr int nb = (MB - I0 < HB) ? MB - I0 : HB;
                constexpr int ci = G::chbase(K) + ch;
                const double2_t zero = {0.0, 0.0};
#pragma unroll
                for (int w = 0; w < 2 * nb; w++) ((double2_t*)stage_())[w * 64 + lane] = zero;
                wave_lds_sync();
                scatter_entries<false>(stage_(), __builtin_amdgcn_readfirstlane(meta[META_SEG + ci]),
                                       __builtin_amdgcn_readfirstlane(meta[META_SEG + ci + 1]));
                wave_lds_sync();
#pragma unroll
                for (int bi = 0; bi < nb; bi++) {
                    double4_t blk;
#pragma unroll
                    for (int r = 0; r < 4; r++) blk[r] = stage_()[bi * 256 + 64 * r + lane];
                    park(P[G::bix(K, I0 + bi)], blk);
                }
                wave_lds_sync();
            });
        });
    }

    // tile += diagonal block K of M = A diag(d) A' (strict lower triangle from the entry tables, the diagonal from Md)
    template <int K>
    __device__ __forceinline__ void diag_from_tables(double* tile, const double (&Md)[MR]) const {
        scatter_entries<true>(tile, __builtin_amdgcn_readfirstlane(meta[META_DSEG + K]),
                              __builtin_amdgcn_readfirstlane(meta[META_DSEG + K + 1]));
        if (q == (K & 3)) tile[c16 * 18] += Md[K >> 2];   // row 16K + c16 lives in lane 16(K&3) + c16 of register K>>2
    }

    // W_K element [row 4s + q][column c16] -- the TRANSPOSED operand layout -- from the packed copy in LDS
    template <int K>
    __device__ __forceinline__ double w_elemT(int s) const {
        const int row = 4 * s + q;
        const double v = wl_()[K * WL + ((c16 < row) ? row * (row - 1) / 2 + c16 : 0)];
        return (c16 < row) ? v : ((c16 == row) ? 1.0 : 0.0);
    }

    // Blocked LDL' of the matrix whose off-diagonal blocks are in U; `diag_add(Kc, tile)` adds the original diagonal
    // block K (element [i][k], k <= i, at tile[17 i + k]) to the tile that already holds its Schur update.
    // RELF: pivot floor of column j is flr_()[j] (LDS) instead of floor_.
    // Returns (wave-uniform) whether the Nocedal-Wright guard would have bitten anywhere.
    template <bool RELF, typename DiagAdd>
    __device__ __forceinline__ bool factor(double beta2, double floor_, DiagAdd&& diag_add STAMP_ARGS) {
        // guard verdict, kept as a per-lane integer that every test is folded into AT ONCE (asm pin): left as a boolean the
        // compiler sinks the 16 + 112 compares to the end of the sweep and keeps their operands alive until then
        int viol = 0;
        double* tile = stage_() + TILE_OFF;
        static_for<0, MB>([&](auto Kc) {
            constexpr int K = decltype(Kc)::value;
            // ---- diagonal block K, left-looking: Schur update -sum_{K'<K} (D U_K'K)' U_K'K on the matrix cores ----
            double4_t sch = {0.0, 0.0, 0.0, 0.0};
            static_for<0, K>([&](auto Kp) {
                constexpr int K2 = decltype(Kp)::value;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const double y = unpark(P[G::bix(K2, K)], s);
                    sch = __builtin_amdgcn_mfma_f64_16x16x4f64(-(y * rdv_()[16 * K2 + 4 * s + q]), y, sch, 0, 0, 0);
                }
            });
            // accumulator layout -> tile, + original block -> lane = row (each 16-lane row of the wave a redundant copy)
#pragma unroll
            for (int r = 0; r < 4; r++) tile[(4 * r + q) * 17 + c16] = sch[r];
            wave_lds_sync();
            STAMP(2)
            diag_add(Kc, tile);
            wave_lds_sync();
            STAMP(3)
            double Wd[16], Ld[16];
#pragma unroll
            for (int k = 0; k < 16; k++) Wd[k] = tile[c16 * 17 + k];
            const double myf = RELF ? flr_()[16 * K + c16] : floor_;
            wave_lds_sync();
            double rDr[4] = {1.0, 1.0, 1.0, 1.0}, aDr[4] = {1.0, 1.0, 1.0, 1.0}, rdiag = 1.0, adiag = 1.0;
            static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double u = Wd[j];
                const double piv = row_bcast<j>(u);
                const double aD = fmax(fabs(piv), RELF ? row_bcast<j>(myf) : floor_);
                const double rD = fast_rcp(aD);
                const bool below = c16 > j;
                viol |= (below & (u * u > beta2 * aD)) ? 1 : 0;
                const double li = below ? u * rD : 0.0;
                static_for<j + 1, 16>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    Wd[k] = fma(-li, row_bcast<k>(u), Wd[k]);
                });
                Ld[j] = li;
                rDr[j >> 2] = (q == (j & 3)) ? rD : rDr[j >> 2];
                aDr[j >> 2] = (q == (j & 3)) ? aD : aDr[j >> 2];
                rdiag = (c16 == j) ? rD : rdiag;
                adiag = (c16 == j) ? aD : adiag;
                // select NOW: deferred to the end of the chain (where the scheduler sinks them) the 64 selects keep all
                // 16 pivots and reciprocals alive and the chain spills
                asm volatile("" : "+v"(rDr[j >> 2]), "+v"(aDr[j >> 2]), "+v"(rdiag), "+v"(adiag), "+v"(viol));
            });
            if (q == 0) { rdv_()[16 * K + c16] = rdiag; adv_()[16 * K + c16] = adiag; }
            STAMP(4)
            // ---- W = L_KK^-1 in the A-operand layout: Ws[s] = W[row c16][column 4s + q]; packed copy to LDS ----
            double Ws[4];
#pragma unroll
            for (int s = 0; s < 4; s++) Ws[s] = (c16 == 4 * s + q) ? 1.0 : 0.0;
            static_for<0, 15>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<0, 4>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (4 * s <= j) Ws[s] = fma(-Ld[j], row_bcast<j>(Ws[s]), Ws[s]);
                });
            });
#pragma unroll
            for (int s = 0; s < 4; s++) if (4 * s + q < c16) wl_()[K * WL + c16 * (c16 - 1) / 2 + 4 * s + q] = Ws[s];
            // ---- panel: Y_KI = W M_KI = D_K L_IK' on the matrix cores.  The block stays UNSCALED in its accumulator
            //      registers (every use below is an MFMA operand or folds 1/D into a vector): nothing ever writes a
            //      resident block from the VALU side.  Guard test: Y^2 > beta^2 D. ----
            static_for<K + 1, MB>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; s++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[s], (K == 0) ? unpark(P[G::bix(K, I)], s) : U[G::bix(K, I)][s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) viol |= (acc[r] * acc[r] > beta2 * aDr[r]) ? 1 : 0;
                asm volatile("" : "+v"(viol));
                park(P[G::bix(K, I)], acc);
            });
            STAMP(5)
            // ---- trailing update of the off-diagonal blocks: M_JI -= Y_KJ' D_K^-1 Y_KI, J < I; the A operand
            //      -D^-1 Y_KJ is formed per block row J ----
            static_for<K + 1, MB>([&](auto Jc) {
                constexpr int J = decltype(Jc)::value;
                double yn[4];
#pragma unroll
                for (int r = 0; r < 4; r++) yn[r] = -(unpark(P[G::bix(K, J)], r) * rDr[r]);
                static_for<J + 1, MB>([&](auto Ic) {
                    constexpr int I = decltype(Ic)::value;
                    double4_t acc;
                    if constexpr (K == 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[r] = unpark(P[G::bix(J, I)], r);
                    } else {
                        acc = U[G::bix(J, I)];
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yn[s], unpark(P[G::bix(K, I)], s), acc, 0, 0, 0);
                    U[G::bix(J, I)] = acc;
                });
            });
            STAMP(6)
            __builtin_amdgcn_sched_barrier(0);   // one panel at a time: nothing of panel K+1 is hoisted above this line
        });
        wave_lds_sync();
        return __any(viol != 0);
    }

    // um <- (L D L')^-1 um.  Vectors of a 16-row block appear in two forms: "column form" (lane (c16, q) holds element
    // c16, identical in the four quads) and "row form" (register r of lane (c16, q) holds element 4r + q, identical in the
    // 16 lanes of a quad).  With W_K read in the TRANSPOSED operand layout (element [4s+q][c16]) every product maps one
    // form onto the other with a DPP row reduction or a quad reduction and NO layout conversion through LDS:
    //   forward   t_I = W_I r_I:   r column form -> products -> row_sum  -> t in row form  (what the Y blocks multiply)
    //   backward  x_K = W_K' v_K:  v row form    -> products -> quad_sum -> x in column form (what the Y blocks multiply)
    // Forward substitution is column oriented (t_K, once known, is folded into the partial sums of all later block rows
    // and dropped), backward substitution row oriented: at most 8 + 4 doubles of vector state live.
    __device__ __forceinline__ void solve() {
        double p[MB];
#pragma unroll
        for (int I = 0; I < MB; I++) p[I] = 0.0;
        // forward: t_I = W_I (s_I - sum_{K<I} L_IK t_K), L_IK t_K = Y_KI' (D_K^-1 t_K)
        static_for<0, MB>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            double rC = um_()[16 * I + c16];
            if constexpr (I > 0) rC -= quad_sum(p[I]);
            double tR[4];
#pragma unroll
            for (int s = 0; s < 4; s++) tR[s] = row_sum(w_elemT<I>(s) * rC);
#pragma unroll
            for (int s = 0; s < 4; s++) if (c16 == 0) um_()[16 * I + 4 * s + q] = tR[s];
            if constexpr (I + 1 < MB) {
#pragma unroll
                for (int r = 0; r < 4; r++) tR[r] *= rdv_()[16 * I + 4 * r + q];     // D_I^-1 t_I: the resident blocks are Y = D L'
                static_for<I + 1, MB>([&](auto Jc) {
                    constexpr int J = decltype(Jc)::value;
#pragma unroll
                    for (int r = 0; r < 4; r++) p[J] = fma(unpark(P[G::bix(I, J)], r), tR[r], p[J]);
                });
            }
        });
        wave_lds_sync();
        // backward: x_K = W_K' D_K^-1 (t_K - sum_{I>K} Y_KI x_I)
        double xCL[MB];
        static_for<0, MB>([&](auto Kr) {
            constexpr int K = MB - 1 - decltype(Kr)::value;
            double pr[4] = {0.0, 0.0, 0.0, 0.0};
            static_for<K + 1, MB>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
#pragma unroll
                for (int r = 0; r < 4; r++) pr[r] = fma(unpark(P[G::bix(K, I)], r), xCL[I], pr[r]);
            });
            double px = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = um_()[16 * K + 4 * r + q];
                if constexpr (K < MB - 1) v -= row_sum(pr[r]);
                px = fma(w_elemT<K>(r), v * rdv_()[16 * K + 4 * r + q], px);
            }
            xCL[K] = quad_sum(px);
        });
        wave_lds_sync();
#pragma unroll
        for (int K = 0; K < MB; K++) if (q == 0) um_()[16 * K + c16] = xCL[K];
        wave_lds_sync();
    }
};

template <int MB, int NQ>
__device__ __forceinline__ void wreg_carve(WReg<MB, NQ>& w, double* W0, int tid) {
    using G = WGeo<MB>;
    w.W0 = W0;
    w.lane = tid & 63; w.q = w.lane >> 4; w.c16 = w.lane & 15;
}

template <int MB, int NQ>
__device__ __forceinline__ void wreg_setup(WReg<MB, NQ>& w, const WregTab& T, unsigned char* lraw, int tid) {
    using G = WGeo<MB>;
    double* s_csr_val = (double*)(lraw + T.o_csr_val);
    double* s_ec_val = (double*)(lraw + T.o_ec_val);
    double* s_t_w = (double*)(lraw + T.o_t_w);
    unsigned* s_e_ptr = (unsigned*)(lraw + T.o_e_ptr);
    int* s_meta = (int*)(lraw + T.o_meta);
    unsigned short* s_csr_col = (unsigned short*)(lraw + T.o_csr_col);
    unsigned short* s_csr_ptr = (unsigned short*)(lraw + T.o_csr_ptr);
    unsigned short* s_csr_len = (unsigned short*)(lraw + T.o_csr_len);
    unsigned short* s_ec_row = (unsigned short*)(lraw + T.o_ec_row);
    unsigned short* s_colmap = (unsigned short*)(lraw + T.o_colmap);
    unsigned short* s_e_dst = (unsigned short*)(lraw + T.o_e_dst);
    unsigned short* s_t_col = (unsigned short*)(lraw + T.o_t_col);
    const int nth = blockDim.x;
    for (int i = tid; i < T.nnz; i += nth) {
        s_csr_val[i] = T.csr_val[i]; s_csr_col[i] = T.csr_col[i];
    }
    for (int i = tid; i < G::MPL; i += nth) { s_csr_ptr[i] = T.csr_ptr[i]; s_csr_len[i] = T.csr_len[i]; }
    for (int i = tid; i < T.ctot * 64; i += nth) { s_ec_val[i] = T.ec_val[i]; s_ec_row[i] = T.ec_row[i]; }
    for (int i = tid; i < 64 * NQ; i += nth) s_colmap[i] = T.colmap[i];
    for (int i = tid; i < T.n_term; i += nth) { s_t_w[i] = T.t_w[i]; s_t_col[i] = T.t_col[i]; }
    for (int i = tid; i < T.n_ent; i += nth) s_e_dst[i] = T.e_dst[i];
    for (int i = tid; i <= T.n_ent; i += nth) s_e_ptr[i] = T.e_ptr[i];
    for (int i = tid; i < META_N; i += nth) s_meta[i] = T.meta[i];
    __syncthreads();
    w.csr_val = s_csr_val; w.csr_col = s_csr_col; w.csr_ptr = s_csr_ptr; w.csr_len = s_csr_len;
    w.ec_val = s_ec_val; w.ec_row = s_ec_row; w.colmap = s_colmap;
    w.e_ptr = s_e_ptr; w.e_dst = s_e_dst; w.t_w = s_t_w; w.t_col = s_t_col; w.meta = s_meta;
    wreg_carve(w, (double*)(lraw + T.o_wave) + (size_t)(tid >> 6) * T.wave_doubles, tid);
    w.m = T.m; w.n = T.n; w.rmax = T.rmax;
}

// Newton step of the primal normal equations for the point (x, z, y) of this wave's LP (ldl.cl:656-712 with the x-space
// refinement of oracle newton_dy):  M dy = A(d t) - rho,  dx = d (t - A'dy),  then  e = rho - A dx;  M eta = e;
// dx += d A'eta;  dy -= eta  while max|e| > etol, at most max_refine times.  The first solve is written as pass 0 of that
// loop so that the kernel holds ONE copy of the (fully unrolled) block substitution.
// In: t (per column, parked in the stage), rho (per row), um = A(d t) - rho in LDS, the factor in w.U / w.wl_().
// Out: dy (per row), dx, wv = A'dy.  Returns the refinement passes used; `bad` reports a non-finite dy.
template <int MB, int NQ>
__device__ __forceinline__ int newton_solve(WReg<MB, NQ>& w, const double (&x)[NQ], const double (&z)[NQ],
                                            const bool (&okc)[NQ], const bool (&okr)[WGeo<MB>::MR],
                                            const double (&rho)[WGeo<MB>::MR], double etol, int max_refine,
                                            double (&dy)[WGeo<MB>::MR], double (&dx)[NQ], double (&wv)[NQ], bool& bad STAMP_ARGS) {
    constexpr int MR = WGeo<MB>::MR, MP = WGeo<MB>::MP;
    const int lane = w.lane;
    double* vx = w.stage_();
    double d[NQ];
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) d[qq] = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;
    int pass = 0;
    bad = false;
    for (;;) {
        w.solve();
        STAMP(7)
        double w2[NQ];
        w.At(w.um_(), w2);
        if (pass == 0) {
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const double tq = w.stage_()[lane + 64 * qq];       // t, parked there by the caller
                wv[qq] = w2[qq];
                dx[qq] = (tq - w2[qq]) * d[qq];
            }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                dy[r2] = (lane + 64 * r2 < MP) ? w.um_()[lane + 64 * r2] : 0.0;
                bad = bad | !isfinite(dy[r2]);
            }
        } else {
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) { dx[qq] = fma(d[qq], w2[qq], dx[qq]); wv[qq] -= w2[qq]; }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) dy[r2] -= (lane + 64 * r2 < MP) ? w.um_()[lane + 64 * r2] : 0.0;
        }
        wave_lds_sync();
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = okc[qq] ? dx[qq] : 0.0;
        wave_lds_sync();
        double Adx[MR], e[MR], dummy[MR], me = 0.0;
        w.template Arow<false>(vx, Adx, dummy);
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            e[r2] = okr[r2] ? rho[r2] - Adx[r2] : 0.0;
            me = fmax(me, fabs(e[r2]));
        }
        const double maxe = wmax(me);
        STAMP(8)
        if (!(maxe > etol) || pass >= max_refine) break;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) if (lane + 64 * r2 < MP) w.um_()[lane + 64 * r2] = e[r2];
        wave_lds_sync();
        pass++;
    }
    bad = __any(bad);
    return pass;
}

// ------------------------------------------------------------------------------------------------------------------
// solve kernel: sparse_standard_primal_normal (primal_normal.cl:287-375), one LP per wavefront
// ------------------------------------------------------------------------------------------------------------------
template <int MB, int NQ>
__global__ void __launch_bounds__(256, 1)
ipm_wreg_kernel(WregTab T, long B, const double* __restrict__ bg, const double* __restrict__ cg,
                double* __restrict__ xg, double* __restrict__ yg, double* __restrict__ zg, double* __restrict__ pobj,
                double* __restrict__ dobj, int* __restrict__ status, int* __restrict__ iters, int* __restrict__ queue,
                int* __restrict__ defer, DevOpts o) {
    using G = WGeo<MB>;
    constexpr int MR = G::MR, MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, NQ> w;
    USE_AGPR_FORM();
    wreg_setup(w, T, lraw, threadIdx.x);
    const int lane = w.lane;
    const int m = w.m, n = w.n;
    const bool warm = (o.flags & PYCLLP_FLAG_WARM_START) != 0;
    const double nm = (double)(n + m);
    double* vx = w.stage_();
    bool okc[NQ], okr[MR];
    int jc[NQ];       // the column that lives at position lane + 64 q of the N-vectors
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) { okc[qq] = lane + 64 * qq < n; jc[qq] = w.colmap[lane + 64 * qq]; }
#pragma unroll
    for (int r2 = 0; r2 < MR; r2++) okr[r2] = lane + 64 * r2 < m;

    long lp;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
    STAMP_DECL
    while (lp < B) {
        double x[NQ], z[NQ];
        double c2 = 0.0;
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const int j = jc[qq];
            const double cj = okc[qq] ? cg[lp * n + j] : 0.0;
            c2 = fma(cj, cj, c2);
            x[qq] = (warm && okc[qq]) ? xg[lp * n + j] : 1.0;
            z[qq] = (warm && okc[qq]) ? zg[lp * n + j] : 1.0;
        }
        double b2 = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            const double bi = okr[r2] ? bg[lp * m + i] : 0.0;
            b2 = fma(bi, bi, b2);
            if (i < MP) {
                w.bs_()[i] = bi;
                w.ys_()[i] = okr[r2] ? ((warm && yg) ? yg[lp * m + i] : 1.0) : 0.0;
            }
        }
        wave_lds_sync();
        const double nb2 = wsum(b2), nc2 = wsum(c2);
        const double tol_r = uni(o.eps * (1.0 + sqrt(nb2))), tol_s = uni(o.eps * (1.0 + sqrt(nc2)));
        const double etol = uni(o.refine_tol * (1.0 + sqrt(nb2)));
        double normr0 = 1e300, norms0 = 1e300, po = 0.0, du = 0.0;
        int stat = PYCLLP_STATUS_ITERATION_LIMIT, it = 0;
        bool running = true;

        while (running) {
            // ---- sigma, gamma, objectives (primal_normal.cl:96-120, 245-248) ----
            double v[NQ], cq[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) cq[qq] = okc[qq] ? cg[lp * n + jc[qq]] : 0.0;   // in flight (vmcnt) while A'y runs on LDS
            w.At(w.ys_(), v);
            double s2 = 0.0, gam = 0.0, pp = 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const double sg = okc[qq] ? cq[qq] - v[qq] + z[qq] : 0.0;
                s2 = fma(sg, sg, s2);
                gam += okc[qq] ? x[qq] * z[qq] : 0.0;
                pp += cq[qq] * (okc[qq] ? x[qq] : 0.0);
            }
            double dd = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                dd += (i < MP) ? w.bs_()[i] * w.ys_()[i] : 0.0;
            }
            s2 = wsum(s2); gam = wsum(gam); po = wsum(pp); du = wsum(dd);
            const double norms = uni(sqrt(s2));
            const double mu = uni(o.delta * gam / nm);
            // ---- d, t; rho = b - A x (primal_normal.cl:50-74) ----
            double t[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int j = lane + 64 * qq;
                const double dq = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;      // v_rcp_f64 + 2 Newton steps (<= 2 ulp), as
                t[qq] = okc[qq] ? cq[qq] - v[qq] + mu * fast_rcp(x[qq]) : 0.0;   // the dense group kernel
                vx[j] = okc[qq] ? x[qq] : 0.0;
                w.vd_()[j] = dq;
            }
            wave_lds_sync();
            double rho[MR], Ax[MR], Md[MR];
            w.template Arow<false>(vx, Ax, Md);
            double r2s = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                rho[r2] = okr[r2] ? w.bs_()[i] - Ax[r2] : 0.0;
                r2s = fma(rho[r2], rho[r2], r2s);
            }
            const double normr = uni(sqrt(wsum(r2s)));
            // ---- stop tests (primal_normal.cl:256-269; oracle ipm_one_path) ----
            if (!(isfinite(normr) && isfinite(norms) && isfinite(gam))) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
            else if (normr <= tol_r && norms <= tol_s && gam <= o.eps * (1.0 + fabs(po))) { stat = PYCLLP_STATUS_OPTIMAL; running = false; }
            else if (normr > 10.0 * normr0 && normr > PYCLLP_GROWTH_FLOOR * tol_r) { stat = PYCLLP_STATUS_PRIMAL_INFEASIBLE; running = false; }
            else if (norms > 10.0 * norms0 && norms > PYCLLP_GROWTH_FLOOR * tol_s) { stat = PYCLLP_STATUS_DUAL_INFEASIBLE; running = false; }
            if (running) {
                // ---- rhs = A (d t) - rho, diag(M) ----
                wave_lds_sync();
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * t[qq];
                wave_lds_sync();
                double Adt[MR];
                w.template Arow<true>(vx, Adt, Md);
                double bmax = 0.0;
#pragma unroll
                for (int r2 = 0; r2 < MR; r2++) {
                    const int i = lane + 64 * r2;
                    if (i < MP) w.um_()[i] = okr[r2] ? Adt[r2] - rho[r2] : 0.0;
                    bmax = fmax(bmax, okr[r2] ? fabs(Md[r2]) : 0.0);
                }
                const double beta2 = wmax(bmax);     // ldl.cl:296-311
                wave_lds_sync();
                STAMP(0)
                // ---- M = A diag(d) A' into registers, t parked in the stage, factor ----
                w.gram();
                STAMP(1)
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) w.stage_()[lane + 64 * qq] = t[qq];
                const bool viol = w.template factor<false>(beta2, o.pivot_floor, [&](auto Kc, double* tile) {
                    w.template diag_from_tables<decltype(Kc)::value>(tile, Md); } STAMP_PASS);
                if (viol || (o.flags & PYCLLP_FLAG_FORCE_GUARD_PATH)) { stat = -1; running = false; }
                else {
                    double dy[MR], wv[NQ], dx[NQ];
                    bool bad;
                    (void)newton_solve(w, x, z, okc, okr, rho, etol, o.max_refine, dy, dx, wv, bad STAMP_PASS);
                    if (bad) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
                    else {
                        // ---- step (primal_normal.cl:158-198) ----
                        double dz[NQ], th = 0.0;
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double rx = fast_rcp(x[qq]), rz = fast_rcp(z[qq]);
                            dz[qq] = okc[qq] ? (mu - z[qq] * dx[qq]) * rx - z[qq] : 0.0;
                            if (okc[qq]) th = fmax(th, fmax(-dz[qq] * rz, -dx[qq] * rx));
                        }
                        th = wmax(th);
                        const double theta = uni(fmin(o.r / th, 1.0));
                        wave_lds_sync();
#pragma unroll
                        for (int r2 = 0; r2 < MR; r2++) {
                            const int i = lane + 64 * r2;
                            if (i < MP) w.ys_()[i] = fma(theta, dy[r2], w.ys_()[i]);
                        }
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) { x[qq] = fma(theta, dx[qq], x[qq]); z[qq] = fma(theta, dz[qq], z[qq]); }
                        normr0 = normr; norms0 = norms;
                        wave_lds_sync();
                        it++;
                        if (it >= o.max_iter) running = false;   // status stays ITERATION_LIMIT
                        STAMP(9)
                    }
                }
            }
        }
        wave_lds_sync();
        if (stat == -1) {   // the guard would have bitten: hand the LP to the guarded kernel
            if (lane == 0) { const int k = atomicAdd(defer, 1); defer[1 + k] = (int)lp; status[lp] = -1; }
        } else {
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int j = jc[qq];
                if (okc[qq]) { xg[lp * n + j] = x[qq]; if (zg) zg[lp * n + j] = z[qq]; }
            }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                if (yg && okr[r2]) yg[lp * m + i] = w.ys_()[i];
            }
            if (lane == 0) {
                if (pobj) pobj[lp] = po;
                if (dobj) dobj[lp] = du;
                status[lp] = stat;
                if (iters) iters[lp] = it;
            }
        }
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
        STAMP(9)
    }
    STAMP_FLUSH(o, blockIdx.x * 4 + (threadIdx.x >> 6))
}

// ------------------------------------------------------------------------------------------------------------------
// the same solve on the homogeneous self-dual embedding (PYCLLP_FLAG_HSD; oracle hsd_one_raw, ipm_block_kernel's run-time
// branch, csrc/ipm_group_hsd.inc): tau and kappa are wave-uniform scalars, one factorisation serves the two right-hand
// sides  M p = A(d c) - b  and  M q = A(d r1) - eta rho, the pivot floor of column j is pivot_floor^2 |M_jj|
// ------------------------------------------------------------------------------------------------------------------
template <int MB, int NQ>
__global__ void __launch_bounds__(256, 1)
hsd_wreg_kernel(WregTab T, long B, const double* __restrict__ bg, const double* __restrict__ cg,
                double* __restrict__ xg, double* __restrict__ yg, double* __restrict__ zg, double* __restrict__ pobj,
                double* __restrict__ dobj, int* __restrict__ status, int* __restrict__ iters, int* __restrict__ queue,
                int* __restrict__ defer, DevOpts o) {
    using G = WGeo<MB>;
    constexpr int MR = G::MR, MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, NQ> w;
    USE_AGPR_FORM();
    wreg_setup(w, T, lraw, threadIdx.x);
    const int lane = w.lane;
    const int m = w.m, n = w.n;
    const bool warm = (o.flags & PYCLLP_FLAG_WARM_START) != 0;
    const double eta = 1.0 - o.delta, einf = 100.0 * o.eps;
    double* vx = w.stage_();
    double* pv = w.flr_();          // p = M^-1 (A(d c) - b): the floor vector is dead once the factor exists
    bool okc[NQ], okr[MR];
    int jc[NQ];       // the column that lives at position lane + 64 q of the N-vectors
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) { okc[qq] = lane + 64 * qq < n; jc[qq] = w.colmap[lane + 64 * qq]; }
#pragma unroll
    for (int r2 = 0; r2 < MR; r2++) okr[r2] = lane + 64 * r2 < m;

    long lp;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
    STAMP_DECL
    while (lp < B) {
        double x[NQ], z[NQ];
        double c2 = 0.0, g0 = 0.0;
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const int j = jc[qq];
            const double cj = okc[qq] ? cg[lp * n + j] : 0.0;
            c2 = fma(cj, cj, c2);
            x[qq] = (warm && okc[qq]) ? xg[lp * n + j] : 1.0;
            z[qq] = (warm && okc[qq]) ? zg[lp * n + j] : 1.0;
            g0 += okc[qq] ? x[qq] * z[qq] : 0.0;
        }
        double b2 = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            const double bi = okr[r2] ? bg[lp * m + i] : 0.0;
            b2 = fma(bi, bi, b2);
            if (i < MP) {
                w.bs_()[i] = bi;
                w.ys_()[i] = (okr[r2] && warm && yg) ? yg[lp * m + i] : 0.0;
            }
        }
        wave_lds_sync();
        const double nbn = uni(sqrt(wsum(b2))), ncn = uni(sqrt(wsum(c2)));
        const double tol_r = uni(o.eps * (1.0 + nbn)), tol_s = uni(o.eps * (1.0 + ncn));
        double tau = 1.0, kap = 1.0;
        if (warm) kap = uni(wsum(g0) / (double)n);
        double po = 0.0, du = 0.0;
        int stat = PYCLLP_STATUS_ITERATION_LIMIT, it = 0;
        bool running = true;

        while (running) {
            // ---- sigma = c tau - A'y + z, gamma, objectives ----
            double v[NQ], cq[NQ], sg[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) cq[qq] = okc[qq] ? cg[lp * n + jc[qq]] : 0.0;   // in flight (vmcnt) while A'y runs on LDS
            w.At(w.ys_(), v);
            double s2 = 0.0, gam = 0.0, pp = 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                sg[qq] = okc[qq] ? cq[qq] * tau - v[qq] + z[qq] : 0.0;
                s2 = fma(sg[qq], sg[qq], s2);
                gam += okc[qq] ? x[qq] * z[qq] : 0.0;
                pp += cq[qq] * (okc[qq] ? x[qq] : 0.0);
            }
            double dd = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                dd += (i < MP) ? w.bs_()[i] * w.ys_()[i] : 0.0;
            }
            s2 = wsum(s2); gam = wsum(gam); po = wsum(pp); du = wsum(dd);
            const double norms = uni(sqrt(s2));
            const double mu = uni(o.delta * (gam + tau * kap) / (double)(n + 1));
            const double phi = uni(du - po + kap);
            // ---- d, t = r1 = mu/x - z + eta sigma; rho = b tau - A x ----
            double t[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int j = lane + 64 * qq;
                const double dq = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;
                t[qq] = okc[qq] ? fma(eta, sg[qq], mu * fast_rcp(x[qq]) - z[qq]) : 0.0;
                vx[j] = okc[qq] ? x[qq] : 0.0;
                w.vd_()[j] = dq;
            }
            wave_lds_sync();
            double rho[MR], Ax[MR], Md[MR];
            w.template Arow<false>(vx, Ax, Md);
            double r2s = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                rho[r2] = okr[r2] ? w.bs_()[i] * tau - Ax[r2] : 0.0;
                r2s = fma(rho[r2], rho[r2], r2s);
            }
            const double normr = uni(sqrt(wsum(r2s)));
            // ---- stop tests (oracle hsd_one_raw): optimal, or a primal / dual ray ----
            const bool p_ray = po > 0.0 && fma(nbn, tau, normr) <= einf * po;
            const bool d_ray = du < 0.0 && fma(ncn, tau, norms) <= einf * -du;
            if (!(isfinite(normr) && isfinite(norms) && isfinite(gam) && isfinite(tau) && isfinite(kap))) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
            else if (normr <= tol_r * tau && norms <= tol_s * tau && gam <= o.eps * tau * (tau + fabs(po))) { stat = PYCLLP_STATUS_OPTIMAL; running = false; }
            else if (p_ray || d_ray) {
                stat = (p_ray && d_ray) ? ((-du > po) ? PYCLLP_STATUS_PRIMAL_INFEASIBLE : PYCLLP_STATUS_DUAL_INFEASIBLE)
                                        : (p_ray ? PYCLLP_STATUS_DUAL_INFEASIBLE : PYCLLP_STATUS_PRIMAL_INFEASIBLE);
                running = false;
            }
            if (running) {
                // ---- A(d r1), diag(M), A(d c) ----
                wave_lds_sync();
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * t[qq];
                wave_lds_sync();
                double Adt[MR], Adc[MR], dummy[MR];
                w.template Arow<true>(vx, Adt, Md);
                wave_lds_sync();
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * cq[qq];
                wave_lds_sync();
                w.template Arow<false>(vx, Adc, dummy);
                double rq[MR], bmax = 0.0;
#pragma unroll
                for (int r2 = 0; r2 < MR; r2++) {
                    const int i = lane + 64 * r2;
                    rq[r2] = okr[r2] ? fma(-eta, rho[r2], Adt[r2]) : 0.0;
                    if (i < MP) {
                        w.um_()[i] = okr[r2] ? Adc[r2] - w.bs_()[i] : 0.0;                   // right-hand side of p
                        w.flr_()[i] = o.pivot_floor * o.pivot_floor * fabs(Md[r2]);         // floor of column i
                    }
                    bmax = fmax(bmax, okr[r2] ? fabs(Md[r2]) : 0.0);
                }
                const double beta2 = uni(wmax(bmax));
                wave_lds_sync();
                STAMP(0)
                w.gram();
                STAMP(1)
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) w.stage_()[lane + 64 * qq] = t[qq];
                const bool viol = w.template factor<true>(beta2, 0.0, [&](auto Kc, double* tile) {
                    w.template diag_from_tables<decltype(Kc)::value>(tile, Md); } STAMP_PASS);
                if (viol || (o.flags & PYCLLP_FLAG_FORCE_GUARD_PATH)) { stat = -1; running = false; }
                else {
                    // ---- one loop around the ONE copy of the block substitution: pass 0 solves for p, pass 1 for q and
                    //      combines them through dtau, the following passes are the x-space refinement ----
                    double d[NQ], dx[NQ], u[NQ] /* c - A'p */, dy[MR], rhot[MR];
#pragma unroll
                    for (int qq = 0; qq < NQ; qq++) d[qq] = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;
                    double dtau = 0.0, etol_it = 0.0;
                    bool bad = false;
                    int pass = 0;
                    for (;;) {
                        w.solve();
                        STAMP(7)
                        double w2[NQ];
                        w.At(w.um_(), w2);
                        bool more = true;
                        if (pass == 0) {
                            // c - A'p is kept; p moves to pv, q's right-hand side into um
#pragma unroll
                            for (int qq = 0; qq < NQ; qq++) u[qq] = cq[qq] - w2[qq];
                            wave_lds_sync();
#pragma unroll
                            for (int r2 = 0; r2 < MR; r2++) {
                                const int i = lane + 64 * r2;
                                if (i < MP) { pv[i] = w.um_()[i]; w.um_()[i] = rq[r2]; }
                            }
                            wave_lds_sync();
                        } else {
                            if (pass == 1) {
                                double dsum = 0.0, nsum = 0.0, bq = 0.0;
#pragma unroll
                                for (int qq = 0; qq < NQ; qq++) {
                                    const double tq = w.stage_()[lane + 64 * qq];
                                    dx[qq] = d[qq] * (tq - w2[qq]);                       // v = d (r1 - A'q)
                                    dsum = fma(d[qq] * u[qq], u[qq], dsum);               // |sqrt(d)(c - A'p)|^2
                                    nsum = fma(cq[qq], dx[qq], nsum);                     // c'v
                                }
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) {
                                    const int i = lane + 64 * r2;
                                    bq += (i < MP) ? w.bs_()[i] * w.um_()[i] : 0.0;       // b'q
                                }
                                const double den = wsum(dsum) + kap / tau;
                                const double num = fma(eta, phi, mu / tau - kap) + wsum(bq) - wsum(nsum);
                                dtau = uni(num / den);
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) {
                                    const int i = lane + 64 * r2;
                                    dy[r2] = (i < MP) ? fma(pv[i], dtau, w.um_()[i]) : 0.0;
                                    rhot[r2] = okr[r2] ? fma(w.bs_()[i], dtau, eta * rho[r2]) : 0.0;   // A dx - b dtau = eta rho
                                    bad = bad | !isfinite(dy[r2]);
                                }
#pragma unroll
                                for (int qq = 0; qq < NQ; qq++) dx[qq] = fma(d[qq] * u[qq], dtau, dx[qq]);   // dx = u dtau + v, u = d (c - A'p)
                                etol_it = uni(o.refine_tol * (1.0 + nbn) * fmax(tau, kap));
                            } else {
#pragma unroll
                                for (int qq = 0; qq < NQ; qq++) dx[qq] = fma(d[qq], w2[qq], dx[qq]);
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) dy[r2] -= (lane + 64 * r2 < MP) ? w.um_()[lane + 64 * r2] : 0.0;
                            }
                            wave_lds_sync();
#pragma unroll
                            for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = okc[qq] ? dx[qq] : 0.0;
                            wave_lds_sync();
                            double Adx[MR], e[MR], dm[MR], me = 0.0;
                            w.template Arow<false>(vx, Adx, dm);
#pragma unroll
                            for (int r2 = 0; r2 < MR; r2++) {
                                e[r2] = okr[r2] ? rhot[r2] - Adx[r2] : 0.0;
                                me = fmax(me, fabs(e[r2]));
                            }
                            const double maxe = wmax(me);
                            STAMP(8)
                            if (!(maxe > etol_it) || pass - 1 >= o.max_refine) more = false;
                            else {
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) if (lane + 64 * r2 < MP) w.um_()[lane + 64 * r2] = e[r2];
                                wave_lds_sync();
                            }
                        }
                        if (!more) break;
                        pass++;
                    }
                    if (__any(bad) || !isfinite(dtau)) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
                    else {
                        // ---- step: ratio test over x, z, tau, kappa ----
                        const double dkap = mu / tau - kap - kap / tau * dtau;
                        double dz[NQ], th = fmax(fmax(-dtau / tau, -dkap / kap), 0.0);
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double rx = fast_rcp(x[qq]), rz = fast_rcp(z[qq]);
                            dz[qq] = okc[qq] ? (mu - z[qq] * dx[qq]) * rx - z[qq] : 0.0;
                            if (okc[qq]) th = fmax(th, fmax(-dz[qq] * rz, -dx[qq] * rx));
                        }
                        th = wmax(th);
                        const double theta = uni(fmin(o.r / th, 1.0));
                        wave_lds_sync();
#pragma unroll
                        for (int r2 = 0; r2 < MR; r2++) {
                            const int i = lane + 64 * r2;
                            if (i < MP) w.ys_()[i] = fma(theta, dy[r2], w.ys_()[i]);
                        }
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) { x[qq] = fma(theta, dx[qq], x[qq]); z[qq] = fma(theta, dz[qq], z[qq]); }
                        tau = uni(fma(theta, dtau, tau)); kap = uni(fma(theta, dkap, kap));
                        wave_lds_sync();
                        it++;
                        if (it >= o.max_iter) running = false;
                        STAMP(9)
                    }
                }
            }
        }
        wave_lds_sync();
        if (stat == -1) {
            if (lane == 0) { const int k = atomicAdd(defer, 1); defer[1 + k] = (int)lp; status[lp] = -1; }
        } else {
            // optimal (and iteration-limit) points leave the homogeneous scaling (hsd.c:266-273); certificates stay
            const double rt = (stat == PYCLLP_STATUS_OPTIMAL || stat == PYCLLP_STATUS_ITERATION_LIMIT) ? 1.0 / tau : 1.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int j = jc[qq];
                if (okc[qq]) { xg[lp * n + j] = x[qq] * rt; if (zg) zg[lp * n + j] = z[qq] * rt; }
            }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                if (yg && okr[r2]) yg[lp * m + i] = w.ys_()[i] * rt;
            }
            if (lane == 0) {
                if (pobj) pobj[lp] = po * rt;
                if (dobj) dobj[lp] = du * rt;
                status[lp] = stat;
                if (iters) iters[lp] = it;
            }
        }
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
        STAMP(9)
    }
    STAMP_FLUSH(o, blockIdx.x * 4 + (threadIdx.x >> 6))
}

// ------------------------------------------------------------------------------------------------------------------
// stand-alone Newton step: sparse_solve_primal_normal (ldl.cl:656-712) as launched by the reference's
// tests/test_ldl.py:276-361, one state per wavefront
// ------------------------------------------------------------------------------------------------------------------
template <int MB, int NQ>
__global__ void __launch_bounds__(256, 1)
newton_wreg_kernel(WregTab T, long B, const double* __restrict__ xg, const double* __restrict__ zg,
                   const double* __restrict__ yg, const double* __restrict__ bg, const double* __restrict__ cg, double mu,
                   double* __restrict__ dyg, int* __restrict__ nrefg, int* __restrict__ queue, DevOpts o) {
    using G = WGeo<MB>;
    constexpr int MR = G::MR, MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, NQ> w;
    USE_AGPR_FORM();
    wreg_setup(w, T, lraw, threadIdx.x);
    const int lane = w.lane, m = w.m, n = w.n;
    double* vx = w.stage_();
    bool okc[NQ], okr[MR];
    int jc[NQ];       // the column that lives at position lane + 64 q of the N-vectors
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) { okc[qq] = lane + 64 * qq < n; jc[qq] = w.colmap[lane + 64 * qq]; }
#pragma unroll
    for (int r2 = 0; r2 < MR; r2++) okr[r2] = lane + 64 * r2 < m;
    long lp;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
    while (lp < B) {
        double x[NQ], z[NQ], t[NQ], v[NQ];
        double b2 = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            const double bi = okr[r2] ? bg[lp * m + i] : 0.0;
            b2 = fma(bi, bi, b2);
            if (i < MP) { w.bs_()[i] = bi; w.ys_()[i] = okr[r2] ? yg[lp * m + i] : 0.0; }
        }
        wave_lds_sync();
        const double etol = o.refine_tol * (1.0 + sqrt(wsum(b2)));
        w.At(w.ys_(), v);
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const int j = lane + 64 * qq, jg = jc[qq];
            x[qq] = okc[qq] ? xg[lp * n + jg] : 1.0;
            z[qq] = okc[qq] ? zg[lp * n + jg] : 1.0;
            const double cj = okc[qq] ? cg[lp * n + jg] : 0.0;
            t[qq] = okc[qq] ? cj - v[qq] + mu * fast_rcp(x[qq]) : 0.0;
            vx[j] = okc[qq] ? x[qq] : 0.0;
            w.vd_()[j] = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;
        }
        wave_lds_sync();
        double rho[MR], Ax[MR], Adt[MR], Md[MR];
        w.template Arow<false>(vx, Ax, Md);
        wave_lds_sync();
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * t[qq];
        wave_lds_sync();
        w.template Arow<true>(vx, Adt, Md);
        double bmax = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            rho[r2] = okr[r2] ? w.bs_()[i] - Ax[r2] : 0.0;
            if (i < MP) w.um_()[i] = okr[r2] ? Adt[r2] - rho[r2] : 0.0;
            bmax = fmax(bmax, okr[r2] ? fabs(Md[r2]) : 0.0);
        }
        const double beta2 = wmax(bmax);
        wave_lds_sync();
        w.gram();
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) w.stage_()[lane + 64 * qq] = t[qq];
#ifdef PYCLLP_PROFILE
        unsigned long long t_prev_ = 0, t_acc_[NPHASE] = {0};
#endif
        (void)w.template factor<false>(beta2, o.pivot_floor, [&](auto Kc, double* tile) {
            w.template diag_from_tables<decltype(Kc)::value>(tile, Md); } STAMP_PASS);
        double dy[MR], wv[NQ], dx[NQ];
        bool bad;
        const int nref = newton_solve(w, x, z, okc, okr, rho, etol, o.max_refine, dy, dx, wv, bad STAMP_PASS);
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) if (okr[r2]) dyg[lp * m + lane + 64 * r2] = dy[r2];
        if (nrefg && lane == 0) nrefg[lp] = nref;
        wave_lds_sync();
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// stand-alone LDL' solve of explicit dense symmetric matrices: the register factor + block substitution on their own
// (pycllp/ldl.py:202-239 solve_ldl; also the bring-up check of factor()/solve())
// ------------------------------------------------------------------------------------------------------------------
template <int MB>
__global__ void __launch_bounds__(256, 1)
ldl_solve_wreg_kernel(int n, long B, const double* __restrict__ Ag, const double* __restrict__ rhs, double* __restrict__ out,
                      double floor_, int* __restrict__ queue) {
    using G = WGeo<MB>;
    constexpr int MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, 1> w;
    USE_AGPR_FORM();
    const int tid = threadIdx.x;
    wreg_carve(w, (double*)lraw + (size_t)(tid >> 6) * G::WAVE_D(1), tid);
    const int lane = w.lane, q = w.q, c16 = w.c16;
    long mat;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        mat = __builtin_amdgcn_readfirstlane(nxt);
    }
    while (mat < B) {
        const double* A = Ag + mat * (long)n * n;
        // off-diagonal blocks: U[K][I] register r = M[i = 16I + c16][k = 16K + 4r + q] (i > k); padded rows are zero
        static_for<0, MB>([&](auto Kc) {
            constexpr int K = decltype(Kc)::value;
            static_for<K + 1, MB>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                double4_t blk;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 16 * I + c16, k = 16 * K + 4 * r + q;
                    blk[r] = (i < n) ? A[(long)i * n + k] : 0.0;
                }
                park(w.P[G::bix(K, I)], blk);
            });
        });
        for (int i = lane; i < MP; i += 64) w.um_()[i] = (i < n) ? rhs[mat * n + i] : 0.0;
        wave_lds_sync();
#ifdef PYCLLP_PROFILE
        unsigned long long t_prev_ = 0, t_acc_[NPHASE] = {0};
#endif
        // diagonal blocks come straight from memory when their turn comes: element [row][col], col <= row, of block K
        (void)w.template factor<false>(1e300, floor_, [&](auto Kc, double* tile) {
            constexpr int K = decltype(Kc)::value;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * K + 4 * r + q, col = 16 * K + c16;
                double v = 0.0;
                if (col <= row) v = (row < n) ? A[(long)row * n + col] : ((row == col) ? 1.0 : 0.0);
                tile[(4 * r + q) * 17 + c16] += v;
            }
        } STAMP_PASS);
        w.solve();
        for (int i = lane; i < n; i += 64) out[mat * n + i] = w.um_()[i];
        wave_lds_sync();
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        mat = __builtin_amdgcn_readfirstlane(nxt);
    }
}

// ---- selftest of the cross-lane primitives (bring-up aid) --------------------------------------------------------
__global__ void wreg_selftest_kernel(double* out) {
    const int lane = threadIdx.x & 63;
    const double v = 1.0 + lane;
    out[lane] = quad_sum(v);                 // expect sum over l' = l mod 16 + 16 k
    out[64 + lane] = row_sum(v);             // expect sum over the 16-lane row
    out[128 + lane] = wsum(v);               // expect 2080
    out[192 + lane] = row_bcast<5>(v);       // expect 1 + (lane & ~15) + 5
    // MFMA layout: A[m][k] = 100 m + k (k = 0..3), B[k][n] = (k == 1) ? n + 1 : 0  ->  C[m][n] = (100 m + 1)(n + 1)
    const int c16 = lane & 15, qd = lane >> 4;
    const double a = 100.0 * c16 + qd, b = (qd == 1) ? c16 + 1.0 : 0.0;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[256 + 64 * r + lane] = acc[r];   // expect (100 (4r + q) + 1)(c16 + 1)
    out[512 + lane] = wmax(v);               // expect 64
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
struct WregPlan {
    WregTab tab;
    int mb, nq;
    void* dev_blob;
};

namespace {

template <typename T>
size_t put(std::vector<char>& host, const std::vector<T>& v) {
    size_t off = (host.size() + 15) & ~(size_t)15;
    host.resize(off + v.size() * sizeof(T));
    if (!v.empty()) memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

int nch_host(int MB, int K) { return (MB - 1 - K + HB - 1) / HB; }

typedef hipError_t (*wsolve_fn)(const WregTab&, long, const double*, const double*, double*, double*, double*, double*,
                                double*, int*, int*, int*, int*, DevOpts, int, hipStream_t);
typedef hipError_t (*wnewton_fn)(const WregTab&, long, const double*, const double*, const double*, const double*,
                                 const double*, double, double*, int*, int*, DevOpts, int, hipStream_t);

template <int MB, int NQ>
hipError_t do_solve(const WregTab& T, long B, const double* b, const double* c, double* x, double* y, double* z,
                    double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o, int grid,
                    hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void*)ipm_wreg_kernel<MB, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, T.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ipm_wreg_kernel<MB, NQ>), dim3(grid), dim3(256), T.lds_bytes, st, T, B, b, c, x, y, z, pobj, dobj,
                       status, iters, qhead, defer, o);
    return hipGetLastError();
}
template <int MB, int NQ>
hipError_t do_solve_hsd(const WregTab& T, long B, const double* b, const double* c, double* x, double* y, double* z,
                        double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o, int grid,
                        hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void*)hsd_wreg_kernel<MB, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, T.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((hsd_wreg_kernel<MB, NQ>), dim3(grid), dim3(256), T.lds_bytes, st, T, B, b, c, x, y, z, pobj, dobj,
                       status, iters, qhead, defer, o);
    return hipGetLastError();
}
template <int MB, int NQ>
hipError_t do_newton(const WregTab& T, long B, const double* x, const double* z, const double* y, const double* b,
                     const double* c, double mu, double* dy, int* nref, int* qhead, DevOpts o, int grid, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void*)newton_wreg_kernel<MB, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, T.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((newton_wreg_kernel<MB, NQ>), dim3(grid), dim3(256), T.lds_bytes, st, T, B, x, z, y, b, c, mu, dy,
                       nref, qhead, o);
    return hipGetLastError();
}

struct WVariant { int mb, nq; wsolve_fn solve, solve_hsd; wnewton_fn newton; };
#define WVARIANT(MB, NQ) { MB, NQ, do_solve<MB, NQ>, do_solve_hsd<MB, NQ>, do_newton<MB, NQ> }
// ordered by cost; the first variant with 16 mb >= m and 64 nq >= n is used
const WVariant kWVariants[] = { WVARIANT(8, 6) };
const int kNumWVariants = sizeof(kWVariants) / sizeof(kWVariants[0]);

}  // namespace

int wreg_plan_create(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds,
                     hipStream_t st, WregPlan** out) {
    int vi = -1;
    for (int i = 0; i < kNumWVariants; i++)
        if (m <= 16 * kWVariants[i].mb && n <= 64 * kWVariants[i].nq) { vi = i; break; }
    if (vi < 0 || nnz >= 65536) return 1;
    const int MB = kWVariants[vi].mb, NQ = kWVariants[vi].nq;
    const int MP = 16 * MB, MPL = 64 * ((MP + 63) / 64), NP = 64 * NQ;
    WregPlan* P = new WregPlan();
    WregTab& T = P->tab;
    memset(&T, 0, sizeof(T));
    T.m = m; T.n = n; T.nnz = nnz;
    // ---- column positions: the columns sorted by length (longest first) are dealt to positions 0, 1, ...; position p
    //      is element p % 64 of N-vector register p / 64, so every register holds 64 columns of similar length ----
    std::vector<int> cptr(n + 1, 0), crow(nnz);
    std::vector<double> csc_val(nnz);
    for (int e = 0; e < nnz; e++) cptr[col[e] + 1]++;
    for (int j = 0; j < n; j++) cptr[j + 1] += cptr[j];
    {
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int i = 0; i < m; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++) { const int p = fill[col[e]]++; crow[p] = i; csc_val[p] = val[e]; }
    }
    std::vector<int> order(n), posof(n);
    for (int j = 0; j < n; j++) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cptr[a + 1] - cptr[a] > cptr[b + 1] - cptr[b]; });
    std::vector<unsigned short> colmap(NP, 0);
    for (int p = 0; p < n; p++) { colmap[p] = (unsigned short)order[p]; posof[order[p]] = p; }
    // ---- A by rows (column indices = positions), compact ----
    std::vector<double> csr_val(val, val + nnz);
    std::vector<unsigned short> csr_col(nnz), csr_ptr(MPL, 0), csr_len(MPL, 0);
    int rmax = 0;
    for (int i = 0; i < m; i++) {
        csr_ptr[i] = (unsigned short)ptr[i]; csr_len[i] = (unsigned short)(ptr[i + 1] - ptr[i]);
        rmax = std::max(rmax, ptr[i + 1] - ptr[i]);
    }
    for (int e = 0; e < nnz; e++) csr_col[e] = (unsigned short)posof[col[e]];
    T.rmax = rmax;
    // ---- A by columns, ELL over positions ----
    int ctot = 0;
    for (int q = 0; q < NQ; q++) {
        int cm = 0;
        for (int p = 64 * q; p < std::min(n, 64 * q + 64); p++) cm = std::max(cm, cptr[order[p] + 1] - cptr[order[p]]);
        T.meta[q] = cm; T.meta[META_COFF + q] = ctot; ctot += cm;
    }
    T.ctot = ctot;
    std::vector<double> ec_val((size_t)std::max(ctot, 1) * 64, 0.0);
    std::vector<unsigned short> ec_row((size_t)std::max(ctot, 1) * 64, 0);
    for (int p = 0; p < n; p++) {
        const int j = order[p], q = p / 64, l = p % 64;
        for (int e = cptr[j], t = 0; e < cptr[j + 1]; e++, t++) {
            ec_val[(size_t)(T.meta[META_COFF + q] + t) * 64 + l] = csc_val[e];
            ec_row[(size_t)(T.meta[META_COFF + q] + t) * 64 + l] = (unsigned short)crow[e];
        }
    }
    // ---- Gram entries (strictly lower triangle of M): off-diagonal blocks grouped by staging chunk, then the entries
    //      inside the diagonal blocks grouped by block ----
    struct Term { int group, dst, colj; double w; };
    std::vector<Term> terms;
    std::vector<int> chbase(MB + 1, 0);
    for (int K = 0; K < MB; K++) chbase[K + 1] = chbase[K] + nch_host(MB, K);
    const int nchunk = chbase[MB];
    if (nchunk + 1 > 24 || MB + 1 > 24) { delete P; return 1; }
    for (int j = 0; j < n; j++)
        for (int a = cptr[j]; a < cptr[j + 1]; a++)
            for (int b2 = cptr[j]; b2 < a; b2++) {
                const int i = crow[a], k = crow[b2];   // rows ascend inside a column: i > k
                const int K = k / 16, I = i / 16;
                if (I == K) {          // diagonal block K: element [i%16][k%16] of the stride-17 tile
                    terms.push_back({nchunk + K, (i % 16) * 17 + (k % 16), posof[j], csc_val[a] * csc_val[b2]});
                } else {               // block (K, I) of U: element [k%16][i%16], block (I-K-1) % HB of its chunk
                    const int ch = (I - K - 1) / HB, bi = (I - K - 1) % HB;
                    terms.push_back({chbase[K] + ch, bi * 256 + (k % 16) * 16 + (i % 16), posof[j], csc_val[a] * csc_val[b2]});
                }
                if (terms.size() > ((size_t)1 << 22)) { delete P; return 1; }
            }
    std::stable_sort(terms.begin(), terms.end(), [](const Term& a, const Term& b) {
        return a.group != b.group ? a.group < b.group : a.dst < b.dst; });
    std::vector<unsigned> e_ptr; std::vector<unsigned short> e_dst, t_col(terms.size());
    std::vector<double> t_w(terms.size());
    const int ngroup = nchunk + MB;
    std::vector<int> seg(ngroup + 1, 0);
    for (size_t t = 0; t < terms.size(); t++) {
        if (t == 0 || terms[t].group != terms[t - 1].group || terms[t].dst != terms[t - 1].dst) {
            e_ptr.push_back((unsigned)t); e_dst.push_back((unsigned short)terms[t].dst);
            seg[terms[t].group + 1] = (int)e_dst.size();
        }
        t_col[t] = (unsigned short)terms[t].colj; t_w[t] = terms[t].w;
    }
    e_ptr.push_back((unsigned)terms.size());
    for (int i = 1; i <= ngroup; i++) seg[i] = std::max(seg[i], seg[i - 1]);
    for (int i = 0; i <= nchunk; i++) T.meta[META_SEG + i] = seg[i];
    for (int K = 0; K <= MB; K++) T.meta[META_DSEG + K] = seg[nchunk + K];
    T.n_ent = (int)e_dst.size(); T.n_term = (int)terms.size();
    if (t_w.empty()) { t_w.push_back(0.0); t_col.push_back(0); }
    if (e_dst.empty()) e_dst.push_back(0);
    // ---- LDS plan ----
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 15) & ~(size_t)15; const size_t o_ = off; off += bytes; return (int)o_; };
    T.o_csr_val = take(sizeof(double) * csr_val.size());
    T.o_ec_val = take(sizeof(double) * ec_val.size());
    T.o_t_w = take(sizeof(double) * t_w.size());
    T.wave_doubles = STAGE_D + 64 * NQ + 6 * MP + MB * WL;
    T.o_wave = take(sizeof(double) * 4 * (size_t)T.wave_doubles);
    T.o_e_ptr = take(sizeof(unsigned) * e_ptr.size());
    T.o_meta = take(sizeof(int) * META_N);
    T.o_csr_col = take(sizeof(unsigned short) * csr_col.size());
    T.o_csr_ptr = take(sizeof(unsigned short) * csr_ptr.size());
    T.o_csr_len = take(sizeof(unsigned short) * csr_len.size());
    T.o_ec_row = take(sizeof(unsigned short) * ec_row.size());
    T.o_colmap = take(sizeof(unsigned short) * colmap.size());
    T.o_e_dst = take(sizeof(unsigned short) * e_dst.size());
    T.o_t_col = take(sizeof(unsigned short) * t_col.size());
    T.lds_bytes = (int)((off + 15) & ~(size_t)15);
    if (T.lds_bytes > max_lds) { delete P; return 1; }
    // ---- device copies ----
    std::vector<char> host;
    const size_t a1 = put(host, csr_val), a2 = put(host, ec_val), a3 = put(host, t_w), a4 = put(host, e_ptr),
                 a5 = put(host, csr_col), a6 = put(host, csr_ptr), a7 = put(host, csr_len), a8 = put(host, ec_row),
                 a9 = put(host, colmap), a11 = put(host, e_dst), a12 = put(host, t_col);
    hipError_t e = hipMalloc(&P->dev_blob, host.size());
    if (e == hipSuccess) e = hipMemcpyAsync(P->dev_blob, host.data(), host.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { if (P->dev_blob) (void)hipFree(P->dev_blob); delete P; return 1000 + (int)e; }
    char* db = (char*)P->dev_blob;
    T.csr_val = (const double*)(db + a1); T.ec_val = (const double*)(db + a2); T.t_w = (const double*)(db + a3);
    T.e_ptr = (const unsigned*)(db + a4);
    T.csr_col = (const unsigned short*)(db + a5); T.csr_ptr = (const unsigned short*)(db + a6);
    T.csr_len = (const unsigned short*)(db + a7); T.ec_row = (const unsigned short*)(db + a8);
    T.colmap = (const unsigned short*)(db + a9);
    T.e_dst = (const unsigned short*)(db + a11); T.t_col = (const unsigned short*)(db + a12);
    P->mb = MB; P->nq = NQ;
    *out = P;
    return 0;
}

void wreg_plan_free(WregPlan* p) {
    if (!p) return;
    if (p->dev_blob) (void)hipFree(p->dev_blob);
    delete p;
}

int wreg_lds_bytes(const WregPlan* p) { return p ? p->tab.lds_bytes : 0; }

static const WVariant* find_variant(const WregPlan* p) {
    for (int i = 0; i < kNumWVariants; i++)
        if (kWVariants[i].mb == p->mb && kWVariants[i].nq == p->nq) return &kWVariants[i];
    return nullptr;
}

hipError_t wreg_launch_solve(WregPlan* p, long B, const double* b, const double* c, double* x, double* y, double* z,
                             double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o,
                             int num_cu, hipStream_t st, int* grid_out) {
    const WVariant* v = find_variant(p);
    if (!v) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(defer, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    long cus = (long)num_cu - o.reserve_cus > 0 ? (long)num_cu - o.reserve_cus : 1;
    long grid = std::min(cus, (B + 3) / 4);
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = (int)grid;
    return ((o.flags & PYCLLP_FLAG_HSD) ? v->solve_hsd : v->solve)(p->tab, B, b, c, x, y, z, pobj, dobj, status, iters, qhead, defer, o, (int)grid, st);
}

hipError_t wreg_launch_newton(WregPlan* p, long B, const double* x, const double* z, const double* y, const double* b,
                              const double* c, double mu, double* dy, int* nref, DevOpts o, int num_cu, hipStream_t st) {
    const WVariant* v = find_variant(p);
    if (!v) return hipErrorInvalidValue;
    int* qhead = nullptr;
    hipError_t e = hipMallocAsync((void**)&qhead, sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(qhead, 0, sizeof(int), st);
    long grid = std::min((long)num_cu, (B + 3) / 4);
    if (grid < 1) grid = 1;
    if (e == hipSuccess) e = v->newton(p->tab, B, x, z, y, b, c, mu, dy, nref, qhead, o, (int)grid, st);
    hipError_t e2 = hipFreeAsync(qhead, st);
    return e != hipSuccess ? e : e2;
}

hipError_t wreg_launch_ldl_solve(int n, long B, const double* A, const double* rhs, double* out, double floor_,
                                 int num_cu, hipStream_t st) {
    if (n > 128) return hipErrorInvalidValue;
    int* qhead = nullptr;
    hipError_t e = hipMallocAsync((void**)&qhead, sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(qhead, 0, sizeof(int), st);
    long grid = std::min((long)num_cu, (B + 3) / 4);
    if (grid < 1) grid = 1;
    const int lds = (int)(sizeof(double) * 4 * WGeo<8>::WAVE_D(1));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ldl_solve_wreg_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((ldl_solve_wreg_kernel<8>), dim3((unsigned)grid), dim3(256), lds, st, n, B, A, rhs, out, floor_, qhead);
        e = hipGetLastError();
    }
    hipError_t e2 = hipFreeAsync(qhead, st);
    return e != hipSuccess ? e : e2;
}

extern "C" int pycllp_hip_debug_wreg_selftest(double* out_dev, void* stream) {
    hipLaunchKernelGGL(wreg_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
    return (int)hipGetLastError();
}
