#!/usr/bin/env python
"""bench.py -- LPs solved/sec on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (HipDensePrimalNormalSolver.solve_device -> pycllp_hip_dense_solve) over one
batch of synthetic LPs that is already resident in HBM, followed -- for N > 1 -- by the final result gather to
rank 0 (RCCL).  Workload per GPU: 65 536 random dense LPs, StandardLP (m=32, n=64) -> equality form N=96, fp64
(BASELINE.json configs[2]; with 8 GPUs this is configs[3], 524 288 LPs).  Weak scaling: per-GPU work is fixed.

Rank 0 prints ONE JSON line.  `roofline` prices the solve kernel against the FP64 FMA/MFMA peak (the path is
compute bound: ~1500 flop per compulsory HBM byte) and also reports the algorithmic HBM rate;
`cpu_baseline` is the reference's own CPU solver (pycllp/ipo.py -> ipo/hsd.c, built from the reference sources
into oracle/_ref) timed on one host core on a bounded sample of the same workload.

With one GPU and the default workload the line also carries `secondary`: the other single-GPU configurations of this
path (configs[1] 4 096 x (16, 32); configs[4]'s per-GPU share on the reference's algorithm and on the homogeneous
self-dual variant; per-problem values of A on that structure; a dense LP beyond the lane-group kernels, m = 100) --
each measured the same way (resident inputs, HIP events around the launches, barrier-free wall clock with a device
synchronise either side), each with its own roofline figures, its parity against the committed golden vectors of the
reference solver and the reference solver's own time on a bounded sample of THAT workload.  They are records, not the
headline: `value` is always the configs[2] figure.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, N_STD, B_PER_GPU = 32, 64, 65536
PEAK_FP64_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (datasheet)
PEAK_HBM_GBS = 8000.0       # HBM3E spec (MI355X_MICROARCH.md)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def flops_per_lp(m, N, iters, refine=0.0):
    """F_alg = I*[m(m+1)N + 8mN + m^3/3 + 4m^2(1+r) + 14N + 3m]."""
    return iters * (m * (m + 1) * N + 8 * m * N + m ** 3 / 3.0 + 4 * m * m * (1 + refine) + 14 * N + 3 * m)


def flops_per_lp_sparse(m, N, iters, n_terms, nnz_e):
    """Term-list Gram assembly and CSR/CSC products instead of the dense m(m+1)N + 8mN; dense m^3/3 LDL'."""
    return iters * (2 * n_terms + 8 * nnz_e + m ** 3 / 3.0 + 4 * m * m + 14 * N + 3 * m)


def bytes_per_lp(m, N):
    """Compulsory HBM traffic of the fused solve: b, c in; x, y, z out; objectives, status, iters."""
    return 8 * (m + N) + 8 * (N + m) + 8 * N + 16 + 8


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows all of the
    host's hardware threads in the mask but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


# ---- CPU legs (rank 0, one GPU only): the reference's own solver on a bounded sample of each workload -------------------
def cpu_reference(matrices, b, c, seconds, what, chunk=64):
    """Reference CPU solver (oracle/_ref: pycllp/ipo.py -> ipo/hsd.c) on the first LPs of (b, c), one core, for about
    `seconds`.  `matrices`: one shared matrix (dense array: every entry handed over, as a dense reference LP does; scipy
    sparse: structural non-zeros only, as the reference's tocsc_arrays does, lp.py:289-299), or a callable k -> the matrix
    of LP k (per-problem values: one solver call per LP, the only way the reference can run such a batch)."""
    from oracle import hsd_ref
    if not hsd_ref.available():
        return None
    per_lp = callable(matrices)
    get = matrices if per_lp else (lambda k: matrices)
    hsd_ref.solve_standard(get(0), b[:1], c[:1])     # warm the library
    done, t0 = 0, time.perf_counter()
    step = 1 if per_lp else chunk
    while done < b.shape[0] and time.perf_counter() - t0 < seconds:
        r = hsd_ref.solve_standard(get(done), b[done:done + step], c[done:done + step])
        assert (r["status"] == 0).all()
        done += step
    done = min(done, b.shape[0])
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "LPs/s", "cores": 1, "kind": "reference",
            "sample": "first %d LPs of %s through oracle/_ref (reference ipo/hsd.c), %.1f s, single thread "
                      "(the C solver keeps global state: not thread-safe)" % (done, what, dt)}


def cpu_baseline(seconds=10.0):
    """The headline workload's CPU leg."""
    from oracle import hsd_ref
    from pycllp_amd import problems
    if not hsd_ref.available():
        from oracle import port
        A, b, c = problems.random_dense_arrays(M, N_STD, 2048, seed=0)
        Ae, be, ce = problems.equality_arrays(A, b, c)
        t = time.perf_counter(); port.dense_solve(Ae, be, ce, nthreads=1); dt = time.perf_counter() - t
        return {"value": 2048 / dt, "unit": "LPs/s", "cores": 1, "kind": "port",
                "sample": "first 2048 LPs of the workload, oracle/ipm_dense_ref.c single thread"}
    A, b, c = problems.random_dense_arrays(M, N_STD, 16384, seed=0)
    return cpu_reference(A, b, c, seconds, "the workload", chunk=256)


def cpu_port(ref_rate=None, nlp_single=1024, nlp_all=16384):
    """The oracle restatement (same algorithm as the kernel) on the host: single thread, and OpenMP over LPs on every core
    the process may use -- the all-core figure of SURVEY 8d, reported NEXT TO cpu_baseline (the reference solver itself
    cannot use threads).  `vs_reference_1core` is the port / reference single-thread ratio SURVEY 8d asks to carry."""
    from oracle import port
    from pycllp_amd import problems
    cores = usable_cores()
    A, b, c = problems.random_dense_arrays(M, N_STD, nlp_all, seed=0)
    Ae, be, ce = problems.equality_arrays(A, b, c)
    port.dense_solve(Ae, be[:64], ce[:64], nthreads=1)
    t = time.perf_counter(); port.dense_solve(Ae, be[:nlp_single], ce[:nlp_single], nthreads=1); d1 = time.perf_counter() - t
    port.dense_solve(Ae, be[:256], ce[:256], nthreads=cores)
    t = time.perf_counter(); r = port.dense_solve(Ae, be, ce, nthreads=cores); dt = time.perf_counter() - t
    assert (r["status"] == 0).all()
    one = nlp_single / d1
    return {"value": nlp_all / dt, "unit": "LPs/s", "cores": cores, "kind": "port",
            "single_thread": one, "vs_reference_1core": (one / ref_rate) if ref_rate else None,
            "sample": "oracle/ipm_dense_ref.c: first %d LPs of the workload on 1 thread (%.1f s), first %d LPs with OpenMP on "
                      "%d threads = the CPU share of this box (%.1f s)" % (nlp_single, d1, nlp_all, cores, dt)}


# ---- workloads ----------------------------------------------------------------------------------------------------------
class Workload(object):
    """One synthetic batch resident on the device + the solver that runs it.  Everything the JSON record of a workload
    needs is derived here, so that the headline and the secondary records are built by the same code."""

    def __init__(self, name, B, rank, dev, reserve=0, hsd=False, predcorr=False, r=None):
        import torch
        from pycllp_amd import problems
        from pycllp_amd.lp import SparseMatrix, EqualityLP, StandardLP
        from pycllp_amd.solvers import solver_registry
        self.name, self.B, self.hsd, self.dev = name, B, bool(hsd), dev
        self.predcorr = bool(predcorr)      # PYCLLP_FLAG_PREDCORR: Mehrotra's predictor-corrector (an option, never the headline)
        self.r = r                          # step fraction other than the reference's R = 0.9 (primal_normal.cl:11); option records only
        extra = {} if r is None else {"r": float(r)}
        self.per_a = name == "perA"
        self.sparse = name in ("sparse5", "perA")
        self.a_values = None
        if self.sparse:
            m_, n_ = 128, 256
            A, b, c = problems.random_sparse_arrays(m_, n_, B, density=0.025, seed=0)
            if rank:
                rs = np.random.RandomState(1000003 * rank)
                b = 0.5 + rs.rand(B, m_); c = 0.5 + rs.rand(B, n_)
            be, ce = b, np.hstack([c, np.zeros((B, m_))])
            if self.per_a:
                # the structure of configs[4]'s A, values of LP k = the shared ones x U[0.75, 1.25) per entry (seed 7 + rank)
                rows, cols, data = problems.per_problem_values(A, B, seed=7 + rank)
                self.coo = (rows, cols, data)
                lp = StandardLP(SparseMatrix(rows, cols, data), b, c, 0.0).to_equality_form()
            else:
                lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
            # hsd=False: the reference's algorithm (primal_normal.cl path following), what the plugin's default hsd='auto'
            # runs first; hsd=True: the homogeneous self-dual variant (41 instead of 52 iterations, two solves per iteration)
            solver = solver_registry["hip_sparse_primal_normal"](device=dev, hsd=self.hsd, predcorr=self.predcorr, reserve_cus=reserve, **extra)
            self.what = ("%d random LPs per GPU, shared SPARSE A (m=%d, n=%d, density 0.025, rows>=3 and columns>=1 non-zeros) "
                         "-> equality form N=%d, b,c~U[0.5,1.5), seed 0 (BASELINE.json configs[4] per-GPU share)%s"
                         % (B, m_, n_, n_ + m_, "; PER-PROBLEM VALUES of A on that structure (SURVEY 8f-4; shared x U[0.75,1.25))"
                            if self.per_a else ""))
            if self.predcorr:
                self.what += "; OPTION predcorr=True (Mehrotra predictor-corrector: one factorisation, two solves per iteration)"
            if r is not None:
                self.what += "; OPTION step fraction r = %g (reference: 0.9)" % r
        else:
            m_, n_ = {"dense3": (M, N_STD), "dense2": (16, 32), "dense100": (100, 80), "dense200": (200, 200)}[name]
            A, b, c = problems.random_dense_arrays(m_, n_, B, seed=0, shard=rank)
            Ae, be, ce = problems.equality_arrays(A, b, c)
            lp = EqualityLP(SparseMatrix(matrix=Ae), be[:1], ce[:1], 0.0)
            solver = solver_registry["hip_dense_primal_normal"](device=dev, hsd=self.hsd, predcorr=self.predcorr, reserve_cus=reserve, **extra)
            cfg = {"dense3": "BASELINE.json configs[2]", "dense2": "BASELINE.json configs[1]",
                   "dense100": "a dense LP beyond the lane-group kernels; m = 100 is the reference's own kernel-test size, "
                               "tests/test_ldl.py:226-238",
                   "dense200": "a dense LP beyond the wavefront-per-LP kernel: the large-LP kernel csrc/ipm_big.hip, one LP per "
                               "workgroup; the reference's hosts take any size, solvers/cl.py:28-83"}[name]
            self.what = ("%d random dense LPs per GPU, StandardLP (m=%d, n=%d) -> equality form N=%d, A~U[0,1) shared, "
                         "b,c~U[0.5,1.5), seed 0 (%s)" % (B, m_, n_, n_ + m_, cfg))
            if self.predcorr:
                self.what += "; OPTION predcorr=True (Mehrotra predictor-corrector: one factorisation, two solves per iteration)"
            if r is not None:
                self.what += "; OPTION step fraction r = %g (reference: 0.9)" % r
        self.m, self.n, self.N = m_, n_, n_ + m_
        self.A, self.b_std, self.c_std = A, b, c
        self.lp, self.solver = lp, solver
        lp.init(solver)
        if self.per_a:      # the values travel once, like b and c: resident in HBM when the timed region starts
            solver._a_values = torch.as_tensor(np.ascontiguousarray(np.asarray(lp.A.data, dtype=np.float64)[:, solver._a_perm]), device=dev)
        self.bd = torch.as_tensor(be, device=dev)
        self.cd = torch.as_tensor(ce, device=dev)

    def step(self, slot=0):
        return self.solver.solve_device(self.bd, self.cd, slot=slot)

    # -- parity on the committed golden LPs (outputs of the reference's CPU solver; outside any timed region) ----------
    def parity(self):
        import torch
        from pycllp_amd import problems
        s, dev = self.solver, self.dev
        fname = {"dense3": "config_32x64.npz", "dense2": "config_16x32.npz", "dense100": "config_dense_100x80.npz",
                 "dense200": "config_dense_200x200.npz",
                 "sparse5": "config_sparse_128x256.npz", "perA": "config_perA_128x256.npz"}[self.name]
        path = os.path.join(GOLDEN, fname)
        if not os.path.exists(path):
            return None
        g = np.load(path)
        if self.per_a:
            k = int(g["nobj"])       # the first LPs of the batch ARE the golden LPs (values: problems.per_problem_values, seed 7)
            assert np.allclose(g["input_checksum"], [self.coo[2][:k].sum(), self.b_std[:k].sum(), self.c_std[:k].sum()], rtol=1e-12)
            keep = s._a_values
            s._a_values = keep[:k].contiguous()
            try:
                r = s.solve_device(self.bd[:k].contiguous(), self.cd[:k].contiguous(), slot=2)
                torch.cuda.synchronize(dev)
            finally:
                s._a_values = keep
            src = ("reference ipo.py (hsd.c) called LP by LP with each LP's own matrix, via tests/golden/%s (the reference's LP "
                   "classes refuse a per-problem-A batch, lp.py:335-336; its solver takes one matrix per call)" % fname)
        elif self.sparse:
            k = int(g["pobj"].shape[0])
            r = s.solve_device(g["b"], np.hstack([g["c"], np.zeros((k, self.m))]), slot=2)
            torch.cuda.synchronize(dev)
            src = "reference ipo.py (hsd.c) via tests/golden/%s" % fname
        else:
            k = int(g["nobj"])
            A2, b2, c2 = problems.random_dense_arrays(self.m, self.n, k, seed=0)
            assert np.allclose(g["input_checksum"], [A2.sum(), b2.sum(), c2.sum()], rtol=1e-12)
            _, b2e, c2e = problems.equality_arrays(A2, b2, c2)
            r = s.solve_device(b2e, c2e, slot=2)
            torch.cuda.synchronize(dev)
            src = "reference ipo.py (hsd.c) via tests/golden/%s" % fname
        ep = np.abs(r["pobj"].cpu().numpy() - g["pobj"]) / np.maximum(1.0, np.abs(g["pobj"]))
        ed = np.abs(r["dobj"].cpu().numpy() - g["dobj"]) / np.maximum(1.0, np.abs(g["dobj"]))
        return {"golden_lps": k, "max_rel_err_primal_obj": float(ep.max()), "max_rel_err_dual_obj": float(ed.max()),
                "tolerance": 1e-8, "ok": bool(ep.max() <= 1e-8 and ed.max() <= 1e-8 and (r["status"].cpu().numpy() == 0).all()),
                "source": src}

    # -- roofline of the dominant kernel ----------------------------------------------------------------------------------
    def roofline(self, kern_ms, iters_mean, world=1):
        import scipy.sparse as sp
        m_, n_, Nn, B = self.m, self.n, self.N, self.B
        if self.sparse:
            nnz_e = int(self.A.nnz) + m_
            coln = np.diff(sp.csc_matrix(self.lp.A.tocsr(0) if self.per_a else self.lp.A.tocsr()).indptr)
            n_terms = int((coln * (coln + 1) // 2).sum())
            f_alg = f_exec = flops_per_lp_sparse(m_, Nn, iters_mean, n_terms, nnz_e)
        else:
            f_alg = flops_per_lp(m_, Nn, iters_mean)
            # executed work: the slack-aware kernels (lane-group kernel and the dense-image wave kernel alike) run the Gram
            # product and the mat-vecs on the n = N - m dense columns only (the identity columns of [A | I] bypass them)
            f_exec = iters_mean * (m_ * (m_ + 1) * n_ + 8 * m_ * n_ + m_ ** 3 / 3.0 + 4 * m_ * m_ + 14 * Nn + 3 * m_)
            # (the large-LP kernel forms both triangles of the diagonal blocks and multiplies whole 16-row blocks; still priced
            # as the lower triangle: what it executes beyond that is overhead, not work)
        if self.predcorr:      # + one forward/back substitution (4 m^2) and A'u, A v (4 nnz resp. 4 m n) per iteration
            extra = iters_mean * (4 * m_ * m_ + (4 * (int(self.A.nnz) + m_) if self.sparse else 4 * m_ * n_))
            f_alg += extra; f_exec += extra
        t = kern_ms * 1e-3
        tf_alg, tf_exec = f_alg * B / t / 1e12, f_exec * B / t / 1e12
        a_bytes = 8 * (int(self.A.nnz) + m_) if self.per_a else 0   # per-problem A: every LP reads its own values (equality form)
        b_survey = 16 * (m_ + Nn) + 24 + a_bytes                    # b, c in; x, y out; objectives; status, iters
        b_with_z = bytes_per_lp(m_, Nn) + a_bytes                   # + the dual slacks z this library also returns
        gbs = b_survey * B / t / 1e9
        # HBM traffic of the dominant kernel from PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
        # passes): a profile-sourced figure, attached only when this run is the profiled configuration
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            ent = json.load(open(tpath)).get(self.name + ("_hsd" if self.hsd else "") + ("_predcorr" if self.predcorr else ""))
            if ent and ent.get("lps_per_launch") == B and world == 1:
                traffic, traffic_src = ent.get("bytes_per_launch"), ent.get("source")
        return {"bound": "mfma", "achieved": tf_exec, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": tf_exec / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": kern_ms, "flop_per_lp": f_exec,
                "frac_algorithmic": tf_alg / PEAK_FP64_TFLOPS, "achieved_algorithmic": tf_alg, "flop_per_lp_algorithmic": f_alg,
                "note": "FP64 FMA/MFMA peak.  achieved / frac: the flops the kernel EXECUTES with the measured mean iteration "
                        "count, refinement passes counted as 0 (dense workloads: the identity columns of [A | I] are never "
                        "multiplied; sparse workloads: term-list Gram + CSR products + dense m^3/3 LDL').  frac_algorithmic: "
                        "SURVEY 8d's formula, which prices every column of the equality form densely (rounds 1-2 reported "
                        "that figure as frac; ADVICE r2)",
                "hbm_algorithmic": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                    "bytes_per_lp": b_survey, "bytes_per_lp_with_z": b_with_z}}

    def kernel_name(self, info):
        if info.get("kernel") == "big":
            return ("ipm_big_kernel (one LP per workgroup, 16 x 16 blocks of the factor in %s, Gram: %s)%s, grid %d x block %d, %d B LDS"
                    % ("LDS" if info["lds_bytes"] > 100000 else "an L2-resident workspace", info.get("variant"),
                       ", PYCLLP_FLAG_HSD" if self.hsd else (", PYCLLP_FLAG_PREDCORR" if self.predcorr else ""),
                       info["grid"], info["block"], info["lds_bytes"]))
        if self.sparse or "kernel" in info:
            if info.get("kernel") == "wave":
                return ("%s (one LP per wavefront, factor in registers, A as %s)%s, grid %d x block %d, %d B LDS"
                        % ("hsd_wreg_kernel" if self.hsd else "ipm_wreg_kernel", info.get("variant", "tables"),
                           ", PYCLLP_FLAG_HSD" if self.hsd else (", PYCLLP_FLAG_PREDCORR" if self.predcorr else ""),
                           info["grid"], info["block"], info["lds_bytes"]))
            return ("ipm_block_kernel (one LP per 256-thread workgroup)%s, grid %d x block %d, %d B LDS"
                    % (", PYCLLP_FLAG_HSD" if self.hsd else "", info["grid"], info["block"], info["lds_bytes"]))
        return ("%s<%d,%d%s> grid %d x block %d, %d B LDS" % ("hsd_group_kernel" if self.hsd else "ipm_group_kernel",
                                                              info["m_pad"], info["n_pad"], ",predictor-corrector" if self.predcorr else "",
                                                              info["grid"], info["block"], info["lds_bytes"]))

    def launch_info(self):
        return self.solver.launch_info()

    def cpu(self, seconds):
        import scipy.sparse as sp
        if self.per_a:
            rows, cols, data = self.coo
            shape = (self.m, self.n)
            return cpu_reference(lambda k: sp.csr_matrix((data[k], (rows, cols)), shape=shape), self.b_std, self.c_std, seconds,
                                 "this workload, each with its own matrix")
        return cpu_reference(self.A, self.b_std, self.c_std, seconds, "this workload",
                             chunk=256 if self.m <= 32 else (16 if self.m <= 128 else 4))


def measure_secondary(name, B, dev, steps, warmup, hsd=False, cpu_seconds=4.0, cpu_from=None, predcorr=False, r=None):
    """One secondary record: the workload resident in HBM, `warmup` untimed and `steps` timed passes (HIP events around
    every launch on the launch stream, wall clock between two device synchronisations), parity, roofline, CPU reference."""
    import torch
    t_build = time.perf_counter()
    w = Workload(name, B, 0, dev, hsd=hsd, predcorr=predcorr, r=r)
    for k in range(warmup):
        w.step(k % 2)
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        buf = w.step(k % 2)
        ev[k][1].record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in ev]))
    info = w.launch_info()
    status = buf["status"].cpu().numpy(); iters = buf["iters"].cpu().numpy()
    pobj = buf["pobj"].cpu().numpy(); dobj = buf["dobj"].cpu().numpy()
    rec = {"workload": name + ("_hsd" if hsd else "") + ("_predcorr" if predcorr else "") + ("_r%g" % r if r is not None else ""),
           "value": B * steps / elapsed, "unit": "LPs/s", "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "kernel_ms": kern_ms,
           "config": {"workload": w.what, "lps_per_gpu": B, "m": w.m, "n": w.n, "N_equality": w.N, "kernel": w.kernel_name(info)},
           "solved_optimal": int((status == 0).sum()), "mean_ipm_iterations": float(iters.mean()),
           "max_rel_duality_gap": float(np.max(np.abs(pobj - dobj) / np.maximum(1.0, np.abs(pobj)))),
           "parity": w.parity(), "roofline": w.roofline(kern_ms, float(iters.mean()))}
    rec["cpu_baseline"] = cpu_from if cpu_from is not None else (w.cpu(cpu_seconds) if cpu_seconds else None)
    rec["record_seconds"] = time.perf_counter() - t_build
    return rec


# (workload, LPs, hsd, predcorr, step fraction or None = the reference's 0.9)
SECONDARY = (("dense2", 4096, False, False, None), ("sparse5", 16384, False, False, None), ("sparse5", 16384, True, False, None),
             ("perA", 16384, False, False, None), ("dense100", 16384, False, False, None), ("dense200", 4096, False, False, None),
             ("dense3", 65536, True, False, None), ("dense3", 65536, False, True, None), ("dense3", 65536, False, True, 0.99),
             ("sparse5", 16384, False, True, None), ("sparse5", 16384, False, True, 0.99))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="LPs per GPU (default: the BASELINE workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline record only (no `secondary` list)")
    ap.add_argument("--hsd", action="store_true", help="time the homogeneous self-dual variant (PYCLLP_FLAG_HSD) of --workload")
    ap.add_argument("--predcorr", action="store_true", help="time the predictor-corrector option (PYCLLP_FLAG_PREDCORR) of --workload")
    ap.add_argument("--workload", choices=("dense3", "sparse5", "perA", "dense2", "dense100", "dense200"), default="dense3",
                    help="dense3 (default): BASELINE.json configs[2], the headline workload; sparse5: configs[4]'s per-GPU "
                         "share (16 384 LPs, shared sparse A m=128, n=256, density 0.025) through hip_sparse_primal_normal; perA: "
                         "the same with per-problem values of A; dense2: configs[1]; dense100: 16 384 dense LPs (m=100, n=80)")
    ap.add_argument("--sync-gather", action="store_true", help="block on the result gather after every solve (no overlap)")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="compute units the solve kernel leaves idle (opts.reserve_cus).  Default: 8 (one per XCD) when "
                         "results are gathered across ranks -- the persistent solve kernel otherwise fills every CU and the "
                         "RCCL copy kernels of the overlapped gather would have to wait for it to end -- else 0")
    ap.add_argument("--force-collectives", action="store_true",
                    help="development aid: take the multi-rank code path (process group, result gather, barriers) even "
                         "with one rank -- exercises the RCCL calls on a one-GPU box")
    ap.add_argument("--rehearse", action="store_true",
                    help="development aid: run the N-rank path on ONE GPU (all ranks share cuda:0, gloo backend, "
                         "results gathered through host memory); the numbers it prints are not benchmark results")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON result of rank 0.  Libraries write there too (RCCL prints a five-line
    # version banner to stdout when its communicator comes up), so file descriptor 1 is pointed at stderr for the whole
    # run and the result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pycllp_amd.dist import PackedGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU fallback)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse else dev     # where collectives run
    multi = world > 1 or args.force_collectives
    reserve = args.reserve_cus if args.reserve_cus is not None else (8 if multi else 0)
    if multi:
        # (a bare `python bench.py --force-collectives` has no launcher that sets the rendezvous)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B = args.batch
    if args.batch == B_PER_GPU and args.workload != "dense3":
        B = 4096 if args.workload in ("dense2", "dense200") else 16384
    wl = Workload(args.workload, B, rank, dev, reserve=reserve, hsd=args.hsd, predcorr=args.predcorr)
    solver, bd, cd, Nn = wl.solver, wl.bd, wl.cd, wl.N
    sparse, per_a, m_, n_ = wl.sparse, wl.per_a, wl.m, wl.n

    cpu = cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.workload == "dense3":
            cpu = cpu_baseline()
            cpu_all = cpu_port(cpu["value"] if cpu["kind"] == "reference" else None)
        else:
            cpu = wl.cpu(8.0)

    fields = ("pobj", "dobj", "status", "iters", "x", "y")
    # Result gather to rank 0 (the path's one collective).  It is issued asynchronously and overlaps the NEXT step's
    # solve: outputs are double-buffered (slot k%2), receive buffers on rank 0 too, and a slot is reused only after the
    # gather that read it has completed.  --sync-gather falls back to a blocking gather after every solve.
    # The six result arrays of a solve are views into one packed allocation (HipDensePrimalNormalSolver._buffers), so the
    # gather is ONE collective of gather_bytes per rank per step.
    layout = solver._pack_layout(B)
    pg = PackedGather(layout, world, rank, cdev, dst=0, slots=2) if multi else None

    def step(k, e0=None, e1=None):
        sl = k % 2
        if multi:
            pg.wait(sl)                  # slot sl (solver outputs + receive buffers) is free again
        if e0 is not None:
            e0.record()
        buf = solver.solve_device(bd, cd, slot=sl)     # the dominant kernel, on torch's current stream
        if e1 is not None:
            e1.record()
        if multi:
            t = buf["packed"]
            pg.post(k, t.to(cdev) if args.rehearse else t, sync=args.sync_gather)
        return buf

    for k in range(args.warmup):
        step(k)
    if multi:
        pg.wait()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        buf = step(k, ev[k][0], ev[k][1])
    if multi:
        pg.wait()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    gathered = pg.results(args.steps - 1, names=fields) if multi else None

    kern_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in ev]))
    info = wl.launch_info()      # of the timed launches (the parity solve below launches again, with another batch)
    status = buf["status"].cpu().numpy()
    iters = buf["iters"].cpu().numpy()
    pobj = buf["pobj"].cpu().numpy(); dobj = buf["dobj"].cpu().numpy()
    ok_local = int((status == 0).sum())
    gap = float(np.max(np.abs(pobj - dobj) / np.maximum(1.0, np.abs(pobj))))
    if multi:
        agg = torch.tensor([ok_local, float(iters.sum())], dtype=torch.float64, device=cdev)
        dist.all_reduce(agg)
        ok_total, iters_mean = int(agg[0].item()), float(agg[1].item()) / (B * world)
        if rank == 0:
            assert gathered["x"].shape == (B * world, Nn) and gathered["status"].shape == (B * world,)
            # rank 0's own shard must have come through the collective unchanged
            assert torch.equal(gathered["pobj"][:B].to(dev), buf["pobj"]) and torch.equal(gathered["x"][:B].to(dev), buf["x"])
            assert int((gathered["status"] == 0).sum()) == ok_total
    else:
        ok_total, iters_mean = ok_local, float(iters.mean())

    if rank == 0:
        parity = wl.parity()     # on the committed golden LPs (outside the timed region)
        total = B * world * args.steps
        value = total / elapsed
        out = {
            "metric": "LPs solved/sec", "value": value, "unit": "LPs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not args.rehearse else "synthetic (REHEARSAL on one shared GPU: not a result)",
            "config": {"workload": wl.what + ("; x8 = configs[3]" if (world == 8 and args.workload == "dense3") else ""),
                       "lps_per_gpu": B, "lps_total": B * world, "m": m_, "n": n_, "N_equality": Nn,
                       "parallelism": "batch sharded over %d GPU(s), result gather to rank 0" % world, "reserve_cus": reserve,
                       "kernel": wl.kernel_name(info)},
            "solved_optimal": ok_total, "mean_ipm_iterations": iters_mean, "max_rel_duality_gap_rank0": gap,
            "parity": parity,
            "roofline": wl.roofline(kern_ms, iters_mean, world),
            "cpu_baseline": cpu,
            "cpu_port": cpu_all,
        }
        if (world == 1 and not multi and args.workload == "dense3" and not args.hsd and not args.predcorr and not args.no_secondary
                and B == B_PER_GPU):
            # the other single-GPU configurations under the same clock (records; `value` above stays the headline's)
            del wl
            sec, cpu5 = [], None
            for name, Bs, hsd, pc, rr in SECONDARY:
                try:
                    # same LPs, same reference solver: timed once (the headline's own CPU leg serves dense3_predcorr)
                    reuse = cpu5 if (name == "sparse5" and (hsd or pc)) else (cpu if name == "dense3" else None)
                    r = measure_secondary(name, Bs, dev, steps=5, warmup=2, hsd=hsd, predcorr=pc, r=rr,
                                          cpu_seconds=None if args.no_cpu_baseline else 4.0, cpu_from=reuse)
                    if name == "sparse5" and not hsd and not pc:
                        cpu5 = r["cpu_baseline"]
                    sec.append(r)
                except Exception as exc:      # a secondary record must never cost the headline its line
                    sec.append({"workload": name + ("_hsd" if hsd else "") + ("_predcorr" if pc else "") + ("_r%g" % rr if rr else ""),
                                "error": "%s: %s" % (type(exc).__name__, exc)})
            out["secondary"] = sec
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
