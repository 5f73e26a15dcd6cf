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


def flops_per_lp(m, N, iters, refine=0.0):
    """F_alg = I*[m(m+1)N + 8mN + m^3/3 + 4m^2(1+r) + 14N + 3m]."""
    return iters * (m * (m + 1) * N + 8 * m * N + m ** 3 / 3.0 + 4 * m * m * (1 + refine) + 14 * N + 3 * m)


def bytes_per_lp(m, N):
    """Compulsory HBM traffic of the fused solve: b, c in; x, y, z out; objectives, status, iters."""
    return 8 * (m + N) + 8 * (N + m) + 8 * N + 16 + 8


def cpu_baseline(seconds=12.0):
    """Reference CPU solver (oracle/_ref: ipo.py's hsd.c) on the first LPs of the same batch, 1 core."""
    from oracle import hsd_ref
    from pycllp_amd import problems
    if not hsd_ref.available():
        from oracle import port
        A, b, c = problems.random_dense_arrays(M, N_STD, 2048, seed=0)
        Ae, be, ce = problems.equality_arrays(A, b, c)
        t = time.perf_counter(); port.dense_solve(Ae, be, ce, nthreads=1); dt = time.perf_counter() - t
        return {"value": 2048 / dt, "unit": "LPs/s", "cores": 1, "kind": "port",
                "sample": "first 2048 LPs of the workload, oracle/ipm_dense_ref.c single thread"}
    nmax = 16384
    A, b, c = problems.random_dense_arrays(M, N_STD, nmax, seed=0)
    hsd_ref.solve_standard(A, b[:8], c[:8])   # warm the library
    done, t0 = 0, time.perf_counter()
    chunk = 256
    while done < nmax and time.perf_counter() - t0 < seconds:
        r = hsd_ref.solve_standard(A, b[done:done + chunk], c[done:done + chunk])
        assert (r["status"] == 0).all()
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "LPs/s", "cores": 1, "kind": "reference",
            "sample": "first %d LPs of the workload through oracle/_ref (reference ipo/hsd.c), %.1f s, single thread "
                      "(the C solver keeps global state: not thread-safe)" % (done, dt)}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="LPs per GPU (default: the BASELINE workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hsd", action="store_true", help="sparse5 only: time the homogeneous self-dual variant (PYCLLP_FLAG_HSD)")
    ap.add_argument("--workload", choices=("dense3", "sparse5", "perA"), default="dense3",
                    help="dense3 (default): BASELINE.json configs[2], the headline workload; sparse5: configs[4]'s per-GPU "
                         "share (16 384 LPs, shared sparse A m=128, n=256, density 0.025) through hip_sparse_primal_normal")
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
    from pycllp_amd import problems
    from pycllp_amd.lp import SparseMatrix, EqualityLP
    from pycllp_amd.solvers import solver_registry
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

    cpu = cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        cpu_all = cpu_port(cpu["value"] if cpu["kind"] == "reference" else None)

    B = args.batch
    per_a = args.workload == "perA"       # one structure, every LP its own values of A (read from HBM per LP)
    sparse = args.workload == "sparse5" or per_a
    if sparse:
        if args.batch == B_PER_GPU:
            B = 16384
        m_, n_ = 128, 256
        A, b, c = problems.random_sparse_arrays(m_, n_, B, density=0.025, seed=0)
        if rank:
            rs = np.random.RandomState(1000003 * rank)
            b = 0.5 + rs.rand(B, m_); c = 0.5 + rs.rand(B, n_)
        be, ce = b, np.hstack([c, np.zeros((B, m_))])
        Nn = n_ + m_
        from pycllp_amd.lp import StandardLP
        if per_a:
            # the structure of configs[4]'s A, values of LP k = the shared ones x U[0.75, 1.25) per entry (seed 7 + rank)
            Ac = A.tocoo()
            data = Ac.data[None, :] * (0.75 + 0.5 * np.random.RandomState(7 + rank).rand(B, Ac.nnz))
            lp = StandardLP(SparseMatrix(Ac.row, Ac.col, data), b, c, 0.0).to_equality_form()
        else:
            lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
        # the reference's algorithm (primal_normal.cl path following), what the plugin's default hsd='auto' runs first;
        # --hsd: the homogeneous self-dual variant (41 instead of 52 iterations on this workload, two solves per iteration)
        solver = solver_registry["hip_sparse_primal_normal"](device=dev, hsd=bool(args.hsd), reserve_cus=reserve)
        cpu = None
    else:
        m_, n_ = M, N_STD
        A, b, c = problems.random_dense_arrays(M, N_STD, B, seed=0, shard=rank)
        Ae, be, ce = problems.equality_arrays(A, b, c)
        Nn = Ae.shape[1]
        lp = EqualityLP(SparseMatrix(matrix=Ae), be[:1], ce[:1], 0.0)
        solver = solver_registry["hip_dense_primal_normal"](device=dev, reserve_cus=reserve)
    lp.init(solver)
    if per_a:      # the values travel once, like b and c: resident in HBM when the timed region starts
        solver._a_values = torch.as_tensor(np.ascontiguousarray(np.asarray(lp.A.data, dtype=np.float64)[:, solver._a_perm]), device=dev)
    bd = torch.as_tensor(be, device=dev)
    cd = torch.as_tensor(ce, device=dev)
    sizes = [B] * world
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
    info = solver.launch_info()      # of the timed launches (the parity solve below launches again, with another batch)
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
        # parity on the committed golden LPs (outside the timed region)
        parity = None
        gpath = os.path.join(ROOT, "tests", "golden", "config_32x64.npz")
        if sparse:
            import scipy.sparse as sp
        if per_a:
            # no reference fixture can exist for this extension (the reference's LP classes refuse per-problem A, lp.py:335-336)
            # and the oracle belongs to tests/ (tests/test_hip_parity.py::test_per_problem_values_of_A checks every LP against
            # it with ITS OWN matrix): here the first LPs are checked through the optimality conditions themselves
            kk = 64
            xs = buf["x"][:kk].cpu().numpy(); ys = buf["y"][:kk].cpu().numpy(); zs = buf["z"][:kk].cpu().numpy()
            pv = buf["pobj"][:kk].cpu().numpy(); dv = buf["dobj"][:kk].cpu().numpy()
            rp = rd = 0.0
            for k in range(kk):
                Ak = np.asarray(lp.A.todense(k))
                rp = max(rp, np.linalg.norm(be[k] - Ak @ xs[k]) / (1.0 + np.linalg.norm(be[k])))
                rd = max(rd, np.linalg.norm(ce[k] - Ak.T @ ys[k] + zs[k]) / (1.0 + np.linalg.norm(ce[k])))
            parity = {"kkt_lps": kk, "max_rel_primal_residual": float(rp), "max_rel_dual_residual": float(rd),
                      "max_rel_gap": float(np.max(np.abs(pv - dv) / np.maximum(1.0, np.abs(pv)))), "tolerance": 1e-8,
                      "source": "optimality conditions with each LP's own matrix (parity unpinned by the reference: it has no "
                                "per-problem-A path; oracle parity LP by LP is in tests/test_hip_parity.py)"}
        elif sparse:
            g = np.load(os.path.join(ROOT, "tests", "golden", "config_sparse_128x256.npz"))
            r = solver.solve_device(g["b"], np.hstack([g["c"], np.zeros((g["c"].shape[0], m_))]))
            torch.cuda.synchronize(dev)
            ep = np.abs(r["pobj"].cpu().numpy() - g["pobj"]) / np.maximum(1.0, np.abs(g["pobj"]))
            ed = np.abs(r["dobj"].cpu().numpy() - g["dobj"]) / np.maximum(1.0, np.abs(g["dobj"]))
            parity = {"golden_lps": int(g["pobj"].shape[0]), "max_rel_err_primal_obj": float(ep.max()),
                      "max_rel_err_dual_obj": float(ed.max()), "tolerance": 1e-8,
                      "source": "reference ipo.py (hsd.c) via tests/golden/config_sparse_128x256.npz"}
        elif os.path.exists(gpath):
            g = np.load(gpath)
            A2, b2, c2 = problems.random_dense_arrays(M, N_STD, int(g["nobj"]), seed=0)
            _, b2e, c2e = problems.equality_arrays(A2, b2, c2)
            r = solver.solve_device(b2e, c2e)
            torch.cuda.synchronize(dev)
            ep = np.abs(r["pobj"].cpu().numpy() - g["pobj"]) / np.maximum(1.0, np.abs(g["pobj"]))
            ed = np.abs(r["dobj"].cpu().numpy() - g["dobj"]) / np.maximum(1.0, np.abs(g["dobj"]))
            parity = {"golden_lps": int(g["nobj"]), "max_rel_err_primal_obj": float(ep.max()),
                      "max_rel_err_dual_obj": float(ed.max()), "tolerance": 1e-8,
                      "source": "reference ipo.py (hsd.c) via tests/golden/config_32x64.npz"}
        total = B * world * args.steps
        value = total / elapsed
        if sparse:   # term-list Gram assembly and CSR/CSC products instead of the dense m(m+1)N + 8mN
            nnz_e = int(A.nnz) + m_
            coln = np.diff(sp.csc_matrix(lp.A.tocsr()).indptr)
            n_terms = int((coln * (coln + 1) // 2).sum())
            f_lp = iters_mean * (2 * n_terms + 8 * nnz_e + m_ ** 3 / 3.0 + 4 * m_ * m_ + 14 * Nn + 3 * m_)
        else:
            f_lp = flops_per_lp(m_, Nn, iters_mean)
        tflops = f_lp * B / (kern_ms * 1e-3) / 1e12
        # executed work: the slack-aware dense kernel runs the Gram product and the mat-vecs on the n = N - m dense
        # columns only (the identity columns of [A | I] bypass them); the sparse formula above already counts executed work
        f_exec = f_lp if sparse else iters_mean * (m_ * (m_ + 1) * n_ + 8 * m_ * n_ + m_ ** 3 / 3.0 + 4 * m_ * m_ + 14 * Nn + 3 * m_)
        tflops_exec = f_exec * B / (kern_ms * 1e-3) / 1e12
        a_bytes = 8 * (int(A.nnz) + m_) if per_a else 0      # per-problem A: every LP reads its own values (equality form)
        b_survey = 16 * (m_ + Nn) + 24 + a_bytes             # b, c in; x, y out; objectives; status, iters
        b_with_z = bytes_per_lp(m_, Nn) + a_bytes            # + the dual slacks z this library also returns
        gbs = b_survey * B / (kern_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
        # passes): a profile-sourced figure, attached only when this run is the profiled configuration
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get(args.workload)
            if ent and ent.get("lps_per_launch") == B and world == 1 and not args.hsd:
                traffic, traffic_src = ent.get("bytes_per_launch"), ent.get("source")
        out = {
            "metric": "LPs solved/sec", "value": value, "unit": "LPs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not args.rehearse else "synthetic (REHEARSAL on one shared GPU: not a result)",
            "config": {"workload": ("%d random LPs per GPU, shared SPARSE A (m=%d, n=%d, density 0.025, rows>=3 and columns>=1 "
                                    "non-zeros) -> equality form N=%d, b,c~U[0.5,1.5), seed 0 (BASELINE.json configs[4] per-GPU share)%s"
                                    % (B, m_, n_, Nn, "; PER-PROBLEM VALUES of A on that structure (SURVEY 8f-4; shared x U[0.75,1.25))"
                                       if per_a else "")) if sparse else
                                   ("%d random dense LPs per GPU, StandardLP (m=%d, n=%d) -> equality form N=%d, "
                                    "A~U[0,1) shared, b,c~U[0.5,1.5), seed 0 (BASELINE.json configs[2]%s)"
                                    % (B, M, N_STD, Nn, "; x8 = configs[3]" if world == 8 else "")),
                       "lps_per_gpu": B, "lps_total": B * world, "m": m_, "n": n_, "N_equality": Nn,
                       "parallelism": "batch sharded over %d GPU(s), result gather to rank 0" % world, "reserve_cus": reserve,
                       "kernel": ("%s%s, grid %d x block %d, %d B LDS"
                                  % (("hsd_wreg_kernel" if args.hsd else "ipm_wreg_kernel") + " (one LP per wavefront, factor in registers)"
                                     if info["kernel"] == "wave" else "ipm_block_kernel (one LP per 256-thread workgroup)",
                                     ", PYCLLP_FLAG_HSD" if args.hsd else "", info["grid"], info["block"], info["lds_bytes"]))
                                 if sparse else
                                 "ipm_group_kernel<%d,%d> grid %d x block %d, %d B LDS"
                                 % (info["m_pad"], info["n_pad"], info["grid"], info["block"], info["lds_bytes"])},
            "solved_optimal": ok_total, "mean_ipm_iterations": iters_mean, "max_rel_duality_gap_rank0": gap,
            "parity": parity,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": kern_ms, "flop_per_lp": f_lp,
                         "frac_executed": tflops_exec / PEAK_FP64_TFLOPS, "flop_per_lp_executed": f_exec,
                         "note": "FP64 FMA/MFMA peak.  frac: algorithmic flops of SURVEY 8d (every column of the equality "
                                 "form priced densely; sparse workload: term-list Gram + CSR products + dense m^3/3 LDL') with "
                                 "the measured mean iteration count, refinement passes counted as 0.  frac_executed: the "
                                 "flops the kernel executes (dense workload: identity columns excluded)",
                         "hbm_algorithmic": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                             "frac": gbs / PEAK_HBM_GBS, "bytes_per_lp": b_survey,
                                             "bytes_per_lp_with_z": b_with_z}},
            "cpu_baseline": cpu,
            "cpu_port": cpu_all,
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
