"""HIP host for the batched dense primal-normal interior-point solver.

Mirrors the contract of the reference's OpenCL host ``ClDensePrimalNormalSolver``
(``pycllp/solvers/cl.py:12-124``): the constructor takes optional device handles, ``init(lp)`` captures
the shared constraint matrix once, ``solve(lp)`` consumes the LP's current ``b``/``c`` and leaves
``self.x [nproblems, ncols]`` and ``self.status [nproblems]`` as attributes.  Differences, all
supersets: ``solve`` also returns ``self.status`` (as the reference's CPU solvers do,
``pycllp/solvers/normal_eqns.py:33``) and sets ``y, z, primal_obj, dual_obj, iters``.

PyTorch is plumbing only (device memory, streams); the compute is ``csrc/libpycllp_hip.so``.
"""
import ctypes

import numpy as np
import torch

from . import BaseSolver
from .. import _native


# order of the result arrays inside the one allocation that holds them (HipDensePrimalNormalSolver._buffers)
PACK_ORDER = (("pobj", torch.float64), ("dobj", torch.float64), ("status", torch.int32), ("iters", torch.int32),
              ("y", torch.float64), ("x", torch.float64), ("z", torch.float64))


def pack_layout(B, m, n):
    """(offset, nbytes, dtype, shape) of every result array of a B-LP solve in its packed allocation, 256-byte aligned
    sections; everything up to and including x -- what a result gather ships -- forms a contiguous prefix of
    ``_gather_bytes`` bytes (``pycllp_amd.dist.PackedGather``)."""
    shapes = {"pobj": (B,), "dobj": (B,), "status": (B,), "iters": (B,), "y": (B, m), "x": (B, n), "z": (B, n)}
    layout, off = {}, 0
    for name, dt in PACK_ORDER:
        nb = int(np.prod(shapes[name], dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
        layout[name] = (off, nb, dt, shapes[name])
        off = (off + nb + 255) & ~255
        if name == "x":
            layout["_gather_bytes"] = off
    layout["_total_bytes"] = off
    return layout


def unpack(packed, layout, names=("pobj", "dobj", "status", "iters", "y", "x")):
    """Views of the result arrays inside a packed byte buffer (e.g. one received from another rank)."""
    out = {}
    for name in names:
        off, nb, dt, shape = layout[name]
        out[name] = packed[off:off + nb].view(dt).view(shape)
    return out


# pycllp_hip_sparse_launch_info's `kernel` -> (variant, kernel family): 'wave' = the register-resident one-LP-per-wavefront
# kernel (csrc/ipm_wreg.hip) on Gram term tables or on a dense image of A, 'block' = the one-LP-per-workgroup kernel of
# csrc/ipm_block.inc (m <= 128), 'big' = the one-LP-per-workgroup kernel for large LPs (csrc/ipm_big.hip: 128 < m <= 256 or
# 512 < n <= 1280), Gram from a term list or on the matrix cores
SPARSE_KERNEL_KINDS = {0: ("block", "block"), 1: ("tables", "wave"), 2: ("dense image", "wave"),
                       3: ("term list", "big"), 4: ("MFMA Gram", "big")}
AUTOSCALE_BAND = (0.1, 10.0)


def autoscale_wanted(b, c):
    """The ``autoscale='auto'`` rule: True when, for any LP of the batch, max|b| or max|c| lies outside [0.1, 10] -- the
    start x = z = y = 1 and the unit floors of the tolerances (eps (1 + |b|)) are tuned to data of order 1 (DESIGN.md section
    2, scaling caveat: decades away from 1 cost 40-170 iterations and objective accuracy).  numpy arrays or torch tensors."""
    lo, hi = AUTOSCALE_BAND
    for v in (b, c):
        if isinstance(v, torch.Tensor):
            if v.numel() == 0:
                continue
            mx = v.abs().amax(dim=-1)
            if bool(((mx < lo) | (mx > hi)).any()):
                return True
        else:
            v = np.asarray(v)
            if v.size == 0:
                continue
            mx = np.abs(v).max(axis=-1)
            if ((mx < lo) | (mx > hi)).any():
                return True
    return False


def warm_lifted(x, lift=1e-3):
    """The start point ``warm_start=True`` makes of a previous x (or z): max(x, lift * max(1, |x|_inf)) per LP (numpy)."""
    x = np.asarray(x, dtype=np.float64)
    return np.maximum(x, lift * np.maximum(1.0, np.abs(x).max(axis=1, keepdims=True)))


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError("pycllp_amd: no ROCm device visible -- the HIP solvers have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


class HipDensePrimalNormalSolver(BaseSolver):
    """Drop-in for ``cl_dense_primal_normal`` on MI355X.  The LP must be in equality form
    (callers use ``StandardLP.to_equality_form()`` first, as for the OpenCL solver)."""
    name = 'hip_dense_primal_normal'

    def __init__(self, device=None, stream=None, keep_on_device=False, autoscale="auto", hsd="auto", warm_start=False,
                 predcorr=False, warm_lift=1e-3, **options):
        """``hsd=True`` (PYCLLP_FLAG_HSD) solves on the homogeneous self-dual embedding, the model of the reference's
        CPU solver ``pycllp/ipo/hsd.c``: infeasible (status 2) and unbounded (status 4) LPs are then detected reliably,
        after ~12-15 iterations, and ``x`` / ``y, z`` hold the certificate.  ``hsd="auto"`` (default) runs the reference's
        own path (``pycllp/cl/primal_normal.cl:201-284``) and then re-solves, on the embedding, exactly those LPs that did
        not end optimal: an optimal LP gets the drop-in behaviour, every other one a verdict with a certificate instead of
        the outcome of the reference kernel's 10x-growth heuristic (which labels most infeasible LPs "iteration limit").
        ``hsd=False`` is the reference's path alone.
        ``warm_start=True``: repeat solve -- the library's stated purpose (reference ``README.md:5-6``, the intent recorded
        at ``pycllp/cl/primal_normal.cl:213-219``): the solver keeps x, z, y of the previous ``solve()`` on the device and
        starts every LP that was optimal there from its previous point (PYCLLP_FLAG_WARM_START); the others, and a first
        solve or one with a different batch size, start from x = z = y = 1.  The previous optimum sits ON the boundary
        (x_j z_j ~ 1e-10), where an interior-point method restarts badly: measured on 1 %-perturbed data, 12-18 iterations on
        average with tails of 50-100 and about one jammed LP in 300.  ``warm_lift`` (default 1e-3; 0 = the raw point) lifts the
        start into the interior first -- x <- max(x, warm_lift max(1, |x|_inf)), likewise z: 10 iterations on average, at most
        19, no jam (cold: 20-23).
        ``autoscale=True`` (PYCLLP_FLAG_AUTOSCALE, not in the reference) solves every LP with b/max|b| and c/max|c| and
        scales the results back: for b or c orders of magnitude away from 1.  ``autoscale="auto"`` (default) switches it on
        in ``solve(lp)`` for a batch in which some LP has max|b| or max|c| outside [0.1, 10] (``autoscale_wanted``) and
        leaves a batch inside that band -- the reference's test and benchmark regime -- on the reference's arithmetic bit
        for bit; ``solve_device`` (asynchronous, no look at the data) treats "auto" as off.
        ``predcorr=True`` (PYCLLP_FLAG_PREDCORR, not in the reference's kernel): Mehrotra's predictor-corrector step on the same
        Gram / LDL' machinery -- one factorisation and two solves per iteration, the same optimum in ~27 % fewer iterations at
        the reference's step fraction r = 0.9 and ~45 % fewer at r = 0.99; the default stays the reference's rule.  With the
        default ``hsd="auto"`` the LPs that do not end optimal are still re-solved on the embedding.  Other keyword arguments are
        the fields of ``pycllp_hip_opts`` (eps, delta, r, pivot_floor, refine_tol, max_iter, max_refine, flags)."""
        super(HipDensePrimalNormalSolver, self).__init__()
        if isinstance(hsd, str):
            if hsd != "auto":
                raise ValueError("hsd must be True, False or 'auto'")
        elif isinstance(hsd, (bool, np.bool_, int, np.integer)):
            hsd = bool(hsd)           # 1 / np.bool_(True) mean True (`hsd is True` tests below)
        else:
            raise ValueError("hsd must be True, False or 'auto'")
        if predcorr:
            if hsd is True:
                raise ValueError("predcorr is an option of the reference's path; it cannot be combined with hsd=True")
            options["flags"] = int(options.get("flags", 0)) | _native.FLAG_PREDCORR
        if isinstance(autoscale, str):
            if autoscale != "auto":
                raise ValueError("autoscale must be True, False or 'auto'")
        else:
            autoscale = bool(autoscale)
        if int(options.get("flags", 0)) & _native.FLAG_AUTOSCALE:
            autoscale = True
        if autoscale is True:
            options["flags"] = int(options.get("flags", 0)) | _native.FLAG_AUTOSCALE
        elif autoscale == "auto" and int(options.get("flags", 0)) & _native.FLAG_WAVE_KERNEL:
            autoscale = False
        self.autoscale = autoscale
        self._extra_flags = 0        # flags solve(lp) adds for one call (autoscale='auto')
        if hsd is True or (int(options.get("flags", 0)) & _native.FLAG_HSD):
            options["flags"] = int(options.get("flags", 0)) | _native.FLAG_HSD
            hsd = True
        if int(options.get("flags", 0)) & _native.FLAG_WAVE_KERNEL and hsd == "auto":
            hsd = False          # the first-generation kernel has no embedding
        self.hsd = hsd
        self.warm_start = bool(warm_start)
        self.warm_lift = float(warm_lift)
        if not (self.warm_lift >= 0.0):
            raise ValueError("warm_lift must be >= 0")
        self._prev_B = None
        self.device = device
        self.stream = stream
        self.keep_on_device = keep_on_device
        self.options = dict(options)
        _native.default_opts(**self.options) if options else None  # validate names early
        self._handle = None
        self.buffers = {}
        self._a_row0 = None

    # -- helpers ---------------------------------------------------------------------------------
    def _stream_ptr(self):
        st = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(st.cuda_stream)

    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

    def _free(self):
        if self._handle is not None:
            _native.lib().pycllp_hip_dense_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    @staticmethod
    def consume(lp):
        """Everything a HIP host reads from an LP object -- the reference's ``EqualityLP`` or this package's -- as plain
        numpy: the attribute surface of ``pycllp/solvers/cl.py:35-39,99,102`` (``nrows, ncols, nproblems, A.todense(), b, c``)
        plus ``f``.  tools/check_reference_boundary.py runs this on reference-built objects."""
        m, n = int(lp.nrows), int(lp.ncols)
        A = lp.A.todense() if hasattr(lp.A, "todense") else lp.A
        A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        if A.shape != (m, n):
            raise ValueError("A has shape %r, expected (%d, %d)" % (A.shape, m, n))
        return dict(m=m, n=n, nproblems=int(lp.nproblems), A=A,
                    b=np.ascontiguousarray(np.asarray(lp.b, dtype=np.float64)), c=np.ascontiguousarray(np.asarray(lp.c, dtype=np.float64)),
                    f=np.atleast_1d(np.asarray(getattr(lp, "f", 0.0), dtype=np.float64)))

    # -- plugin API ------------------------------------------------------------------------------
    def init(self, lp, verbose=0):
        """Densify and upload A once (``pycllp/solvers/cl.py:39,46``)."""
        self.device = _require_gpu(self.device)
        L = _native.lib()
        self._delegate = None
        if type(self) is HipDensePrimalNormalSolver and getattr(lp.A, "nproblems", 1) > 1:
            # per-problem values of A: served by the sparse path's per-problem kernel (one LP per workgroup, values from HBM)
            d = HipSparsePrimalNormalSolver(device=self.device, stream=self.stream, keep_on_device=self.keep_on_device,
                                            autoscale=self.autoscale, hsd=self.hsd, warm_start=self.warm_start, warm_lift=self.warm_lift,
                                            **self.options)
            d.init(lp, verbose=verbose)
            self._delegate, self.m, self.n = d, d.m, d.n
            return
        m, n = int(lp.nrows), int(lp.ncols)
        A = lp.A.todense() if hasattr(lp.A, "todense") else lp.A
        A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        if A.shape != (m, n):
            raise ValueError("A has shape %r, expected (%d, %d)" % (A.shape, m, n))
        if verbose > 0:
            print("Initializing HipDensePrimalNormalSolver (m=%d, n=%d) on %s" % (m, n, self.device))
        self._free()
        with torch.cuda.device(self.device):
            A_dev = self._dev(A)
            h = ctypes.c_void_p()
            _native.check(L.pycllp_hip_dense_init(m, n, self._ptr(A_dev), self._stream_ptr(), ctypes.byref(h)),
                          "pycllp_hip_dense_init")
        self._handle = h
        self.m, self.n = m, n
        self.buffers = {}

    def _pack_layout(self, B):
        return pack_layout(B, self.m, self.n)

    @staticmethod
    def unpack(packed, layout, names=("pobj", "dobj", "status", "iters", "y", "x")):
        return unpack(packed, layout, names)

    def _buffers(self, B, slot=0):
        """Output tensors of one solve: views into ONE device allocation (``packed``, bytes), so that a caller who ships
        the results elsewhere -- the multi-GPU gather -- moves a single contiguous buffer (its first ``gather_bytes``
        bytes hold pobj, dobj, status, iters, y, x) instead of six.  ``slot`` selects one of several independent sets so
        that the results of solve k stay alive (e.g. while they are being gathered) during solve k+1."""
        key = "set%d" % slot
        cur = self.buffers.get(key)
        if cur is None or cur["B"] != B:
            layout = self._pack_layout(B)
            packed = torch.empty(max(layout["_total_bytes"], 256), dtype=torch.uint8, device=self.device)
            cur = dict(B=B, packed=packed, layout=layout, gather_bytes=layout["_gather_bytes"])
            cur.update(self.unpack(packed, layout, names=[n for n, _ in PACK_ORDER]))
            self.buffers[key] = cur
        return cur

    def solve_device(self, b, c, warm_start=False, slot=0, **options):
        """Device-resident entry: b [B,m], c [B,n] (torch CUDA or numpy) -> dict of CUDA tensors.
        Asynchronous on the solver's stream."""
        if getattr(self, "_delegate", None) is not None:
            return self._delegate.solve_device(b, c, warm_start=warm_start, slot=slot, **options)
        if self._handle is None:
            raise RuntimeError("solve() called before init()")
        b = self._dev(b); c = self._dev(c)
        if b.ndim != 2 or c.ndim != 2 or b.shape[1] != self.m or c.shape[1] != self.n or b.shape[0] != c.shape[0]:
            raise ValueError("b must be [B,%d] and c [B,%d] with equal B; got %r and %r"
                             % (self.m, self.n, tuple(b.shape), tuple(c.shape)))
        B = int(b.shape[0])
        buf = self._buffers(B, slot)
        opts = dict(self.options); opts.update(options)
        opts["flags"] = int(opts.get("flags", 0)) | self._extra_flags
        if warm_start:
            opts["flags"] = int(opts.get("flags", 0)) | _native.FLAG_WARM_START
        o = _native.default_opts(**opts)
        if o.max_iter < 1 or o.max_refine < _native.MAX_REFINE_AUTO or not (o.eps > 0):
            raise ValueError("max_iter must be >= 1, max_refine >= 0 (or -1 = auto) and eps > 0")
        with torch.cuda.device(self.device):
            self._launch(B, b, c, buf, o)
        self._keepalive = (b, c)
        return buf

    def _launch(self, B, b, c, buf, o):
        _native.check(_native.lib().pycllp_hip_dense_solve(
            self._handle, B, self._ptr(b), self._ptr(c), self._ptr(buf["x"]), self._ptr(buf["y"]),
            self._ptr(buf["z"]), self._ptr(buf["pobj"]), self._ptr(buf["dobj"]), self._ptr(buf["status"]),
            self._ptr(buf["iters"]), ctypes.byref(o), self._stream_ptr()), "pycllp_hip_dense_solve")

    # -- host-to-host path -----------------------------------------------------------------------
    PIPELINE_MIN_BATCH = 8192   # below this one upload / launch / download is as fast
    PIPELINE_CHUNKS = 8

    def _host_staging(self, B):
        """Page-locked host images of b, c and of every result, allocated once per batch size (the counterpart of
        the host arrays ``pycllp/solvers/cl.py:40-41`` keeps) plus the two copy streams of the pipeline."""
        st = self.buffers.get("host")
        if st is None or st["B"] != B:
            f64, i32 = torch.float64, torch.int32
            pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            st = dict(B=B, b=pin((B, self.m), f64), c=pin((B, self.n), f64),
                      x=pin((B, self.n), f64), z=pin((B, self.n), f64), y=pin((B, self.m), f64),
                      pobj=pin((B,), f64), dobj=pin((B,), f64), status=pin((B,), i32), iters=pin((B,), i32),
                      db=torch.empty((B, self.m), dtype=f64, device=self.device),
                      dc=torch.empty((B, self.n), dtype=f64, device=self.device),
                      s_in=torch.cuda.Stream(self.device), s_out=torch.cuda.Stream(self.device))
            st["hb"], st["hc"] = st["b"].numpy(), st["c"].numpy()
            self.buffers["host"] = st
        return st

    def _solve_host(self, b, c, warm=False):
        """numpy in, numpy out, as a three-stage pipeline over row chunks of the batch: upload chunk k+1, solve
        chunk k and download chunk k-1 run on three HIP streams, so the PCIe time hides behind the kernel.  The
        returned arrays are views of the solver's page-locked buffers and are overwritten by the next solve()
        (the reference's ``self.x`` behaves the same way, ``pycllp/solvers/cl.py:40,119``)."""
        b = np.ascontiguousarray(b, dtype=np.float64); c = np.ascontiguousarray(c, dtype=np.float64)
        if b.ndim != 2 or c.ndim != 2 or b.shape[1] != self.m or c.shape[1] != self.n or b.shape[0] != c.shape[0]:
            raise ValueError("b must be [B,%d] and c [B,%d] with equal B; got %r and %r"
                             % (self.m, self.n, tuple(b.shape), tuple(c.shape)))
        B = int(b.shape[0])
        st, buf = self._host_staging(B), self._buffers(B, 0)
        nchunk = max(1, min(self.PIPELINE_CHUNKS, B // self.PIPELINE_MIN_BATCH))
        edges = [B * k // nchunk for k in range(nchunk + 1)]
        compute = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        outs = ("x", "y", "z", "pobj", "dobj", "status", "iters")

        def stage_and_upload(lo, hi):
            # plain memcpy on this thread: torch's CPU copy_ would wake its whole intra-op thread pool, whose
            # spin-waiting burns a container's CPU quota and shows up as sporadic ~80 ms stalls
            np.copyto(st["hb"][lo:hi], b[lo:hi]); np.copyto(st["hc"][lo:hi], c[lo:hi])
            with torch.cuda.stream(st["s_in"]):
                st["db"][lo:hi].copy_(st["b"][lo:hi], non_blocking=True)
                st["dc"][lo:hi].copy_(st["c"][lo:hi], non_blocking=True)
                return st["s_in"].record_event()

        opts = dict(self.options)
        if warm:
            opts["flags"] = int(opts.get("flags", 0)) | _native.FLAG_WARM_START
        o = _native.default_opts(**opts)
        if o.max_iter < 1 or o.max_refine < _native.MAX_REFINE_AUTO or not (o.eps > 0):
            raise ValueError("max_iter must be >= 1, max_refine >= 0 (or -1 = auto) and eps > 0")
        base_flags = int(o.flags)
        # autoscale='auto' decided on the device (self._extra_flags is None): the chunks are solved WITHOUT the flag while their
        # band tests (two row reductions each, queued behind the chunk's upload) accumulate; if some LP of the batch turns out
        # to lie outside the band -- b or c decades away from 1: LPs that take 40-170 iterations unscaled -- the batch is solved
        # once more with the flag from the data already on the device.  Data inside the band (the common case) keeps the
        # full overlap of staging, upload, solve and download.
        if self._extra_flags is None and warm:
            # (a warm start reads the x, z, y the first pass would overwrite: decide before launching anything)
            self._extra_flags = _native.FLAG_AUTOSCALE if autoscale_wanted(b, c) else 0
        speculative = self._extra_flags is None
        o.flags = base_flags | (0 if speculative else int(self._extra_flags))
        lo_, hi_ = AUTOSCALE_BAND
        tests = []
        with torch.cuda.device(self.device):
            ups = []
            for lo, hi in zip(edges[:-1], edges[1:]):
                up = stage_and_upload(lo, hi)
                ups.append(up)
                if speculative and hi > lo:
                    with torch.cuda.stream(st["s_in"]):
                        for v in (st["db"][lo:hi], st["dc"][lo:hi]):
                            mx = v.abs().amax(dim=-1)
                            tests.append(((mx < lo_) | (mx > hi_)).any())
                compute.wait_event(up)
                part = {k: buf[k][lo:hi] for k in outs}
                self._a_row0 = lo          # (per-problem A values of the sparse solver: rows of this chunk)
                self._launch(hi - lo, st["db"][lo:hi], st["dc"][lo:hi], part, o)
                self._a_row0 = None
                done = compute.record_event()
                with torch.cuda.stream(st["s_out"]):
                    st["s_out"].wait_event(done)
                    for k in outs:
                        st[k][lo:hi].copy_(part[k], non_blocking=True)
            if speculative:
                self._extra_flags = 0
                if tests and bool(torch.stack(tests).any()):
                    self._extra_flags = _native.FLAG_AUTOSCALE        # (the hsd='auto' re-solve of this call runs with it too)
                    o.flags = base_flags | _native.FLAG_AUTOSCALE
                    st["s_out"].synchronize()
                    for lo, hi in zip(edges[:-1], edges[1:]):
                        part = {k: buf[k][lo:hi] for k in outs}
                        self._a_row0 = lo
                        self._launch(hi - lo, st["db"][lo:hi], st["dc"][lo:hi], part, o)
                        self._a_row0 = None
                        done = compute.record_event()
                        with torch.cuda.stream(st["s_out"]):
                            st["s_out"].wait_event(done)
                            for k in outs:
                                st[k][lo:hi].copy_(part[k], non_blocking=True)
            st["s_out"].synchronize()
        return {k: st[k].numpy() for k in outs}

    def _prepare_warm(self, B):
        """True when this solve may start from the previous solution: same batch size, a previous solve exists.  LPs that
        did not end optimal last time are reset to the cold start x = z = y = 1 (their x, z may hold a certificate)."""
        if not self.warm_start or self._prev_B != B or "set0" not in self.buffers:
            return False
        buf = self.buffers["set0"]
        with self._on_solver_stream():       # the kernel that reads x, z, y is launched on the solver's stream
            bad = buf["status"] != 0
            if bool(bad.any()):
                buf["x"][bad] = 1.0; buf["z"][bad] = 1.0; buf["y"][bad] = 0.0 if self.hsd is True else 1.0
            if self.warm_lift > 0.0:      # off the boundary (see __init__): the same rule as warm_lifted() below
                for k in ("x", "z"):
                    v = buf[k]
                    floor = self.warm_lift * torch.clamp(v.abs().amax(dim=1, keepdim=True), min=1.0)
                    torch.maximum(v, floor, out=v)
        return True

    def _on_solver_stream(self):
        """Context in which torch tensor work is queued on the stream the solve kernels run on (gathers and
        resets issued on torch's current stream raced a kernel launched on a caller-supplied stream)."""
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _resolve_non_optimal(self, b, c, res):
        """hsd='auto': the LPs that did not end optimal are solved again on the homogeneous self-dual embedding and their
        results replace the first verdict.  ``res``: dict of numpy arrays or CUDA tensors of the first solve."""
        status = res["status"]
        idx = torch.nonzero(status != 0).flatten() if isinstance(status, torch.Tensor) else np.flatnonzero(status != 0)
        if len(idx) == 0:
            return
        flags = (int(self.options.get("flags", 0)) | _native.FLAG_HSD) & ~(_native.FLAG_WARM_START | _native.FLAG_PREDCORR)
        full_values = getattr(self, "_a_values", None)
        with self._on_solver_stream():       # gathers, launch and scatters in ONE stream order
            if isinstance(b, torch.Tensor) or isinstance(c, torch.Tensor):
                it = idx.to(self.device) if isinstance(idx, torch.Tensor) else torch.as_tensor(idx, device=self.device)
                b2, c2 = self._dev(b)[it], self._dev(c)[it]
            else:
                hi = idx.cpu().numpy() if isinstance(idx, torch.Tensor) else idx
                b2, c2 = np.asarray(b)[hi], np.asarray(c)[hi]
            if full_values is not None:      # per-problem A values (sparse solver): the rows of the LPs being re-solved
                it2 = idx.to(self.device) if isinstance(idx, torch.Tensor) else torch.as_tensor(idx, device=self.device)
                self._a_values = full_values[it2].contiguous()
            try:
                r2 = self.solve_device(b2, c2, slot=3, flags=flags)
            finally:
                if full_values is not None:
                    self._a_values = full_values
            for k in ("x", "y", "z", "pobj", "dobj", "status", "iters"):
                if isinstance(res[k], torch.Tensor):
                    res[k][idx.to(res[k].device) if isinstance(idx, torch.Tensor) else torch.as_tensor(idx, device=res[k].device)] = r2[k]
            torch.cuda.synchronize(self.device)
        for k in ("x", "y", "z", "pobj", "dobj", "status", "iters"):
            if not isinstance(res[k], torch.Tensor):
                res[k][idx.cpu().numpy() if isinstance(idx, torch.Tensor) else idx] = r2[k].cpu().numpy()

    def solve(self, lp, verbose=0):
        """Solve every problem of ``lp`` (current ``lp.b``, ``lp.c``); results in attributes."""
        if getattr(self, "_delegate", None) is not None:
            d = self._delegate
            d.solve(lp, verbose=verbose)
            for k in ("x", "y", "z", "status", "iters", "primal_obj", "dual_obj"):
                setattr(self, k, getattr(d, k))
            return self.status
        if int(lp.nrows) != self.m or int(lp.ncols) != self.n:
            raise ValueError("LP shape changed since init(): (%d,%d) vs (%d,%d)" % (lp.nrows, lp.ncols, self.m, self.n))
        if self._handle is None:
            raise RuntimeError("solve() called before init()")
        if verbose > 0:
            print("Solving %d LPs with %s..." % (lp.nproblems, type(self).__name__))
        f = np.asarray(getattr(lp, "f", 0.0), dtype=np.float64)
        B = int(lp.nproblems)
        # autoscale='auto': the band test of the whole batch.  For host arrays headed for the host pipeline it is left to the
        # device (_solve_host: after the upload, two reductions): on 65 536 x (32 + 96) doubles the numpy version of the test
        # took 24 ms of a 9 ms host-to-host solve.
        host_pipeline = (not isinstance(lp.b, torch.Tensor) and not isinstance(lp.c, torch.Tensor) and not self.keep_on_device)
        if self.autoscale == "auto" and host_pipeline and B * (self.m + self.n) >= (1 << 16):
            self._extra_flags = None       # (small batches: the numpy test costs microseconds, the device test ~8 launches)
        else:
            self._extra_flags = _native.FLAG_AUTOSCALE if (self.autoscale == "auto" and autoscale_wanted(lp.b, lp.c)) else 0
        try:
            return self._solve_plugin(lp, f, B, verbose)
        finally:
            self._extra_flags = 0

    def _solve_plugin(self, lp, f, B, verbose):
        warm = self._prepare_warm(B)
        on_host = not isinstance(lp.b, torch.Tensor) and not isinstance(lp.c, torch.Tensor)
        if on_host and not self.keep_on_device:
            res = self._solve_host(lp.b, lp.c, warm)
            if self.hsd == "auto":
                self._resolve_non_optimal(lp.b, lp.c, res)
            self.x, self.y, self.z = res["x"], res["y"], res["z"]
            self.status, self.iters = res["status"], res["iters"]
            # objective offset f is added when reporting (as pycllp/solvers/pathfollowing.py:113-114)
            self.primal_obj, self.dual_obj = res["pobj"] + f, res["dobj"] + f
        else:
            buf = self.solve_device(lp.b, lp.c, warm_start=warm)
            torch.cuda.synchronize(self.device)
            res = {k: buf[k] for k in ("x", "y", "z", "pobj", "dobj", "status", "iters")}
            if self.hsd == "auto":
                self._resolve_non_optimal(lp.b, lp.c, res)
            if self.keep_on_device:
                self.x, self.y, self.z = res["x"], res["y"], res["z"]
                self.status, self.iters = res["status"], res["iters"]
                ft = torch.as_tensor(np.broadcast_to(f, (buf["B"],)).copy(), device=self.device)
                self.primal_obj, self.dual_obj = res["pobj"] + ft, res["dobj"] + ft
            else:
                self.x = res["x"].cpu().numpy(); self.y = res["y"].cpu().numpy(); self.z = res["z"].cpu().numpy()
                self.status = res["status"].cpu().numpy(); self.iters = res["iters"].cpu().numpy()
                self.primal_obj = res["pobj"].cpu().numpy() + f
                self.dual_obj = res["dobj"].cpu().numpy() + f
        self._prev_B = B
        if verbose > 0:
            print("Solve complete.")
        return self.status

    def newton_step(self, x, z, y, b, c, mu, **options):
        """Stand-alone Newton step dy for B states (the reference's ``solve_primal_normal`` kernel,
        ``pycllp/cl/ldl.cl:602-653``, as launched by its ``tests/test_ldl.py:219-273``)."""
        if getattr(self, "_delegate", None) is not None:
            return self._delegate.newton_step(x, z, y, b, c, mu, **options)
        if self._handle is None:
            raise RuntimeError("newton_step() called before init()")
        x, z, y, b, c = [self._dev(np.atleast_2d(v) if not isinstance(v, torch.Tensor) else v) for v in (x, z, y, b, c)]
        B = int(x.shape[0])
        dy = torch.empty((B, self.m), dtype=torch.float64, device=self.device)
        nref = torch.empty(B, dtype=torch.int32, device=self.device)
        opts = dict(self.options); opts.update(options)
        o = _native.default_opts(**opts)
        with torch.cuda.device(self.device):
            _native.check(_native.lib().pycllp_hip_dense_newton(
                self._handle, B, self._ptr(x), self._ptr(z), self._ptr(y), self._ptr(b), self._ptr(c), float(mu),
                self._ptr(dy), self._ptr(nref), ctypes.byref(o), self._stream_ptr()), "pycllp_hip_dense_newton")
        torch.cuda.synchronize(self.device)
        self.nrefine = nref.cpu().numpy()
        return dy.cpu().numpy()

    def launch_info(self):
        if getattr(self, "_delegate", None) is not None:
            return self._delegate.launch_info()
        vals = [ctypes.c_int() for _ in range(5)]
        _native.check(_native.lib().pycllp_hip_dense_launch_info(self._handle, *[ctypes.byref(v) for v in vals]),
                      "pycllp_hip_dense_launch_info")
        d = dict(zip(("grid", "block", "lds_bytes", "m_pad", "n_pad"), [v.value for v in vals]))
        kind = _native.lib().pycllp_hip_dense_kernel_kind(self._handle)
        if kind >= 0:      # beyond the lane-group kernels: which of the sparse path's kernels served the last launch
            d["variant"], d["kernel"] = SPARSE_KERNEL_KINDS[kind]
        return d


class HipSparsePrimalNormalSolver(HipDensePrimalNormalSolver):
    """Drop-in for ``cl_sparse_primal_normal`` (``pycllp/solvers/cl.py:127-278``): shared SPARSE constraint matrix, one LP
    per workgroup, m <= 128 rows and n <= 512 columns (equality form).  Same contract and result attributes as the
    dense solver; ``init`` takes the CSR arrays of ``lp.A`` (the reference builds them from the densified A,
    ``cl.py:175-178``), the transposed copy and the structure of A diag(x/z) A' are derived inside the library."""
    name = 'hip_sparse_primal_normal'

    def _free(self):
        if self._handle is not None:
            _native.lib().pycllp_hip_sparse_free(self._handle)
            self._handle = None

    def init(self, lp, verbose=0):
        import scipy.sparse as sp
        self.device = _require_gpu(self.device)
        L = _native.lib()
        m, n = int(lp.nrows), int(lp.ncols)
        self._a_perm = None
        if getattr(lp.A, "nproblems", 1) > 1:
            # per-problem values on one structure (SparseMatrix.data[nproblems, nnz], pycllp/lp.py:16-54): the structure
            # comes from the coordinate lists, the values of problem 0 only stand in at init
            rows, cols = np.asarray(lp.A._rows), np.asarray(lp.A._cols)
            perm = np.lexsort((cols, rows))
            if perm.size and ((np.diff(rows[perm]) == 0) & (np.diff(cols[perm]) == 0)).any():
                raise ValueError("per-problem A: duplicate (row, column) entries in the structure")
            self._a_perm = perm
            indptr = np.zeros(m + 1, dtype=np.int64)
            np.add.at(indptr, rows + 1, 1)
            A = sp.csr_matrix((np.asarray(lp.A.data[0], dtype=np.float64)[perm], cols[perm], np.cumsum(indptr)), shape=(m, n))
        else:
            if hasattr(lp.A, "tocsr"):
                A = sp.csr_matrix(lp.A.tocsr())
            else:
                A = sp.csr_matrix(np.asarray(lp.A.todense() if hasattr(lp.A, "todense") else lp.A, dtype=np.float64))
            A = sp.csr_matrix(A, shape=(m, n))
            A.sum_duplicates(); A.eliminate_zeros(); A.sort_indices()
        if verbose > 0:
            print("Initializing HipSparsePrimalNormalSolver (m=%d, n=%d, nnz=%d) on %s" % (m, n, A.nnz, self.device))
        self._free()
        with torch.cuda.device(self.device):
            data = torch.as_tensor(np.ascontiguousarray(A.data, dtype=np.float64), device=self.device)
            indptr = torch.as_tensor(np.ascontiguousarray(A.indptr, dtype=np.int32), device=self.device)
            indices = torch.as_tensor(np.ascontiguousarray(A.indices, dtype=np.int32), device=self.device)
            h = ctypes.c_void_p()
            _native.check(L.pycllp_hip_sparse_init(m, n, int(A.nnz), self._ptr(data), self._ptr(indptr), self._ptr(indices),
                                                   self._stream_ptr(), ctypes.byref(h)), "pycllp_hip_sparse_init")
        self._handle = h
        self.m, self.n = m, n
        self.buffers = {}
        self._a_values = None

    def solve(self, lp, verbose=0):
        if self._a_perm is not None:
            data = np.asarray(lp.A.data, dtype=np.float64)
            if data.shape != (int(lp.nproblems), self._a_perm.size):
                raise ValueError("per-problem A: lp.A.data must be [nproblems, nnz] = (%d, %d); got %r"
                                 % (lp.nproblems, self._a_perm.size, data.shape))
            self._a_values = torch.as_tensor(np.ascontiguousarray(data[:, self._a_perm]), device=self.device)
        return super(HipSparsePrimalNormalSolver, self).solve(lp, verbose=verbose)

    def _launch(self, B, b, c, buf, o):
        if self._a_perm is not None:
            av = self._a_values
            if av is None or av.shape[0] < B:
                raise ValueError("per-problem A: no values for this batch (use lp.solve(solver))")
            # a chunk of the host pipeline, or the sub-batch of an hsd='auto' re-solve, addresses its rows of the values
            # through the offset of its b inside the full batch
            lo = 0
            if getattr(self, "_a_row0", None) is not None:
                lo = self._a_row0
            _native.check(_native.lib().pycllp_hip_sparse_solve_batch(
                self._handle, B, self._ptr(av[lo:lo + B]), self._ptr(b), self._ptr(c), self._ptr(buf["x"]), self._ptr(buf["y"]),
                self._ptr(buf["z"]), self._ptr(buf["pobj"]), self._ptr(buf["dobj"]), self._ptr(buf["status"]),
                self._ptr(buf["iters"]), ctypes.byref(o), self._stream_ptr()), "pycllp_hip_sparse_solve_batch")
            return
        _native.check(_native.lib().pycllp_hip_sparse_solve(
            self._handle, B, self._ptr(b), self._ptr(c), self._ptr(buf["x"]), self._ptr(buf["y"]),
            self._ptr(buf["z"]), self._ptr(buf["pobj"]), self._ptr(buf["dobj"]), self._ptr(buf["status"]),
            self._ptr(buf["iters"]), ctypes.byref(o), self._stream_ptr()), "pycllp_hip_sparse_solve")

    def newton_step(self, x, z, y, b, c, mu, **options):
        """Stand-alone Newton step dy for B states with the sparse shared A: the reference's ``sparse_solve_primal_normal``
        kernel (``pycllp/cl/ldl.cl:656-712``) as launched by its ``tests/test_ldl.py:276-361``."""
        if self._handle is None:
            raise RuntimeError("newton_step() called before init()")
        if self._a_perm is not None:
            # the handle holds problem 0's values only: a step for LP k > 0 would silently use the wrong matrix
            raise NotImplementedError("newton_step() with per-problem values of A: the stand-alone Newton entry takes one "
                                      "shared matrix (pycllp/cl/ldl.cl:656-712); init a solver per matrix instead")
        x, z, y, b, c = [self._dev(np.atleast_2d(v) if not isinstance(v, torch.Tensor) else v) for v in (x, z, y, b, c)]
        B = int(x.shape[0])
        dy = torch.empty((B, self.m), dtype=torch.float64, device=self.device)
        nref = torch.empty(B, dtype=torch.int32, device=self.device)
        opts = dict(self.options); opts.update(options)
        o = _native.default_opts(**opts)
        with torch.cuda.device(self.device):
            _native.check(_native.lib().pycllp_hip_sparse_newton(
                self._handle, B, self._ptr(x), self._ptr(z), self._ptr(y), self._ptr(b), self._ptr(c), float(mu),
                self._ptr(dy), self._ptr(nref), ctypes.byref(o), self._stream_ptr()), "pycllp_hip_sparse_newton")
        torch.cuda.synchronize(self.device)
        self.nrefine = nref.cpu().numpy()
        return dy.cpu().numpy()

    def launch_info(self):
        """grid / block / LDS bytes of the last solve and which kernel ran it (``SPARSE_KERNEL_KINDS``)."""
        vals = [ctypes.c_int() for _ in range(4)]
        _native.check(_native.lib().pycllp_hip_sparse_launch_info(self._handle, *[ctypes.byref(v) for v in vals]),
                      "pycllp_hip_sparse_launch_info")
        d = dict(zip(("grid", "block", "lds_bytes", "kernel"), [v.value for v in vals]))
        d["variant"], d["kernel"] = SPARSE_KERNEL_KINDS[d["kernel"]]
        return d
