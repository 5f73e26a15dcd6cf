"""Batched LDL' / modified LDL' of explicit matrices on the GPU.

Mirrors the reference's ``pycllp/ldl.py`` prototypes ``ldl(A)`` (:115-128) and ``modified_ldl(A, delta)``
(:58-90) -- same names, same ``(D, L)`` return order, L unit-lower-triangular -- and its OpenCL test kernels
``ldl`` / ``modified_ldl`` (``pycllp/cl/ldl.cl:28-107``).  ``A`` may be one matrix ``[n, n]`` or a batch ``[B, n, n]``
(numpy or CUDA tensor); the work is done by ``pycllp_hip_ldl`` in ``csrc/libpycllp_hip.so`` (no CPU fallback).
"""
import ctypes

import numpy as np
import torch

from . import _native


def _run(A, modified, beta, delta, device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("pycllp_amd: no ROCm device visible -- the HIP LDL' kernels have no CPU fallback")
    single = (A.ndim == 2)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    At = A if isinstance(A, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(A, dtype=np.float64))
    At = At.to(device=dev, dtype=torch.float64).contiguous()
    if single:
        At = At.unsqueeze(0)
    if At.ndim != 3 or At.shape[1] != At.shape[2]:
        raise ValueError("A must be [n, n] or [B, n, n]; got %r" % (tuple(A.shape),))
    B, n = int(At.shape[0]), int(At.shape[1])
    if modified and beta is None:
        # beta = sqrt(max A) over the matrix (pycllp/ldl.py:72); one value per launch, as the reference test uses
        beta = float(torch.sqrt(At.max()).item())
    Lp = torch.empty((B, n * (n + 1) // 2), dtype=torch.float64, device=dev)
    D = torch.empty((B, n), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().pycllp_hip_ldl(n, B, ctypes.c_void_p(At.data_ptr()), ctypes.c_void_p(Lp.data_ptr()),
                                                  ctypes.c_void_p(D.data_ptr()), int(bool(modified)),
                                                  float(beta if beta is not None else 1.0), float(delta),
                                                  ctypes.c_void_p(st.cuda_stream)), "pycllp_hip_ldl")
    torch.cuda.synchronize(dev)
    L = torch.zeros((B, n, n), dtype=torch.float64, device=dev)
    ti = torch.tril_indices(n, n, device=dev)
    L[:, ti[0], ti[1]] = Lp
    if isinstance(A, torch.Tensor):
        return (D[0], L[0]) if single else (D, L)
    D, L = D.cpu().numpy(), L.cpu().numpy()
    return (D[0], L[0]) if single else (D, L)


def ldl(A, device=None):
    """(D, L) with A = L diag(D) L'."""
    return _run(A, False, None, 0.0, device)


def modified_ldl(A, delta=1e-6, beta=None, device=None):
    """(D, L) of the modified factorisation (Nocedal & Wright alg. 3.4 diagonal guard)."""
    return _run(A, True, beta, delta, device)
