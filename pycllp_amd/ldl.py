"""Batched LDL' / modified LDL' of explicit matrices on the GPU, and the solves built on them.

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


def _vec(b, B, n, dev):
    bt = b if isinstance(b, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(b, dtype=np.float64))
    bt = bt.to(device=dev, dtype=torch.float64).contiguous()
    if bt.ndim == 1:
        bt = bt.unsqueeze(0)
    if tuple(bt.shape) != (B, n):
        raise ValueError("right-hand side must be [%d] or [%d, %d]; got %r" % (n, B, n, tuple(bt.shape)))
    return bt


def _solve(A, b, modified, beta, delta, device=None):
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
    bt = _vec(b, B, n, dev)
    if modified and beta is None:
        beta = float(torch.sqrt(At.max()).item())      # pycllp/ldl.py:255
    x = torch.empty((B, n), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().pycllp_hip_ldl_solve(n, B, ctypes.c_void_p(At.data_ptr()), ctypes.c_void_p(bt.data_ptr()),
                                                        ctypes.c_void_p(x.data_ptr()), int(bool(modified)),
                                                        float(beta if beta is not None else 1.0), float(delta),
                                                        ctypes.c_void_p(st.cuda_stream)), "pycllp_hip_ldl_solve")
    torch.cuda.synchronize(dev)
    if not isinstance(A, torch.Tensor):
        x = x.cpu().numpy()
    return x[0] if single else x


def solve_ldl(A, b, device=None):
    """x = A^-1 b through A = L D L' (``pycllp/ldl.py:202-239``); one matrix per wavefront, factor in registers."""
    return _solve(A, b, False, None, 0.0, device)


def forward_backward_modified_ldl(A, b, delta=1e-6, beta=None, device=None):
    """x = A^-1 b through the modified factorisation formed on the fly (``pycllp/ldl.py:242-281``)."""
    return _solve(A, b, True, beta, delta, device)


def forward_backward_ldl(L, D, b, device=None):
    """x = (L D L')^-1 b for given factors (``pycllp/ldl.py:165-180``).  L: unit lower triangular ``[n, n]`` / ``[B, n, n]``."""
    if not torch.cuda.is_available():
        raise RuntimeError("pycllp_amd: no ROCm device visible -- the HIP LDL' kernels have no CPU fallback")
    single = (L.ndim == 2)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))).to(
        device=dev, dtype=torch.float64).contiguous()
    Lt, Dt = as_t(L), as_t(D)
    if single:
        Lt, Dt = Lt.unsqueeze(0), Dt.unsqueeze(0)
    B, n = int(Lt.shape[0]), int(Lt.shape[1])
    ti = torch.tril_indices(n, n, device=dev)
    Lp = Lt[:, ti[0], ti[1]].contiguous()
    bt = _vec(b, B, n, dev)
    x = torch.empty((B, n), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().pycllp_hip_forward_backward_ldl(n, B, ctypes.c_void_p(Lp.data_ptr()), ctypes.c_void_p(Dt.data_ptr()),
                                                                   ctypes.c_void_p(bt.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                                                   ctypes.c_void_p(st.cuda_stream)), "pycllp_hip_forward_backward_ldl")
    torch.cuda.synchronize(dev)
    if not isinstance(L, torch.Tensor):
        x = x.cpu().numpy()
    return x[0] if single else x


def forward_backward(L, U, b, device=None):
    """x = (L U)^-1 b for a lower/upper pair that is a scaled LDL' factorisation, U1 = L1' with L1 = L diag(l_ii)^-1 and
    U1 = diag(u_ii)^-1 U -- the two ways the reference's tests call it (``tests/test_ldl.py:119-130``): Cholesky factors
    (C, C') and (L D, L').  Then L U = L1 diag(l_ii u_ii) L1' (``pycllp/ldl.py:147-162``)."""
    Ln = L.cpu().numpy() if isinstance(L, torch.Tensor) else np.asarray(L, dtype=np.float64)
    Un = U.cpu().numpy() if isinstance(U, torch.Tensor) else np.asarray(U, dtype=np.float64)
    dl = np.diagonal(Ln, axis1=-2, axis2=-1)
    du = np.diagonal(Un, axis1=-2, axis2=-1)
    L1 = Ln / dl[..., None, :]
    U1 = Un / du[..., :, None]
    if not np.allclose(np.swapaxes(L1, -1, -2), U1, rtol=1e-12, atol=1e-14):
        raise NotImplementedError("forward_backward is provided for pairs with U = D2 L1' , L = L1 D1 only")
    return forward_backward_ldl(L1, dl * du, b, device)
