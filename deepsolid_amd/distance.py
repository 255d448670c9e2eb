"""Minimal-image dispatch and the batched PBC wrap (reference DeepSolid/distance.py).

The device kernels implement the three distance paths; this module decides,
exactly like ``MinimalImageDistance.__init__`` (distance.py:41-61), which one a
lattice takes and offers the batched ``enforce_pbc`` (distance.py:144-163) on
the GPU.
"""
import numpy as np

DIAGONAL, ORTHOGONAL, GENERAL = 0, 1, 2
ORTHO_TOL = 1e-10


def minimal_image_mode(latvec):
    """Reproduces the reference test verbatim -- including that the
    orthogonality test compares signed dot products, so lattices whose pairwise
    dot products are all negative (bcc primitive) take the `orthogonal` path."""
    latvec = np.asarray(latvec, dtype=np.float64)
    if np.all(np.abs(latvec - np.diag(np.diagonal(latvec))) < ORTHO_TOL):
        return DIAGONAL
    if (np.dot(latvec[0], latvec[1]) < ORTHO_TOL and np.dot(latvec[1], latvec[2]) < ORTHO_TOL
            and np.dot(latvec[2], latvec[0]) < ORTHO_TOL):
        return ORTHOGONAL
    return GENERAL


def minimal_image_host(latvec, mode, d):
    """Host version for a single displacement (used for the ion-ion constant only)."""
    latvec = np.asarray(latvec, dtype=np.float64)
    if mode == DIAGONAL:
        ld = np.diagonal(latvec)
        return np.mod(d + ld / 2, ld) - ld / 2
    if mode == ORTHOGONAL:
        frac = d @ np.linalg.inv(latvec)
        return (np.mod(frac + 0.5, 1.0) - 0.5) @ latvec
    mesh = np.meshgrid(*[np.array([0, 1, 2])] * 3)                # default 'xy' indexing, distance.py:66
    pts = np.stack([m.ravel() for m in mesh], axis=0).T - 1
    cand = d[None, :] + pts @ latvec
    return cand[np.argmin(np.linalg.norm(cand, axis=1))]


def enforce_pbc(latvec, epos):
    """Batched wrap into the cell spanned by ``latvec`` (distance.py:144-163):
    epos (B, 3N) device tensor -> (wrapped (B, 3N), wrap (B, N, 3)).  HIP kernel
    ``k_enforce_pbc`` through ``ds_enforce_pbc``."""
    import ctypes as C

    import torch

    from . import _lib
    lib = _lib.load()
    if not epos.is_cuda:
        raise RuntimeError('enforce_pbc: walkers must live on the ROCm device (no CPU path)')
    lv = np.ascontiguousarray(np.asarray(latvec.detach().cpu() if isinstance(latvec, torch.Tensor) else latvec,
                                         dtype=np.float64).reshape(3, 3))
    x = epos.contiguous()
    n_elec = x.numel() // 3
    out = torch.empty_like(x)
    wrap = torch.empty(x.shape[:-1] + (x.shape[-1] // 3, 3), dtype=x.dtype, device=x.device)
    dt = {torch.float64: 0, torch.float32: 1}[x.dtype]
    _lib.check(lib.ds_enforce_pbc(lv.ctypes.data_as(C.POINTER(C.c_double)), dt, C.c_void_p(x.data_ptr()), n_elec,
                                  C.c_void_p(out.data_ptr()), C.c_void_p(wrap.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'ds_enforce_pbc')
    return out, wrap
