"""Fast-diagonalisation preconditioner data for the pressure solve (host-side setup, NumPy).

The pressure matrix of the reference, ``P x = sum_f (alpha_P/A_P + alpha_N/A_N)/2 (x_N - x_P)``
(``PISO_multiblock_cuda_kernel.cu:4842-4889``), differs from the constant-coefficient operator
``L = P|_{A = 1}`` only through ``1/A``, and ``A dt = 1 + O(CFL)``.  ``L`` is therefore spectrally
equivalent to ``P`` (condition number of ``L^-1 P`` ~ max(A)/min(A), independent of the grid), and on
a rectilinear grid it is *separable*:

    (L x)_{ijk} = hy_j hz_k (Tx x)_i + hx_i hz_k (Ty x)_j + hx_i hy_j (Tz x)_k ,

with 1-D second-difference matrices ``T_a`` (off-diagonals ``(1/h_i + 1/h_{i+-1})/2``, periodic wrap or
no entry at a FIXED end).  With the generalised eigen-decompositions ``T_a Q_a = H_a Q_a Lambda_a``
(``Q_a^T H_a Q_a = I``) of the transform axes (x, and z in 3-D) the system decouples into one
symmetric tridiagonal system along y per mode:

    hy_j (lambda^x_a + lambda^z_c) u_j + (Ty u)_j = (Qx^T (x) Qz^T b)_j ,      x = (Qx (x) Qz) u .

This module builds ``Q_a``, ``Lambda_a`` and the per-mode LU factors of those tridiagonal systems
once per grid; the device applies them with two dense basis changes (fp32 MFMA GEMMs) and one
Thomas sweep per CG iteration (``csrc/fg_fdprecond.hip``).  The reference has no preconditioner on
this path (plain CG, ``cg_solver_kernel.cu:129-471``; its ILU0 branch is unreachable,
``PISO_multiblock_cuda_kernel.cu:7061-7070``) -- the converged answer is the same, the iteration
count drops from O(100) to a handful.

Requires the y axis to be FIXED (the tridiagonal axis); otherwise the caller falls back to plain CG.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np


def axis_operator(h: np.ndarray, fixed: bool) -> np.ndarray:
    """Dense 1-D operator ``T`` of an axis with cell widths ``h`` (float64)."""
    n = len(h)
    rh = 1.0 / np.asarray(h, dtype=np.float64)
    T = np.zeros((n, n))
    for i in range(n):
        for s in (-1, 1):
            j = i + s
            if j < 0 or j >= n:
                if fixed:
                    continue
                j %= n
            w = 0.5 * (rh[i] + rh[j])
            T[i, j] += w
            T[i, i] -= w
    return T


def cosine_basis(n: int, h: float):
    """Eigenpairs of a FIXED axis with uniform width ``h``, in DCT order: ``Q[i, k] = s_k cos(pi (i + 1/2) k / n) /
    sqrt(h)`` (orthonormal DCT-II basis; ``Q^T H Q = I``) and ``lam_k = -(2 - 2 cos(pi k / n)) / h^2``.  Same subspace
    decomposition as :func:`generalized_eig` gives (tests/test_abi_and_host.py), but in an order and sign the device
    can apply as a fast cosine transform (``csrc/fg_fdfft.hip``)."""
    i = np.arange(n)[:, None]
    k = np.arange(n)[None, :]
    s = np.full(n, np.sqrt(2.0 / n))
    s[0] = np.sqrt(1.0 / n)
    Q = s[None, :] * np.cos(np.pi * (i + 0.5) * k / n) / np.sqrt(h)
    lam = -(2.0 - 2.0 * np.cos(np.pi * np.arange(n) / n)) / (h * h)
    return Q, lam


def fourier_basis(n: int, h: float):
    """Eigenpairs of a PERIODIC axis with uniform width ``h`` in the order a real FFT produces them: mode 0 the constant,
    modes ``k`` and ``n - k`` (``0 < k < n/2``) the cosine and the sine of wavenumber ``k``, mode ``n/2`` the alternating vector;
    ``Q^T H Q = I`` and ``lam_m = -(2 - 2 cos(2 pi k / n)) / h^2`` with ``k = min(m, n - m)``.  The device applies this basis as
    one FFT per row (``csrc/fg_fdfft.hip``, ``PERIODIC`` form) instead of the dense n x n GEMM."""
    i = np.arange(n)[:, None]
    m = np.arange(n)[None, :]
    k = np.minimum(m, n - m)
    ang = 2.0 * np.pi * k * i / n
    Q = np.where(m <= n // 2, np.cos(ang), np.sin(ang)) * np.sqrt(2.0 / n)
    Q[:, 0] = np.sqrt(1.0 / n)
    Q[:, n // 2] = np.sqrt(1.0 / n) * (1.0 - 2.0 * (np.arange(n) % 2))
    lam = -(2.0 - 2.0 * np.cos(2.0 * np.pi * k[0] / n)) / (h * h)
    return Q / np.sqrt(h), lam


def is_uniform(h: np.ndarray, rtol: float = 1e-6) -> bool:
    h = np.asarray(h, dtype=np.float64)
    return bool(np.abs(h - h[0]).max() <= rtol * abs(h[0]))


def generalized_eig(T: np.ndarray, h: np.ndarray):
    """``T Q = H Q Lambda`` with ``Q^T H Q = I`` via the symmetric standard problem
    ``H^-1/2 T H^-1/2``; returns ``Q [n,n]`` (columns = modes) and ``lam [n]`` (<= 0)."""
    s = 1.0 / np.sqrt(np.asarray(h, dtype=np.float64))
    S = T * s[:, None] * s[None, :]
    lam, V = np.linalg.eigh(0.5 * (S + S.T))
    return V * s[:, None], lam


class FDPreconditioner:
    """Host-side factors; arrays are float32, laid out like cell fields ``[(nz,) ny, nx]``."""

    def __init__(self, widths: Sequence[np.ndarray], fixed_faces: Sequence[int]):
        d = len(widths)
        self.dims = d
        fixed_axis = [(2 * a) in fixed_faces for a in range(d)]
        if not fixed_axis[1]:
            raise ValueError("fast-diagonalisation preconditioner needs FIXED y faces (tridiagonal axis)")
        h = [np.asarray(w, dtype=np.float64) for w in widths]
        nx, ny = len(h[0]), len(h[1])
        nz = len(h[2]) if d == 3 else 1
        # uniform FIXED x axis of a power-of-two length: cosine basis in DCT order, the device applies it as an FFT
        self.x_cosine_width: Optional[float] = None
        self.x_fourier_width: Optional[float] = None
        if fixed_axis[0] and is_uniform(h[0]) and nx in (64, 128, 256, 512):
            self.x_cosine_width = float(np.float32(widths[0][0]))
            Qx, lx = cosine_basis(nx, float(h[0][0]))
        elif (not fixed_axis[0]) and is_uniform(h[0], rtol=1e-3) and nx in (64, 128, 256, 512):
            # periodic uniform x (RBC, TCF): real Fourier basis in FFT order (uniform to 1e-3: vertex coordinates that went through
            # fp32 differ by 1e-5 of a width; the mean width defines the basis, and a preconditioner does not care)
            self.x_fourier_width = float(np.float32(h[0].mean()))
            Qx, lx = fourier_basis(nx, float(h[0].mean()))
        else:
            Qx, lx = generalized_eig(axis_operator(h[0], fixed_axis[0]), h[0])
        if d == 3:
            Qz, lz = generalized_eig(axis_operator(h[2], fixed_axis[2]), h[2])
        else:
            Qz, lz = np.ones((1, 1)), np.zeros(1)
        lam = lz[:, None] + lx[None, :]  # [nz, nx] mode eigenvalue sums
        # for the Helmholtz preconditioner of the advection-diffusion solves (csrc/fg_fdprecond.hip fg_fd_helmholtz_apply): the
        # eigenvalue sums themselves, and whether the transform axes are periodic and uniform (then velocity and pressure share
        # the 1-D operators of those axes -- a FIXED axis has different boundary rows for a Dirichlet variable)
        self.lam = np.ascontiguousarray(lam, dtype=np.float32)
        # (uniform to 1e-3: vertex coordinates that went through fp32 differ by 1e-5 of a width, and a preconditioner does not
        #  care -- the eigenvectors above are those of the widths as they are)
        self.transform_axes_periodic_uniform = bool((not fixed_axis[0]) and is_uniform(h[0], rtol=1e-3) and
                                                    (d == 2 or ((not fixed_axis[2]) and is_uniform(h[2], rtol=1e-3))))
        Ty = axis_operator(h[1], True)
        # The mode that is constant along every transform axis has lambda = 0 and meets the singular Neumann
        # operator Ty (constant null space of the all-Neumann/periodic pressure system).  Its last pivot
        # would vanish; shifting that single diagonal entry keeps M symmetric negative definite and changes
        # M^-1 only by a rank-one term along the null space, which CG never sees (r is mean-free).
        zero_mode = np.abs(lam) <= 1e-9 * max(np.abs(lam).max(), 1e-300)
        lam = np.where(zero_mode, 0.0, lam)
        lower = np.concatenate([[0.0], np.diag(Ty, -1)])  # l_j = Ty(j, j-1)
        upper = np.concatenate([np.diag(Ty, 1), [0.0]])
        diag = np.diag(Ty)
        inv = np.empty((nz, ny, nx))
        cp = np.empty((nz, ny, nx))
        dj = diag[0] + h[1][0] * lam
        inv[:, 0, :] = 1.0 / dj
        cp[:, 0, :] = upper[0] / dj
        for j in range(1, ny):
            dj = diag[j] + h[1][j] * lam - lower[j] * cp[:, j - 1, :]
            if j == ny - 1:
                dj = np.where(zero_mode, dj + diag[j], dj)  # diag < 0: pivot ~0 -> ~diag
            inv[:, j, :] = 1.0 / dj
            cp[:, j, :] = upper[j] / dj
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.Qx, self.QxT = f32(Qx), f32(Qx.T)
        self.Qz, self.QzT = f32(Qz), f32(Qz.T)
        self.lower = f32(lower)
        self.inv = f32(inv if d == 3 else inv[0])
        self.cp = f32(cp if d == 3 else cp[0])
        self.shape = (nz, ny, nx)

    # ---- NumPy application (tests / documentation of what the kernels do) ---------------------
    def apply(self, r: np.ndarray) -> np.ndarray:
        """``z = M^-1 r`` for one env; ``r`` shaped ``[(nz,) ny, nx]``."""
        nz, ny, nx = self.shape
        x = np.asarray(r, dtype=np.float64).reshape(nz, ny, nx)
        x = x @ self.Qx.astype(np.float64)  # forward x transform with Q^T: sum_i r_i Q[i, a]
        if self.dims == 3:
            x = np.einsum("kc,kjm->cjm", self.Qz.astype(np.float64), x)
        inv, cp = self.inv.reshape(nz, ny, nx).astype(np.float64), self.cp.reshape(nz, ny, nx).astype(np.float64)
        lo = self.lower.astype(np.float64)
        y = np.empty_like(x)
        y[:, 0] = x[:, 0] * inv[:, 0]
        for j in range(1, ny):
            y[:, j] = (x[:, j] - lo[j] * y[:, j - 1]) * inv[:, j]
        for j in range(ny - 2, -1, -1):
            y[:, j] -= cp[:, j] * y[:, j + 1]
        if self.dims == 3:
            y = np.einsum("kc,cjm->kjm", self.Qz.astype(np.float64), y)
        y = y @ self.QxT.astype(np.float64)
        return y.reshape(r.shape)
