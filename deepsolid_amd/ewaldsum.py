"""Ewald sums: host-side table setup + device energy (reference DeepSolid/ewaldsum.py).

``EwaldSum(cell)`` mirrors the reference class: the constructor builds, once and
on the host, everything that depends on the cell only (alpha, the half-space G
mesh and its weights, the ion structure factor, the constants), and
``energy(configs)`` evaluates (ee, ei, ii) for a batch of walkers with the HIP
kernel ``k_ewald`` through the C ABI (``ds_ewald``).
"""
import numpy as np

from . import distance

EWALD_GMAX = 200          # reference ewaldsum.py:34
WEIGHT_CUTOFF = 1e-12     # reference ewaldsum.py:199


def _half_space_mesh(recvec, cellvolume, alpha, gmax, latvec=None):
    """G vectors with x>0, or x=0,y>0, or x=y=0,z>0 and weight above the cutoff
    (reference ewaldsum.py:67-89,194-200).  Order = the reference's: the three
    groups concatenated, each in C order of its integer mesh.

    The reference scans all (2 gmax + 1)^3 / 2 = 3.2e7 integer triples; the weight falls monotonically with |G|^2, so only
    triples inside the sphere w(|G|^2) = cutoff can pass, and n_j = G . a_j / 2 pi bounds each integer by |G|max |a_j| / 2 pi:
    the scan runs over that box only (same points, same order: a sub-box of a C-ordered mesh keeps the order of what it keeps)."""
    lo, hi = 0.0, 1.0
    wfun = lambda g2: 4 * np.pi * np.exp(-g2 / (4 * alpha ** 2)) / (cellvolume * g2)
    while wfun(hi) > WEIGHT_CUTOFF:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if wfun(mid) > WEIGHT_CUTOFF else (lo, mid)
    gm = np.sqrt(hi) * (1 + 1e-9)
    if latvec is None:
        latvec = np.linalg.inv(recvec).T
    nmax = [int(min(gmax, np.floor(gm * np.linalg.norm(latvec[j]) / (2 * np.pi)) + 1)) for j in range(3)]
    fy, fz = np.arange(-nmax[1], nmax[1] + 1), np.arange(-nmax[2], nmax[2] + 1)
    pts, wts = [], []

    def scan(ix, iy, iz):
        X, Y, Z = np.meshgrid(ix, iy, iz, indexing='ij')
        n = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float64)
        g = (n @ recvec) * (2 * np.pi)
        g2 = np.einsum('ik,ik->i', g, g)
        w = 4 * np.pi * np.exp(-g2 / (4 * alpha ** 2)) / (cellvolume * g2)
        keep = w > WEIGHT_CUTOFF
        if keep.any():
            pts.append(g[keep])
            wts.append(w[keep])

    for x0 in range(1, nmax[0] + 1, 8):
        scan(np.arange(x0, min(x0 + 8, nmax[0] + 1)), fy, fz)
    scan(np.array([0]), np.arange(1, nmax[1] + 1), fz)
    scan(np.array([0]), np.array([0]), np.arange(1, nmax[2] + 1))
    return np.concatenate(pts, axis=0), np.concatenate(wts, axis=0)


class EwaldTables:
    """Cell-only quantities of the Ewald sum (reference ewaldsum.py:34-136)."""

    def __init__(self, cell, ewald_gmax=EWALD_GMAX):
        self.nelec = tuple(int(n) for n in cell.nelec)
        self.atom_coords = np.asarray(cell.atom_coords(), dtype=np.float64)
        self.atom_charges = np.asarray(cell.atom_charges(), dtype=np.float64)
        self.latvec = np.asarray(cell.lattice_vectors(), dtype=np.float64)
        self.dist_mode = distance.minimal_image_mode(self.latvec)
        volume = np.linalg.det(self.latvec)
        recvec = np.linalg.inv(self.latvec).T
        self.alpha = 5.0 / np.amin(1.0 / np.linalg.norm(recvec, axis=1))          # :63-64
        self.gpoints, self.gweight = _half_space_mesh(recvec, volume, self.alpha, ewald_gmax, self.latvec)
        q = self.atom_charges
        self.i_sum = q.sum()
        ii_sum2 = (q ** 2).sum()
        ii_sum = (self.i_sum ** 2 - ii_sum2) / 2
        self.ijconst = -np.pi / (volume * self.alpha ** 2)                         # :97
        self.squareconst = -self.alpha / np.sqrt(np.pi) + self.ijconst / 2         # :98
        self.ii_const = ii_sum * self.ijconst + ii_sum2 * self.squareconst          # :100
        self.ion_exp = np.exp(1j * (self.gpoints @ self.atom_coords.T)) @ q         # :131-132
        self.ion_ion = self._ion_ion_real() + float(self.gweight @ np.abs(self.ion_exp) ** 2)

    def _ion_ion_real(self):
        """Real-space ion-ion sum over pairs a<b and the 27 neighbour cells (:122-129)."""
        from scipy.special import erfc
        n = len(self.atom_charges)
        if n == 1:
            return 0.0
        shifts = np.stack(np.meshgrid(*[np.arange(-1, 2)] * 3, indexing='ij'), -1).reshape(-1, 3) @ self.latvec
        tot = 0.0
        for a in range(n):
            for b in range(a + 1, n):
                d = distance.minimal_image_host(self.latvec, self.dist_mode,
                                                self.atom_coords[a] - self.atom_coords[b])
                r = np.linalg.norm(d[None, :] + shifts, axis=1)
                tot += self.atom_charges[a] * self.atom_charges[b] * np.sum(erfc(self.alpha * r) / r)
        return float(tot)

    def ee_const(self, ne):
        return ne * (ne - 1) / 2 * self.ijconst + ne * self.squareconst             # :109-110

    def ei_const(self, ne):
        return -ne * self.i_sum * self.ijconst                                      # :112-113


class EwaldSum(EwaldTables):
    """Reference-shaped class: ``energy(configs)`` -> (ee, ei, ii) on the GPU.

    ``configs`` is a torch tensor on the device, (3N,) or (B, 3N)."""

    def __init__(self, cell, ewald_gmax=EWALD_GMAX, nlatvec=1, dtype=None):
        if nlatvec != 1:
            raise ValueError('the device kernel sums the 27 neighbour cells (nlatvec=1), like every reference caller')
        super().__init__(cell, ewald_gmax)
        self._cell = cell
        self._dtype = dtype
        self._system = None

    def _sys(self, like):
        from .device import DeviceSystem
        if self._system is None:
            self._system = DeviceSystem.for_ewald(self._cell, tables=self, dtype=self._dtype or like.dtype)
        return self._system

    def energy(self, configs):
        single = configs.dim() == 1
        out = self._sys(configs).ewald(configs.reshape(1, -1) if single else configs)
        if single:
            return out[0, 0], out[0, 1], out[0, 2]
        return out[:, 0], out[:, 1], out[:, 2]
