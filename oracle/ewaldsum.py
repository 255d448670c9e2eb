"""Oracle (test infrastructure): Ewald sums for a periodic cell.

Restates /root/reference/DeepSolid/ewaldsum.py (PyQMC-derived) in numpy/torch
float64.  Setup is numpy, the per-walker energy is torch so it can be vmapped
or looped by the caller.
"""
import math

import numpy as np
import torch

from .distance import MinimalImageDistance
from .network import _t


# ewaldsum.py:194-200
def select_big(gpts, cellvolume, recvec, alpha):
    gpoints = np.einsum('j...,jk->...k', gpts, recvec) * 2 * np.pi
    gsquared = np.einsum('...k,...k->...', gpoints, gpoints)
    gweight = 4 * np.pi * np.exp(-gsquared / (4 * alpha ** 2))
    gweight /= cellvolume * gsquared
    big = gweight > 1e-12
    return gpoints[big], gweight[big]


class EwaldSum:
    def __init__(self, cell, ewald_gmax=200, nlatvec=1):
        # ewaldsum.py:34-46
        self.nelec = tuple(int(n) for n in cell.nelec)
        self.atom_coords_np = np.asarray(cell.atom_coords(), dtype=np.float64)
        self.atom_charges_np = np.asarray(cell.atom_charges(), dtype=np.float64)
        self.latvec_np = np.asarray(cell.lattice_vectors(), dtype=np.float64)
        self.dist = MinimalImageDistance(self.latvec_np)
        # ewaldsum.py:48-56
        XYZ = np.meshgrid(*[np.arange(-nlatvec, nlatvec + 1)] * 3, indexing='ij')
        xyz = np.stack(XYZ, axis=-1).reshape((-1, 3))
        self.lattice_displacements_np = np.dot(xyz, self.latvec_np)
        self._setup_reciprocal(ewald_gmax)
        self.atom_coords = _t(self.atom_coords_np)
        self.atom_charges = _t(self.atom_charges_np)
        self.lattice_displacements = _t(self.lattice_displacements_np)
        self.gpoints = _t(self.gpoints_np)
        self.gweight = _t(self.gweight_np)
        self.ion_exp_re = _t(self.ion_exp_np.real)
        self.ion_exp_im = _t(self.ion_exp_np.imag)

    # ewaldsum.py:58-90
    def _setup_reciprocal(self, gmax):
        cellvolume = np.linalg.det(self.latvec_np)
        recvec = np.linalg.inv(self.latvec_np).T
        smallestheight = np.amin(1 / np.linalg.norm(recvec, axis=1))
        self.alpha = 5.0 / smallestheight
        zero = np.asarray([0])
        gX = np.meshgrid(np.arange(1, gmax + 1), *[np.arange(-gmax, gmax + 1)] * 2, indexing='ij')
        gY = np.meshgrid(zero, np.arange(1, gmax + 1), np.arange(-gmax, gmax + 1), indexing='ij')
        gZ = np.meshgrid(zero, zero, np.arange(1, gmax + 1), indexing='ij')
        sel = [select_big(np.stack(g), cellvolume, recvec, self.alpha) for g in (gX, gY, gZ)]
        self.gpoints_np = np.concatenate([s[0] for s in sel], axis=0)
        self.gweight_np = np.concatenate([s[1] for s in sel], axis=0)
        self._set_constants(cellvolume)

    # ewaldsum.py:92-118
    def _set_constants(self, cellvolume):
        q = self.atom_charges_np
        self.i_sum = np.sum(q)
        ii_sum2 = np.sum(q ** 2)
        ii_sum = (self.i_sum ** 2 - ii_sum2) / 2
        self.ijconst = -np.pi / (cellvolume * self.alpha ** 2)
        self.squareconst = -self.alpha / np.sqrt(np.pi) + self.ijconst / 2
        self.ii_const = ii_sum * self.ijconst + ii_sum2 * self.squareconst
        self.ion_ion = self._ewald_ion()

    def ee_const(self, ne):
        return ne * (ne - 1) / 2 * self.ijconst + ne * self.squareconst

    def ei_const(self, ne):
        return -ne * self.i_sum * self.ijconst

    # ewaldsum.py:120-136
    def _ewald_ion(self):
        q = _t(self.atom_charges_np)
        if len(self.atom_charges_np) == 1:
            real = 0.0
        else:
            coords = _t(self.atom_coords_np).reshape(-1)
            d = self.dist.dist_matrix(coords)
            rvec = d[None] + _t(self.lattice_displacements_np)[:, None, None, :]
            r = torch.linalg.norm(rvec, dim=-1)
            cij = q[:, None] * q[None, :]
            real = float(torch.sum(torch.triu(cij * torch.erfc(self.alpha * r) / r, diagonal=1)))
        GdotR = self.gpoints_np @ self.atom_coords_np.T
        self.ion_exp_np = np.exp(1j * GdotR) @ self.atom_charges_np
        rec = float(np.dot(self.gweight_np, np.abs(self.ion_exp_np) ** 2))
        return real + rec

    # ewaldsum.py:138-142
    def _real_cij(self, dists):
        r = dists[:, :, None, :] + self.lattice_displacements
        r = torch.linalg.norm(r, dim=-1)
        return torch.sum(torch.erfc(self.alpha * r) / r, dim=-1)

    # ewaldsum.py:144-172
    def ewald_electron(self, configs):
        nelec = sum(self.nelec)
        ei_d = self.dist.dist_i(self.atom_coords.reshape(-1), configs)
        ei_cij = self._real_cij(ei_d)
        ei_real = torch.sum(-self.atom_charges[None, :] * ei_cij)
        ee_real = torch.zeros((), dtype=configs.dtype)
        if nelec > 1:
            ee_d = self.dist.dist_matrix(configs)
            rvec = ee_d[None] + self.lattice_displacements[:, None, None, :]
            r = torch.linalg.norm(rvec, dim=-1)
            # triu(k=1) on the (27,N,N) tensor: also drops every i==j image term
            ee_real = torch.sum(torch.triu(torch.erfc(self.alpha * r) / r, diagonal=1))
        ee_rec, ei_rec = self.reciprocal_space_electron(configs)
        return ee_real + ee_rec, ei_real + ei_rec

    # ewaldsum.py:174-183
    def reciprocal_space_electron(self, configs):
        g = configs.reshape(sum(self.nelec), -1) @ self.gpoints.T
        s_sin = torch.sin(g).sum(0)
        s_cos = torch.cos(g).sum(0)
        ee = torch.dot(s_sin ** 2 + s_cos ** 2, self.gweight)
        cc = -self.ion_exp_re * s_cos - self.ion_exp_im * s_sin
        ei = 2 * torch.dot(cc, self.gweight)
        return ee, ei

    # ewaldsum.py:185-191
    def energy(self, configs):
        nelec = sum(self.nelec)
        ee, ei = self.ewald_electron(configs)
        ee = ee + self.ee_const(nelec)
        ei = ei + self.ei_const(nelec)
        ii = torch.as_tensor(self.ion_ion + self.ii_const, dtype=configs.dtype)
        return ee, ei, ii
