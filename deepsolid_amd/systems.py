"""The five BASELINE.json systems as PySCF-free cells.

Geometry follows the reference config builders (cited per function); lattice
constants the reference leaves to the command line are the SURVEY's choices.
Everything is synthetic-benchmark input: uniform-in-cell walkers and
random-init parameters are generated in ``synthetic_walkers`` /
``network.init`` with numpy ``default_rng`` seeds.
"""
import numpy as np

from .cell import Cell, angstrom2bohr
from .supercell import get_supercell, make_klist

DETNET_DEFAULTS = dict(                       # reference base_config.py:129-139
    envelope_type='isotropic', bias_orbitals=False, use_last_layer=False, full_det=False,
    hidden_dims=((256, 32), (256, 32), (256, 32)), determinants=8, after_determinants=1,
    distance_type='nu')


def _smat(S):
    """int -> S*I, 3-vector -> diag, 3x3 -> as is."""
    S = np.asarray(S)
    if S.ndim == 0:
        return np.eye(3) * int(S)
    return np.diag(S) if S.ndim == 1 else S


def two_hydrogen_cell(L=2.0, S=(1, 1, 1)):
    """reference config/two_hydrogen_cell.py:15-44 with 'H,1,1,1,2.0,0,ccpvdz'."""
    prim = Cell(np.diag([2 * L, 100.0, 100.0]), [('H', [L, 0, 0]), ('H', [0, 0, 0])], spin=0)
    return get_supercell(prim, _smat(S))


def lih_rocksalt(L_ang=4.0, S=1, X='Li', Y='H'):
    """reference config/rock_salt.py:15-36."""
    L = angstrom2bohr(L_ang)
    prim = Cell((np.ones((3, 3)) - np.eye(3)) * L / 2,
                [(X, [0.0, 0.0, 0.0]), (Y, [0.5 * L, 0.5 * L, 0.5 * L])])
    return get_supercell(prim, _smat(S))


def bcc_li(S=2, a0_ang=3.4268178940, nelec=None):
    """One-atom primitive bcc Li cell (lattice constant from the reference's
    config/poscar/bcc_li.vasp:3), tiled S x S x S.  BASELINE.json's
    "24 electrons, 2x2x2" fixes nelec = (12, 12) (SURVEY.md section 8(d) note)."""
    a0 = angstrom2bohr(a0_ang)
    prim = Cell(0.5 * a0 * np.array([[-1.0, 1, 1], [1, -1, 1], [1, 1, -1]]),
                [('Li', [0.0, 0.0, 0.0])], spin=1)
    if nelec is None:
        ne = 3 * abs(int(round(np.linalg.det(_smat(S)))))            # three electrons per atom, det S atoms
        nelec = (ne - ne // 2, ne // 2)
    return get_supercell(prim, _smat(S), nelec=nelec)


def graphene(L_ang=2.46, S=2, z=20.0, X='C', Y='C'):
    """reference config/graphene.py:15-40 (the "graphite 2x2x1" of BASELINE.json)."""
    L = angstrom2bohr(L_ang)
    a = np.array([[L * np.cos(np.pi / 6), -L * 0.5, 0],
                  [L * np.cos(np.pi / 6), L * 0.5, 0],
                  [0, 0, z]])
    prim = Cell(a, [(X, [3 ** (-0.5) * L, 0.0, 0.0]), (Y, [2 * 3 ** (-0.5) * L, 0.0, 0.0])])
    return get_supercell(prim, np.diag([int(S), int(S), 1]) if np.ndim(S) == 0 else _smat(S))


def diamond(L_ang=3.567, S=2, X='C', Y='C'):
    """reference config/diamond.py:15-36."""
    L = angstrom2bohr(L_ang)
    prim = Cell((np.ones((3, 3)) - np.eye(3)) * L / 2,
                [(X, [0.0, 0.0, 0.0]), (Y, [0.25 * L, 0.25 * L, 0.25 * L])])
    return get_supercell(prim, _smat(S))


SYSTEMS = {
    'h2': two_hydrogen_cell,          # BASELINE config 1:  2 e-
    'lih': lih_rocksalt,              # BASELINE config 2:  4 e-
    'bcc_li': bcc_li,                 # BASELINE config 3: 24 e-  (headline)
    'graphene': graphene,             # BASELINE config 4: 48 e-
    'diamond': diamond,               # BASELINE config 5: 96 e-
}


def build(name, twist=(0.0, 0.0, 0.0), **kw):
    """Returns (simulation_cell, klist) for one of the benchmark systems."""
    cell = SYSTEMS[name](**kw)
    return cell, make_klist(cell, twist)


def synthetic_walkers(cell, batch, seed=1234):
    """Uniform-in-cell walkers, x = U[0,1)^(B,N,3) @ a  (SURVEY.md section 8(d))."""
    rng = np.random.default_rng(seed)
    n = cell.nelec[0] + cell.nelec[1]
    return (rng.uniform(size=(batch, n, 3)) @ cell.a).reshape(batch, 3 * n)
