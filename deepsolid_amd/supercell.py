"""Supercell helpers without PySCF.

Mirrors the public functions of reference ``DeepSolid/supercell.py``:
``get_supercell_kpts`` (:32), ``get_supercell_copies`` (:51), ``get_supercell``
(:64), ``set_symmetry_lat`` (:98).  ``make_klist`` stands in for the k-list the
reference obtains from a Hartree-Fock run (``hf.py:84-104``): it fills the
supercell k-points in order, which is all a synthetic benchmark needs.
"""
import numpy as np

from .cell import Cell


def _unit_box_range(S_like):
    corners = np.stack([c.ravel() for c in np.meshgrid(*[[0, 1]] * 3, indexing='ij')]).T
    img = corners @ S_like
    return np.stack([img.min(axis=0), img.max(axis=0)]).T


def get_supercell_kpts(supercell):
    """k-points of the supercell folded into the primitive reciprocal unit box
    (reference supercell.py:32-48)."""
    S = np.asarray(supercell.S, dtype=np.float64)
    Sinv_T = np.linalg.inv(S).T
    rng = _unit_box_range(S.T)
    mesh = np.meshgrid(*[np.arange(lo, hi) for lo, hi in rng], indexing='ij')
    cand = np.stack([m.ravel() for m in mesh]).T @ Sinv_T
    inside = np.all((cand >= 0) & (cand < 1 - 1e-12), axis=1)
    recip = np.linalg.inv(supercell.original_cell.lattice_vectors()).T * 2 * np.pi
    return cand[inside] @ recip


def get_supercell_copies(latvec, S):
    """Translations of the primitive cell that tile the supercell
    (reference supercell.py:51-61)."""
    S = np.asarray(S, dtype=np.float64)
    Sinv = np.linalg.inv(S)
    rng = _unit_box_range(S)
    mesh = np.meshgrid(*[np.arange(lo, hi) for lo, hi in rng], indexing='ij')
    cand = np.stack([m.ravel() for m in mesh]).T @ Sinv
    inside = np.all((cand >= 0) & (cand < 1 - 1e-12), axis=1)
    return np.linalg.multi_dot((cand[inside], S, np.asarray(latvec)))


_SYM_MATS = {
    'minimal': np.eye(3),
    'fcc': np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], dtype=np.float64),
    'bcc': np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, -1, 0], [1, 0, -1], [0, 1, -1]],
                    dtype=np.float64),
    'hexagonal': np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, 0]], dtype=np.float64),
}


def set_symmetry_lat(supercell, sym_type='minimal'):
    """Attach the feature lattices AV/BV to both cells (reference supercell.py:98-140)."""
    mat = _SYM_MATS.get(sym_type, np.eye(3))
    prim_bv = mat @ supercell.original_cell.reciprocal_vectors()
    sim_bv = mat @ supercell.reciprocal_vectors()
    supercell.BV = sim_bv
    supercell.AV = np.linalg.pinv(sim_bv).T
    supercell.original_cell.BV = prim_bv
    supercell.original_cell.AV = np.linalg.pinv(prim_bv).T
    return supercell


def get_supercell(cell, S, sym_type='minimal', nelec=None):
    """Tile ``cell`` by the integer matrix ``S`` (reference supercell.py:64-95).

    ``nelec`` overrides the (n_up, n_dn) PySCF would derive from
    ``cell.spin * scale`` -- needed for the synthetic 24-electron bcc-Li case
    of BASELINE.json (SURVEY.md section 8(d), config 3 note)."""
    S = np.asarray(S, dtype=np.float64)
    scale = int(abs(int(np.round(np.linalg.det(S)))))
    superlattice = S @ cell.lattice_vectors()
    Rpts = get_supercell_copies(cell.lattice_vectors(), S)
    atom = []
    for name, xyz in cell._atom:
        atom.extend([(name, xyz + R) for R in Rpts])
    sc = Cell(superlattice, atom, spin=cell.spin * scale, nelec=nelec)
    sc.original_cell = cell
    sc.S = S
    sc.scale = scale
    return set_symmetry_lat(sc, sym_type)


def make_klist(simulation_cell, twist=(0.0, 0.0, 0.0)):
    """Occupied k-point list per spin, shaped like ``hf.SCF.klist``
    (reference hf.py:61-62 for the twist shift, :99-104 for the grouping):
    k-points are filled in order, each repeated by its occupation, so that
    every spin channel gets exactly ``n_s`` rows."""
    kpts = get_supercell_kpts(simulation_cell)
    # hf.py:61: the twist is a fraction of the SIMULATION cell's reciprocal vectors (SCF is built on the simulation cell,
    # process.py:87), so that moving one electron by a supercell vector multiplies psi by exp(2 pi i twist) (test_network.py:86-106)
    kpts = kpts + np.dot(np.linalg.inv(simulation_cell.a), np.mod(np.asarray(twist, dtype=np.float64), 1.0)) * 2 * np.pi
    nk = kpts.shape[0]
    occ = []
    for ns in simulation_cell.nelec:
        base, rem = divmod(int(ns), nk)
        occ.append([base + (1 if i < rem else 0) for i in range(nk)])
    return klist_from_occupations(kpts, occ)


def klist_from_occupations(kpts, n_occ):
    """hf.py:99-104: every k-point repeated by the number of orbitals occupied there, per spin.
    kpts (nk, 3); n_occ [spin][k] integers.  -> [klist_up (n_up, 3), klist_dn (n_dn, 3)]."""
    kpts = np.asarray(kpts, dtype=np.float64).reshape(-1, 3)
    klist = []
    for occ in n_occ:
        rows = [np.tile(k[None, :], (int(o), 1)) for k, o in zip(kpts, occ) if int(o) > 0]
        klist.append(np.concatenate(rows, axis=0) if rows else np.zeros((0, 3)))
    return klist


def klist_from_scf(kpts, mo_occ):
    """`klist` of a finished PySCF k-point SCF exactly as hf.SCF.init_scf builds it (hf.py:84-104), from plain arrays:
    `kpts` = kmf.kpts (nk, 3) and `mo_occ` = kmf.mo_occ, either restricted [k][mo] (occupations 0 / 2: spin up counts
    occupations > 0.9, spin down > 1.1, hf.py:93-95) or unrestricted [spin][k][mo] (> 0.9, hf.py:91).  Ragged per-k lists
    are accepted."""
    occ = [np.asarray(o, dtype=np.float64) for o in mo_occ] if not isinstance(mo_occ, np.ndarray) else mo_occ
    unrestricted = (isinstance(occ, np.ndarray) and occ.ndim == 3) or \
                   (not isinstance(occ, np.ndarray) and len(occ) == 2 and np.asarray(occ[0], dtype=object).ndim >= 1
                    and all(np.ndim(o) == 2 or (np.ndim(o) == 1 and np.asarray(o).dtype == object) for o in occ))
    nk = np.asarray(kpts).reshape(-1, 3).shape[0]
    if unrestricted:
        n_occ = [[int(np.sum(np.asarray(occ[s][k]) > 0.9)) for k in range(nk)] for s in range(2)]
    else:
        n_occ = [[int(np.sum(np.asarray(occ[k]) > thr)) for k in range(nk)] for thr in (0.9, 1.1)]
    return klist_from_occupations(kpts, n_occ)


def load_hf_klist(path):
    """`klist` from a saved HF run: an .npz written as ``np.savez(path, kpts=kmf.kpts, mo_occ=np.asarray(kmf.mo_occ))``
    (restricted: (nk, nmo); unrestricted: (2, nk, nmo)).  PySCF's own chkfile is HDF5, which this image cannot read."""
    with np.load(path, allow_pickle=True) as f:
        return klist_from_scf(f['kpts'], f['mo_occ'])
