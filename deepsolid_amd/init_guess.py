"""Initial walkers (reference DeepSolid/init_guess.py:27-80): Gaussian blobs of electrons around the
atoms, spin-assigned atom by atom, wrapped into the simulation cell."""
import numpy as np

from .cell import _Z

# ground-state (n_alpha, n_beta) of the neutral atoms of the first three rows (utils/elements.py)
_UNPAIRED = {1: 1, 2: 0, 3: 1, 4: 0, 5: 1, 6: 2, 7: 3, 8: 2, 9: 1, 10: 0, 11: 1, 12: 0, 13: 1, 14: 2, 15: 3, 16: 2, 17: 1, 18: 0}


def _atomic_spin(z):
    un = _UNPAIRED.get(z, z % 2)
    return ((z + un) // 2, (z - un) // 2)


def init_electrons(key, cell, latvec, electrons, batch_size, init_width=0.5):
    """-> numpy (batch, 3N) walkers.  `cell` is any object with atom_coords() / atom_charges()
    (the reference passes its internal Atom list); `key` an int seed or numpy Generator."""
    rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
    coords = np.asarray(cell.atom_coords(), dtype=np.float64)
    charges = [int(z) for z in cell.atom_charges()]
    electrons = tuple(int(e) for e in electrons)
    if sum(charges) != sum(electrons):
        if len(charges) == 1:
            configs = [electrons]
        else:
            raise NotImplementedError('No initialization policy yet exists for charged molecules.')   # init_guess.py:49
    else:
        configs = [_atomic_spin(z) for z in charges]
        assert sum(sum(c) for c in configs) == sum(electrons)
        while tuple(sum(c) for c in zip(*configs)) != electrons:        # init_guess.py:61-65: flip spins until the totals match
            i = int(rng.integers(len(configs)))
            na, nb = configs[i]
            if tuple(sum(c) for c in zip(*configs))[0] > electrons[0] and na > 0:
                configs[i] = (na - 1, nb + 1)
            elif tuple(sum(c) for c in zip(*configs))[0] < electrons[0] and nb > 0:
                configs[i] = (na + 1, nb - 1)
    pos = []
    for s in range(2):
        for j, xyz in enumerate(coords):
            pos.append(np.tile(xyz, configs[j][s]))
    pos = np.concatenate(pos)
    guess = pos + init_width * rng.standard_normal((batch_size, pos.size))
    lat = np.asarray(latvec, dtype=np.float64)
    frac = guess.reshape(batch_size, -1, 3) @ np.linalg.inv(lat)
    return ((frac - np.floor(frac)) @ lat).reshape(batch_size, -1)     # distance.enforce_pbc, :79


def read_poscar(fname='POSCAR'):
    """VASP POSCAR -> deepsolid_amd.cell.Cell in Bohr (reference utils/poscar_to_cell.py:31-91)."""
    from .cell import ANGSTROM_BOHR, Cell
    with open(fname) as f:
        lines = f.readlines()
    factor = float(lines[1].split()[0])
    a = np.array([[float(v) for v in lines[i].split()[:3]] for i in range(2, 5)]) * factor / ANGSTROM_BOHR
    names = lines[5].split()
    if all(n.isdigit() for n in names):
        nums = [int(n) for n in names]
        names = ['X'] * len(nums)
        ln = 6
    else:
        nums = [int(n) for n in lines[6].split()]
        ln = 7
    cart = lines[ln].split()[0][0] in 'CKck'
    ln += 1
    atoms = []
    for name, num in zip(names, nums):
        for _ in range(num):
            c = np.array([float(v) for v in lines[ln].split()[:3]])
            c = c * factor / ANGSTROM_BOHR if cart else c @ a
            atoms.append((name, c))
            ln += 1
    spin = sum(_Z[n] for n, _ in atoms) % 2
    return Cell(a, atoms, spin=spin)
