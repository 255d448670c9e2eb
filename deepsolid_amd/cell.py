"""PySCF-free periodic cell: the attribute bag the VMC hot path reads.

The reference hands a ``pyscf.pbc.gto.Cell`` around but the hot path only reads
a handful of attributes from it: ``.a``, ``.nelec``,
``.atom_coords()``, ``.atom_charges()``, ``.lattice_vectors()``,
``.reciprocal_vectors()``, ``.original_cell``, ``.S``, ``.scale``, ``.AV``,
``.BV`` and (assertion only) ``.energy_nuc()``.  This class provides exactly
that surface, in Bohr, with no SCF machinery.  A real PySCF cell can be passed
to every factory in this package instead.
"""
import numpy as np

ANGSTROM_BOHR = 0.52917721067      # reference utils/units.py:25

# symbol -> atomic number, first four rows are enough for the shipped configs
_Z = {s: i + 1 for i, s in enumerate(
    "H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn "
    "Ga Ge As Se Br Kr".split())}


def angstrom2bohr(x):
    """reference utils/units.py:40."""
    return x * (1.0 / ANGSTROM_BOHR)


class Cell:
    """Minimal stand-in for ``pyscf.pbc.gto.Cell`` (already 'built', unit Bohr)."""

    def __init__(self, a, atom, spin=0, nelec=None, charge=0):
        self.a = np.asarray(a, dtype=np.float64).reshape(3, 3)
        self._atom = [(str(s), np.asarray(xyz, dtype=np.float64).reshape(3)) for s, xyz in atom]
        self.charge = int(charge)
        self.spin = int(spin)
        self.unit = 'Bohr'
        ne = int(sum(_Z[s] for s, _ in self._atom)) - self.charge
        if nelec is None:
            if (ne + self.spin) % 2:
                raise ValueError(f'electron number {ne} and spin {self.spin} are inconsistent')
            nelec = ((ne + self.spin) // 2, (ne - self.spin) // 2)
        self.nelec = (int(nelec[0]), int(nelec[1]))
        self.original_cell = self
        self.S = np.eye(3)
        self.scale = 1
        self.AV = None
        self.BV = None

    # --- the PySCF accessors the hot path uses -------------------------------
    @property
    def natm(self):
        return len(self._atom)

    @property
    def nelectron(self):
        return self.nelec[0] + self.nelec[1]

    def lattice_vectors(self):
        return self.a

    def reciprocal_vectors(self):
        return 2 * np.pi * np.linalg.inv(self.a).T

    def atom_coords(self):
        return np.stack([xyz for _, xyz in self._atom]) if self._atom else np.zeros((0, 3))

    def atom_charges(self):
        return np.asarray([_Z[s] for s, _ in self._atom], dtype=np.int64)

    def atom_symbol(self, i):
        return self._atom[i][0]

    def energy_nuc(self):
        """PySCF's nuclear-repulsion Ewald energy is not available without
        PySCF; ``None`` makes ``hamiltonian.local_ewald_energy`` skip the
        cross-check the reference performs at hamiltonian.py:170."""
        return None
