"""Local energy = kinetic (Laplacian of log psi) + Ewald (reference DeepSolid/hamiltonian.py).

``local_energy_seperate(f, simulation_cell, mode, partition_number)`` keeps the
reference signature (hamiltonian.py:194).  All four Laplacian modes of the
reference (``for`` :45, ``dim_batch`` :73, ``hessian`` :104, ``partition`` :127)
are different schedules of the same number, so they all map to the one
forward-Laplacian HIP kernel chain (``ds_local_energy``); the argument is
validated and otherwise only kept for drop-in compatibility.
"""
import torch

from .network import NetworkApply

MODES = ('for', 'hessian', 'dim_batch', 'partition')


def _system_of(f):
    if not isinstance(f, NetworkApply):
        raise TypeError('f must be the .apply of deepsolid_amd.network.make_solid_fermi_net: the HIP path '
                        'differentiates its own kernels, it cannot trace an arbitrary Python function')
    if f.method_name != 'eval_logdet':
        raise ValueError("the local energy needs the complex log psi: build f with method_name='eval_logdet'")
    return f.system


def local_ewald_energy(simulation_cell, dtype=torch.float64):
    """x -> ee + ei + ii (hamiltonian.py:163-179).  The reference asserts the ion-ion
    constant against PySCF's ``cell.energy_nuc()``; kept whenever the cell provides it."""
    from .device import DeviceSystem
    system = DeviceSystem.for_ewald(simulation_cell, dtype=dtype)
    ref = simulation_cell.energy_nuc() if hasattr(simulation_cell, 'energy_nuc') else None
    if ref is not None:
        mine = system.tables.ion_ion + system.tables.ii_const
        assert abs(ref - mine) <= 1e-5 + 1e-8 * abs(mine), (ref, mine)

    def _local_ewald_energy(x):
        single = x.dim() == 1
        e = system.ewald(x.reshape(1, -1) if single else x).sum(-1)
        return e[0] if single else e
    return _local_ewald_energy


def local_energy_seperate(f, simulation_cell, mode='for', partition_number=3):
    """-> (params, x) -> (kinetic: complex, ewald: real); x is (3N,) or (B, 3N)."""
    if mode not in MODES:
        raise ValueError('Unrecognized laplacian evaluation mode.')
    system = _system_of(f)
    if simulation_cell is not system.cell:
        raise ValueError('simulation_cell differs from the one the network was built for')
    if mode == 'partition' and (3 * system.n) % int(partition_number):
        raise ValueError('partition_number must divide 3 * N_e (reference README.md:96-103)')

    def _local_energy(params, x):
        single = x.dim() == 1
        ke, ew, _, _ = system.local_energy(params, x.reshape(1, -1) if single else x)
        ke = torch.view_as_complex(ke)
        return (ke[0], ew[0]) if single else (ke, ew)
    return _local_energy


def local_energy(f, simulation_cell):
    """kinetic + ewald as one complex number (hamiltonian.py:182-191)."""
    sep = local_energy_seperate(f, simulation_cell)

    def _local_energy(params, x):
        ke, ew = sep(params, x)
        return ke + ew
    return _local_energy
