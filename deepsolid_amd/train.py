"""Energy evaluation of a batch of walkers (reference DeepSolid/train.py:37-89).

Only the primal of ``total_energy`` is implemented: it is what the north-star
metric times (local-energy evaluations per second).  The custom JVP that turns
it into an energy gradient (train.py:91-142) belongs to the optimiser side and
is out of scope.
"""
from collections import namedtuple

import torch

from . import constants, hamiltonian

AuxiliaryLossData = namedtuple('AuxiliaryLossData', ['variance', 'local_energy', 'imaginary', 'kinetic', 'ewald'])


def make_loss(network, batch_network, simulation_cell, clip_local_energy=5.0, clip_type='real', mode='for',
              partition_number=3):
    """Same signature as train.py:37-43; ``batch_network`` is accepted and unused (the
    HIP network is batched natively)."""
    del batch_network, clip_local_energy, clip_type
    el_fun = hamiltonian.local_energy_seperate(network, simulation_cell, mode=mode,
                                               partition_number=partition_number)

    def total_energy(params, data):
        ke, ew = el_fun(params, data)                                  # train.py:74
        e_l = ke + ew                                                  # :75
        mean_e_l = e_l.mean()                                          # :76
        var_local = (e_l.abs() ** 2).mean() - mean_e_l.real.abs() ** 2  # :79 (per-device variance, then pmean)
        re, im, var = constants.pmean_packed(mean_e_l.real, mean_e_l.imag, var_local)   # :78-80 in one message
        return re, AuxiliaryLossData(variance=var, local_energy=e_l, imaginary=im, kinetic=ke, ewald=ew)
    return total_energy
