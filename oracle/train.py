"""Oracle (test infrastructure): primal of train.make_loss.total_energy.

Restates /root/reference/DeepSolid/train.py:67-89 (batch local energy -> mean,
variance); the custom JVP (:91-142) is out of scope.
"""
from types import SimpleNamespace

import torch

from . import hamiltonian


def make_loss(network, simulation_cell, mode='for', partition_number=3):
    el_fun = hamiltonian.local_energy_seperate(network, simulation_cell, mode=mode,
                                               partition_number=partition_number)

    def total_energy(params, data):
        kes, ews = zip(*[el_fun(params, x) for x in data])        # vmap(el_fun, (None, 0)) :64
        ke = torch.stack([k.to(torch.complex128) for k in kes])
        ew = torch.stack(list(ews))
        e_l = ke + ew                                             # :75
        mean_e_l = e_l.mean()                                     # :76
        variance = (e_l.abs() ** 2).mean() - mean_e_l.real.abs() ** 2   # :79
        return mean_e_l.real, SimpleNamespace(variance=variance, local_energy=e_l,
                                              imaginary=mean_e_l.imag, kinetic=ke, ewald=ew)
    return total_energy
