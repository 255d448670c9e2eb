"""Oracle (test infrastructure): train.make_loss.total_energy and its gradient.

Restates /root/reference/DeepSolid/train.py:67-89 (batch local energy -> mean,
variance) and the custom JVP :91-142, whose parameter gradient is
    d loss / d theta = mean_b Re( clip_diff_b * conj(d log psi_b / d theta) ),
with clip_diff treated as a constant.  The derivative of log psi comes from torch
autograd over the restated forward (oracle/network.py), the same role jax.jvp
plays at train.py:131.
"""
from types import SimpleNamespace

import torch

from . import hamiltonian
from .network import params_to_torch


def clip_difference(diff, clip_local_energy, clip_type):
    """train.py:105-129 on one device (pmean = identity)."""
    if clip_local_energy <= 0.0:
        return diff
    if clip_type == 'complex':
        radius, phase = diff.abs(), torch.angle(diff)
        radius_tv = radius.std(unbiased=False)                     # jnp .std() is the population std
        radius_mean = torch.quantile(radius, 0.5)                  # jnp.median: mean of the two middle values
        clip_radius = torch.clamp(radius, min=float(radius_mean - radius_tv * clip_local_energy),
                                  max=float(radius_mean + radius_tv * clip_local_energy))
        return clip_radius * torch.exp(1j * phase)
    if clip_type == 'real':
        tv_re = diff.real.abs().mean()
        tv_im = diff.imag.abs().mean()
        re = torch.clamp(diff.real, min=float(-clip_local_energy * tv_re), max=float(clip_local_energy * tv_re))
        im = torch.clamp(diff.imag, min=float(-clip_local_energy * tv_im), max=float(clip_local_energy * tv_im))
        return torch.complex(re, im)
    raise ValueError('Unrecognized clip type.')


def logpsi_vjp(network, params, data, cot):
    """Gradient w.r.t. every parameter leaf of sum_b Re(conj(cot_b) * network(params, x_b)),
    network = eval_logdet (log|psi| + i arg psi).  -> tree shaped like params."""
    leaves = []

    def req(o):
        if isinstance(o, dict):
            return {k: req(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [req(v) for v in o]
        t = o.clone().detach().requires_grad_(True)
        leaves.append(t)
        return t
    p = req(params_to_torch(params))
    tot = 0.0
    for x, c in zip(data, cot):
        lp = network(p, x)
        tot = tot + (torch.conj(c) * lp).real
    grads = torch.autograd.grad(tot, leaves, allow_unused=True)
    it = iter(grads)

    def build(o):
        if isinstance(o, dict):
            return {k: build(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [build(v) for v in o]
        g = next(it)
        return torch.zeros_like(o) if g is None else g
    return build(p)


def make_loss(network, simulation_cell, mode='for', partition_number=3, clip_local_energy=0.0, clip_type='real'):
    el_fun = hamiltonian.local_energy_seperate(network, simulation_cell, mode=mode,
                                               partition_number=partition_number)

    def total_energy(params, data):
        params = params_to_torch(params)
        kes, ews = zip(*[el_fun(params, x) for x in data])        # vmap(el_fun, (None, 0)) :64
        ke = torch.stack([k.to(torch.complex128) for k in kes])
        ew = torch.stack(list(ews))
        e_l = ke + ew                                             # :75
        mean_e_l = e_l.mean()                                     # :76
        variance = (e_l.abs() ** 2).mean() - mean_e_l.real.abs() ** 2   # :79
        return mean_e_l.real, SimpleNamespace(variance=variance, local_energy=e_l,
                                              imaginary=mean_e_l.imag, kinetic=ke, ewald=ew)

    def value_and_grad(params, data):
        """jax.value_and_grad(total_energy, argnums=0, has_aux=True) (process.py:204) through the custom JVP."""
        loss, aux = total_energy(params, data)
        diff = aux.local_energy - loss                            # :101
        clip_diff = clip_difference(diff, clip_local_energy, clip_type)
        cot = clip_diff / len(data)                               # :136 mean over the batch
        return (loss, aux), logpsi_vjp(network, params, data, cot)
    total_energy.value_and_grad = value_and_grad
    return total_energy
