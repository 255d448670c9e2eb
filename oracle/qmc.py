"""Oracle (test infrastructure): Metropolis walker update.

Restates /root/reference/DeepSolid/qmc.py (symmetric all-electron branch, the
only one process.py:183-189 reaches).  The JAX threefry stream cannot be
reproduced, so the Gaussian proposal noise and the uniform acceptance numbers
are explicit arguments; everything else follows the reference line by line.
"""
import torch

from . import distance
from .network import _t


# qmc.py:153-224 (atoms=None branch :191-196, accept/select :217-222)
def mh_update(params, f, x1, lp_1, num_accepts, latvec, stddev=0.02, normal=None, uniform=None,
              atoms=None):
    if atoms is not None:
        raise NotImplementedError('asymmetric proposals are not restated (untested in the reference)')
    x2 = x1 + stddev * normal                        # :192
    x2, _ = distance.enforce_pbc(latvec, x2)         # :193
    lp_2 = 2.0 * f(params, x2)                       # :195
    ratio = lp_2 - lp_1                              # :196
    rnd = torch.log(uniform)                         # :218
    cond = ratio > rnd                               # :219
    x_new = torch.where(cond[..., None], x2, x1)     # :220
    lp_new = torch.where(cond, lp_2, lp_1)           # :221
    num_accepts = num_accepts + cond.sum()           # :222
    return x_new, lp_new, num_accepts


# qmc.py:290-364
def make_mcmc_step(batch_slog_network, batch_per_device, latvec, steps=10, atoms=None,
                   importance_sampling=None, one_electron_moves=False):
    if importance_sampling is not None:
        if one_electron_moves:
            raise ValueError('Importance sampling for one elec move is not implemented yet')
        raise NotImplementedError('importance sampling is not restated (untested in the reference)')
    if one_electron_moves:
        raise NotImplementedError('one-electron moves are not restated (untested in the reference)')

    def mcmc_step(params, data, noise, width):
        """noise = (normals (steps,B,3N), uniforms (steps,B)) replaces the PRNG key."""
        normals, uniforms = noise
        data = _t(data)
        logprob = 2.0 * batch_slog_network(params, data)          # :357
        num_accepts = torch.zeros((), dtype=data.dtype)
        for i in range(steps):                                    # :358 fori_loop
            data, logprob, num_accepts = mh_update(params, batch_slog_network, data, logprob,
                                                   num_accepts, latvec, stddev=width,
                                                   normal=normals[i], uniform=uniforms[i])
        pmove = num_accepts / (steps * batch_per_device)          # :360
        return data, pmove
    return mcmc_step
