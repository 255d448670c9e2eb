"""Oracle (test infrastructure): Metropolis walker update.

Restates /root/reference/DeepSolid/qmc.py (all-electron moves: the symmetric branch
process.py:183-189 reaches and the asymmetric `atoms=` proposal).  The JAX threefry stream cannot be
reproduced, so the Gaussian proposal noise and the uniform acceptance numbers
are explicit arguments; everything else follows the reference line by line.
"""
import torch

from . import distance
from .network import _t


# qmc.py:26-42
def _log_prob_gaussian(x, mu, sigma):
    numer = torch.sum(-0.5 * ((x - mu) ** 2) / (sigma ** 2), dim=(1, 2, 3))
    denom = x.shape[-1] * torch.sum(torch.log(sigma), dim=(1, 2, 3))
    return numer - denom


# qmc.py:45-60: harmonic mean of the (non-periodic) electron-nucleus distances
def _harmonic_mean(x, atoms):
    ae = x - atoms[None, ...]
    r_ae = torch.linalg.norm(ae, dim=-1, keepdim=True)
    return 1.0 / torch.mean(1.0 / r_ae, dim=-2, keepdim=True)


# qmc.py:153-224 (atoms=None branch :191-196, asymmetric branch :197-215, accept/select :217-222)
def mh_update(params, f, x1, lp_1, num_accepts, latvec, stddev=0.02, normal=None, uniform=None,
              atoms=None):
    if atoms is None:
        x2 = x1 + stddev * normal                        # :192
        x2, _ = distance.enforce_pbc(latvec, x2)         # :193
        lp_2 = 2.0 * f(params, x2)                       # :195
        ratio = lp_2 - lp_1                              # :196
    else:
        atoms = _t(atoms)
        n = x1.shape[0]
        x1 = x1.reshape(n, -1, 1, 3)                     # :199
        hmean1 = _harmonic_mean(x1, atoms)               # :200
        x2 = x1 + stddev * hmean1 * normal.reshape(x1.shape)   # :202
        x2, _ = distance.enforce_pbc(latvec, x2.reshape(n, -1))   # :203-204
        lp_2 = 2.0 * f(params, x2)                       # :205
        x2 = x2.reshape(n, -1, 1, 3)
        hmean2 = _harmonic_mean(x2, atoms)               # :208
        lq_1 = _log_prob_gaussian(x1, x2, stddev * hmean1)   # :210
        lq_2 = _log_prob_gaussian(x2, x1, stddev * hmean2)   # :211
        ratio = lp_2 + lq_2 - lp_1 - lq_1                # :212
        x1 = x1.reshape(n, -1)
        x2 = x2.reshape(n, -1)
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


# qmc.py:63-81
def limdrift(g, cutoff=1):
    shape = g.shape
    g = g.reshape(-1, 3)
    tot = torch.linalg.norm(g, dim=-1)
    normalize = torch.clamp(tot, min=cutoff, max=float(tot.max()))
    return (cutoff * g / normalize[:, None]).reshape(shape)


# qmc.py:227-287 (symmetric branch)
def mh_one_electron_update(params, f, x1, lp_1, num_accepts, latvec, stddev=0.02, i=0, normal=None, uniform=None):
    n = x1.shape[0]
    x = x1.reshape(n, -1, 3)
    ii = i % x.shape[1]
    x2 = x.clone()
    x2[:, ii] = x2[:, ii] + stddev * normal                  # x1.at[:, ii].add(...)
    x2, _ = distance.enforce_pbc(latvec, x2.reshape(n, -1))
    lp_2 = 2.0 * f(params, x2)
    cond = (lp_2 - lp_1) > torch.log(uniform)
    return (torch.where(cond[..., None], x2, x1), torch.where(cond, lp_2, lp_1), num_accepts + cond.sum())


# qmc.py:83-150 (symmetric branch); f(params, x) -> (log|psi| (B,), grad (B,3N))
def importance_update(params, f, x1, lp_1, num_accepts, latvec, stddev=0.02, normal=None, uniform=None):
    _, grad = f(params, x1)
    grad = limdrift(grad)
    gauss = stddev * normal
    x2, _ = distance.enforce_pbc(latvec, x1 + gauss + stddev ** 2 * grad)
    lpsi_2, new_grad = f(params, x2)
    new_grad = limdrift(new_grad)
    forward = (gauss ** 2).sum(-1)
    backward = ((gauss + stddev ** 2 * (grad + new_grad)) ** 2).sum(-1)
    lp_2 = 2 * lpsi_2 + 1 / (2 * stddev ** 2) * (forward - backward)
    cond = (lp_2 - lp_1) > torch.log(uniform)
    return (torch.where(cond[..., None], x2, x1), torch.where(cond, lp_2, lp_1), num_accepts + cond.sum())
