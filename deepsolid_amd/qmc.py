"""Metropolis walker update (reference DeepSolid/qmc.py).

``make_mcmc_step`` keeps the signature of qmc.py:290-297 and returns
``mcmc_step(params, data, key, width) -> (data, pmove)``.  The proposal + wrap
(qmc.py:192-193) and the accept/select (qmc.py:217-222) are HIP kernels
(``ds_mh_propose`` / ``ds_mh_accept``); the wavefunction call in between is the
batched log-psi kernel chain.  ``key`` is a ``torch.Generator`` on the device
(or an int seed): the Philox stream of torch replaces JAX's threefry, or explicit
noise ``(normals, uniforms)`` can be supplied for reproducible tests.
"""
import torch

from . import constants
from .network import NetworkApply


def _generator(key, device):
    if isinstance(key, torch.Generator):
        return key
    g = torch.Generator(device=device)
    g.manual_seed(int(key))
    return g


def mh_update(params, f, x1, key, lp_1, num_accepts, latvec=None, stddev=0.02, atoms=None, i=0,
              normal=None, uniform=None):
    """One all-electron Metropolis step (qmc.py:153-224, symmetric branch).
    Returns (x_new, key, lp_new, num_accepts) like the reference."""
    del i, latvec                                  # the lattice lives in f.system
    if atoms is not None:
        raise NotImplementedError('asymmetric proposals (atoms != None) are flagged untested in the reference '
                                  '(base_config.py:122-126) and are not implemented on the device')
    system = f.system
    if normal is None:
        normal = torch.randn(x1.shape, dtype=x1.dtype, device=x1.device, generator=key)
    if uniform is None:
        uniform = torch.rand(lp_1.shape, dtype=lp_1.dtype, device=lp_1.device, generator=key)
    x2 = system.mh_propose(x1, normal, stddev)
    lp_2 = 2.0 * f(params, x2)
    x_new, lp_new = x1.clone(), lp_1.clone()
    system.mh_accept(x_new, lp_new, x2, lp_2.contiguous(), uniform.contiguous(), num_accepts)
    return x_new, key, lp_new, num_accepts


def make_mcmc_step(batch_slog_network, batch_per_device, latvec, steps=10, atoms=None,
                   importance_sampling=None, one_electron_moves=False):
    if importance_sampling is not None:
        if one_electron_moves:
            raise ValueError('Importance sampling for one elec move is not implemented yet')
        raise NotImplementedError('importance sampling is flagged untested in the reference and not implemented')
    if one_electron_moves:
        raise NotImplementedError('one-electron moves are flagged untested in the reference and not implemented')
    if not isinstance(batch_slog_network, NetworkApply) or batch_slog_network.method_name != 'eval_slogdet':
        raise TypeError("batch_slog_network must be the .apply of make_solid_fermi_net(method_name='eval_slogdet')")
    f = batch_slog_network
    del latvec                                     # equals f.system.cell.a (process.py:185)

    def mcmc_step(params, data, key, width):
        """noise: `key` = torch.Generator / int seed, or a tuple (normals (steps,B,3N), uniforms (steps,B))."""
        explicit = isinstance(key, (tuple, list))
        gen = None if explicit else _generator(key, data.device)
        logprob = 2.0 * f(params, data)                                           # qmc.py:357
        num_accepts = torch.zeros(1, dtype=data.dtype, device=data.device)
        for i in range(steps):                                                    # qmc.py:358
            nz, un = (key[0][i], key[1][i]) if explicit else (None, None)
            data, _, logprob, num_accepts = mh_update(params, f, data, gen, logprob, num_accepts,
                                                      stddev=width, atoms=atoms, normal=nz, uniform=un)
        pmove = num_accepts[0] / (steps * batch_per_device)                       # qmc.py:360
        pmove = constants.pmean_if_pmap(pmove)                                    # qmc.py:361
        return data, pmove
    return mcmc_step
