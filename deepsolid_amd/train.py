"""Energy and energy gradient of a batch of walkers (reference DeepSolid/train.py:37-184).

``make_loss`` returns ``total_energy(params, data) -> (loss, AuxiliaryLossData)`` like
train.py:67-89.  JAX differentiates that function through its custom JVP (train.py:91-142);
here the same derivative is the method ``total_energy.value_and_grad(params, data)``
(= ``jax.value_and_grad(total_energy, argnums=0, has_aux=True)``, process.py:204):

    d loss / d theta = mean_b Re( clip_diff_b * conj(d log psi_b / d theta) ),

one reverse sweep of the HIP value chain (`ds_logpsi_vjp`) with cotangent clip_diff / B.
``make_training_step`` is train.py:147-184 with the gradient averaged over ranks in ONE
all-reduce of the packed buffer (RCCL), and ``adam`` a packed-buffer Adam for it.
"""
from collections import namedtuple

import torch

from . import constants, hamiltonian

# the reference's five fields (train.py:28-34) + the non-finite count behind its check_nan step rejection (process.py:303-318)
AuxiliaryLossData = namedtuple('AuxiliaryLossData', ['variance', 'local_energy', 'imaginary', 'kinetic', 'ewald', 'n_nonfinite'],
                               defaults=[None])


def clip_difference(diff, clip_local_energy, clip_type):
    """train.py:105-129: clip E_L - E around the batch statistics (statistics pmean'd over ranks).

    The two statistics of a clip type travel in ONE packed all-reduce (the reference sends two pmeans).  They are taken of
    `diff = E_L - loss` with the ALREADY all-reduced loss, so they cannot ride on the energy statistics' message: a training step
    is three collectives (energy statistics, clip statistics, packed gradient), an energy evaluation one."""
    if clip_local_energy <= 0.0:
        return diff
    if clip_type == 'complex':
        radius, phase = diff.abs(), torch.angle(diff)
        radius_tv, radius_mean = constants.pmean_packed(radius.std(unbiased=False), torch.quantile(radius, 0.5))
        radius_tv, radius_mean = radius_tv.to(radius.dtype), radius_mean.to(radius.dtype)
        lo, hi = radius_mean - radius_tv * clip_local_energy, radius_mean + radius_tv * clip_local_energy
        clip_radius = torch.minimum(torch.maximum(radius, lo), hi)
        return torch.polar(clip_radius, phase)
    if clip_type == 'real':
        tv_re, tv_im = constants.pmean_packed(diff.real.abs().mean(), diff.imag.abs().mean())
        tv_re, tv_im = tv_re.to(diff.real.dtype), tv_im.to(diff.real.dtype)
        re = torch.minimum(torch.maximum(diff.real, -clip_local_energy * tv_re), clip_local_energy * tv_re)
        im = torch.minimum(torch.maximum(diff.imag, -clip_local_energy * tv_im), clip_local_energy * tv_im)
        return torch.complex(re, im)
    raise ValueError('Unrecognized clip type.')


def make_loss(network, batch_network, simulation_cell, clip_local_energy=5.0, clip_type='real', mode='for',
              partition_number=3):
    """Same signature as train.py:37-43; ``batch_network`` is accepted and unused (the
    HIP network is batched natively)."""
    del batch_network
    if clip_type not in ('real', 'complex'):
        raise ValueError('Unrecognized clip type.')
    el_fun = hamiltonian.local_energy_seperate(network, simulation_cell, mode=mode,
                                               partition_number=partition_number)
    system = hamiltonian._system_of(network)

    def total_energy(params, data):
        ke, ew = el_fun(params, data)                                  # train.py:74
        e_l = ke + ew                                                  # :75
        # one deterministic device reduction (ds_energy_stats) -> [sum Re, sum Im, sum |E|^2, n, n_nonfinite, ...]
        st = system.energy_stats(torch.view_as_real(ke), ew)
        n = st[3].clamp(min=1.0)
        mean_re, mean_im = st[0] / n, st[1] / n                        # :76
        var_local = st[2] / n - mean_re.abs() ** 2                     # :79 (per-device variance, then pmean)
        # :78-80 (+ the non-finite count, summed not averaged) in ONE all-reduce of a packed vector
        re, im, var, bad = constants.pmean_packed(mean_re, mean_im, var_local, st[4] * constants.world_size())
        return re, AuxiliaryLossData(variance=var, local_energy=e_l, imaginary=im, kinetic=ke, ewald=ew, n_nonfinite=bad)

    def value_and_grad_packed(params, data):
        """-> ((loss, aux), packed gradient of this rank's walkers)."""
        loss, aux = total_energy(params, data)
        diff = aux.local_energy - loss                                 # :101
        clip_diff = clip_difference(diff, clip_local_energy, clip_type)
        cot = clip_diff / data.shape[0]                                # :136 mean over the batch
        flat, _, _ = system.logpsi_vjp(params, data, cot)
        return (loss, aux), flat

    def value_and_grad(params, data):
        out, flat = value_and_grad_packed(params, data)
        return out, system.unpack_grad(flat, params)

    total_energy.value_and_grad = value_and_grad
    total_energy.value_and_grad_packed = value_and_grad_packed
    total_energy.system = system
    return total_energy


def adam(learning_rate=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """Plain Adam on parameter trees (the role optax.adam plays at process.py:209-219).
    -> (init(params) -> state, update(t, grads, params, state) -> (state, params))."""
    def leaves(o, out):
        if isinstance(o, dict):
            for k in sorted(o):
                leaves(o[k], out)
        elif isinstance(o, (list, tuple)):
            for v in o:
                leaves(v, out)
        else:
            out.append(o)
        return out

    def init(params):
        ps = leaves(params, [])
        return {'count': 0, 'm': [torch.zeros_like(p) for p in ps], 'v': [torch.zeros_like(p) for p in ps]}

    def update(t, grads, params, state):
        lr = learning_rate(t) if callable(learning_rate) else learning_rate
        ps, gs = leaves(params, []), leaves(grads, [])
        # the bias correction follows the optimiser's own step count (optax keeps `count` in its state), not the
        # driver's iteration index: a restored state continues where it stopped
        state['count'] = step = int(state.get('count', int(t))) + 1
        state['m'] = [torch.as_tensor(m, dtype=p.dtype, device=p.device) for m, p in zip(state['m'], ps)]
        state['v'] = [torch.as_tensor(v, dtype=p.dtype, device=p.device) for v, p in zip(state['v'], ps)]
        for p, g, m, v in zip(ps, gs, state['m'], state['v']):
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            mhat = m / (1 - b1 ** step)
            vhat = v / (1 - b2 ** step)
            p.sub_(lr * mhat / (vhat.sqrt() + eps))      # in place: bumps the tensor version, the packed cache refreshes
        return state, params
    return init, update


def _tree_all_finite(tree):
    if isinstance(tree, dict):
        return all(_tree_all_finite(v) for v in tree.values())
    if isinstance(tree, (list, tuple)):
        return all(_tree_all_finite(v) for v in tree)
    return bool(torch.isfinite(tree).all())


def make_training_step(mcmc_step, val_and_grad, opt_update, check_nan=False):
    """train.py:147-184.  ``val_and_grad`` is ``total_energy.value_and_grad`` (tree gradient) or, to send one
    message per step, ``total_energy`` itself: its packed gradient is then averaged over the ranks with a
    single all-reduce before it is unpacked.

    ``check_nan`` (cfg.debug.check_nan, process.py:303-318): a step whose loss, local energies or search direction are
    not finite is DISCARDED -- walkers, parameters and optimiser state keep their previous values and the step returns
    ``loss = aux_data = None``, like the reference's ``except AssertionError`` branch.  The decision is taken BEFORE the
    (in-place) optimiser update from quantities that are already identical on every rank (the all-reduced loss,
    non-finite count and gradient), so all ranks skip together; it costs two device -> host reads (the finiteness of loss and
    gradient, the non-finite count of the local energies)."""
    packed = getattr(val_and_grad, 'value_and_grad_packed', None)

    def step(t, data, params, state, key, mcmc_width):
        new_data, pmove = mcmc_step(params, data, key, mcmc_width)
        if packed is not None:
            (loss, aux_data), flat = packed(params, new_data)
            flat = constants.pmean_if_pmap(flat)                       # :176-177, one RCCL all-reduce
            finite = not check_nan or bool(torch.isfinite(flat).all() & torch.isfinite(loss))
            search_direction = val_and_grad.system.unpack_grad(flat, params) if finite else None
        else:
            (loss, aux_data), search_direction = val_and_grad(params, new_data)
            search_direction = _tree_pmean(search_direction)
            finite = not check_nan or (bool(torch.isfinite(loss)) and _tree_all_finite(search_direction))
        if check_nan and finite and aux_data.n_nonfinite is not None:
            finite = float(aux_data.n_nonfinite) == 0.0
        if not finite:
            # data, params, opt_state are not updated (process.py:314-318)
            return data, params, state, None, None, pmove, None
        state, params = opt_update(t, search_direction, params, state)
        return new_data, params, state, loss, aux_data, pmove, search_direction
    return step


def _tree_pmean(tree):
    if constants.world_size() == 1:
        return tree
    if isinstance(tree, dict):
        return {k: _tree_pmean(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return [_tree_pmean(v) for v in tree]
    return constants.pmean_if_pmap(tree)
