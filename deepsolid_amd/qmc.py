"""Metropolis walker update (reference DeepSolid/qmc.py).

``make_mcmc_step`` keeps the signature of qmc.py:290-297 and returns
``mcmc_step(params, data, key, width) -> (data, pmove)``.  The proposal + wrap
(qmc.py:192-193) and the accept/select (qmc.py:217-222) are HIP kernels
(``ds_mh_propose`` / ``ds_mh_accept``); the wavefunction call in between is the
batched value-only log-psi kernel chain.  The default sampler (all-electron,
symmetric) runs as ONE C-ABI call, ``ds_mcmc_step``: all `steps` moves are enqueued
without touching the host and the noise is a counter-based Philox4x32-10 stream
evaluated inside the kernels (replacing JAX's threefry, qmc.py:190-192,217-218).
``key`` is an int seed or a ``torch.Generator``; explicit noise ``(normals,
uniforms)`` can be supplied instead for replay tests.  One-electron moves and
the drift-biased importance-sampled move run the same way
(``ds_mcmc_step_one_electron``, ``ds_mcmc_step_importance``), and so does the
asymmetric ``atoms=`` proposal of ``mh_update`` (``ds_mcmc_step_asymmetric``).
The per-move functions (``mh_update`` ... ``importance_update``) keep the
reference's signatures and call the propose / accept kernels move by move.

All three samplers of the reference are available: all-electron Metropolis
(``mh_update``, the default), one-electron moves (``mh_one_electron_update``) and
drift-biased importance sampling (``importance_update``; the latter two are
flagged "untested" in base_config.py:122-126), plus the asymmetric ``atoms=`` proposal
of ``mh_update`` (qmc.py:197-215).
"""
import torch

from . import constants
from . import distance
from .network import NetworkApply


def _generator(key, device):
    if isinstance(key, torch.Generator):
        return key
    g = torch.Generator(device=device)
    g.manual_seed(int(key))
    return g


def _noise(key, x1, lp_1, normal, uniform, normal_shape=None):
    if normal is None:
        normal = torch.randn(normal_shape or x1.shape, dtype=x1.dtype, device=x1.device, generator=key)
    if uniform is None:
        uniform = torch.rand(lp_1.shape, dtype=lp_1.dtype, device=lp_1.device, generator=key)
    return normal, uniform


def _check_latvec(latvec, system):
    """The fused propose kernels wrap with the simulation cell the system was built for; a different `latvec`
    (reference qmc.py:193 wraps with whatever it is given) must not be ignored silently."""
    if latvec is None:
        return
    a = torch.as_tensor(latvec, dtype=torch.float64).cpu().reshape(3, 3)
    if not torch.allclose(a, torch.as_tensor(system.cell.a, dtype=torch.float64).reshape(3, 3), rtol=1e-12, atol=1e-12):
        raise ValueError('latvec differs from the lattice of the simulation cell the network was built for')


def limdrift(g, cutoff=1):
    """Limit each electron's drift vector to magnitude `cutoff` (qmc.py:63-81)."""
    shape = g.shape
    g = g.reshape(-1, 3)
    tot = torch.linalg.norm(g, dim=-1)
    normalize = torch.clamp(tot, min=cutoff, max=float(tot.max()))
    return (cutoff * g / normalize[:, None]).reshape(shape)


def _harmonic_mean(x, atoms):
    """qmc.py:45-60: harmonic mean of each electron's (non-periodic) distances to the nuclei; x (B, N, 1, 3)."""
    r_ae = torch.linalg.norm(x - atoms[None, ...], dim=-1, keepdim=True)
    return 1.0 / torch.mean(1.0 / r_ae, dim=-2, keepdim=True)


def _log_prob_gaussian(x, mu, sigma):
    """qmc.py:26-42."""
    return torch.sum(-0.5 * ((x - mu) ** 2) / (sigma ** 2), dim=(1, 2, 3)) - x.shape[-1] * torch.sum(torch.log(sigma), dim=(1, 2, 3))


def _mh_update_asymmetric(params, f, x1, key, lp_1, num_accepts, latvec, stddev, atoms, normal, uniform):
    """qmc.py:197-215: proposal width scaled per electron by the harmonic mean of its nuclear distances, with the
    forward / reverse proposal densities in the acceptance ratio: `ds_mh_propose_ex` / `ds_mh_accept_ex` (mode 1) around
    the value-chain forward."""
    system = f.system
    _check_latvec(latvec, system)
    atoms = torch.as_tensor(atoms, dtype=x1.dtype, device=x1.device).reshape(-1, 3)
    normal, uniform = _noise(key, x1, lp_1, normal, uniform)
    x2 = system.mh_propose_ex(1, x1, normal.reshape(x1.shape), stddev, atoms)     # :200-204
    la2 = f(params, x2)                                                           # :205 (lp_2 = 2 f)
    x_new, lp_new = x1.clone(), lp_1.clone()
    system.mh_accept_ex(1, x_new, lp_new, x2, la2, uniform, None, stddev, atoms, None, num_accepts)   # :208-222
    return x_new, key, lp_new, num_accepts


def mh_update(params, f, x1, key, lp_1, num_accepts, latvec=None, stddev=0.02, atoms=None, i=0,
              normal=None, uniform=None):
    """One all-electron Metropolis step (qmc.py:153-224; `atoms` selects the asymmetric proposal :197-215).
    Returns (x_new, key, lp_new, num_accepts) like the reference."""
    del i
    if atoms is not None:
        return _mh_update_asymmetric(params, f, x1, key, lp_1, num_accepts, latvec, stddev, atoms, normal, uniform)
    system = f.system
    _check_latvec(latvec, system)
    normal, uniform = _noise(key, x1, lp_1, normal, uniform)
    x2 = system.mh_propose(x1, normal, stddev)                                   # :192-193
    lp_2 = 2.0 * f(params, x2)                                                   # :195
    x_new, lp_new = x1.clone(), lp_1.clone()
    system.mh_accept(x_new, lp_new, x2, lp_2.contiguous(), uniform.contiguous(), num_accepts)   # :217-222
    return x_new, key, lp_new, num_accepts


def mh_one_electron_update(params, f, x1, key, lp_1, num_accepts, latvec=None, stddev=0.02, atoms=None, i=0,
                           normal=None, uniform=None):
    """Metropolis step that moves electron i % N only (qmc.py:227-287)."""
    if atoms is not None:
        raise NotImplementedError('Still need to work out reverse probabilities for asymmetric moves.')   # qmc.py:275
    system = f.system
    _check_latvec(latvec, system)
    nelec = x1.shape[1] // 3
    ii = i % nelec
    normal, uniform = _noise(key, x1, lp_1, normal, uniform, normal_shape=(x1.shape[0], 3))
    full = torch.zeros_like(x1)
    full[:, 3 * ii:3 * ii + 3] = normal                                          # x1.at[:, ii].add(...)  :269
    x2 = system.mh_propose(x1, full, stddev)                                     # + wrap of ALL electrons, :271
    lp_2 = 2.0 * f(params, x2)
    x_new, lp_new = x1.clone(), lp_1.clone()
    system.mh_accept(x_new, lp_new, x2, lp_2.contiguous(), uniform.contiguous(), num_accepts)
    return x_new, key, lp_new, num_accepts


def importance_update(params, f, x1, key, lp_1, num_accepts, latvec, stddev=0.02, atoms=None, i=0,
                      normal=None, uniform=None):
    """Drift-biased all-electron move (qmc.py:83-150, symmetric branch).  `f(params, x)` must
    return (log|psi|, grad log|psi|): ``NetworkApply.value_and_grad``.  The drift limiter, the proposal, the
    forward / backward densities and the selection run in `ds_mh_propose_ex` / `ds_mh_accept_ex` (mode 2)."""
    del i
    if atoms is not None:
        raise NotImplementedError('asymmetric importance sampling is not implemented')
    system = f.__self__.system if hasattr(f, '__self__') else f.system
    _check_latvec(latvec, system)
    normal, uniform = _noise(key, x1, lp_1, normal, uniform)
    _, grad = f(params, x1)                                                       # :111
    grad = grad.contiguous()
    scratch = torch.empty(2, dtype=x1.dtype, device=x1.device)                    # batch maxima of |grad| for limdrift's clip (:78)
    x2 = system.mh_propose_ex(2, x1, normal, stddev, grad, scratch)               # :112-115
    lpsi_2, new_grad = f(params, x2)                                              # :118
    x_new, lp_new = x1.clone(), lp_1.clone()
    system.mh_accept_ex(2, x_new, lp_new, x2, lpsi_2, uniform, normal, stddev, grad, new_grad.contiguous(), num_accepts, scratch)   # :119-137
    return x_new, key, lp_new, num_accepts


import collections

_HOST_GENERATORS = collections.OrderedDict()        # id(device generator) -> (generator, seed it was paired at, host generator); <= 8 entries


def _host_generator(gen):
    """CPU generator paired with a device generator: the Philox keys of the in-kernel noise are drawn from IT (no device round
    trip per `mcmc_step`), not from the device generator's own stream.  It is seeded from `gen.initial_seed()` and re-created
    whenever that seed changes, so `gen.manual_seed(s)` restarts the noise reproducibly; `set_state` on the device generator does
    not (its state never drives the noise).  torch generators cannot be weakly referenced: the cache keeps the eight most
    recently used pairs."""
    seed = int(gen.initial_seed()) & (2 ** 63 - 1)
    g = _HOST_GENERATORS.get(id(gen))
    if g is None or g[0] is not gen or g[1] != seed:
        h = torch.Generator(device='cpu')
        h.manual_seed(seed)
        g = (gen, seed, h)
    _HOST_GENERATORS[id(gen)] = g
    _HOST_GENERATORS.move_to_end(id(gen))
    while len(_HOST_GENERATORS) > 8:
        _HOST_GENERATORS.popitem(last=False)
    return g[2]


def make_mcmc_step(batch_slog_network, batch_per_device, latvec, steps=10, atoms=None,
                   importance_sampling=None, one_electron_moves=False):
    if not isinstance(batch_slog_network, NetworkApply) or batch_slog_network.method_name != 'eval_slogdet':
        raise TypeError("batch_slog_network must be the .apply of make_solid_fermi_net(method_name='eval_slogdet')")
    if importance_sampling is not None:
        if one_electron_moves:
            raise ValueError('Importance sampling for one elec move is not implemented yet')      # qmc.py:321
        if not isinstance(importance_sampling, NetworkApply):
            raise TypeError('importance_sampling must be the .apply of make_solid_fermi_net (process.py:182)')
        func, inner_fun = importance_sampling.value_and_grad, importance_update                   # qmc.py:324-325
    else:
        func = batch_slog_network
        inner_fun = mh_one_electron_update if one_electron_moves else mh_update                  # qmc.py:327-333

    system = batch_slog_network.system
    # every sampler runs as ONE C-ABI call; the importance-sampled one when its drift comes from this system's own network
    # (process.py:182 passes the same wavefunction) -- a foreign gradient function keeps the per-move path
    fused = (importance_sampling is None and not (one_electron_moves and atoms is not None)) or \
            (atoms is None and importance_sampling.system is system)
    if fused:
        _check_latvec(latvec, system)

    def fused_step(params, data, key, width):
        """The default sampler (`ds_mcmc_step`), the one-electron sampler (`ds_mcmc_step_one_electron`), the importance-sampled
        one (`ds_mcmc_step_importance`) or the asymmetric proposal (`ds_mcmc_step_asymmetric`) as ONE C-ABI call:
        proposal, wrap, log|psi|, accept/select for all moves are enqueued back to back; the noise is Philox evaluated inside the kernels.  `key`: an int is a
        pure key like a JAX PRNGKey (same key -> same moves; the caller passes a fresh one per iteration, the rank is
        folded in); a torch.Generator is stateful -- the Philox key of a call is DRAWN from it, so the generator's own state
        carries the stream position: closures sharing one generator never replay each other's noise, and a generator
        restored to a saved state continues the same stream; a tuple (normals, uniforms) replays explicit noise."""
        data = data.clone()
        lp = torch.empty(data.shape[0], dtype=data.dtype, device=data.device)
        # one-electron moves (`ds_mcmc_step_one_electron`): N * steps moves, move i displaces electron i % N   qmc.py:355-358
        nsteps = (data.shape[-1] // 3) * steps if one_electron_moves else steps
        first = 0 if one_electron_moves else None
        imp = importance_sampling is not None
        if isinstance(key, (tuple, list)):
            nacc = system.mcmc_step(params, data, lp, nsteps, width, normals=key[0], uniforms=key[1], first_electron=first,
                                    importance=imp, atoms=atoms)
        else:
            if isinstance(key, torch.Generator):
                # the Philox key is drawn on the HOST (no device synchronisation per call): a CPU generator directly; a device
                # generator through a CPU generator that is seeded from it once and travels with it
                g = key if key.device.type == 'cpu' else _host_generator(key)
                r = torch.randint(0, 2 ** 31 - 1, (2,), generator=g).tolist()
                seed, off = (r[0] << 31) | r[1], 0
            else:
                seed, off = int(key) * max(1, constants.world_size()) + constants.rank(), 0
            nacc = system.mcmc_step(params, data, lp, nsteps, width, seed=seed, offset=off, first_electron=first, importance=imp,
                                    atoms=atoms)
        pmove = nacc[0] / (nsteps * batch_per_device)                             # qmc.py:360
        return data, constants.pmean_if_pmap(pmove)                              # :361

    def mcmc_step(params, data, key, width):
        """noise: `key` = torch.Generator / int seed, or a tuple (normals (nsteps,B,...), uniforms (nsteps,B))."""
        if fused and steps > 0:
            return fused_step(params, data, key, width)
        explicit = isinstance(key, (tuple, list))
        gen = None if explicit else _generator(key, data.device)
        nelec = data.shape[-1] // 3
        nsteps = nelec * steps if one_electron_moves else steps                   # qmc.py:355-356
        logprob = 2.0 * batch_slog_network(params, data)                          # :357
        num_accepts = torch.zeros(1, dtype=data.dtype, device=data.device)
        for i in range(nsteps):                                                   # :358
            nz, un = (key[0][i], key[1][i]) if explicit else (None, None)
            data, _, logprob, num_accepts = inner_fun(params, func, data, gen, logprob, num_accepts, latvec=latvec,
                                                      stddev=width, atoms=atoms, i=i, normal=nz, uniform=un)
        pmove = num_accepts[0] / (nsteps * batch_per_device)                      # :360
        pmove = constants.pmean_if_pmap(pmove)                                    # :361
        return data, pmove
    return mcmc_step
