"""Periodic FermiNet wavefunction behind the reference's factory signature.

Mirrors ``DeepSolid/network.py``: ``make_solid_fermi_net`` (network.py:609-667)
returns an object with ``.init(key, data=None) -> params`` and
``.apply(params, x)``; ``init_solid_fermi_net_params`` (network.py:60-186)
builds the same nested parameter tree (torch tensors on the device).

``.apply`` accepts one walker ``(3N,)`` (what the reference's per-walker
function takes) or a batch ``(B, 3N)`` (what the reference obtains with
``jax.vmap(..., in_axes=(None, 0))``, process.py:116-118) and runs the HIP
kernel chain through the C ABI (``ds_logpsi`` / ``ds_orbitals``).
"""
import math

import numpy as np
import torch

METHODS = ('eval_slogdet', 'eval_logdet', 'eval_mats', 'eval_phase_and_slogdet')


def _rng(key):
    """`key` may be an int seed, a numpy Generator or None (the JAX PRNG key of the
    reference has no equivalent here; the threefry stream cannot be reproduced)."""
    if isinstance(key, np.random.Generator):
        return key
    if key is None:
        return np.random.default_rng()
    if isinstance(key, torch.Tensor):
        key = int(key.reshape(-1)[-1].item())
    return np.random.default_rng(int(np.asarray(key).reshape(-1)[-1]))


def init_solid_fermi_net_params(key, data=None, atoms=None, spins=None, envelope_type='full',
                                bias_orbitals=False, use_last_layer=False, eps=0.01, full_det=True,
                                hidden_dims=((256, 32), (256, 32), (256, 32)), determinants=16,
                                after_determinants=1, distance_type='nu', dtype=torch.float64, device=None):
    """Parameter tree of network.py:60-186: ``single[l]{w,b}``, ``double[l]{w,b}``,
    ``orbital[s]{w[,b]}``, ``envelope[s]{pi,sigma}``; w ~ N(0,1)/sqrt(fan_in), b ~ N(0,1),
    envelope = ones."""
    del data, after_determinants, eps
    rng = _rng(key)
    natom = np.asarray(atoms).reshape(-1, 3).shape[0]
    if distance_type == 'nu':
        in_dims = (natom * 4, 4)
    elif distance_type == 'tri':
        in_dims = (natom * 7, 7)
    else:
        raise ValueError('Unrecognized distance function.')
    active = [int(s) for s in spins if s > 0]
    nch = len(active)
    dims_one_in = ([(nch + 1) * in_dims[0] + nch * in_dims[1]] +
                   [(nch + 1) * h[0] + nch * h[1] for h in hidden_dims])
    if not use_last_layer:
        dims_one_in[-1] = hidden_dims[-1][0]
    dims_one_out = [h[0] for h in hidden_dims]
    dims_two = [in_dims[1]] + [h[1] for h in hidden_dims]
    len_double = len(hidden_dims) if use_last_layer else len(hidden_dims) - 1

    def t(a):
        return torch.as_tensor(np.asarray(a), dtype=dtype, device=device)
    params = {'single': [], 'double': [], 'orbital': [], 'envelope': []}
    for s in active:
        nparam = sum(spins) * determinants if full_det else s * determinants
        env = {'pi': t(np.ones((natom, nparam)))}
        if envelope_type == 'isotropic':
            env['sigma'] = t(np.ones((natom, nparam)))
        elif envelope_type == 'diagonal':
            env['sigma'] = t(np.ones((natom, 3, nparam)))
        elif envelope_type == 'full':
            env['sigma'] = t(np.tile(np.eye(3)[..., None, None], [1, 1, natom, nparam]))
        params['envelope'].append(env)
    for i in range(len(hidden_dims)):
        params['single'].append({
            'w': t(rng.standard_normal((dims_one_in[i], dims_one_out[i])) / math.sqrt(dims_one_in[i])),
            'b': t(rng.standard_normal((dims_one_out[i],)))})
        if i < len_double:
            params['double'].append({
                'w': t(rng.standard_normal((dims_two[i], dims_two[i + 1])) / math.sqrt(dims_two[i])),
                'b': t(rng.standard_normal((dims_two[i + 1],)))})
    for s in active:
        nparam = sum(spins) * determinants if full_det else s * determinants
        p = {'w': t(rng.standard_normal((dims_one_in[-1], 2 * nparam)) / math.sqrt(dims_one_in[-1]))}
        if bias_orbitals:
            p['b'] = t(rng.standard_normal((2 * nparam,)))
        params['orbital'].append(p)
    return params


class NetworkApply:
    """The ``.apply`` of the reference's haiku-like module, bound to a device system."""

    def __init__(self, simulation_cell, klist, net_kw, method_name, dtype):
        self.simulation_cell = simulation_cell
        self.klist = klist
        self.net_kw = net_kw
        self.method_name = method_name
        self.dtype = dtype
        self._system = None

    @property
    def system(self):
        if self._system is None:
            from .device import DeviceSystem
            self._system = DeviceSystem.for_network(self.simulation_cell, self.klist, self.net_kw, self.dtype)
        return self._system

    def value_and_grad(self, params, x):
        """(log|psi|, d log|psi| / dx) for a batch: the `jax.vmap(jax.value_and_grad(f, argnums=1))` the
        reference builds for importance sampling (qmc.py:324)."""
        single = x.dim() == 1
        la, g = self.system.logpsi_grad(params, x.reshape(1, -1) if single else x)
        return (la[0], g.real[0]) if single else (la, g.real)

    def __call__(self, params, x):
        single = x.dim() == 1
        xb = x.reshape(1, -1) if single else x
        sysd = self.system
        if self.method_name == 'eval_mats':
            mats = sysd.orbitals(params, xb)
            return [m[0] for m in mats] if single else mats
        logabs, ph = sysd.logpsi(params, xb)
        phase = torch.view_as_complex(ph)
        if self.method_name == 'eval_slogdet':
            out = logabs
        elif self.method_name == 'eval_logdet':
            out = torch.complex(logabs, torch.angle(phase))          # log(sign) + slogdet, network.py:597-598
        else:
            out = (phase, logabs)
            return (phase[0], logabs[0]) if single else out
        return out[0] if single else out


class SolidFermiNet:
    def __init__(self, init, apply):
        self.init = init
        self.apply = apply


def make_solid_fermi_net(envelope_type='full', bias_orbitals=False, use_last_layer=False, klist=None,
                         simulation_cell=None, full_det=True, hidden_dims=((256, 32), (256, 32), (256, 32)),
                         determinants=16, after_determinants=1, distance_type='nu', method_name='eval_logdet',
                         dtype=torch.float64):
    """Same arguments and defaults as network.py:609-621 (note: the function defaults differ from
    base_config.py:129-139; process.py:106-110 passes the config).  ``dtype`` is the extra knob that
    replaces the global ``jax_enable_x64`` switch (bin/deepsolid:31-32)."""
    if method_name not in METHODS:
        raise ValueError('Method name is not in class dir.')
    if distance_type not in ('nu', 'tri'):
        raise ValueError('Unrecognized distance function.')
    net_kw = dict(envelope_type=envelope_type, bias_orbitals=bias_orbitals, use_last_layer=use_last_layer,
                  full_det=full_det, hidden_dims=tuple(tuple(h) for h in hidden_dims),
                  determinants=determinants, distance_type=distance_type)
    atoms = np.asarray(simulation_cell.original_cell.atom_coords())
    spins = tuple(int(s) for s in simulation_cell.nelec)

    def init(key, data=None):
        dev = torch.device(f'cuda:{torch.cuda.current_device()}') if torch.cuda.is_available() else None
        return init_solid_fermi_net_params(key, data, atoms=atoms, spins=spins, envelope_type=envelope_type,
                                           bias_orbitals=bias_orbitals, use_last_layer=use_last_layer,
                                           full_det=full_det, hidden_dims=hidden_dims, determinants=determinants,
                                           after_determinants=after_determinants, distance_type=distance_type,
                                           dtype=dtype, device=dev)
    return SolidFermiNet(init, NetworkApply(simulation_cell, klist, net_kw, method_name, dtype))
