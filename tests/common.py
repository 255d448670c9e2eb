"""Shared helpers for the test-suite (fixtures, system construction)."""
import functools
import os

import numpy as np
import torch

from deepsolid_amd import systems
from oracle import network as onet
from oracle.testing import CASES, make_test_params, params_checksum

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_case(name):
    """-> (fixture dict, simulation cell, klist, net_kw, numpy params)."""
    fx = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    case = CASES[name]
    cell, klist = systems.build(case['system'], twist=case.get('twist', (0, 0, 0)),
                                **case.get('system_kw', {}))
    if case.get('sym_type'):
        from deepsolid_amd import supercell
        supercell.set_symmetry_lat(cell, case['sym_type'])
    net_kw = dict(systems.DETNET_DEFAULTS)
    net_kw.update(case.get('net_kw', {}))
    params = make_test_params(case['seed'], cell.original_cell.atom_coords(), cell.nelec, net_kw)
    np.testing.assert_allclose(params_checksum(params), fx['params_checksum'], rtol=1e-13,
                               err_msg='numpy default_rng stream differs from the fixture generator')
    return fx, cell, klist, net_kw, params


def oracle_net(cell, klist, net_kw, method):
    return onet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name=method, **net_kw)


def tt(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


@functools.lru_cache(maxsize=None)
def _float32_budget_walker(name, b):
    """(ref, loss) of fixture walker b from the oracle: the committed numbers of tests/golden/f32_budget.npz where they exist (made by
    tools/f32_budget_fixture.py with exactly the computation below: the 96-electron walkers cost 7 s of CPU each), else computed."""
    path = os.path.join(GOLDEN, 'f32_budget.npz')
    if os.path.exists(path):
        fxb = np.load(path)
        if f'{name}_ref' in fxb and b < len(fxb[f'{name}_ref']):
            return complex(fxb[f'{name}_ref'][b]), float(fxb[f'{name}_loss'][b])
    return compute_float32_budget_walker(name, b)


def float32_reference_run(name, nb):
    """The REFERENCE'S OWN float32 run (tests/golden/f32_reference.npz, tools/make_f32_reference.py: its hamiltonian.py over its
    network.py executed in float32 / complex64, as JAX runs it by default, and in float64 at the same float32-rounded walkers):
    -> (ref64, loss) with ref64[b] the reference's float64 E_kin at the rounded walker and loss[b] =
    |E_kin(reference, float32) - ref64| / max(1, |ref64|).  Diamond: 2.0e-5, 3.5e-4, 3.6e-3, 2.1e-4 -- the reference's 3N
    forward-over-reverse sweeps lose MORE digits in float32 than the forward-Laplacian chain (oracle: 1.4e-5 ... 6.2e-4)."""
    fxr = np.load(os.path.join(GOLDEN, 'f32_reference.npz'))
    ref = [complex(v) for v in fxr[f'{name}_ke_f64_at_x32'][:nb]]
    loss = [float(abs(complex(fxr[f'{name}_ke_f32'][b]) - ref[b]) / max(1.0, abs(ref[b]))) for b in range(nb)]
    return tuple(ref), tuple(loss)


def compute_float32_budget_walker(name, b):
    from oracle import forward_laplacian as ofl
    fx, cell, klist, net_kw, params = load_case(name)
    p64 = onet.params_to_torch(params)
    p32 = onet.params_to_torch(params, dtype=torch.float32)
    x32 = torch.as_tensor(fx['x'][b], dtype=torch.float32)
    r = complex(ofl.stages(p64, x32.double(), klist, cell, net_kw)['ke'])
    with onet.working_dtype(torch.float32):
        e = abs(complex(ofl.stages(p32, x32, klist, cell, net_kw)['ke']) - r) / max(1.0, abs(r))
    return r, e


def float32_budget(name, nb):
    """What a straight float32 evaluation of the reference ALGORITHM loses on the first `nb` fixture walkers of a case:
    -> (ref, loss) with ref[b] the float64 forward-Laplacian oracle E_kin at the float32-ROUNDED walker and loss[b] the
    relative error |E_kin(f32 oracle) - ref| / max(1, |ref|) of the same restatement run in float32 on the CPU.
    (What the reference's own code loses in float32 is in `float32_reference_run`: more, at the worst walker 6 x more.)
    The loss is a property of the walker (conditioning), not of an implementation: on diamond it ranges from 1e-5 to 6e-4
    over the four fixture walkers and walker 1's moves between 1e-4 and 5e-4 with the host's BLAS summation order
    (tools/f32_budget.py), so float32 tests bound the HIP chain by `float32_tolerance`, not by one number per case."""
    pairs = [_float32_budget_walker(name, b) for b in range(nb)]
    return tuple(p[0] for p in pairs), tuple(p[1] for p in pairs)


def float32_tolerance(loss, b):
    """Relative E_kin tolerance of a float32 implementation at walker b: 3x what the oracle's own float32 run loses there,
    or 3x the case's mean loss where that run happens to land close (+1e-6 for the 4-electron cases at round-off)."""
    return 3 * max(loss[b], sum(loss) / len(loss)) + 1e-6
