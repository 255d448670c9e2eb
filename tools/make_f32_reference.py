#!/usr/bin/env python3
"""tests/golden/f32_reference.npz: the reference's OWN hamiltonian.py over its own network.py, executed in SINGLE precision.

The float32 GPU tests used to bound the HIP float32 chain by what the ORACLE's float32 run loses (tests/common.py::float32_budget);
the round-4 review asked for a float32 run of the reference itself.  JAX computes in float32 by default, so this is how the
reference's local kinetic energy actually behaves at BASELINE config 5 (diamond, fp32): `local_energy_seperate(f, cell,
mode='for')` (hamiltonian.py:45-70,194-228) under the torch-backed `jax` stand-in with its working precision set to float32 /
complex64 (tools/jax_torch_standin.py::working_dtype), parameters, cell and walker rounded to float32.  Stored per case:
`ke_f32` (complex64 results), `x32` (the rounded walkers) -- next to the float64 `ke_ref` of the main fixture, evaluated at the
UNROUNDED walker, and `ke_f64_at_x32`, the same float64 run at the rounded walker (what a float32 implementation should be
compared with).  Runs only in the build container (needs /root/reference).  Usage: python tools/make_f32_reference.py [case ...]
"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools import make_golden as mg                      # noqa: E402
from tools import jax_torch_standin as standin           # noqa: E402

CASES32 = {'lih': 2, 'bcc_li': 2, 'diamond': 4}


def main():
    import torch
    mg._install_shim()
    from DeepSolid import network as rnet, ewaldsum as rewald, hamiltonian as rham, supercell as rsc
    from deepsolid_amd import systems
    from deepsolid_amd.ewaldsum import EwaldTables
    from oracle.testing import make_test_params, klist_from_kpts, CASES
    out_path = os.path.join(REPO, 'tests', 'golden', 'f32_reference.npz')
    out = dict(np.load(out_path)) if os.path.exists(out_path) else {}
    only = set(sys.argv[1:])
    for name, nw in CASES32.items():
        if only and name not in only:
            continue
        case = CASES[name]
        my_cell = systems.SYSTEMS[case['system']](**case.get('system_kw', {}))
        prim0 = my_cell.original_cell
        prim = mg.FakeCell(prim0.a, prim0.atom_coords(), prim0.atom_charges(), prim0.nelec)
        sim = mg.ref_supercell(rsc, prim, my_cell.S, my_cell.nelec, case.get('sym_type', 'minimal'))
        kpts = rsc.get_supercell_kpts(sim)
        klist = klist_from_kpts(kpts, sim.nelec)
        N = sum(sim.nelec)
        net_kw = dict(systems.DETNET_DEFAULTS)
        net_kw.update(case.get('net_kw', {}))
        params = make_test_params(case['seed'], prim.atom_coords(), sim.nelec, net_kw)
        pnp = mg.to_np_params(params)
        fx = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
        x = fx['x'][:nw]
        tab = EwaldTables(my_cell)
        e_nuc = float(tab.ion_ion + tab.ii_const)

        class _NoEwald:                                   # the kinetic energy is what is compared; the Ewald term is not differentiated
            def __init__(self, _cell):
                self.ion_ion, self.ii_const = tab.ion_ion, tab.ii_const

            def energy(self, xt):
                return [torch.zeros((), dtype=standin.WORK[0])] * 3
        rham.ewaldsum = SimpleNamespace(EwaldSum=_NoEwald)
        res = {}
        for tag, real in (('f64_at_x32', torch.float64), ('f32', torch.float32)):
            with standin.working_dtype(real), standin.torch_mode():
                cast = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).to(real)      # noqa: E731
                tprim = mg.TorchCell(prim)
                tsim = mg.TorchCell(sim, original=tprim, energy_nuc=e_nuc)
                for c in (tprim, tsim):
                    c.a, c.AV, c.BV, c._coords = c.a.to(real), c.AV.to(real), c.BV.to(real), c._coords.to(real)
                tklist = [cast(k) for k in klist]
                tparams = standin.tree_map(cast, pnp)
                tnet = rnet.make_solid_fermi_net(klist=tklist, simulation_cell=tsim, method_name='eval_logdet', **net_kw)
                el = rham.local_energy_seperate(tnet.apply, tsim, mode='for', partition_number=3 if (3 * N) % 3 == 0 else 1)
                kes = []
                for b in range(nw):
                    t0 = time.time()
                    xb = torch.as_tensor(x[b].astype(np.float32)).to(real)          # the float32-ROUNDED walker in both runs
                    k_, _ = el(tparams, xb)
                    want = torch.complex64 if real == torch.float32 else torch.complex128
                    assert k_.dtype == want, f'{name}: the {tag} run produced {k_.dtype}: a float64 operand leaked into it'
                    kes.append(complex(k_))
                    print(f'{name} {tag} walker {b}: {complex(k_):.10f}  ({time.time() - t0:.1f} s)', flush=True)
                res[tag] = np.asarray(kes)
        out[name + '_x32'] = x.astype(np.float32)
        out[name + '_ke_f32'] = res['f32'].astype(np.complex64)
        out[name + '_ke_f64_at_x32'] = res['f64_at_x32']
        out[name + '_ke_ref'] = np.asarray(fx['ke_ref'][:nw])
        loss = np.abs(res['f32'] - res['f64_at_x32']) / np.maximum(1.0, np.abs(res['f64_at_x32']))
        print(name, 'relative loss of the reference in float32:', loss, flush=True)
        np.savez(out_path, **out)


if __name__ == '__main__':
    main()
