"""Shared helpers for the test-suite (fixtures, system construction)."""
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
