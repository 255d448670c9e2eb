"""ctypes binding of libdeepsolid_hip.so (include/deepsolid_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be
loaded, every call raises.  Build it with ``python -c "import __graft_entry__
as g; g.build()"`` or ``make -C deepsolid_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEEPSOLID_HIP_LIB: load another build of the SAME library (the sanitizer build of tools/asan_check.sh); still no fallback.
LIB_PATH = os.environ.get('DEEPSOLID_HIP_LIB') or os.path.join(_HERE, 'libdeepsolid_hip.so')

DS_MAX_LAYERS = 8
DS_MAX_SYM = 6
_PD = C.POINTER(C.c_double)


class SystemDesc(C.Structure):
    _fields_ = [
        ('dtype', C.c_int32), ('n_up', C.c_int32), ('n_dn', C.c_int32), ('n_atoms_prim', C.c_int32),
        ('prim_atoms', _PD), ('prim_a', C.c_double * 9), ('sim_a', C.c_double * 9), ('n_sym', C.c_int32),
        ('prim_AV', C.c_double * (DS_MAX_SYM * 3)), ('prim_BV', C.c_double * (DS_MAX_SYM * 3)),
        ('sim_AV', C.c_double * (DS_MAX_SYM * 3)), ('sim_BV', C.c_double * (DS_MAX_SYM * 3)),
        ('n_layers', C.c_int32), ('hidden_single', C.c_int32 * DS_MAX_LAYERS),
        ('hidden_double', C.c_int32 * DS_MAX_LAYERS), ('n_det', C.c_int32), ('distance_type', C.c_int32),
        ('envelope_type', C.c_int32), ('full_det', C.c_int32), ('use_last_layer', C.c_int32),
        ('bias_orbitals', C.c_int32), ('klist_up', _PD), ('klist_dn', _PD),
        ('n_atoms_sim', C.c_int32), ('sim_atoms', _PD), ('sim_charges', _PD), ('dist_mode', C.c_int32),
        ('n_g', C.c_int32), ('gpoints', _PD), ('gweight', _PD), ('ion_exp_re', _PD), ('ion_exp_im', _PD),
        ('ewald_alpha', C.c_double), ('ee_const', C.c_double), ('ei_const', C.c_double), ('ii_total', C.c_double),
    ]


class ParamBlock(C.Structure):
    _fields_ = [('offset', C.c_int64), ('rows', C.c_int32), ('cols', C.c_int32)]


# name -> (restype, argtypes); every symbol declared in include/deepsolid_hip.h
_VP = C.c_void_p
SIGNATURES = {
    'ds_device_widths': (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'ds_system_create': (C.c_int, [C.POINTER(SystemDesc), C.POINTER(_VP)]),
    'ds_system_destroy': (None, [_VP]),
    'ds_last_error': (C.c_char_p, []),
    'ds_param_count': (C.c_int64, [_VP]),
    'ds_int8_layers': (C.c_int, [_VP]),
    'ds_param_layout': (C.c_int, [_VP, C.POINTER(ParamBlock), C.c_int]),
    'ds_workspace_bytes': (C.c_int64, [_VP, C.c_int64]),
    'ds_logpsi': (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, C.c_int64, _VP]),
    'ds_logpsi_grad': (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    'ds_vjp_workspace_bytes': (C.c_int64, [_VP, C.c_int64]),
    'ds_logpsi_vjp': (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    'ds_orbitals': (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, C.c_int64, _VP]),
    'ds_ewald': (C.c_int, [_VP, _VP, C.c_int64, _VP, _VP]),
    'ds_local_energy': (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    'ds_enforce_pbc': (C.c_int, [_PD, C.c_int, _VP, C.c_int64, _VP, _VP, _VP]),
    'ds_mh_propose': (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64, _VP, _VP]),
    'ds_mh_accept': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP, _VP]),
    'ds_mh_propose_ex': (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_double, _VP, C.c_int, C.c_int64, _VP, _VP, _VP]),
    'ds_mh_accept_ex': (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, C.c_double, _VP, _VP, C.c_int, C.c_int64, _VP, _VP, _VP]),
    'ds_mcmc_workspace_bytes': (C.c_int64, [_VP, C.c_int64]),
    'ds_mcmc_step': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int, C.c_double, C.c_uint64, C.c_uint64, _VP, _VP, C.c_int, _VP, _VP,
                              C.c_int64, _VP]),
    'ds_mcmc_step_one_electron': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64, _VP, _VP,
                                           C.c_int, _VP, _VP, C.c_int64, _VP]),
    'ds_mcmc_step_importance': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int, C.c_double, C.c_uint64, C.c_uint64, _VP, _VP, C.c_int,
                                         _VP, _VP, C.c_int64, _VP]),
    'ds_mcmc_step_asymmetric': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int, C.c_double, _VP, C.c_int, C.c_uint64, C.c_uint64, _VP, _VP,
                                         C.c_int, _VP, _VP, C.c_int64, _VP]),
    'ds_philox_host': (None, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]),
    'ds_energy_stats': (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP]),
    'ds_debug_stage': (C.c_int64, [_VP, _VP, _VP, C.c_int64, C.c_char_p, _VP, C.c_int64, _VP, C.c_int64, _VP]),
    'ds_profile_enable': (C.c_int, [_VP, C.c_int]),
    'ds_profile_read': (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'ds_profile_read_clock': (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'ds_debug_timeline': (C.c_int, [_VP, C.POINTER(C.c_uint64), C.c_int]),
    'ds_calib_copy': (C.c_int, [_VP, _VP, C.c_int64, _VP]),
    'ds_mfma_f64_peak': (C.c_int64, [C.c_int64, C.c_int, C.c_int, _VP, _VP]),
}

PROF_KINDS = ('features', 'm2_expand', 'two_layer', 'single_first', 'single_hidden', 'orbital', 'det_inverse',
              'det_trace', 'combine', 'ewald', 'shared_term', 'single_lr')

_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP extension is not built and deepsolid_amd has no CPU '
            'fallback.  Run `make -C deepsolid_amd/csrc` (hipcc, --offload-arch=gfx950).')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ds_last_error()
        raise RuntimeError(f'{what} failed: {msg.decode() if msg else rc}')
