"""On-disk compatibility with the reference's checkpoints (DeepSolid/checkpoint.py:39-165).

A reference checkpoint ``qmcjax_ckpt_NNNNNN.npz`` holds ``t``, ``data`` (walkers with a leading local-device
axis, (ndev, B/ndev, 3N)), ``params`` (the parameter tree as a pickled object array; every leaf replicated
over the same leading device axis), ``opt_state`` and ``mcmc_width``.  JAX device arrays pickle as numpy
arrays, so the files load without JAX.  Here one process drives one GPU: ``restore`` returns the reference's
tuple unchanged and ``to_single_device`` drops the device axis (replica 0 of the parameters, all walkers
concatenated); ``save`` writes the same layout with a device axis of length 1 so the reference can read it.
"""
import datetime
import os
import pickle
import zipfile

import numpy as np


def get_restore_path(restore_path=None):
    return restore_path if restore_path else None          # checkpoint.py:27-36


_UNREADABLE = (OSError, EOFError, zipfile.BadZipFile, pickle.UnpicklingError, ValueError)


def _readable(fname):
    try:
        with open(fname, 'rb') as f:
            np.load(f, allow_pickle=True)
        return True
    except _UNREADABLE:
        return False          # empty / truncated file (a run killed while writing)


def find_last_checkpoint(ckpt_path=None):
    """Newest readable ``qmcjax_ckpt_*`` of a directory, or None; unreadable files are skipped so a run killed
    while saving falls back to the previous checkpoint (the behaviour of checkpoint.py:39-68)."""
    if not ckpt_path or not os.path.isdir(ckpt_path):
        return None
    names = sorted((n for n in os.listdir(ckpt_path) if 'qmcjax_ckpt_' in n), reverse=True)
    return next((os.path.join(ckpt_path, n) for n in names if _readable(os.path.join(ckpt_path, n))), None)


def create_save_path(save_path=None):
    timestamp = datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
    path = save_path or os.path.join(os.getcwd(), f'DeepSolid_{timestamp}')          # checkpoint.py:71-91
    if path and not os.path.isdir(path):
        os.makedirs(path)
    return path


def _map(tree, fn):
    if isinstance(tree, dict):
        return {k: _map(v, fn) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_map(v, fn) for v in tree) if not hasattr(tree, '_fields') else type(tree)(*[_map(v, fn) for v in tree])
    return fn(tree)


def _to_numpy(a):
    if a is None:
        return None
    if isinstance(a, (int, float)):
        return a
    if hasattr(a, 'detach'):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def save(save_path, t, data, params, opt_state, mcmc_width, add_device_axis=True, prefix='qmcjax_ckpt_'):
    """checkpoint.py:94-124.  `prefix`: file name stem; anything but the default is NOT picked up by `find_last_checkpoint`
    (the training driver's post-mortem state of an aborted run).  Tensors are converted to numpy; with ``add_device_axis`` every array gets the
    leading (1, ...) axis the reference expects from its pmap-replicated state."""
    lead = (lambda a: a if a is None or isinstance(a, (int, float)) else _to_numpy(a)[None]) if add_device_axis else _to_numpy
    # the scalar bypass above is for the optimiser state only (Adam's python step count).  mcmc_width ALWAYS gets the device
    # axis: the reference restores it with jax.pmap(lambda x: x)(jnp.asarray(mcmc_width_ckpt)) (process.py:252,
    # constants.py:29), which needs shape (n_devices,) -- a python float from the training driver included
    # (a scalar of any kind -> shape (1,); an array that already has the device axis is kept)
    width = (lambda a: np.asarray(_to_numpy(a), dtype=np.float64).reshape(-1)) if add_device_axis else _to_numpy
    ckpt_filename = os.path.join(save_path, f'{prefix}{t:06d}.npz')
    with open(ckpt_filename, 'wb') as f:
        np.savez(f, t=t, data=lead(data), params=_map(params, lead),
                 opt_state=None if opt_state is None else _map(opt_state, lead),
                 mcmc_width=None if mcmc_width is None else width(mcmc_width))
    return ckpt_filename


def restore(restore_filename, batch_size=None, shape_check=True, n_devices=1):
    """-> (t, data, params, opt_state, mcmc_width) from a file in the layout of checkpoint.py:92-120; `t` is the
    number of completed iterations (stored index + 1, checkpoint.py:143).  The same two consistency checks as
    checkpoint.py:151-163: the leading axis of `data` is the device count (``n_devices`` stands in for
    jax.local_device_count()) and devices x per-device batch is the configured batch."""
    with np.load(restore_filename, allow_pickle=True) as ck:
        stored = {k: ck[k] for k in ('t', 'data', 'params', 'opt_state', 'mcmc_width')}
    data = stored['data']
    if shape_check:
        ndev, per_dev = data.shape[0], data.shape[1]
        if ndev != n_devices:
            raise ValueError(f'Incorrect number of devices found. Expected {ndev}, found {n_devices}.')
        if batch_size and ndev * per_dev != batch_size:
            raise ValueError(f'Wrong batch size in loaded data. Expected {batch_size}, found {ndev * per_dev}.')
    unbox = lambda a: a.tolist()          # 0-d object / scalar arrays back to the python objects that were saved
    return int(stored['t']) + 1, data, unbox(stored['params']), unbox(stored['opt_state']), unbox(stored['mcmc_width'])


def opt_state_to_single_device(opt_state):
    """Adam state written by `save` (leaves with the leading device axis) -> replica 0; scalars unchanged."""
    if opt_state is None:
        return None
    return _map(opt_state, lambda a: a if isinstance(a, (int, float)) else np.asarray(a)[0])


def to_single_device(data, params, mcmc_width=None):
    """Drop the reference's device axis: walkers of all devices concatenated -> (B, 3N); replica 0 of every
    parameter leaf (the replicas are identical, process.py:134-138); scalar width."""
    data = np.asarray(data)
    data = data.reshape(-1, data.shape[-1])
    params = _map(params, lambda a: np.asarray(a)[0])
    if mcmc_width is not None:
        mcmc_width = float(np.asarray(mcmc_width).reshape(-1)[0])
    return data, params, mcmc_width
