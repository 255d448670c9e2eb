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


def find_last_checkpoint(ckpt_path=None):
    """Most recent readable ``qmcjax_ckpt_*`` in a directory, or None (checkpoint.py:39-68)."""
    if ckpt_path and os.path.exists(ckpt_path):
        files = [f for f in os.listdir(ckpt_path) if 'qmcjax_ckpt_' in f]
        for file in sorted(files, reverse=True):
            fname = os.path.join(ckpt_path, file)
            with open(fname, 'rb') as f:
                try:
                    np.load(f, allow_pickle=True)
                    return fname
                except (OSError, EOFError, zipfile.BadZipFile, pickle.UnpicklingError, ValueError):
                    pass        # empty / truncated file: try the next one
    return None


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
    if hasattr(a, 'detach'):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def save(save_path, t, data, params, opt_state, mcmc_width, add_device_axis=True):
    """checkpoint.py:94-124.  Tensors are converted to numpy; with ``add_device_axis`` every array gets the
    leading (1, ...) axis the reference expects from its pmap-replicated state."""
    lead = (lambda a: None if a is None else _to_numpy(a)[None]) if add_device_axis else _to_numpy
    ckpt_filename = os.path.join(save_path, f'qmcjax_ckpt_{t:06d}.npz')
    with open(ckpt_filename, 'wb') as f:
        np.savez(f, t=t, data=lead(data), params=_map(params, lead),
                 opt_state=None if opt_state is None else _map(opt_state, lead),
                 mcmc_width=None if mcmc_width is None else lead(mcmc_width))
    return ckpt_filename


def restore(restore_filename, batch_size=None, shape_check=True, n_devices=1):
    """checkpoint.py:127-165: -> (t, data, params, opt_state, mcmc_width) exactly as stored (t + 1 iterations
    completed).  ``n_devices`` plays the role of jax.local_device_count() in the shape check."""
    with open(restore_filename, 'rb') as f:
        ckpt_data = np.load(f, allow_pickle=True)
        t = ckpt_data['t'].tolist() + 1
        data = ckpt_data['data']
        params = ckpt_data['params'].tolist()
        opt_state = ckpt_data['opt_state'].tolist()
        mcmc_width = ckpt_data['mcmc_width'].tolist()
        if shape_check:
            if data.shape[0] != n_devices:
                raise ValueError('Incorrect number of devices found. Expected {}, found {}.'.format(data.shape[0], n_devices))
            if batch_size and data.shape[0] * data.shape[1] != batch_size:
                raise ValueError('Wrong batch size in loaded data. Expected {}, found {}.'.format(
                    batch_size, data.shape[0] * data.shape[1]))
    return t, data, params, opt_state, mcmc_width


def to_single_device(data, params, mcmc_width=None):
    """Drop the reference's device axis: walkers of all devices concatenated -> (B, 3N); replica 0 of every
    parameter leaf (the replicas are identical, process.py:134-138); scalar width."""
    data = np.asarray(data)
    data = data.reshape(-1, data.shape[-1])
    params = _map(params, lambda a: np.asarray(a)[0])
    if mcmc_width is not None:
        mcmc_width = float(np.asarray(mcmc_width).reshape(-1)[0])
    return data, params, mcmc_width
