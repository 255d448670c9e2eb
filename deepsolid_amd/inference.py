"""Driver loops.  `run_inference`: energy evaluation of a fixed wavefunction, the `optimizer='none'` branch of the reference driver
(reference DeepSolid/process.py:256-374), i.e. burn-in, then per iteration
``mcmc_step -> total_energy -> one CSV row -> MCMC width adaptation``.

This is SURVEY.md section 8 row f1: the smallest step from "kernel" to "usable VMC energy of a trained
wavefunction".  Everything numeric runs in the HIP chain; this module is host bookkeeping only.
"""
import logging
import os

import numpy as np
import torch

from . import qmc, train

TRAIN_SCHEMA = ['step', 'energy', 'variance', 'pmove', 'imaginary', 'kinetic', 'ewald']   # process.py:276


class Writer:
    """CSV writer with the reference's file layout (utils/writers.py:27-91, iteration_key=None)."""

    def __init__(self, name, schema, directory='logs/'):
        self._schema = list(schema)
        os.makedirs(directory, exist_ok=True)
        self._filename = os.path.join(directory, name + '.csv')

    def __enter__(self):
        header = not os.path.exists(self._filename)
        self._file = open(self._filename, 'a+')
        if header:
            self._file.write(','.join(self._schema) + '\n')
        return self

    def write(self, t, **data):
        for key in data:
            if key not in self._schema:
                raise ValueError('Not a recognized key for writer: %s' % key)
        self._file.write(','.join(str(data.get(key, '')) for key in self._schema) + '\n')

    def __exit__(self, *exc):
        self._file.flush()
        self._file.close()


def _rank_generator(key, device, t_init=0):
    """Per-rank noise stream: the reference folds the host index into the key and splits it per device
    (process.py:104, constants.py:54-57); here one process drives one GPU, so the rank is folded into the seed.
    Walkers must be initialised per rank as well (`init_guess.init_electrons` takes its own key).
    `t_init` (the first iteration of a resumed run) is folded in as well, so that a run restarted from a checkpoint with
    the same key does not replay the noise of iterations 0..k of the first run (the reference draws a fresh, time-based
    key on every start, process.py:99-104)."""
    if isinstance(key, torch.Generator):
        return key
    from . import constants
    seed = int(key) * max(1, constants.world_size()) + constants.rank()
    seed = (seed + 0x9E3779B97F4A7C15 * int(t_init)) % (1 << 63)
    return torch.Generator(device=device).manual_seed(seed)


def run_inference(slog_net, logdet_net, params, data, simulation_cell, iterations, key=0, move_width=0.02,
                  mcmc_steps=20, burn_in=100, adapt_frequency=100, stats_frequency=1, save_path=None,
                  stats_file_name='train_stats', laplacian_mode='for', partition_number=3):
    """Returns (data, mcmc_width, rows).  `slog_net` / `logdet_net` are the objects returned by
    ``make_solid_fermi_net(method_name='eval_slogdet' | 'eval_logdet')``; `data` is (B, 3N) on the device.
    Energies are reported per primitive cell like process.py:330-334 (divided by ``simulation_cell.scale``)."""
    gen = _rank_generator(key, data.device)
    batch = data.shape[0]
    mcmc_step = qmc.make_mcmc_step(slog_net.apply, batch, latvec=simulation_cell.a, steps=mcmc_steps)
    total_energy = train.make_loss(logdet_net.apply, None, simulation_cell, mode=laplacian_mode,
                                   partition_number=partition_number)
    width = float(move_width)
    for _ in range(burn_in):                                             # process.py:256-261
        data, _ = mcmc_step(params, data, gen, width)
    scale = float(getattr(simulation_cell, 'scale', 1))
    pmoves = np.zeros(adapt_frequency)
    rows = []
    writer = Writer(stats_file_name, TRAIN_SCHEMA, save_path) if save_path else None
    if writer:
        writer.__enter__()
    try:
        for t in range(iterations):
            data, pmove = mcmc_step(params, data, gen, width)            # process.py:320
            loss, aux = total_energy(params, data)                       # :321
            row = {'step': t,
                   'energy': float(loss) / scale,                        # :330-334
                   'variance': float(aux.variance) / scale ** 2,
                   'pmove': float(pmove),
                   'imaginary': float(aux.imaginary) / scale,
                   'kinetic': complex(aux.kinetic.mean().item()) / scale,
                   'ewald': float(aux.ewald.mean()) / scale}
            if t % stats_frequency == 0:
                rows.append(row)
                if writer:
                    writer.write(t, **row)
            if t > 0 and t % adapt_frequency == 0:                       # :368-373
                if np.mean(pmoves) > 0.55:
                    width *= 1.1
                if np.mean(pmoves) < 0.5:
                    width /= 1.1
                pmoves[:] = 0
            pmoves[t % adapt_frequency] = row['pmove']
    finally:
        if writer:
            writer.__exit__(None, None, None)
    return data, width, rows


def learning_rate_schedule(rate=5e-2, decay=1.0, delay=10000.0):
    """process.py:200-202 with the defaults of base_config.py:46-51."""
    return lambda t: rate * (1.0 / (1.0 + t / delay)) ** decay


def run_training(slog_net, logdet_net, params, data, simulation_cell, iterations, key=0, move_width=0.02, mcmc_steps=10,
                 burn_in=100, adapt_frequency=100, learning_rate=None, clip_local_energy=5.0, clip_type='real',
                 save_path=None, save_every=None, stats_file_name='train_stats', laplacian_mode='for',
                 partition_number=3, t_init=0, opt_state=None, check_nan=True, max_rejected=20):
    """The `optimizer='adam'` branch of the reference driver (process.py:204-219, 256-383): burn-in, then per iteration
    ``mcmc_step -> value_and_grad(total_energy) -> gradient pmean -> Adam -> CSV row -> width adaptation``, with
    checkpoints in the reference's layout (`deepsolid_amd.checkpoint.save`) every `save_every` iterations.
    `params` is updated in place.  Returns (data, params, opt_state, mcmc_width, rows).
    `check_nan` (cfg.debug.check_nan, process.py:303-318; on by default here: Adam updates in place, one NaN gradient would
    poison the parameters for good): a step with non-finite local energies / loss / gradient is discarded -- walkers,
    parameters and optimiser state keep their values, no CSV row is written for it (process.py:344), `rows` gets
    ``{'step': t, 'rejected': True, 'pmove': ...}`` and a warning is logged like the reference's (process.py:316).  A walker with a
    non-finite coordinate or non-finite parameters can never recover (the move is never accepted, the step never kept), so
    after `max_rejected` (default 20) rejections IN A ROW the loop writes the state it is stuck in (walkers, parameters, optimiser
    state: nothing of a rejected step was kept) as `aborted_ckpt_<t>.npz` -- a name `find_last_checkpoint` does not pick up, so a
    restart resumes from the last regular checkpoint -- and raises instead of running to the end doing nothing;
    `max_rejected=None` is the reference's behaviour: log and go on (process.py:303-318)."""
    from . import checkpoint
    gen = _rank_generator(key, data.device, t_init)
    batch = data.shape[0]
    mcmc_step = qmc.make_mcmc_step(slog_net.apply, batch, latvec=simulation_cell.a, steps=mcmc_steps)
    total_energy = train.make_loss(logdet_net.apply, None, simulation_cell, clip_local_energy=clip_local_energy,
                                   clip_type=clip_type, mode=laplacian_mode, partition_number=partition_number)
    opt_init, opt_update = train.adam(learning_rate if learning_rate is not None else learning_rate_schedule())
    if opt_state is None:
        opt_state = opt_init(params)
    step = train.make_training_step(mcmc_step, total_energy, opt_update, check_nan=check_nan)
    width = float(move_width)
    if t_init == 0:                                                      # process.py:256: burn-in only on a fresh start
        for _ in range(burn_in):
            data, _ = mcmc_step(params, data, gen, width)
    scale = float(getattr(simulation_cell, 'scale', 1))
    pmoves = np.zeros(adapt_frequency)
    rows = []
    writer = Writer(stats_file_name, TRAIN_SCHEMA, save_path) if save_path else None
    if writer:
        writer.__enter__()
    t_last = t_init + iterations - 1
    n_rejected = 0
    try:
        for t in range(t_init, t_init + iterations):
            data, params, opt_state, loss, aux, pmove, _ = step(t, data, params, opt_state, gen, width)
            if loss is None:                                             # rejected step: nothing was updated, no CSV row
                rows.append({'step': t, 'rejected': True, 'pmove': float(pmove)})
                n_rejected += 1
                logging.warning('step %d: non-finite local energy / loss / gradient, step discarded (%d in a row)', t, n_rejected)
                if max_rejected is not None and n_rejected >= max_rejected:
                    if save_path:
                        # post-mortem state under a name of its own: a restart must not resume from it (the walkers or parameters
                        # that made every step fail are IN it), and it is the state from before the first rejected step, not step t's
                        checkpoint.save(save_path, t, data, params, opt_state, width, prefix='aborted_ckpt_')
                    raise FloatingPointError(f'{n_rejected} consecutive training steps were rejected for non-finite values '
                                             f'(last at step {t}): walkers or parameters are not finite')
            else:
                row = {'step': t, 'energy': float(loss) / scale, 'variance': float(aux.variance) / scale ** 2,
                       'pmove': float(pmove), 'imaginary': float(aux.imaginary) / scale,
                       'kinetic': complex(aux.kinetic.mean().item()) / scale, 'ewald': float(aux.ewald.mean()) / scale}
                n_rejected = 0
                rows.append(row)
                if writer:
                    writer.write(t, **row)
            if t > 0 and t % adapt_frequency == 0:                       # process.py:368-373
                if np.mean(pmoves) > 0.55:
                    width *= 1.1
                if np.mean(pmoves) < 0.5:
                    width /= 1.1
                pmoves[:] = 0
            pmoves[t % adapt_frequency] = float(pmove)
            # process.py:376-383: every `save_every` iterations and always at the last one; the Adam moments and
            # step count travel with the parameters (process.py:381 saves opt_state)
            if save_path and ((save_every and (t + 1) % save_every == 0) or t >= t_last):
                checkpoint.save(save_path, t, data, params, opt_state, width)
    finally:
        if writer:
            writer.__exit__(None, None, None)
    return data, params, opt_state, width, rows

