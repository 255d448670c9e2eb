#!/usr/bin/env python3
"""Headline benchmark: local-energy evaluations per second (BASELINE.json metric).

One "step" = one primal pass of train.make_loss.total_energy (reference train.py:67-89) over a
batch of synthetic walkers already resident in HBM: kinetic energy (forward-Laplacian HIP chain)
+ Ewald energy per walker, batch mean / variance, and -- for N > 1 -- the packed RCCL all-reduce
that replaces the reference's pmean (train.py:78-80).  Walkers shard across ranks with no other
communication ("scaling": "weak": every rank keeps `--batch` walkers).

    python bench.py [--gpus N --steps K --warmup W] [--scaling weak|strong]
        N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or
        -- when no launcher environment is present -- bench.py re-executes itself under that launcher.
        weak (default): every rank keeps --batch walkers; strong: --batch is the global batch, split over the
        ranks like process.py:72-77 (4096 global -> 512 per GPU at N = 8).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the
library over the timed region) and `cpu_baseline` (the oracle's reference-algorithm restatement,
timed on this box's host cores; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_I8_TOPS = 1024 * 2048 * 2.4e9 / 1e12      # v_mfma_i32_16x16x64_i8: 32768 operations per 16 cycles and SIMD
PEAK_TFLOPS = {torch.float64: 78.6, torch.float32: 157.3}   # MI355X dense matrix = vector peak (datasheet)


def layer_flops(n_elec, kloc, nout):
    """Algorithmic FLOPs of the dominant kernel per walker: the N electron tiles of one hidden
    one-electron-stream layer in the forward-Laplacian formulation:
    D = 3N+2 jet slots per scalar, per-electron rows [h_i | mean_j h2_ji] (kloc), multiply-add = 2.
    (The spin-mean term shared by all electrons is a separate, 15x smaller launch.)"""
    d = 3 * n_elec + 2
    return 2.0 * n_elec * d * kloc * nout


def pmc_traffic(kernel_prefix, system, dtype_name, avg_launch_ms):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, calibrated against a copy of known size -- on gfx950 raw FETCH_SIZE is
    half the bytes, see profiles/*_pmc_traffic.json).  bench.py cannot run rocprofv3 around itself, so this is the value
    measured for the same kernel, system, dtype and launch size when the newest committed profile that names this kernel
    was taken.  Bytes per launch do not depend on the clock the box runs at: the figure is always reported, next to the
    kernel's average duration in the profile run and in this run, with `duration_mismatch` set when the two differ by more
    than 5 % (another box, another clock -- the same launches)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):      # (sorted: rNN ascending, the newest wins)
        try:
            d = json.load(open(f))
            if d.get('system', 'bcc_li') != system or d.get('dtype', 'f64') != dtype_name:
                continue
            for name, e in d['kernels'].items():
                if name.startswith(kernel_prefix) and 'read_bytes_per_launch' in e and 'write_bytes_per_launch' in e:
                    ref_ms = e.get('avg_launch_ms', d.get('avg_launch_ms', {}).get(name))
                    best = dict(bytes_per_launch=e['read_bytes_per_launch'] + e['write_bytes_per_launch'],
                                read=e['read_bytes_per_launch'], write=e['write_bytes_per_launch'],
                                walkers_per_launch=d.get('walkers_per_launch'), profile_avg_launch_ms=ref_ms,
                                run_avg_launch_ms=avg_launch_ms,
                                duration_mismatch=bool(ref_ms is None or abs(ref_ms - avg_launch_ms) > 0.05 * avg_launch_ms),
                                source=os.path.basename(f))
                    # all kernels of one evaluation in the same profile (the driver runs several evaluations: k_features runs once in each)
                    n_eval = max([v.get('launches', 0) for k, v in d['kernels'].items() if 'k_features<' in k] or [0])
                    if n_eval:
                        best['step_bytes'] = sum((v.get('read_bytes_per_launch', 0.0) + v.get('write_bytes_per_launch', 0.0)) * v.get('launches', 0)
                                                 for k, v in d['kernels'].items() if 'k_calib_copy' not in k) / n_eval
        except Exception:
            pass
    return best


def gemm_instance(n_elec, dtype, epi):
    """The k_jet_gemm<T, NB, ST, EPI> instantiation ds_api.hip::dispatch_tiles launches for this electron count
    (NB x 16 features and ST x 16 jet slots per wave)."""
    st = (3 * n_elec + 2 + 15) // 16
    nb = 4 if st <= 5 else (2 if st <= 10 else (2 if dtype == torch.float32 else 1))
    tname = 'double' if dtype == torch.float64 else 'float'
    if epi == 5 and st > 10 and dtype == torch.float32 and not os.environ.get('DS_NO_LDSB'):
        return f'k_jet_gemm_lb<float,{st},5,4>', 1, st            # orbital head of the wide float32 cells: jet rows staged in LDS (ds_ldsb.h)
    # 73 .. 76 jets on five slot tiles (24 electrons), float64: the last slot tile as three 4-column groups (k_jet_gemm<.., G4 = 3>:
    # the dense residual layer <4,5,2,3> and the 48-column orbital head <3,5,5,3>; ds_api.hip, DS_NO_G4=1 switches it off)
    if st == 5 and dtype == torch.float64 and 72 < 3 * n_elec + 2 <= 76 and epi in (2, 5) and not os.environ.get('DS_NO_G4'):
        return f'k_jet_gemm<{tname},{nb},{st},{epi},3>', nb, st
    # ... and one group where at most 4 jets sit on the tenth tile (48 electrons: <2,10,2,1>, <2,10,5,1>)
    if st == 10 and dtype == torch.float64 and 3 * n_elec + 2 - 144 <= 4 and epi in (2, 5) and not os.environ.get('DS_NO_G4'):
        return f'k_jet_gemm<{tname},{nb},{st},{epi},1>', nb, st
    return f'k_jet_gemm<{tname},{nb},{st},{epi}>', nb, st


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_worker(job):
    """One of the concurrent processes of the all-cores CPU baseline: `reps` fori_loop iterations (hamiltonian.py:59-66) on the
    vmapped sample with `threads` torch threads; -> seconds per iteration."""
    system, net_kw, params_np, xs_np, threads, reps = job
    import torch as _t
    from torch.func import vmap
    from deepsolid_amd import systems
    from oracle import hamiltonian as oham
    from oracle import network as onet
    _t.set_num_threads(threads)
    cell, klist = systems.build(system)                 # (host-side cell builder: no GPU; the parent's objects do not pickle)
    net = onet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    p = onet.params_to_torch(params_np)
    xs = _t.as_tensor(xs_np)
    one_dir = vmap(lambda xx: oham.local_kinetic_energy_real_imag(net.apply, directions=1)(p, xx)[0])
    one_dir(xs[:2])
    t0 = time.perf_counter()
    for _ in range(reps):
        one_dir(xs)
    return (time.perf_counter() - t0) / reps


def cpu_all_cores(system, net_kw, params_np, xs_np, t_dir_single, value_single, threads=16, max_procs=4):
    """The same CPU restatement on MORE of the host: `n` processes x `threads` threads side by side, each on the whole sample
    (torch's intra-op pool does not scale past ~16 threads on these small contractions, processes do).  The aggregate is
    n x the single-process rate divided by the slow-down of an iteration under that load.  The processes are plain
    `python bench.py --cpu-worker JOB` children with a deadline: whatever happens to them, the bench line is printed."""
    import pickle
    import subprocess
    import tempfile
    n = max(1, min(max_procs, (os.cpu_count() or 1) // threads))
    if n < 2:
        return None
    procs, path = [], None
    try:
        reps = max(1, min(3, int(6.0 / max(t_dir_single, 1e-3))))
        with tempfile.NamedTemporaryFile(suffix='.pkl', delete=False) as f:
            pickle.dump((system, dict(net_kw), params_np, xs_np, threads, reps), f)
            path = f.name
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', path], stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(n)]
        deadline = time.time() + max(120.0, 30.0 * t_dir_single * reps)
        ts = []
        for pr in procs:
            out, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
            ts.append(float([l for l in out.splitlines() if l.startswith('CPU_WORKER ')][-1].split()[1]))
        slow = max(ts) / t_dir_single
        return dict(value=n * value_single / max(slow, 1.0), processes=n, threads_per_process=threads, cores=n * threads,
                    slowdown_per_iteration=slow,
                    sample=f'{n} processes x {threads} threads, each {reps} fori_loop iteration(s) on the same vmapped walkers at once: '
                           f'{max(ts):.2f} s per iteration against {t_dir_single:.2f} s alone')
    except Exception as e:                                  # the baseline is a courtesy number: never fail the bench line over it
        log(f'all-cores cpu baseline skipped: {e!r}')
        return None
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        if path and os.path.exists(path):
            os.unlink(path)


def cpu_baseline(cell, klist, net_kw, params_np, x_np, e_gpu, seconds=20.0, walkers=64, system=None, e_gpu2=None):
    """Reference-algorithm CPU restatement (JAX is not installable here or on the GPU box; SURVEY 8(d) protocol).

    `for` (the reference default, hamiltonian.py:45-70): `walkers` walkers batched with torch.func.vmap the way
    train.py:64 vmaps the local energy, torch CPU float64 on 16 host threads.  Bounded sample (about `seconds` of CPU
    work): one fori_loop iteration (two jvp-of-grad sweeps) is probed on the vmapped batch; when all 3N iterations fit
    the budget (the 24-electron benchmark cell: 72 x 0.25 s) the COMPLETE loop is timed once -- whole evaluations, whose
    energies are also compared with the GPU's --, otherwise as many iterations as fit are timed and scaled to 3N (the
    iterations are identical work); the vmapped Ewald sum is timed in full.
    `hessian` (hamiltonian.py:104-124, the faster schedule when memory allows) is timed on a 4-walker vmap as a
    courtesy number.  Accuracy of the GPU energies: one walker against the autodiff `hessian` oracle (a different
    algorithm from the HIP chain), three more against the forward-Laplacian oracle."""
    from torch.func import vmap
    from oracle import forward_laplacian as ofl
    from oracle import hamiltonian as oham
    from oracle import network as onet
    # 16 threads: the contractions of one walker are small (<= 832 x 256); on the 256-thread host of the GPU box torch's
    # intra-op pool at full width is ~30x SLOWER than at 16 threads (measured: 46 s vs 1.5 s per loop iteration)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    net = onet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    p = onet.params_to_torch(params_np)
    ew = oham.local_ewald_energy(cell)
    nw = min(walkers, x_np.shape[0])
    xs = torch.as_tensor(x_np[:nw])
    n3 = xs.shape[1]
    one_dir = vmap(lambda xx: oham.local_kinetic_energy_real_imag(net.apply, directions=1)(p, xx)[0])
    one_dir(xs[:2])                                                                      # warm-up (first-call overheads)
    t0 = time.perf_counter(); one_dir(xs[:2]); t_probe = time.perf_counter() - t0       # cost probe on two walkers
    if t_probe * nw / 2 * 3 > 2 * seconds:                # keep the whole sample near `seconds`
        nw = max(2, int(nw * seconds / (t_probe * nw / 2 * 3)))
        xs = xs[:nw]
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); one_dir(xs); ts.append(time.perf_counter() - t0)
    t_dir = float(np.median(ts))
    t0 = time.perf_counter(); e_ew = [float(ew(xx)) for xx in xs]; t_ew = time.perf_counter() - t0
    n_it = n3 if t_dir * n3 <= 1.5 * seconds else max(1, min(n3, int(seconds / t_dir)))
    loop = vmap(lambda xx: sum(oham.local_kinetic_energy_real_imag(net.apply, directions=None if n_it == n3 else n_it)(p, xx)))
    t0 = time.perf_counter(); ke_for = loop(xs); t_loop = time.perf_counter() - t0
    t_for = t_loop * n3 / n_it + t_ew
    log(f'cpu baseline `for`: {t_loop:.1f} s for {n_it} of {n3} fori_loop iterations on {nw} vmapped walkers '
        f'({t_dir:.2f} s per iteration probed) -> {t_for / nw:.2f} s per evaluation')
    nh = min(4, nw)
    kh = vmap(lambda xx: sum(oham.local_kinetic_energy_real_imag_hessian(net.apply)(p, xx)))
    t0 = time.perf_counter(); ke_h = kh(xs[:nh]); t_h = time.perf_counter() - t0 + t_ew * nh / nw
    log(f'cpu baseline `hessian`: {t_h / nh:.2f} s per evaluation ({nh} vmapped walkers)')
    errs = [abs(complex(e_gpu[0]) - (complex(ke_h[0]) + e_ew[0]))]                      # autodiff oracle
    if n_it == n3:                                                                       # the timed `for`-mode evaluations themselves
        errs += [abs(complex(e_gpu[b]) - (complex(ke_for[b]) + e_ew[b])) for b in range(min(nw, len(e_gpu)))]
    for b in range(1, min(4, x_np.shape[0])):
        xb = torch.as_tensor(x_np[b])
        errs.append(abs(complex(e_gpu[b]) - (complex(ofl.stages(p, xb, klist, cell, net_kw)['ke']) + float(ew(xb)))))
    err2 = None
    if e_gpu2 is not None:              # a second set of GPU energies (the opt-in int8 layer) against the same CPU evaluations
        e2 = [abs(complex(e_gpu2[0]) - (complex(ke_h[0]) + e_ew[0]))]
        if n_it == n3:
            e2 += [abs(complex(e_gpu2[b]) - (complex(ke_for[b]) + e_ew[b])) for b in range(min(nw, len(e_gpu2)))]
        err2 = max(e2)
    cpu_baseline.err2 = err2
    allc = cpu_all_cores(system, net_kw, params_np, xs.numpy(), t_dir, nw / t_for, threads=cores) if system else None
    return dict(value=nw / t_for, unit='local-energy evals/s', cores=cores, kind='port', cpu=cpu_model(),
                mode='for', hessian_mode_value=nh / t_h, all_cores=allc, host_threads=os.cpu_count(),
                sample=f'{nw} walkers under torch.func.vmap; '
                       + (f'the complete fori_loop ({n3} iterations, hamiltonian.py:59-66) timed once: {t_loop:.1f} s, '
                          if n_it == n3 else f'{n_it} of the {n3} fori_loop iterations (hamiltonian.py:59-66) timed, scaled by {n3}/{n_it}, ')
                       + f'+ the Ewald sums in full; torch CPU float64, {cores} threads; '
                       f'{t_for / nw:.2f} s per evaluation (`hessian` mode on {nh} vmapped walkers: {t_h / nh:.2f} s); '
                       'reference-algorithm CPU restatement (no JAX available)'), max(errs)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
    (process.py:72-77,96 splits one batch over the local devices inside one process; here it is one process per GPU)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world, global_batch, use_dist):
    """The launcher + reduction path without a GPU (tests/test_host_cpu.py): every rank joins the process group,
    the max-over-ranks time and the ranks_seen count go through the same all-reduces as the real run."""
    import torch.distributed as dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    dt = 1e-3 * (1 + rank)
    ranks_seen = 1
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        seen = torch.zeros(world, dtype=torch.float64)
        seen[rank] = 1.0
        dist.all_reduce(seen)
        ranks_seen = int(seen.sum().item())
    if rank == 0:
        print(json.dumps({'metric': 'local-energy evals/sec', 'value': None, 'n_gpus': world, 'ranks_seen': ranks_seen,
                          'scaling': args.scaling, 'dry_run': True, 'max_rank_seconds': dt,
                          'config': {'batch_per_gpu': args.batch, 'global_batch': global_batch}}))
    if use_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--system', default='bcc_li')
    ap.add_argument('--batch', type=int, default=4096, help='walkers per GPU (weak) or in total (strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: every rank keeps --batch walkers (default); strong: --batch walkers are split across the ranks')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mcmc', action='store_true', help='skip the (untimed, reported separately) Metropolis sub-benchmark')
    ap.add_argument('--no-int8', '--no-strict', dest='no_int8', action='store_true',
                    help='skip the timed region with the opt-in int8 hidden layer (DS_I8=1) reported beside the headline')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="process-group backend ('nccl' is RCCL on ROCm; 'gloo' only for the launcher test on CPU hosts)")
    ap.add_argument('--shared-gpu', action='store_true',
                    help='every rank uses device 0 (with --backend gloo: the N > 1 path exercised on a one-GPU box; not a scaling number)')
    ap.add_argument('--single-scaling', action='store_true', help='N > 1: measure only --scaling, not the other mode beside it')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / reduction plumbing only (no GPU work): every rank reports in, rank 0 prints the JSON line')
    ap.add_argument('--cpu-worker', default=None, help=argparse.SUPPRESS)       # child process of the all-cores CPU baseline
    args = ap.parse_args()

    if args.cpu_worker:
        import pickle
        with open(args.cpu_worker, 'rb') as f:
            job = pickle.load(f)
        print(f'CPU_WORKER {_cpu_worker(job):.6f}', flush=True)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} ranks')
    global_batch = args.batch if args.scaling == 'strong' else world * args.batch
    if args.scaling == 'strong':                           # process.py:72-77: one global batch split over the devices
        if args.batch % world:
            raise SystemExit(f'--scaling strong: batch {args.batch} is not divisible by {world} ranks (process.py:75)')
        args.batch = args.batch // world
    local = 0 if args.shared_gpu else int(os.environ.get('LOCAL_RANK', 0))
    import torch.distributed as dist
    use_dist = world > 1 or 'RANK' in os.environ          # launched by torchrun: one process per GPU over RCCL
    if args.dry_run:
        return dry_run(args, rank, world, global_batch, use_dist)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if args.backend == 'gloo':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group(args.backend, device_id=dev)

    from deepsolid_amd import network, systems, train
    dtype = torch.float64 if args.dtype == 'f64' else torch.float32
    cell, klist = systems.build(args.system)
    net_kw = dict(systems.DETNET_DEFAULTS)
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=dtype, **net_kw)
    params = net.init(0)                                   # numpy default_rng(0), identical on every rank
    x_np = systems.synthetic_walkers(cell, args.batch, seed=1234 + rank)
    x = torch.as_tensor(x_np, dtype=dtype, device=dev)
    total_energy = train.make_loss(net.apply, None, cell)
    sysd = net.apply.system

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(xb, steps, warmup, with_profile):
        """W warm-up steps, then exactly `steps` steps between barrier + synchronize; -> (seconds (max over ranks), loss, aux)."""
        for _ in range(warmup):
            loss, aux = total_energy(params, xb)
        sync()
        if with_profile:
            sysd.profile(True, only='single_hidden')  # events + in-kernel clock probe on the hidden-layer kernel only
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            loss, aux = total_energy(params, xb)
        ev1.record()
        sync()
        dt_ = time.perf_counter() - t0
        timed.gpu_ms = ev0.elapsed_time(ev1)          # HIP events on the stream the library launches on: GPU time of the timed region
        if use_dist:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, loss, aux

    log(f'system {args.system}: N={sum(cell.nelec)} batch/GPU={args.batch} world={world}; warm-up')
    dt, loss, aux = timed(x, args.steps, args.warmup, True)
    gpu_ms_main = getattr(timed, 'gpu_ms', None)
    log('timed region done')
    prof = sysd.profile_read()
    clk_cycles, clk_ticks, clk_ghz = sysd.profile_clock()
    # the other scaling mode in the same invocation (N > 1): BASELINE configs 4 / 5 split ONE global batch over the GPUs
    # (process.py:72-77), the tier's convention is a fixed per-GPU batch -- both lines are measured, `value` is --scaling's
    other = None
    if world > 1 and not args.single_scaling:
        # weak run: also ONE global batch of --batch walkers split over the ranks; strong run: also --batch walkers on every rank
        ob = args.batch // world if args.scaling == 'weak' else global_batch
        if ob >= 1 and (args.scaling == 'strong' or args.batch % world == 0):
            xo = torch.as_tensor(systems.synthetic_walkers(cell, ob, seed=4321 + rank), dtype=dtype, device=dev)
            sysd.profile(False)
            dto, _, _ = timed(xo, args.steps, 1, False)
            other = {'scaling': 'strong' if args.scaling == 'weak' else 'weak', 'batch_per_gpu': ob, 'global_batch': ob * world,
                     'value': world * ob * args.steps / dto, 'ms_per_step': dto / args.steps * 1e3, 'steps': args.steps}
    sysd.profile(True)                                 # one extra, untimed step with events around every kernel: the breakdown
    total_energy(params, x)
    torch.cuda.synchronize()
    prof_all = sysd.profile_read()
    sysd.profile(False)
    log(f'{args.steps} steps in {dt:.3f} s; kernels: ' + ', '.join(f'{k}={v[0]:.1f}ms' for k, v in prof_all.items()))
    # The same step with the dense hidden layer as the 47-bit int8 split of csrc/ds_i8.h (opt-in, DS_I8=1 at handle creation; the
    # default and the headline are float64 arithmetic throughout since round 6), 5 timed steps on a handle of its own: reported
    # beside the headline with its own error against the same CPU evaluations
    i8blk = None
    e_i8 = None
    if world == 1 and dtype == torch.float64 and not sysd.int8_layers() and not args.no_int8:
        os.environ['DS_I8'] = '1'
        try:
            cell8, klist8 = systems.build(args.system)             # (a cell object of its own: handles are cached per cell)
            net8 = network.make_solid_fermi_net(klist=klist8, simulation_cell=cell8, method_name='eval_logdet', dtype=dtype, **net_kw)
            sys8 = net8.apply.system                               # (the handle is created on first use: inside the switch's scope)
        finally:
            del os.environ['DS_I8']
        te8 = train.make_loss(net8.apply, None, cell8)
        if sys8.int8_layers() > 0:
            te8(params, x)
            torch.cuda.synchronize()
            n8 = 5
            sys8.profile(True, only='single_hidden')
            t0 = time.perf_counter()
            for _ in range(n8):
                loss8, aux8 = te8(params, x)
            torch.cuda.synchronize()
            dt8 = time.perf_counter() - t0
            ms8, nl8 = sys8.profile_read()['single_hidden']
            sys8.profile(False)
            e_i8 = aux8.local_energy[:64].cpu().numpy()
            d_all = (aux8.local_energy - aux.local_energy).abs()
            n_e8 = sum(cell.nelec)
            kloc8 = net_kw['hidden_dims'][0][0] + (2 if cell.nelec[1] else 1) * net_kw['hidden_dims'][0][1]
            ops8 = 21 * 2.0 * kloc8 * net_kw['hidden_dims'][0][0] * ((3 * n_e8 + 2 + 15) // 16 * 16) * n_e8
            ach8 = ops8 * args.batch * nl8 / max(ms8 * 1e-3, 1e-12) / 1e12
            i8blk = {'value': args.batch * n8 / dt8, 'ms_per_step': dt8 / n8 * 1e3, 'steps': n8, 'max_abs_err_ha': None,
                     'energy_mean_ha': float(loss8), 'int8_layers': sys8.int8_layers(),
                     'int8_vs_float64_max_abs_diff_ha': float(d_all.max()), 'int8_vs_float64_median_abs_diff_ha': float(d_all.median()),
                     'kernel': 'ds::i8::k_layer_i8<5,2>', 'avg_launch_ms': ms8 / max(nl8, 1),
                     'roofline': {'bound': 'mfma', 'achieved': ach8, 'peak': PEAK_I8_TOPS, 'unit': 'TOP/s', 'frac': ach8 / PEAK_I8_TOPS,
                                  'useful_frac': ach8 / PEAK_I8_TOPS * (3 * n_e8 + 2) / ((3 * n_e8 + 2 + 15) // 16 * 16),
                                  'note': 'executed int8 operations (21 plane products, padded slots included) / launch time / (1024 SIMDs x 2048 op/clk x 2.4 GHz)'},
                     'note': 'DS_I8=1: same chain, dense hidden layer as a 47-bit truncating int8 split (opt-in); max_abs_err_ha against the same CPU evaluations as the headline'}
            log(f"int8 split (opt-in): {i8blk['ms_per_step']:.2f} ms per step, layer {i8blk['avg_launch_ms']:.2f} ms, max |dE_L| vs float64 {i8blk['int8_vs_float64_max_abs_diff_ha']:.2e} Ha")
        del te8, net8
        torch.cuda.empty_cache()
    mcmc = None
    if not args.no_mcmc:
        # Metropolis sub-benchmark (width 0.02, 20 moves = base_config.py:110,116), outside the timed region:
        # one `ds_mcmc_step` call = 21 log|psi| forwards of the whole batch, in-kernel Philox noise
        from deepsolid_amd import qmc
        slog = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', dtype=dtype, **net_kw)
        step = qmc.make_mcmc_step(slog.apply, args.batch, cell.a, steps=20)
        xm, pm = step(params, x, 11 + rank, 0.02)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for r in range(reps):
            xm, pm = step(params, xm, 100 + r, 0.02)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        e0.record()
        for r in range(reps):
            lp = slog.apply(params, xm)
        e1.record()
        torch.cuda.synchronize()
        mcmc = {'ms_per_mcmc_step': ms, 'moves': 20, 'width': 0.02, 'pmove': float(pm), 'logpsi_forward_ms': e0.elapsed_time(e1) / reps,
                'walker_moves_per_s': world * args.batch * 20 / (ms * 1e-3)}
        log(f"mcmc_step (20 moves, {args.batch} walkers): {ms:.1f} ms, log-psi forward {mcmc['logpsi_forward_ms']:.2f} ms, pmove {float(pm):.3f}")
    ranks_seen = 1
    if use_dist:
        seen = torch.zeros(world, dtype=torch.float64, device=dev)
        seen[rank] = 1.0
        dist.all_reduce(seen)                               # every rank that ran the timed region reports in
        ranks_seen = int(seen.sum().item())
    if rank != 0:
        dist.destroy_process_group()
        return

    n_e = sum(cell.nelec)
    h1 = net_kw['hidden_dims'][0][0]
    h2 = net_kw['hidden_dims'][0][1]
    nch = 2 if cell.nelec[1] else 1
    ndet = int(net_kw['determinants'])
    peak = PEAK_TFLOPS[dtype]
    n_hidden = len(net_kw['hidden_dims']) - 1
    kms_pre = {k: v[0] for k, v in prof_all.items()}
    lowrank = kms_pre.get('single_lr', 0.0) > 0           # the first hidden layer ran as k_layer1_lr
    n_dense = n_hidden - (1 if lowrank else 0)             # launches of the dense hidden-layer kernel per step
    # algorithmic FLOPs per walker of the two MFMA kernels that can dominate a step
    f_layer = layer_flops(n_e, h1 + nch * h2, h1)                         # one hidden layer, all electrons
    f_orb = sum(2.0 * ns * (3 * n_e + 2) * h1 * (2 * ns * ndet) for ns in cell.nelec if ns)     # orbital head, both spins
    kms = {k: v[0] for k, v in prof_all.items()}                          # ms per kernel kind in the extra profiled step
    hidden_name, nb, st = gemm_instance(n_e, dtype, 2)
    oc = (2 * cell.nelec[0] * ndet + 63) // 64 * 64                       # packed orbital columns of the spin-up head (ds_api.hip)
    orb_name = gemm_instance(n_e, dtype, 5)[0]
    if nb == 4 and st <= 5 and oc % 256 != 0 and oc % 192 == 0:           # the 48-column-per-wave instance
        orb_name = orb_name.replace(f',{nb},{st},5', f',3,{st},5', 1)
    else:
        orb_name = orb_name.replace(',5,3>', ',5>')                      # (the 4-column groups exist for the 48-column instance only)
    # the roofline object describes the kernel with the largest share of the step
    ms_hidden, n_launch = prof['single_hidden']
    flops_total = f_layer * args.batch * n_dense * args.steps            # this rank, timed region (dense hidden layers only)
    achieved = flops_total / (ms_hidden * 1e-3) / 1e12 if ms_hidden > 0 else 0.0
    i8_layers = sysd.int8_layers()                                        # dense hidden layers per step that run as an int8 split (csrc/ds_i8.h)
    if i8_layers and i8_layers == n_dense:
        # the float64 contraction as 21 int8 digit-plane products per 64-row chunk on v_mfma_i32_16x16x64_i8: priced against THAT
        # pipe (1024 SIMDs x 2048 operations per clock x 2.4 GHz; MI355X_MICROARCH.md lists >= 3944 measured), in the operations it
        # executes; the float64 FLOPs it stands for are reported beside it
        kloc = h1 + nch * h2
        p_slots = (3 * n_e + 2 + 15) // 16 * 16
        ops_layer = 21 * 2.0 * kloc * h1 * p_slots * n_e                  # int8 multiply-adds x 2 per walker and layer (executed, padded slots included)
        ach8 = ops_layer * args.batch * n_dense * args.steps / (ms_hidden * 1e-3) / 1e12 if ms_hidden > 0 else 0.0
        hidden_name = f'i8::k_layer_i8<{kloc // 64},2>'
        hidden_obj = {'bound': 'mfma', 'kernel': hidden_name + ' (dense hidden one-electron layer%s: K=%d as a 47-bit truncating int8 split, float64 in / out, fused tanh-jet epilogue)' % ('s' if n_dense != 1 else '', kloc),
                      'mfma': 'i8-split s=6 (47-bit truncating fixed point under one scale per 64-row column chunk, 21 of 36 plane products, float64 recombination)',
                      'achieved': ach8, 'peak': PEAK_I8_TOPS, 'unit': 'TOP/s', 'frac': ach8 / PEAK_I8_TOPS, 'traffic': None, 'traffic_detail': None,
                      # executed operations include the padded jet slots (P = 80 columns for D = 74 jets): the share that is layer arithmetic
                      'useful_frac': ach8 / PEAK_I8_TOPS * (3 * n_e + 2) / p_slots,
                      # the float64 FLOPs the launch stands for (NOT a roofline fraction: the kernel does not run on the float64 pipe)
                      'f64_equivalent': {'tflops': achieved, 'speedup_vs_f64_peak_equivalent': achieved / peak},
                      'avg_launch_ms': ms_hidden / max(n_launch, 1), 'launches': n_launch, 'flops_per_walker_layer': f_layer,
                      'int8_ops_per_walker_layer': ops_layer,
                      'timing': 'HIP events inside the library around every launch of this kernel over the timed region',
                      'share_of_step': kms.get('single_hidden', 0.0) / max(sum(kms.values()), 1e-9)}
    else:
        hidden_obj = {'bound': 'mfma', 'kernel': hidden_name + ' (dense hidden one-electron layer%s: K=%d MFMA GEMM + fused tanh-jet epilogue)' % ('s' if n_dense != 1 else '', h1 + nch * h2),
                      'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': None, 'traffic_detail': None,
                      'avg_launch_ms': ms_hidden / max(n_launch, 1), 'launches': n_launch, 'flops_per_walker_layer': f_layer,
                      'timing': 'HIP events inside the library around every launch of this kernel over the timed region',
                      'share_of_step': kms.get('single_hidden', 0.0) / max(sum(kms.values()), 1e-9)}
    if clk_ghz:
        # measured IN THIS RUN: one wave per workgroup of this kernel reads s_memtime (shader clock) and s_memrealtime (100 MHz)
        # at entry and exit; sum of cycles / sum of ticks over every workgroup of the timed region
        hidden_obj['shader_clock_ghz'] = clk_ghz
        hidden_obj['clock_source'] = 'in-kernel s_memtime / s_memrealtime, all workgroups of the timed region'
        hidden_obj['peak_at_measured_clock'] = peak * clk_ghz / 2.4
        hidden_obj['frac_at_measured_clock'] = achieved / (peak * clk_ghz / 2.4)
    orb_ms = kms.get('orbital', 0.0)
    orb_ach = f_orb * args.batch / (orb_ms * 1e-3) / 1e12 if orb_ms > 0 else 0.0
    orbital_obj = {'bound': 'mfma', 'kernel': orb_name + ' (orbital head: K=%d MFMA GEMM + fused envelope x phase product rule)' % h1,
                   'achieved': orb_ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': orb_ach / peak, 'traffic': None,
                   'flops_per_walker': f_orb, 'timing': 'HIP events, one extra profiled step outside the timed region',
                   'share_of_step': orb_ms / max(sum(kms.values()), 1e-9)}
    lr_obj = None
    if lowrank:
        # first hidden layer on the low-rank form of the layer-0 output: its own (smaller) operation count -- per electron the
        # weights C (Kh x (K0 + 2) x Nout), the products over the K0 + pair-mean rows, the recomputed residual rows
        n_atoms = len(np.asarray(cell.original_cell.atom_coords()).reshape(-1, 3))
        k0loc, k0sh = 4 * n_atoms + 4 * nch, 4 * n_atoms * nch
        d_slots = 3 * n_e + 2
        f_lr = n_e * (2.0 * h1 * h1 * (k0loc + k0sh + 2) + 2.0 * (k0loc + k0sh + nch * h2) * h1 * d_slots + 2.0 * k0loc * h1 * d_slots)
        lr_ms = kms['single_lr']
        lr_ach = f_lr * args.batch / (lr_ms * 1e-3) / 1e12
        lr_obj = {'bound': 'mfma', 'kernel': gemm_instance(n_e, dtype, 2)[0].replace(',2,3>', ',2>').replace('k_jet_gemm', 'k_layer1_lr').replace(',2>', ',NC,true,NG>') +
                  ' (first hidden layer on the rank-%d form of the layer-0 output)' % (k0loc + k0sh),
                  'achieved': lr_ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': lr_ach / peak, 'traffic': None,
                  'flops_per_walker': f_lr, 'dense_layer_flops_per_walker': f_layer,
                  'timing': 'HIP events, one extra profiled step outside the timed region',
                  'share_of_step': lr_ms / max(sum(kms.values()), 1e-9)}
    dominant = 'single_hidden' if kms.get('single_hidden', 0.0) >= orb_ms else 'orbital'
    roofline = dict(hidden_obj if dominant == 'single_hidden' else orbital_obj)
    # the step as a whole: algorithmic float64 FLOPs of every matrix-core kernel of one step / the step's wall time / the float64
    # MFMA peak (the kernel fractions above say how well each kernel uses its pipe, this one what the step delivers)
    d_slots = 3 * n_e + 2
    hd = net_kw['hidden_dims']
    n_at = len(np.asarray(cell.original_cell.atom_coords()).reshape(-1, 3))
    f_step = {'hidden_layers': f_layer * n_dense, 'lowrank_layer': (lr_obj or {}).get('flops_per_walker', 0.0), 'orbital_head': f_orb,
              'shared_terms': sum(2.0 * nch * (hd[l - 1][0] if l else 4 * n_at) * hd[l][0] * d_slots for l in range(len(hd))),
              'layer0': 0.0 if lowrank else 2.0 * n_e * (4 * n_at + nch * 4) * h1 * d_slots,
              'pair_layers': sum(2.0 * n_e * n_e * 5 * (hd[l - 1][1] if l else 4) * hd[l][1] for l in range(len(hd) - 1)),
              'det_traces': sum(2.0 * ndet * (2 * ns) * (2 * ns) * ns * d_slots for ns in cell.nelec if ns)}
    if lowrank:
        f_step['layer0'] = 2.0 * n_e * (4 * n_at + nch * 4) * h1 * d_slots
    f_tot = sum(f_step.values())
    step_ach = f_tot * args.batch / (dt / args.steps) / 1e12
    roofline['step'] = {'achieved': step_ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': step_ach / peak, 'flops_per_walker': f_tot,
                        'flops_per_walker_by_kernel': f_step,
                        'note': 'algorithmic float64 FLOPs of all matrix-core kernels of a step / ms_per_step / float64 MFMA peak'}
    roofline['step_traffic'] = None
    if mcmc:
        # Metropolis path: executed FLOPs of one log|psi| forward (value chain: the same contractions with one column per
        # walker: layers incl. the shared term, pair stream, orbital head) / its duration / peak
        hd = net_kw['hidden_dims']
        k1 = [4 * len(np.asarray(cell.original_cell.atom_coords()).reshape(-1, 3))] + [h[0] for h in hd]
        k2 = [4] + [h[1] for h in hd]
        f_fwd = 0.0
        for l in range(len(hd)):
            f_fwd += 2.0 * n_e * (k1[l] + nch * k2[l]) * k1[l + 1] + 2.0 * nch * k1[l] * k1[l + 1]      # per-electron rows + shared term
            if l < len(hd) - 1:
                f_fwd += 2.0 * n_e * n_e * k2[l] * k2[l + 1]                                            # pair stream
        f_fwd += sum(2.0 * ns * k1[-1] * 2 * ns * ndet for ns in cell.nelec if ns)                       # orbital head
        mcmc['forward_flops_per_walker'] = f_fwd
        mcmc['roofline'] = {'bound': 'mfma', 'achieved': f_fwd * args.batch / (mcmc['logpsi_forward_ms'] * 1e-3) / 1e12, 'peak': peak,
                            'unit': 'TFLOP/s'}
        mcmc['roofline']['frac'] = mcmc['roofline']['achieved'] / peak
    out = {
        'metric': 'local-energy evals/sec', 'value': world * args.batch * args.steps / dt,
        'unit': 'local-energy evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': f'{args.system} {n_e} e- ({cell.nelec[0]},{cell.nelec[1]}), total_energy primal '
                               f'(E_kin forward-Laplacian + Ewald), default detnet ((256,32),)*3, 8 dets',
                   'batch_per_gpu': args.batch, 'global_batch': global_batch, 'parallelism': f'walker-dp{world}'},
        'ranks_seen': ranks_seen,
        'energy_mean_ha': float(loss), 'energy_imag_ha': float(aux.imaginary), 'variance': float(aux.variance),
        'roofline': roofline,
        'roofline_other_kernels': dict(({'orbital_head': orbital_obj} if dominant == 'single_hidden' else {'hidden_layers': hidden_obj}),
                                       **({'lowrank_layer': lr_obj} if lr_obj else {})),
        'kernel_ms_per_step': kms,     # from one extra untimed step
        # evidence of GPU work that does not depend on an smi sample: HIP-event time of the timed region and the kernel sum of a step
        'gpu_ms_timed_region': gpu_ms_main, 'kernel_ms_sum_per_step': sum(kms.values()),
        'mcmc': mcmc,
        # the default IS strict float64 since round 6; the opt-in int8 hidden layer beside it
        'strict_f64': True, 'int8_split': i8blk,
    }
    if other:
        out['other_scaling'] = other
    tr = pmc_traffic('ds::' + hidden_name.split(' ')[0].replace(',', ', '), args.system, args.dtype, hidden_obj['avg_launch_ms'])
    if tr:
        roofline['traffic'] = hidden_obj['traffic'] = tr['bytes_per_launch']
        roofline['step_traffic'] = tr.pop('step_bytes', None)        # counter bytes of every kernel of one step, same profile
        p_slots = (3 * n_e + 2 + 15) // 16 * 16
        # read every layer-input row once, write every output row once (weights and S are L2-resident)
        tr['algorithmic_bytes_per_launch'] = (8.0 if dtype == torch.float64 else 4.0) * tr['walkers_per_launch'] * n_e * p_slots * ((h1 + nch * h2) + h1)
        roofline['traffic_detail'] = hidden_obj['traffic_detail'] = tr
    if world == 1 and not args.no_cpu_baseline:
        params_np = {k: [{kk: vv.cpu().numpy() for kk, vv in d.items()} for d in v] for k, v in params.items()}
        cb, err = cpu_baseline(cell, klist, net_kw, params_np, x_np, aux.local_energy[:64].cpu().numpy(), args.cpu_seconds, system=args.system,
                               e_gpu2=e_i8)
        out['cpu_baseline'] = cb
        out['max_abs_err_ha'] = float(err)
        if i8blk is not None and getattr(cpu_baseline, 'err2', None) is not None:
            i8blk['max_abs_err_ha'] = float(cpu_baseline.err2)
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
