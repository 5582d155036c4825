#!/usr/bin/env python
"""bench.py -- speech-frames/sec/node of the Speech2Text training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full training step (neural_sp/bin/asr/train.py:414-452): H2D of a synthetic
batch, Conformer encoder fwd, CTC + RNN-T loss, backward (RCCL gradient all-reduce under
DDP when N>1), grad clipping, Adam.  Workload = BASELINE.json configs[3] per GPU:
Conformer-L (d=512, d_ff=2048, H=8, k=15, 12 layers, x8 subsampling) + CTC(0.3) + RNN-T
(2x1024 LSTM prediction net, joint 512, V=1000), 128 utterances per GPU by default (the same step at 64 and at
16 -- the per-GPU batch SURVEY.md section 8(d) names -- is measured and reported beside it, the 16 entry with its
own roofline object) of T~U[1200,1600] 80-dim frames, U~U[120,200] labels, dropout 0.1, bf16 MFMA operands / fp32
accumulate.  Beside the GEMM-class roofline the line carries the north-star number itself: `encoder_mfu` = encoder
forward + backward time on the main stream against 161.5 MFLOP per valid input frame and the 2.5 PFLOP/s
dense bf16 peak, and the node's MEASURED peaks (library bf16 GEMM, streaming copy) next to the vendor peaks.
Weak scaling: every rank owns its own batch (seed = rank); value = valid frames of all
ranks / wall time (max over ranks).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=4)
    p.add_argument('--batch', type=int, default=128,
                   help='utterances per GPU (128 x T~1400 frames: sized for 288 GB of HBM, not for a 16-32 GB card; '
                        'the line also carries the same step at 64 and 16 utterances, see DESIGN.md section 7)')
    p.add_argument('--size', default='L', choices=['L', 'M', 'S', 'XS'])
    p.add_argument('--tmin', type=int, default=1200)
    p.add_argument('--tmax', type=int, default=1600)
    p.add_argument('--umin', type=int, default=120)
    p.add_argument('--umax', type=int, default=200)
    p.add_argument('--dropout', type=float, default=0.1)
    p.add_argument('--mode', default='bf16', choices=['bf16', 'f32'])
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-kernel-events', action='store_true')
    p.add_argument('--no-b16', action='store_true', help='skip the secondary batch-64 / batch-16 measurements')
    p.add_argument('--event-stride', type=int, default=10,
                   help='HIP events around every GEMM launch of every k-th timed step (roofline line) and phase events on the '
                        'steps two behind them.  An instrumented step is ~15 %% slower (two event records per launch, ~360 GEMM '
                        'launches) and counts in `value` like any other timed step: 10 = two of the default 20 steps each '
                        '(round 6; 4 = five each cost ~5 %% of the reported throughput)')
    p.add_argument('--cpu-batch', type=int, default=1, help='utterances in the CPU-baseline sample')
    p.add_argument('--dist-backend', default='nccl', help="'nccl' (= RCCL); 'gloo' only for single-GPU smoke tests of the N>1 code path")
    p.add_argument('--same-device', action='store_true', help='testing only: every rank uses cuda:0')
    p.add_argument('--layers', type=int, default=12, help='encoder blocks (12 = the workload the metric is quoted on)')
    p.add_argument('--emulate', action='store_true',
                   help='testing only (CPU tier, tests/test_bench_emu_cpu.py): CPU tensors and the kernels of libnsp_hip.so on '
                        'the host emulator of tests/hipemu, gloo between the ranks -- exercises this file\'s launcher, rank '
                        'binding, step loop, reductions and JSON line without a GPU; its numbers mean nothing and the line says so')
    return p.parse_args()


def _usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota.  os.cpu_count()
    reports the host's 256 hardware threads even inside a container limited to a handful of them;
    a torch thread pool sized from it oversubscribes by 10-30x (measured: 965 s instead of ~5 s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(args, margs, cores):
    """Oracle port (oracle/model_ref.py, fp32 torch-CPU) timed on a BOUNDED sample of the same workload,
    as a full training step: fwd + loss + bwd + clip_grad_norm_(5) + Adam, 1 warm-up step + up to 10 timed steps
    inside a 30-s box, median reported (SURVEY section 8d asks for warm-up, the optimizer inside and >= 10 steps).  A short probe runs
    first; the sample (utterances of the bench's T / U ranges) is sized from it to ~5 s per step."""
    from neural_sp_amd.configs import synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    from oracle import model_ref
    from oracle import rnnt_ref
    threads = max(1, min(cores, 32))     # this model's CPU ops stop scaling long before 32 threads
    torch.set_num_threads(threads)
    model_ref.rnnt_loss_ref = rnnt_ref.rnnt_loss_ref_diag  # vectorised lattice (same arithmetic)
    torch.manual_seed(0)
    m = Speech2Text(margs)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'inv_freq' not in k)
          for k, v in m.state_dict().items()}
    leaves = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(leaves, lr=1e-4, betas=(0.9, 0.98), eps=1e-9)

    def run(batch):
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        loss, _, _, _ = model_ref.speech2text_loss(sd, margs, batch, torch.float32)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 5.0)
        opt.step()
        return time.time() - t0

    # guard against a pathological host (throttled / oversubscribed): a 160-frame probe first
    t_probe = run(synthetic_batch(B=1, t_range=(160, 160), u_range=(16, 16), vocab=margs.vocab, seed=121))
    if t_probe > 8.0:
        return {'value': round(160 / t_probe, 2), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
                'sample': '1 probe utterance (T=160, U=16) only: it took %.1f s on %d torch threads, the host is '
                          'too slow for a larger sample within the time box' % (t_probe, threads)}
    t_cal = run(synthetic_batch(B=1, t_range=(400, 400), u_range=(40, 40), vocab=margs.vocab, seed=122))
    est_full = 6.0 * t_cal           # one full-length utterance: ~4x the frames, ~20x the lattice nodes
    n = int(max(1, min(4, args.cpu_batch if args.cpu_batch > 1 else 5.0 / max(est_full, 1e-3))))
    batch = synthetic_batch(B=n, t_range=(args.tmin, args.tmax), u_range=(args.umin, args.umax),
                            vocab=margs.vocab, seed=123)
    frames = sum(len(x) for x in batch['xs'])
    times = [run(batch)]                      # warm-up (allocator, thread pool)
    budget = 30.0 - times[0]
    timed = []
    while len(timed) < 10 and (not timed or sum(timed) + timed[-1] < budget):
        timed.append(run(batch))
    dt = sorted(timed)[len(timed) // 2]
    return {'value': round(frames / dt, 2), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            # the reference itself cannot travel to the GPU box; calibration of the port against it in the build container
            # NOT measured in this run: a static calibration read from the committed profile (None if the file is missing)
            'port_over_reference_static_calibration': _port_over_reference(),
            'sample': '%d utterance(s) of the bench workload (%d frames): full training step (fwd + CTC/RNN-T loss + '
                      'bwd + clip + Adam) of the same Conformer-%s model through oracle/model_ref.py in fp32 on %d '
                      'torch threads; 1 warm-up + %d timed steps, median %.2f s (all: %s)'
                      % (n, frames, args.size, threads, len(timed), dt, ', '.join('%.2f' % t for t in timed))}


def _port_over_reference():
    """{'ratio', 'source', 'measured'}: port / reference CPU speed as tools/ref_vs_port_cpu.py measured it in the build
    container (the only place both exist), parsed from the committed profile -- a calibration of `cpu_baseline.kind =
    'port'`, not a number of this run"""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_cpu_reference_vs_port.txt')
    try:
        text = open(path).read()
        ref = float(re.search(r'^reference .*-> ([0-9.]+) frames/s', text, re.M).group(1))
        port = float(re.search(r'^port .*-> ([0-9.]+) frames/s', text, re.M).group(1))
    except Exception:
        return None
    return {'ratio': round(port / ref, 3), 'reference_frames_per_s': ref, 'port_frames_per_s': port,
            'source': 'profiles/r03_cpu_reference_vs_port.txt', 'measured': 'round 3, build container, 8 cores; static'}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher in the environment: re-run this file as N ranks
    (one per GPU) through torch.distributed.run on 127.0.0.1; rank 0 prints the JSON line."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '4'))
    return subprocess.call(cmd, env=env)


_PH = {'on': False}


_EMU = [False]


def _ev():
    if _EMU[0]:
        from neural_sp_amd.parallel import HostEvent
        return HostEvent().record()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def rank_binding(gpus, same_device, env, n_devices):
    """(world, rank, local_rank, device index) of this process from the launcher's environment (one rank per GPU of ONE
    node: device = LOCAL_RANK), or SystemExit with the reason: a line is only printed for the job that was asked for.
    Pure function of its arguments (tests/test_host_cpu.py drives it without a GPU)."""
    world = int(env.get('WORLD_SIZE', '1'))
    if gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to print a line whose '
                         'n_gpus would not be what ran' % (gpus, world))
    rank, local_rank = int(env.get('RANK', '0')), int(env.get('LOCAL_RANK', '0'))
    if not (0 <= rank < world):
        raise SystemExit('bench.py: RANK %d outside WORLD_SIZE %d' % (rank, world))
    if world > 1 and 'LOCAL_RANK' not in env:
        raise SystemExit('bench.py: WORLD_SIZE %d without LOCAL_RANK: launch with torch.distributed.run (one rank per GPU)' % world)
    if same_device:
        return world, rank, local_rank, 0
    if local_rank != rank:
        raise SystemExit('bench.py: LOCAL_RANK %d != RANK %d: this benchmark is one node, one rank per GPU' % (local_rank, rank))
    if local_rank >= n_devices:
        raise SystemExit('bench.py: LOCAL_RANK %d but only %d device(s) visible: --gpus %d needs %d GPUs on this node '
                         '(two ranks on one device are a different measurement: --same-device)' % (local_rank, n_devices, gpus, gpus))
    return world, rank, local_rank, local_rank


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_spawn_ranks(a))
    if a.emulate:
        if a.dist_backend != 'gloo' and a.gpus > 1:
            raise SystemExit('bench.py: --emulate runs CPU ranks: use --dist-backend gloo')
        a.no_kernel_events = a.no_b16 = a.no_cpu_baseline = True
        _EMU[0] = True
        torch.set_num_threads(2)
        from tests.cpu_ops_shim import host_logic_on_cpu
        with host_logic_on_cpu(real_kernels=True, real_conv=False, mode=a.mode):
            return _main(a)
    return _main(a)


def _main(a):
    emu = a.emulate
    assert emu or torch.cuda.is_available(), 'bench.py measures the HIP path; no GPU is visible'
    world, rank, local_rank, dev_index = rank_binding(a.gpus, a.same_device or emu, os.environ,
                                                      1 if emu else torch.cuda.device_count())
    distributed = world > 1
    if a.same_device and not emu:
        local_rank = dev_index
        # ranks sharing one device cannot guarantee co-residency of each other's grid-barrier kernels (two half-resident
        # persistent grids wait on each other until the barrier times out): rank 0 keeps the persistent LSTM -- so that
        # the grid-barrier kernel runs beside the other rank's kernels and under the communication hook -- the others
        # take per-stage launches, which always terminate
        if rank != 0:
            os.environ['NSP_LSTM_PERSISTENT'] = '0'
    if emu:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.dist_backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev)
        else:
            dist.init_process_group(backend=a.dist_backend)
        assert dist.get_world_size() == world == a.gpus, (dist.get_world_size(), world, a.gpus)

    from neural_sp_amd import ops
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    from neural_sp_amd import parallel

    ops.set_compute_mode(a.mode)
    margs = conformer_rnnt_args(a.size, n_layers=a.layers, vocab=1000, dropout=a.dropout, ctc_weight=0.3)
    torch.manual_seed(1)
    model = Speech2Text(margs).to(dev)
    n_params = model.total_parameters
    if distributed:
        train_model = parallel.wrap_ddp(model, None, cpu_hook=True) if emu else parallel.wrap_ddp(model, local_rank)
    else:
        train_model = model
    params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, fused=not emu)
    # distinct synthetic batches (different T_max / U_max); at most as many as warm-up steps so
    # that every shape has been through the caching allocator before the timed region
    n_distinct = max(1, min(4, a.warmup))
    batches = [synthetic_batch(B=a.batch, t_range=(a.tmin, a.tmax), u_range=(a.umin, a.umax),
                               vocab=1000, seed=1000 * rank + i) for i in range(n_distinct)]

    reported = []        # what a Reporter would have logged: one dict of python floats per step
    pending = [None]

    def report_pending():
        if pending[0] is not None:
            reported.append(dict(pending[0].materialize()))   # the ONE D2H transfer of that step
            pending[0] = None

    # per-phase wall time on the main stream (events at phase boundaries of the instrumented steps):
    # encoder forward | loss heads forward | backward (+ DDP reduction) | clip + Adam
    phase_ev = []
    enc_done = [None]
    enc_bwd_start = [None]

    def _enc_fwd_hook(_m, _inp, out):
        if not _PH['on']:
            return
        enc_done[0] = _ev()
        # the encoder's backward starts when the gradient of its output is complete (CTC + RNN-T branches summed)
        xs = out['ys']['xs'] if isinstance(out, dict) else None
        if torch.is_tensor(xs) and xs.requires_grad:
            xs.register_hook(lambda g: enc_bwd_start.__setitem__(0, _ev()))
    model.enc.register_forward_hook(_enc_fwd_hook)
    comm = getattr(train_model, 'comm_stats', None)
    comm_log = []

    def step(i):
        batch = batches[i % len(batches)]
        ph = _PH['on']
        e0 = _ev() if ph else None
        loss, obs = train_model(batch, task='all')
        e1 = _ev() if ph else None
        if distributed:
            loss = loss * world  # train.py:423-424
        if comm is not None:
            comm.reset()
        loss.backward()
        e2 = _ev() if (ph or comm is not None) else None
        if comm is not None and comm.last_ready is not None:
            comm_log.append((comm.last_ready, e2, comm.buckets, comm.bytes, comm.waited_side))
        # the previous step's loss values are read here, one step late: reading step i's values
        # right after its own forward (the reference's .item() calls) or at the end of the step
        # drains the HIP queue and lets the GPU idle through the next step's host-bound forward
        report_pending()
        pending[0] = obs
        parallel.clip_grad_norm_(params, 5.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        if ph:
            phase_ev.append((e0, enc_done[0], e1, e2, _ev(), enc_bwd_start[0], sum(batch['xlens']),
                             len(batch['xlens']) * max(batch['xlens'])))
        return sum(batch['xlens']), len(batch['xlens']) * max(batch['xlens'])

    def sync():
        if distributed:
            dist.barrier()
        if not emu:
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    report_pending()
    sync()
    if not a.no_kernel_events:
        ops.kernel_events_start()
    t0 = time.perf_counter()
    frames = 0
    padded = 0
    for i in range(a.steps):
        if not a.no_kernel_events:
            # HIP events around every GEMM launch of every `event_stride`-th timed step; phase events
            # (5 per step) on the steps in between, so that neither perturbs the other's numbers
            ops.kernel_events_enable(i % a.event_stride == 0)
            _PH['on'] = i % a.event_stride == 2
        v, pd = step(a.warmup + i)
        frames += v
        padded += pd
    report_pending()     # the last step's values are fetched inside the timed region too
    sync()
    dt = time.perf_counter() - t0
    assert len(reported) == a.warmup + a.steps and all(np.isfinite(list(r.values())).all() for r in reported)
    kev = ops.kernel_events_stop() if not a.no_kernel_events else None
    _PH['on'] = False
    phases = None
    if phase_ev:
        torch.cuda.synchronize()
        acc = [0.0, 0.0, 0.0, 0.0, 0.0]
        fr_valid = fr_padded = 0
        for e0, ee, e1, e2, e3, eb, nv, npad in phase_ev:
            acc[0] += e0.elapsed_time(ee); acc[1] += ee.elapsed_time(e1)
            acc[2] += e1.elapsed_time(e2); acc[3] += e2.elapsed_time(e3)
            acc[4] += eb.elapsed_time(e2) if eb is not None else 0.0
            fr_valid += nv; fr_padded += npad
        phases = {k: round(v / len(phase_ev), 2) for k, v in zip(
            ('encoder_fwd_ms', 'loss_fwd_ms', 'backward_ms', 'clip_adam_ms', 'encoder_bwd_ms'), acc)}
        phases['steps_sampled'] = len(phase_ev)
        enc_ms = acc[0] + acc[4]
        if enc_ms > 0 and a.size == 'L' and acc[4] > 0:
            # the north-star number: Conformer-L encoder forward + backward against the dense bf16 MFMA peak.
            # 161.5 MFLOP per input frame = SURVEY.md section 8(d) (conv front-end + bridge + 12 blocks at T/2, T/4, T/8,
            # forward x 3); valid = unpadded frames (headline), padded = what the kernels actually multiply.
            # encoder_bwd_ms runs from "gradient of the encoder output complete" to the end of backward on the main
            # stream (the prediction network's backward on its side stream overlaps it and is not encoder work).
            phases['encoder_mfu'] = {
                'definition': '161.5 MFLOP x input frames / (encoder_fwd_ms + encoder_bwd_ms) / peak',
                'flop_per_frame': 161.5e6, 'peak_tflops': 2500.0 if a.mode == 'bf16' else 157.3,
                'valid_frames': round(161.5e6 * fr_valid / (enc_ms * 1e-3) / 1e12 / (2500.0 if a.mode == 'bf16' else 157.3), 4),
                'padded_frames': round(161.5e6 * fr_padded / (enc_ms * 1e-3) / 1e12 / (2500.0 if a.mode == 'bf16' else 157.3), 4),
                'encoder_tflops_padded': round(161.5e6 * fr_padded / (enc_ms * 1e-3) / 1e12, 1)}

    # Secondary measurements (single GPU only, outside the timed region above): the same step at 64 utterances
    # per GPU (the round-1/2 default) and at 16 (the upper end of the per-GPU batch SURVEY.md section 8(d)
    # wrote down for this config), so that all three operating points are on record in one line.
    also = None
    if not distributed and not a.no_b16:
        also = []
        for b_also in (64, 16):
            if b_also >= a.batch:
                continue
            batches = [synthetic_batch(B=b_also, t_range=(a.tmin, a.tmax), u_range=(a.umin, a.umax),
                                       vocab=1000, seed=7000 + i) for i in range(4)]
            for i in range(4):
                step(i)
            report_pending()
            sync()
            with_events = b_also == 16 and not a.no_kernel_events
            if with_events:
                ops.kernel_events_start()
            t1 = time.perf_counter()
            f_also = 0
            for i in range(12):
                if with_events:
                    ops.kernel_events_enable(i == 6)      # ONE instrumented step of the twelve (this step is host-bound)
                f_also += step(4 + i)[0]
            report_pending()
            sync()
            dt_also = time.perf_counter() - t1
            ent = {'per_gpu_batch': b_also, 'value': round(f_also / dt_also, 1), 'unit': 'frames/s', 'steps': 12,
                   'ms_per_step': round(dt_also / 12 * 1e3, 2)}
            if with_events:
                k16 = ops.kernel_events_stop()
                if k16['launches'] > 0:
                    ach16 = k16['flops'] / (k16['ms'] * 1e-3) / 1e12
                    pk = 2500.0 if a.mode == 'bf16' else 157.3
                    ent['roofline'] = {'kernel': 'the same GEMM class (main-stream launches of 1 of the 12 steps)', 'bound': 'mfma',
                                       'achieved': round(ach16, 2), 'peak': pk, 'unit': 'TFLOP/s', 'frac': round(ach16 / pk, 4),
                                       'traffic': None, 'gemm_launches_per_step': round(k16['launches'] / 1.0, 1),
                                       'avg_launch_us': round(k16['ms'] * 1e3 / k16['launches'], 2),
                                       'gemm_share_of_step': round(k16['ms'] / 1.0 / (dt_also / 12 * 1e3), 3)}
            also.append(ent)

    # measured peak denominators of THIS node ("state both"): a library bf16 GEMM (hipBLASLt behind
    # torch.matmul, 8192^3, random operands) and a streaming device copy (1 GiB, read + write counted)
    peaks = None
    if rank == 0 and not a.no_kernel_events:
        def _timed(fn, reps):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            q0.record()
            for _ in range(reps):
                fn()
            q1.record()
            torch.cuda.synchronize()
            return q0.elapsed_time(q1) / reps * 1e-3
        try:
            n = 8192
            xa = torch.randn(n, n, device=dev).bfloat16()
            xb = torch.randn(n, n, device=dev).bfloat16()
            t_lib = _timed(lambda: torch.matmul(xa, xb.t()), 5)
            xc = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
            t_own = _timed(lambda: ops._gemm_raw_untimed(n, n, n, xa, n, 1, xb, 1, n, xc, n), 5)
            del xa, xb, xc
            src = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
            dst = torch.empty_like(src)
            t_cp = _timed(lambda: dst.copy_(src), 5)
            del src, dst
            peaks = {'library_gemm_bf16_8192_tflops': round(2.0 * n ** 3 / t_lib / 1e12, 1),
                     'nsp_gemm_bf16_8192_tflops': round(2.0 * n ** 3 / t_own / 1e12, 1),
                     'stream_copy_tb_per_s': round(2.0 * (1 << 30) / t_cp / 1e12, 2),
                     'vendor': {'mfma_bf16_dense_tflops': 2500.0, 'hbm_tb_per_s': 8.0}}
        except Exception as e:      # never lose the headline line to the side measurement
            peaks = {'error': repr(e)[:200]}

    tot = torch.tensor([dt, float(frames), float(padded)], device=dev, dtype=torch.float64)
    padded_all = float(padded)
    if distributed:
        tmax = tot[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        fsum = tot[1:2].clone()
        dist.all_reduce(fsum, op=dist.ReduceOp.SUM)
        dt, frames = tmax.item(), fsum.item()
        psum = tot[2:3].clone()
        dist.all_reduce(psum, op=dist.ReduceOp.SUM)
        padded_all = psum.item()

    if rank == 0:
        peak_tf = 2500.0 if a.mode == 'bf16' else 157.3
        roof = None
        if kev is not None and kev['launches'] > 0:
            ach = kev.get('algo_flops', kev['flops']) / (kev['ms'] * 1e-3) / 1e12
            side = kev['side']
            roof = {'kernel': 'bf16 MFMA GEMM class of libnsp_hip.so on the main stream: gemm_bf16_kk8p_kernel<S, ., RR> (phase-'
                              'interleaved 256 x 256 tiles: long reductions, stacked QKV, every weight gradient), gemm_bf16_kk_glds_kernel<0> '
                              '(K = 512 activations x weights, data gradients), rnnt_joint_rows_kernel<16, LSE|DLOGITS> (the RNN-T joint\'s '
                              'two logit passes, node-stationary), '
                              'gemm_bf16_kernel<.,.> / gemm_bf16_kk_ring_kernel<NS,MI> (small grids); '
                              '%d launches = every main-stream GEMM of every %d-th timed step, rank 0.  The %d side-stream '
                              'launches of those steps (CTC head, prediction-network projections: %.1f %% of the GEMM flops) run '
                              'beside main-stream kernels, so their event pairs (%.2f ms) measure co-scheduling and are kept out'
                              % (kev['launches'], a.event_stride, side['launches'],
                                 100.0 * side['flops'] / max(1.0, side['flops'] + kev['flops']), side['ms']),
                    'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak_tf, 'unit': 'TFLOP/s',
                    'frac': round(ach / peak_tf, 4), 'traffic': None, 'traffic_measured_in_run': False,
                    # `achieved` / `frac` count ALGORITHMIC flops: 2 M N K of every GEMM the path defines; the
                    # recomputed logits of the RNN-T joint backward count zero and the vocabulary padding (1000 -> 1024
                    # columns) is not work.  What the kernels execute is kept beside it.
                    'flops_basis': 'algorithmic',
                    'executed': {'achieved': round(kev['flops'] / (kev['ms'] * 1e-3) / 1e12, 2),
                                 'frac': round(kev['flops'] / (kev['ms'] * 1e-3) / 1e12 / peak_tf, 4),
                                 'flop_per_launch': round(kev['flops'] / kev['launches'], 1)},
                    'flop_per_launch': round(kev.get('algo_flops', kev['flops']) / kev['launches'], 1),
                    'avg_launch_us': round(kev['ms'] * 1e3 / kev['launches'], 2),
                    'achieved_all_streams': round((kev['flops'] + side['flops']) / ((kev['ms'] + side['ms']) * 1e-3) / 1e12, 2),
                    'gemm_share_of_step': round(kev['ms'] / (dt * 1e3 * len(range(0, a.steps, a.event_stride)) / a.steps), 3)}
            # the other kernel classes of the step, timed the same way (HIP events around every launch of the sampled
            # steps): MFMA classes against the dense bf16 peak, streaming classes against the node's measured copy rate
            steps_sampled = max(1, len(range(0, a.steps, a.event_stride)))
            cls = []
            for cname, c in sorted(kev.get('classes', {}).items(), key=lambda kv: -kv[1]['ms']):
                if c['launches'] == 0 or c['ms'] <= 0:
                    continue
                ent = {'class': cname, 'launches_per_step': round(c['launches'] / steps_sampled, 1),
                       'ms_per_step': round(c['ms'] / steps_sampled, 3)}
                if c['unit'] == 'flop':
                    tf = c['work'] / (c['ms'] * 1e-3) / 1e12
                    ent.update(bound='mfma', achieved=round(tf, 1), unit='TFLOP/s', peak=peak_tf, frac=round(tf / peak_tf, 4))
                else:
                    tb = c['work'] / (c['ms'] * 1e-3) / 1e12
                    ent.update(bound='hbm', achieved=round(tb, 2), unit='TB/s', peak=8.0, frac=round(tb / 8.0, 4))
                if cname.endswith('@side'):
                    ent['note'] = 'side stream: the event pairs include co-scheduling with main-stream kernels'
                if c['unit'] != 'flop':
                    ent['algorithmic_bytes_per_step'] = round(c['work'] / steps_sampled)
                elif c.get('bytes'):
                    # (MFMA-bound classes also state the bytes their operands weigh, read / written once: the PMC passes'
                    # counter bytes over this = the re-read factor of a class, e.g. the two-kernel attention backward)
                    ent['algorithmic_bytes_per_step'] = round(c['bytes'] / steps_sampled)
                cls.append(ent)
            roof['classes'] = cls
            # HBM bytes per GEMM launch cannot be counted from inside the process: it is the committed result of
            # the rocprofv3 PMC passes over this same command and workload (profiles/pmc_gemm_traffic.json:
            # separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read correction, with the git revision and the
            # workload they were taken on), valid only for the default workload in bf16 mode
            tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_gemm_traffic.json')
            tjd0 = json.load(open(tj)) if os.path.exists(tj) else {}
            default_wl = (a.size, a.batch, a.tmin, a.tmax, a.umin, a.umax, a.mode) == ('L', tjd0.get('per_gpu_batch', 64), 1200, 1600, 120, 200, 'bf16')
            if default_wl and tjd0:
                tjd = tjd0
                roof['traffic'] = round(tjd['hbm_bytes_per_launch'])
                if tjd.get('algorithmic_bytes_per_launch'):
                    roof['traffic_algorithmic'] = round(tjd['algorithmic_bytes_per_launch'])
                    roof['traffic_over_algorithmic'] = round(tjd['hbm_bytes_per_launch'] / tjd['algorithmic_bytes_per_launch'], 3)
                roof['traffic_source'] = 'profiles/pmc_gemm_traffic.json (rocprofv3 PMC passes at %s, bytes per launch; NOT measured in this run)' % tjd.get('revision', 'unknown revision')
                # counter bytes of the other classes (same PMC passes): wasted-traffic ratio where the class has algorithmic bytes
                for ent in roof.get('classes', []):
                    pc = tjd.get('classes', {}).get(ent['class'])
                    if pc is None and ent['class'] == 'conv_frontend_fwd':
                        pc = None      # (the counters cannot separate the forward from the data gradient: see classes_pmc)
                    if pc:
                        ent['traffic_per_step'] = round(pc['hbm_bytes_per_step'])
                        if ent.get('algorithmic_bytes_per_step'):
                            ent['traffic_over_algorithmic'] = round(pc['hbm_bytes_per_step'] / ent['algorithmic_bytes_per_step'], 3)
                if tjd.get('classes'):
                    roof['classes_pmc'] = {k: {kk: vv for kk, vv in v.items() if kk != 'kernels'} for k, v in tjd['classes'].items()}
        out = {
            'encoder_mfu': None,     # (the north-star figure leads the line; filled below)
            'metric': 'speech-frames/sec/node (Conformer-L + CTC+RNN-T, 80-d fbank)',
            'value': round(frames / dt, 1), 'unit': 'frames/s', 'n_gpus': world, 'ranks_seen': (dist.get_world_size() if distributed else 1), 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if a.mode == 'bf16' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'Conformer-%s %dL conv-subsample x8 + CTC(0.3)+RNN-T(2x1024 LSTM, joint 512, V=1000), '
                                   'T~U[%d,%d], U~U[%d,%d], dropout %.2f; full train step (fwd+loss+bwd+clip+Adam)'
                                   % (a.size, a.layers, a.tmin, a.tmax, a.umin, a.umax, a.dropout),
                       'per_gpu_batch': a.batch, 'global_batch': a.batch * world, 'params': n_params,
                       'padded_frames_per_s': round(padded_all / dt, 1),
                       'parallelism': 'dp%d' % world, 'phases': phases},
            'roofline': roof,
        }
        if roof is not None and peaks is not None:
            roof['peaks_measured'] = peaks
            if 'library_gemm_bf16_8192_tflops' in peaks:
                roof['frac_of_measured_library_gemm'] = round(roof['achieved'] / peaks['library_gemm_bf16_8192_tflops'], 4)
            if peaks.get('stream_copy_tb_per_s'):
                # streaming classes beside what a plain device-to-device copy reaches on THIS box (the vendor 8 TB/s is not
                # reachable by any kernel here: the copy measures ~5 TB/s)
                for ent in roof.get('classes', []):
                    if ent.get('bound') == 'hbm':
                        ent['frac_of_measured_copy'] = round(ent['achieved'] / peaks['stream_copy_tb_per_s'], 3)
        if phases is not None and 'encoder_mfu' in phases:
            out['encoder_mfu'] = phases['encoder_mfu']['valid_frames']     # the north-star figure (target 0.40)
        if distributed and comm_log:
            if not emu:
                torch.cuda.synchronize()
            ex = [r.elapsed_time(e) for r, e, _, _, _ in comm_log[a.warmup:]] or [r.elapsed_time(e) for r, e, _, _, _ in comm_log]
            out['comm'] = {'backend': a.dist_backend, 'algorithm': os.environ.get('NSP_DDP_ALGO') or 'all_reduce',
                           'compress': os.environ.get('NSP_DDP_COMPRESS') or None,
                           'buckets_per_step': comm_log[-1][2], 'bytes_per_step': comm_log[-1][3],
                           'wire_bytes_per_step': getattr(comm, 'wire_bytes', None),
                           'side_stream_waits_per_step': comm_log[-1][4],
                           'exposed_ms_per_step': round(sum(ex) / len(ex), 3),
                           'definition': 'main-stream time from "last bucket ready" (all of backward enqueued) to the point '
                                         'behind DDP\'s wait for every collective, mean over the timed steps, rank 0'}
        if emu:
            out['emulated'] = 'CPU tier: kernels on the host emulator (tests/hipemu), gloo ranks -- NOT a measurement'
        if also is not None:
            out['also'] = also
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(a, margs, _usable_cpus())
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
