"""world_size=2 gloo test of the data-parallel plumbing (neural_sp_amd/parallel.py) on CPU.

Two tiers: a small stand-in module exercises exactly the logic the N>1 path adds around the model (DDP wrapping with
our settings, rank-strided batch sharding, the train.py loss pre-scaling, bench.py's max-time / sum-units
aggregation); and the PRODUCT model itself runs as two gloo ranks on the host-emulated HIP kernels (the worker of the
device test tests/test_ddp_hip_gpu.py with device='cpu') against the single-process gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, algorithm=None):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from neural_sp_amd import parallel
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 16), nn.Tanh(), nn.Linear(16, 1))
    ddp = parallel.wrap_ddp(model, None)
    if algorithm is not None:      # the package's communication hook (CPU branch) with the chosen collective; 161 parameters: an
        ddp.register_comm_hook(None, parallel.make_comm_hook([], algorithm=algorithm))      # odd bucket -> the padded path
    g = torch.Generator().manual_seed(1)
    X = torch.randn(12, 8, generator=g)
    Y = torch.randn(12, 1, generator=g)
    idx = parallel.shard_batch(list(range(12)), rank, world)
    # per-rank loss normalised by the GLOBAL batch (as Speech2Text's per-rank batch-mean is after
    # the sampler multiplies the batch by num_replicas), then the train.py pre-scaling
    loss = ((ddp(X[idx]) - Y[idx]) ** 2).sum() / 12
    parallel.scale_loss_for_ddp(loss, world).backward()
    grads = [p.grad.tolist() for p in model.parameters()]
    dt, units = parallel.aggregate_timing(1.0 + rank, 100 * (rank + 1))
    q.put((rank, grads, dt, units))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize('algorithm', [None, 'all_reduce', 'rs_ag'])
def test_ddp_gloo_world2_matches_single_process(algorithm):
    """algorithm None: torch DDP's own all-reduce; 'all_reduce' / 'rs_ag': parallel.make_comm_hook with an all-reduce or
    with reduce-scatter + all-gather -- all three must give the single-process gradient"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, algorithm)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the whole batch
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 16), nn.Tanh(), nn.Linear(16, 1))
    g = torch.Generator().manual_seed(1)
    X = torch.randn(12, 8, generator=g)
    Y = torch.randn(12, 1, generator=g)
    (((model(X) - Y) ** 2).sum() / 12).backward()
    ref = [p.grad for p in model.parameters()]
    for rank, grads, dt, units in res:
        for a, b in zip(grads, ref):
            assert torch.allclose(torch.tensor(a), b, atol=1e-6), rank
        assert dt == 2.0 and units == 300.0


def test_two_rank_ddp_on_the_real_model_with_emulated_kernels():
    """The N>1 path with the PRODUCT model: tests/ddp_hip_worker.py -- the worker of the device test
    tests/test_ddp_hip_gpu.py -- with device='cpu': two gloo ranks, each running neural_sp_amd.Speech2Text (Conformer
    with d_k = 64 fused attention, CTC + RNN-T with its 2-layer prediction network) in bf16 mode on the host-emulated HIP
    kernels, gradients all-reduced by parallel.make_comm_hook in 50 kB buckets (rebuilt by arrival order in the second
    iteration), against the single-process gradient on the concatenated batch (train.py:263,423-424)."""
    import tempfile

    import pytest
    from tests.hipemu import build_emu
    if not build_emu.available():
        pytest.skip('no host clang++ for the HIP emulator')
    build_emu.build()                       # once, before the ranks start
    from tests import ddp_hip_worker
    world = 2
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'res.pt')
        mp.spawn(ddp_hip_worker.run, args=(world, _free_port(), out, None, 'cpu'), nprocs=world, join=True)
        res = torch.load(out, weights_only=False)
    assert all(res['same']), 'ranks disagree on the reduced gradient'
    assert set(res['ddp']) == set(res['single'])
    gmax = sorted(g.abs().max().item() for g in res['single'].values())
    floor = 1e-4 * gmax[int(0.9 * (len(gmax) - 1))]
    worst, wn = 0.0, ''
    for n, g in res['single'].items():
        want = g * world
        e = ((res['ddp'][n] - want).abs().max() / max(want.abs().max().item(), floor)).item()
        if e > worst:
            worst, wn = e, n
    print('[ddp 2 ranks on emulated kernels] worst per-tensor gradient error vs single process: %.2e (%s)' % (worst, wn))
    assert worst < 1e-3, (worst, wn)          # the device test's gate
    mean_local = sum(l[0] for l in res['losses']) / world / world
    assert abs(mean_local - res['single_loss']) / abs(res['single_loss']) < 1e-3
