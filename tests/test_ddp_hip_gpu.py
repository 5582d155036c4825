"""N>1 on the REAL HIP model: two ranks (one device, gloo) wrapped by neural_sp_amd.parallel.wrap_ddp
-- prediction network + its backward replay on the side stream, CTC branch on its own stream,
pinned gradient accumulators, the multi-stream comm hook, buckets rebuilt by arrival order -- must
produce the gradient a single process computes on the concatenated batch (train.py:263,423-424:
DDP mean of world x per-rank-mean losses = world x the global-mean gradient for equal shards)."""
import os
import socket
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('compress', [None, 'bf16'])
def test_two_rank_ddp_gradients_equal_single_process(compress):
    import torch.multiprocessing as mp
    from tests import ddp_hip_worker
    world = 2
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'res.pt')
        mp.spawn(ddp_hip_worker.run, args=(world, _free_port(), out, compress), nprocs=world, join=True)
        res = torch.load(out, weights_only=False)
    assert all(res['same']), 'ranks disagree on the reduced gradient'
    assert set(res['ddp']) == set(res['single'])
    tol = 2e-2 if compress == 'bf16' else 1e-3
    gmax = sorted(g.abs().max().item() for g in res['single'].values())
    floor = 1e-4 * gmax[int(0.9 * (len(gmax) - 1))]
    worst, wn = 0.0, ''
    for n, g in res['single'].items():
        want = g * world
        e = ((res['ddp'][n] - want).abs().max() / max(want.abs().max().item(), floor)).item()
        if e > worst:
            worst, wn = e, n
    print('[ddp 2 ranks, compress=%s] worst per-tensor gradient error vs single process: %.2e (%s); losses %s / single %.5f'
          % (compress, worst, wn, res['losses'], res['single_loss']))
    assert worst < tol, (worst, wn)
    # world x mean over ranks of the local means == world x the global mean (equal shard sizes)
    mean_local = sum(l[0] for l in res['losses']) / world / world
    assert abs(mean_local - res['single_loss']) / abs(res['single_loss']) < 1e-3
