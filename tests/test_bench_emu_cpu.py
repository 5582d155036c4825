"""bench.py end to end on the CPU tier: `python bench.py --gpus 2 --same-device --dist-backend gloo
--emulate ...` goes through the file's own launcher (re-execution under torch.distributed.run on 127.0.0.1), rank binding,
DDP wrapping with the package's communication hook, the step loop (forward, loss x world, backward with the bucketed
reduction, clip, Adam), the barrier / max-over-ranks / sum-of-frames reductions and the JSON line -- with the kernels of
libnsp_hip.so on the host emulator (tests/hipemu) and gloo between two CPU ranks.  What a node with N > 1 GPUs adds on top
is RCCL itself; everything else of `bench.py --gpus N` has run here.  Its numbers mean nothing (the line says so)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--emulate', '--size', 'XS', '--layers', '2', '--batch', '2', '--tmin', '90', '--tmax', '120', '--umin', '3',
         '--umax', '6', '--steps', '2', '--warmup', '1']


def _bench(args, env=None, timeout=900):
    from tests.hipemu import build_emu
    if not build_emu.available():
        pytest.skip('no host clang++ for the HIP emulator')
    build_emu.build()                       # once, before the ranks start
    e = dict(os.environ, OMP_NUM_THREADS='2')
    e.pop('WORLD_SIZE', None); e.pop('RANK', None); e.pop('LOCAL_RANK', None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _line(res):
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize('algo,compress', [('all_reduce', ''), ('rs_ag', 'bf16')])
def test_bench_two_gloo_ranks_on_emulated_kernels(algo, compress):
    """(rs_ag x bf16: the reduce-scatter + all-gather form of the hook with compressed buckets, on the product model)"""
    d = _line(_bench(['--gpus', '2', '--same-device', '--dist-backend', 'gloo'] + SMALL,
                     env={'NSP_DDP_ALGO': algo, 'NSP_DDP_COMPRESS': compress}))
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['steps'] == 2 and d['warmup'] == 1
    assert d['scaling'] == 'weak' and d['higher_is_better'] is True and d['unit'] == 'frames/s' and d['value'] > 0
    assert d['config']['per_gpu_batch'] == 2 and d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'dp2'
    assert 'emulated' in d and d['vs_baseline'] is None
    c = d['comm']
    assert c['backend'] == 'gloo' and c['algorithm'] == algo and c['buckets_per_step'] >= 1
    assert c['bytes_per_step'] == 4 * d['config']['params']          # every fp32 gradient crossed the hook once
    assert c['compress'] == (compress or None) and c['exposed_ms_per_step'] >= 0
    if compress:
        assert c['bytes_per_step'] // 2 <= c['wire_bytes_per_step'] <= c['bytes_per_step'] // 2 + 4 * c['buckets_per_step']
    else:
        assert c['wire_bytes_per_step'] == c['bytes_per_step']


def test_bench_single_rank_on_emulated_kernels_and_refusals():
    d = _line(_bench(['--gpus', '1'] + SMALL))
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and 'comm' not in d and d['value'] > 0
    # a launcher that started another number of ranks than --gpus: no line
    r = _bench(['--gpus', '2', '--dist-backend', 'gloo'] + SMALL, env={'WORLD_SIZE': '3', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'refusing to print a line' in (r.stderr + r.stdout)
    # CPU ranks over the device backend: refused before anything runs
    r = _bench(['--gpus', '2', '--dist-backend', 'nccl'] + SMALL, env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'gloo' in (r.stderr + r.stdout)
