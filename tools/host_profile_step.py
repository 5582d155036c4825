#!/usr/bin/env python
"""cProfile of the host side of one training step at a small batch, backward INCLUDED: autograd's worker threads are
switched off (torch.autograd.set_multithreading_enabled(False)) so that the backward functions run -- and are profiled --
in the calling thread.      python tools/host_profile_step.py [batch] [top]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, parallel
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device('cuda:0')
ops.set_compute_mode('bf16')
torch.manual_seed(1)
model = Speech2Text(conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)).to(dev)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, fused=True)
batches = [synthetic_batch(B=B, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=i) for i in range(4)]


def step(i):
    loss, obs = model(batches[i % 4], task='all')
    loss.backward()
    parallel.clip_grad_norm_(params, 5.0)
    opt.step()
    opt.zero_grad(set_to_none=True)


for i in range(5):
    step(i)
torch.cuda.synchronize()
with torch.autograd.set_multithreading_enabled(False):
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    host = []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print('[host profile] batch %d, single-threaded autograd: host enqueue %.2f ms/step (min %.2f)' % (B, sum(host) / len(host) * 1e3, min(host) * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(4):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(top)
    print('[host profile] cProfile over 4 steps, by %s:' % key)
    print('\n'.join(l[:170] for l in s.getvalue().splitlines()[4:] if l.strip()))
