#!/usr/bin/env python
"""Batch 16 (the per-GPU batch SURVEY 8(d) names): is the step host- or device-bound?  Enqueue time of one step (no sync)
against its wall time, main-stream busy time from the profiler-free side: sum of HIP-event-timed phases, and the same with
the loss read one step late (as bench.py does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, parallel
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
B = int(os.environ.get('PB', '16'))
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
torch.manual_seed(1)
model = Speech2Text(margs).cuda(0)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, fused=True)
batches = [synthetic_batch(B=B, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=7000 + i) for i in range(4)]
def step(i):
    loss, obs = model(batches[i % 4], task='all'); loss.backward()
    parallel.clip_grad_norm_(params, 5.0); opt.step(); opt.zero_grad(set_to_none=True)
for i in range(4): step(i)
torch.cuda.synchronize()
enq, tot = [], []
for i in range(8):
    t0 = time.perf_counter(); step(i); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print('B=%d  one step at a time: enqueue %.1f ms (min %.1f), until idle %.1f ms (min %.1f)' % (B, sum(enq) / 8, min(enq), sum(tot) / 8, min(tot)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(12): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('B=%d  12 steps back to back: host done after %.1f ms/step, device after %.1f ms/step' % (B, (t1 - t0) / 12 * 1e3, (t2 - t0) / 12 * 1e3))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for i in range(4): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
