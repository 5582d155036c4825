#!/usr/bin/env python
"""cProfile of the clip + optimizer phase of a training step."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, parallel
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, fused=True)
batch = synthetic_batch(B=16, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
pr = cProfile.Profile()
tt = [0, 0, 0]
def step(prof):
    loss, _ = model(batch, task='all'); loss.backward()
    torch.cuda.synchronize()
    if prof: pr.enable()
    t0 = time.perf_counter()
    parallel.clip_grad_norm_(params, 5.0)
    t1 = time.perf_counter()
    opt.step()
    t2 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    if prof:
        pr.disable(); tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += t3 - t2
for _ in range(3): step(False)
for _ in range(3): step(True)
print('clip %.2f ms, adam %.2f ms, zero_grad %.2f ms (host, GPU idle at start)' % tuple(x / 3 * 1e3 for x in tt))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:4000])
