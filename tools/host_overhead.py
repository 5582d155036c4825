#!/usr/bin/env python
"""How long does the host take to ENQUEUE one training step (no sync) vs. the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
batch = synthetic_batch(B=16, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
def step():
    loss, _ = model(batch, task='all'); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('enqueue %.1f ms, total %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# phases
torch.cuda.synchronize(); t0 = time.perf_counter(); loss, _ = model(batch, task='all'); torch.cuda.synchronize(); t1 = time.perf_counter()
loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step(); opt.zero_grad(set_to_none=True); torch.cuda.synchronize(); t3 = time.perf_counter()
print('fwd %.1f  bwd %.1f  clip+adam %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
