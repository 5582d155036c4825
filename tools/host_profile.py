#!/usr/bin/env python
"""cProfile of the host side of one training step (what Python spends its ~40 ms on)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
batch = synthetic_batch(B=16, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
def step():
    loss, _ = model(batch, task='all'); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
