#!/usr/bin/env python
"""The 16-utterance step (SURVEY 8d's per-GPU batch) taken apart on the host side: (1) wall time of a step with the
device kept busy (steady state) and with the queue drained before every step (host enqueue time alone), (2) cProfile
of the host over a few steps (where the enqueue time goes), (3) launches per step by kernel and by the python line of
neural_sp_amd that caused them (torch.profiler, CPU-side op events joined with their kernels).
    python tools/b16_census.py [batch]"""
import cProfile, io, os, pstats, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
from neural_sp_amd import parallel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
    return obs


for i in range(6):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12):
    step(i)
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / 12
host = []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(i)
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print('[b16 census] batch %d: steady state %.2f ms/step; host enqueue alone (queue drained first) %.2f ms/step (min %.2f)'
      % (B, steady * 1e3, sum(host) / len(host) * 1e3, min(host) * 1e3))

pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(4):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print('[b16 census] cProfile over 4 steps, by own time:')
print('\n'.join(l for l in s.getvalue().splitlines()[4:] if l.strip()))

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
by_kernel = collections.Counter()
by_site = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        continue
    ks = getattr(ev, 'kernels', None) or []
    if not ks:
        continue
    site = 'autograd / torch'
    for fr in (ev.stack or []):
        if 'neural_sp_amd' in fr:
            site = fr.strip().split('neural_sp_amd/')[-1][:70]
            break
    for k in ks:
        by_kernel[k.name[:70]] += 1
        if ev.name.startswith('aten::'):
            by_site[(ev.name, site)] += 1
n = sum(by_kernel.values())
print('[b16 census] %d device launches from ATen ops over 2 steps (ctypes launches of libnsp_hip.so are not ATen ops and not in this list)' % n)
for (nm, site), c in by_site.most_common(40):
    print('  %5.1f/step  %-28s %s' % (c / 2, nm, site))
