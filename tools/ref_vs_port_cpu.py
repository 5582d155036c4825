#!/usr/bin/env python
"""The REAL reference (neural_sp.models.seq2seq.speech2text.Speech2Text, imported from /root/reference) and the oracle
port (oracle/model_ref.py) timed on the same host cores, same model (Conformer-L 12L + CTC 0.3 + RNN-T 2x1024, V=1000),
same one-utterance batch of bench.py's workload, full training step (fwd + loss + bwd + clip + Adam), 1 warm-up + N timed
steps.  Build-container only (the GPU box has no /root/reference): bench.py's cpu_baseline is `kind: port`, this states
the port / reference ratio behind that number.  The reference's RNN-T lattice (warprnnt_pytorch, absent) is served by
oracle/rnnt_ref.py in BOTH arms.
usage: python tools/ref_vs_port_cpu.py [threads] [steps]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.set_num_threads(threads)
from oracle import ref_import, model_ref, rnnt_ref
from oracle.gen_golden import install_rnnt_stub
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text as Port

install_rnnt_stub()
ref_import.import_reference()
from neural_sp.models.seq2seq.speech2text import Speech2Text as Ref
import oracle.gen_golden as gg
gg.rnnt_loss_ref = rnnt_ref.rnnt_loss_ref_diag            # the vectorised lattice (same arithmetic), both arms
model_ref.rnnt_loss_ref = rnnt_ref.rnnt_loss_ref_diag
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
batch = synthetic_batch(B=1, t_range=(1600, 1600), u_range=(200, 200), vocab=1000, seed=123)
batch.update(xlens=[len(x) for x in batch['xs']], ys_sub1=[], ys_sub2=[], trigger_points=None, utt_ids=['u0'],
             speakers=['s'], sessions=['x'], text=[''], feat_path=[''])
frames = sum(len(x) for x in batch['xs'])
torch.manual_seed(0)
ref = Ref(margs, None, None)
ref.train()
opt = torch.optim.Adam(ref.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9)


def ref_step():
    t0 = time.time()
    opt.zero_grad(set_to_none=True)
    loss, _ = ref(batch, task='all')
    loss.backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
    opt.step()
    return time.time() - t0, float(loss)


sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'inv_freq' not in k) for k, v in Port(margs).state_dict().items()}
leaves = [v for v in sd.values() if v.requires_grad]
opt2 = torch.optim.Adam(leaves, lr=1e-4, betas=(0.9, 0.98), eps=1e-9)


def port_step():
    t0 = time.time()
    opt2.zero_grad(set_to_none=True)
    loss, _, _, _ = model_ref.speech2text_loss(sd, margs, batch, torch.float32)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(leaves, 5.0)
    opt2.step()
    return time.time() - t0, float(loss)


for name, fn in (('reference', ref_step), ('port', port_step)):
    fn()
    ts = [fn()[0] for _ in range(steps)]
    med = sorted(ts)[len(ts) // 2]
    print('%-9s %d threads: median %.2f s / step (%s) -> %.1f frames/s' % (name, threads, med, ', '.join('%.2f' % t for t in ts), frames / med), flush=True)
