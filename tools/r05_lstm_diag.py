"""round 5: do the encoder's gradients depend on HOW the prediction network's LSTM backward is launched?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
from neural_sp_amd.speech2text import Speech2Text
from tests import ddp_hip_worker as W
ops.set_compute_mode('bf16')
args = W.model_args(small=False)
torch.manual_seed(7)
model = Speech2Text(args).cuda(0)
batch = W.sub_batch(W.global_batch(args.vocab), [1, 3])
def step():
    model.zero_grad(set_to_none=True)
    loss, _ = model(batch, task='all')
    loss.backward()
    torch.cuda.synchronize()
    ops.lstm_check()
    return loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
def cmp(tag, a, b):
    rows = sorted(((((a[1][n] - b[1][n]).abs().max() / b[1][n].abs().max().clamp_min(1e-30)).item(), n) for n in a[1]), reverse=True)
    print('%-28s loss %.7f vs %.7f | worst: %s' % (tag, a[0], b[0], ', '.join('%s %.1e' % (n, v) for v, n in rows[:4])), flush=True)
os.environ['NSP_LSTM_PERSISTENT'] = '1'
A = step(); A2 = step()
cmp('persistent rerun', A2, A)
os.environ['NSP_LSTM_PERSISTENT'] = '0'
B = step(); B2 = step()
cmp('per-stage rerun', B2, B)
cmp('per-stage vs persistent', B, A)
os.environ['NSP_LSTM_PERSISTENT'] = '1'
C = step()
cmp('persistent again vs first', C, A)
os.environ['NSP_LSTM_TEST_FAKE_TIMEOUT'] = '1'
D = step()
cmp('fake-timeout step vs persistent', D, A)
cmp('fake-timeout step vs per-stage', D, B)
for mode in ('NSP_PREDNET_STREAM', 'NSP_CTC_STREAM'):
    os.environ['NSP_LSTM_PERSISTENT'] = '0'
    os.environ[mode] = '0'
    E = step()
    cmp('per-stage, %s=0 vs persistent' % mode, E, A)
    os.environ[mode] = '1'
