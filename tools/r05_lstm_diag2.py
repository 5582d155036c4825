"""round 5: where do the encoder's gradients start to depend on the LSTM launch mode?"""
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
grabbed = {}
def grab(name):
    def h(g):
        grabbed[name] = g.detach().clone()
    return h
real_linear = ops.linear
def enc_hook(mod, inp, out):
    xs = out['ys']['xs']
    if xs.requires_grad:
        xs.register_hook(grab('d_eouts'))
    return None
model.enc.register_forward_hook(enc_hook)
def mk_layer_hook(i):
    def h(m, inp, out):
        t = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(t) and t.requires_grad:
            t.register_hook(grab('d_layer%d_out' % i))
        return None
    return h
for i, layer in enumerate(model.enc.layers):
    layer.register_forward_hook(mk_layer_hook(i))
def step():
    grabbed.clear()
    model.zero_grad(set_to_none=True)
    loss, _ = model(batch, task='all')
    loss.backward()
    torch.cuda.synchronize()
    ops.lstm_check()
    g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    return loss.item(), g, dict(grabbed)
def cmp(tag, a, b):
    rows = sorted(((((a[1][n] - b[1][n]).abs().max() / b[1][n].abs().max().clamp_min(1e-30)).item(), n) for n in a[1]), reverse=True)
    inter = {k: ((a[2][k] - b[2][k]).abs().max() / b[2][k].abs().max().clamp_min(1e-30)).item() for k in a[2] if k in b[2]}
    print('%-40s worst param %s %.1e | intermediates %s' % (tag, rows[0][1], rows[0][0], {k: '%.1e' % v for k, v in inter.items()}), flush=True)
for env in ({}, {'NSP_LN_PREP': '0'}, {'NSP_PREDNET_PRIORITY': '0'}, {'NSP_REPLAY_SIDE': '0'}, {'NSP_CTC_STREAM': '0', 'NSP_PREDNET_STREAM': '0'}):
    for k, v in env.items():
        os.environ[k] = v
    os.environ['NSP_LSTM_PERSISTENT'] = '1'
    A = step()
    os.environ['NSP_LSTM_PERSISTENT'] = '0'
    B = step()
    cmp('per-stage vs persistent, env %s' % env, B, A)
    for k in env:
        del os.environ[k]
