"""round 5: does any kernel of the XS transducer step write outside its tensors?  (guard bands around every allocation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
from neural_sp_amd.speech2text import Speech2Text
from tests import ddp_hip_worker as W
from tests import poison
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
    return loss.item()
step()
for mode in ('1', '0'):
    os.environ['NSP_LSTM_PERSISTENT'] = mode
    with poison.guards() as g:
        l = step()
        bad, n = g.check()
    print('NSP_LSTM_PERSISTENT=%s: loss %.6f, %d guarded allocations, %d damaged bands' % (mode, l, n, len(bad)), flush=True)
    for b in bad[:12]:
        print('   ', b, flush=True)
