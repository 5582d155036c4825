"""BASELINE configs[0] (TIMIT BLSTM-CTC, 5 x 256 units) in either compute mode on the HOST EMULATOR of the HIP kernels
(tests/hipemu), against oracle/model_ref.py -- the CPU-tier stand-in for tests/test_variants_gpu.py's
test_timit_blstm_ctc_config1_full_size_{fp32,bf16} when no device is at hand.  Development tool (imports tests/ and
oracle/): python tools/emu_config1.py bf16 [B] [Tmin] [Tmax]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    t_range = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (150, 500)
    from tests import test_fullsize_parity_gpu as fs
    from tests.cpu_ops_shim import host_logic_on_cpu
    from neural_sp_amd.configs import blstm_ctc_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(9)
    margs = blstm_ctc_args(n_layers=5, n_units=256, vocab=64)
    model = Speech2Text(margs)
    fs._randomise_biases(model, 7)
    batch = synthetic_batch(B=B, t_range=t_range, u_range=(20, min(60, t_range[0] // 3)), vocab=64, input_dim=40, seed=19)
    t0 = time.time()
    with host_logic_on_cpu(real_kernels=True, mode=mode):
        loss, obs = model(batch, task='all')
        loss.backward()
    t1 = time.time()
    grads = {n: p.grad.detach() for n, p in model.named_parameters() if p.grad is not None}
    ref, robs, rgrads = fs._oracle(model, margs, batch)
    print('[config 1 %s on the emulator, B=%d T~U%s] loss hip %.6f oracle %.6f rel %.2e (%.0f s emulated, %.0f s oracle)' % (
        mode, B, t_range, loss.item(), ref, abs(loss.item() - ref) / abs(ref), t1 - t0, time.time() - t1))
    bad, worst, skipped, n = fs._compare_grads(grads, rgrads, 0.99, 0.05)
    print('%d tensors, worst (cos, norm ratio) %s, outside the (0.99, 5 %%) gate: %s' % (n, worst, bad))
    err = max(((grads[k] - g).abs().max() / g.abs().max().clamp(min=1e-12)).item() for k, g in rgrads.items())
    print('worst max-error / max over tensors: %.2e' % err)


if __name__ == '__main__':
    main()
