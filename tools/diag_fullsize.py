"""Diagnostic (GPU box): per-tensor gradient agreement at Conformer-L size between the CPU oracle,
the exact-fp32 HIP mode and the bf16 HIP mode (flash attention on / off)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from neural_sp_amd import ops
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    import test_fullsize_parity_gpu as T
    which = sys.argv[1] if len(sys.argv) > 1 else 'random_biases'
    torch.manual_seed(3)
    margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.0, ctc_weight=0.3)
    model = Speech2Text(margs)
    if which == 'random_biases':
        T._randomise_biases(model, 5)
    model.cuda(0)
    batch = synthetic_batch(B=3, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=17)
    runs = {}
    runs['bf16'] = T._hip(model, batch, 'bf16')
    os.environ['NSP_FLASH_ATTN'] = '0'
    runs['bf16_noflash'] = T._hip(model, batch, 'bf16')
    os.environ['NSP_FLASH_ATTN'] = '1'
    runs['f32'] = T._hip(model, batch, 'f32')
    ref = T._oracle(model, margs, batch)
    runs['oracle'] = (ref[0], ref[1], ref[2])
    out = {'loss': {k: v[0] for k, v in runs.items()}, 'tensors': {}}
    rg = runs['oracle'][2]
    for n, r in rg.items():
        r = r.flatten().double()
        ent = {'ref_norm': r.norm().item()}
        for k in ('bf16', 'bf16_noflash', 'f32'):
            a = runs[k][2][n].flatten().double()
            ent[k] = (torch.nn.functional.cosine_similarity(a, r, dim=0).item(), (a.norm() / r.norm()).item())
        out['tensors'][n] = ent
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'diag_fullsize_%s.json' % which), 'w'), indent=0)
    print(out['loss'])
    worst = sorted(out['tensors'].items(), key=lambda kv: kv[1]['bf16'][0])[:40]
    for n, e in worst:
        print('%-52s norm %.3e  bf16 %.4f/%.3f  noflash %.4f/%.3f  f32 %.5f/%.4f' % (
            n, e['ref_norm'], e['bf16'][0], e['bf16'][1], e['bf16_noflash'][0], e['bf16_noflash'][1],
            e['f32'][0], e['f32'][1]))


if __name__ == '__main__':
    main()
