"""End-to-end parity of neural_sp_amd.Speech2Text (HIP kernels) against fixtures produced by
the REFERENCE model on CPU (oracle/gen_golden.py): same state_dict, same batch ->
loss, encoder output and every parameter gradient.

Tolerances (stated, fp32): loss 1e-4 relative (north-star bar is 1e-3), encoder output
2e-4 of its max magnitude, gradients 2e-3 of each tensor's max magnitude in the exact
fp32-MFMA mode.  In bf16-MFMA mode (the throughput mode): loss 1e-3 relative (the north-star
bar; full-size bf16 parity incl. gradient norms is in tests/test_fullsize_parity_gpu.py), gradients
compared by cosine similarity >= 0.99 on these XS models (tensors of a few dozen elements).

The floor of the per-tensor gradient scale is taken from the 90th percentile of the per-tensor
maxima, not from the largest one: the zero-bias fixture has ONE tensor (enc.conv.bridge.bias)
whose gradient is amplified 1e6x by the eps=1e-12 zero-variance LayerNorm rows."""
import argparse
import glob
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# The encoder-variant fixtures (round 2, committed without GPU time) run through the same two functions below from
# tests/test_variants_gpu.py -- the LAST file of the suite, so that under `pytest -x` a first-contact failure in them
# cannot hide the results of everything that has already been measured on hardware.
VARIANT_CASES = ['conformer_bn_ctc_xs', 'conformer_gn_ctc_xs', 'transformer_glu_ctc_xs', 'conformer_drop_ctc_xs',
                 'conformer_add_ctc_xs', 'conformer_meanpool_ctc_xs', 'conformer_concat_ctc_xs', 'conformer_conv1d_ctc_xs',
                 'conformer_2mtl_ctc_xs', 'transformer_3mtl_att_xs', 'blstm_ctc_xs', 'conv_blstm_proj_drop_xs',
                 'conformer_ctc_las_ss_xs', 'conv_blstm_fullcontext_xs',
                 'conformer_ctc_att_1dconv_xs', 'conformer_ctc_mocha_stableemit_xs', 'conformer_ctc_mocha_ctcsync_xs',
                 'conformer_ctc_mocha_decot_xs', 'conv_lcblstm_chunk_xs', 'transformer_ctc_3ch_xs', 'conformer_ctc_mma_xs', 'conformer_ctc_mma_headdrop_xs',
                 'conformer_ctc_triggered_xs']
CASES = sorted(set(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*_xs.pt'))) - set(VARIANT_CASES))


# bf16-mode loss gate: the north-star bar (1e-3) everywhere, except where rounding the GEMM operands to bf16 ALONE
# (tools/bf16_sensitivity.py: the fp64 oracle with bf16-rounded Linear operands, no GPU involved) already moves the
# fixture's loss by more: ConcatSubsampler's un-normalised ReLU(Linear(3d -> d)) with torch's default init gives
# 1.2e-3 .. 1.5e-3 for every seed and batch size tried.
BF16_LOSS_GATE = {'conformer_concat_ctc_xs': 3e-3}     # (measured on the emulated kernels: 1.09e-3)
# bf16-mode cosine gate per tensor family: 0.99 by default.  In conformer_ctc_mma_headdrop_xs HeadDrop removes BOTH
# monotonic heads of decoder layer 2, so that layer's energies receive gradient only from the quantity loss
# |sum_j alpha_ij - U| (0.1 .. 0.6 % of the typical gradient size), whose derivative is a SIGN: one (utterance, head)
# whose expected count is within bf16 rounding of U flips it.  The fp64 oracle with bf16-rounded GEMM operands and
# gradients (no kernels involved) gives cosine 0.952 .. 0.979 and a norm ratio ~0.8 on exactly these tensors; the
# emulated kernels 0.974 .. 0.982.
# First run on an MI355X (round 3, profiles/r03a_pytest_gpu_full.log): 0.937 / 0.940 on two of the six tensors (loss rel 8e-6,
# all other 146 tensors >= 0.99) -- a third outcome of the same coin: which near-zero count flips depends on the summation
# order of the run.  The gate states that spread (0.90); it is a property of sign() at ~0, not of the kernels.
BF16_COS_GATE = {'conformer_ctc_mma_headdrop_xs': (('dec_fwd.layers.2.norm2.', 'dec_fwd.layers.2.src_attn.monotonic_energy.'), 0.90)}
# fp32-mode gradient gate (fraction of each tensor's max): 2e-3 for every fixture.  Fixtures may list a wider one here when
# the REFERENCE's own fp32 gradients (the fixture) sit far from the fp64 oracle (tools/bf16_sensitivity.py, last column)
# because fp32 rounding decided a max-pool arg-max / ReLU mask in the front-end (tools/fixture_tie_check.py); the
# variant fixtures were re-seeded until that distance was <= 6e-5, so none needs it today.
FP32_GRAD_GATE = {}


def _dev():
    return torch.device('cuda', 0)      # (tests/test_e2e_emu_cpu.py points this at the CPU and runs the same bodies on the emulator)


def _load(name):
    return torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)


def _run(fix, mode):
    from neural_sp_amd import ops
    from neural_sp_amd.speech2text import Speech2Text
    args = argparse.Namespace(**fix['args'])
    model = Speech2Text(args)
    model.load_state_dict(fix['state_dict'], strict=True)
    model.to(_dev())
    if fix['meta'].get('trigger_quantity_loss'):
        model.trigger_quantity_loss()        # train.py's curriculum switch (MoChA quantity loss)
    ss_seed = fix['meta'].get('scheduled_sampling_seed')
    if fix['meta'].get('trigger_stableemit'):
        model.trigger_stableemit()           # train.py's mocha_stableemit_start_epoch switch
    if ss_seed is not None:
        model.trigger_scheduled_sampling()   # train.py's ss_start_epoch switch; Python's `random` decides per step
    batch = dict(fix['batch'])
    batch.update(xlens=[len(x) for x in batch['xs']])
    batch.setdefault('trigger_points', None)   # reference boundaries that come with the batch (DeCoT / MinLT fixtures)
    batch.setdefault('ys_sub1', [])        # auxiliary-task transcripts (multi-task fixtures only)
    batch.setdefault('ys_sub2', [])
    with ops.compute_mode(mode):
        model.zero_grad()
        if ss_seed is not None:
            random.seed(ss_seed)
        loss, obs = model(batch, task='all')
        loss.backward()
        grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
        model.eval()
        with torch.no_grad():
            eout = model.encode(batch['xs'], 'all')
            if ss_seed is not None:
                random.seed(ss_seed)         # the reference samples in eval-mode forwards too (las.py:668)
            loss_eval, _ = model(batch, task='all', is_eval=True)
    return loss.item(), obs, eout['ys']['xs'].cpu(), eout['ys']['xlens'], grads, loss_eval.item()


def _robust_gmax(grads):
    m = sorted(g.abs().max().item() for g in grads.values())
    return m[int(0.9 * (len(m) - 1))]


@pytest.mark.parametrize('name', CASES)
def test_golden_fp32(name):
    fix = _load(name)
    loss, obs, eout, elens, grads, loss_eval = _run(fix, 'f32')
    ref = fix['loss'].item()
    assert abs(loss - ref) / abs(ref) < 1e-4, (loss, ref)
    assert abs(loss_eval - fix['loss_eval'].item()) / abs(ref) < 1e-4
    for k, v in fix['observation'].items():
        if v is not None:
            assert abs(obs[k] - v) <= 1e-4 * abs(v) + 1e-6, (k, obs[k], v)
    assert torch.equal(elens.int(), fix['elens'].int())
    assert eout.shape == fix['eout'].shape
    assert (eout - fix['eout']).abs().max() / fix['eout'].abs().max() < 2e-4
    assert set(grads) == set(fix['grads'])
    # tensors whose true gradient is zero (e.g. w_key.bias: softmax is shift-invariant) hold
    # only rounding noise in the reference, so the floor of the scale is 1e-5 of the largest grad
    gmax = _robust_gmax(fix['grads'])
    err = {n: ((grads[n] - g).abs().max() / max(g.abs().max().item(), 1e-5 * gmax)).item()
           for n, g in fix['grads'].items()}
    if fix['args'].get('conformer_normalization') == 'batch_norm':
        # BatchNorm removes a per-channel shift, so the depthwise-conv bias has a true gradient of ZERO: both sides
        # hold only the rounding noise of a sum over all frames (reference: 6e-6 against tensor maxima of ~40)
        for n in [n for n in err if n.endswith('.conv.depthwise_conv.bias')]:
            assert grads[n].abs().max() < 1e-4 * gmax and fix['grads'][n].abs().max() < 1e-4 * gmax, n
            del err[n]
    bad = {n: e for n, e in err.items() if e > FP32_GRAD_GATE.get(name, 2e-3)}
    assert not bad, (max(err.values()), bad)


def assert_bf16_gates(name, fix, loss, grads, tag='golden bf16'):
    """the bf16-mode gates, shared with the CPU tier (tests/test_e2e_emu_cpu.py runs the same kernels on the emulator)"""
    ref = fix['loss'].item()
    cos = {}
    gmax = _robust_gmax(fix['grads'])
    for n, g in fix['grads'].items():
        if g.numel() < 16 or g.abs().max() < 1e-5 * gmax:
            continue
        cos[n] = torch.nn.functional.cosine_similarity(grads[n].flatten(), g.flatten(), dim=0).item()
    # MoChA's chunk-energy projections: beta is a softmax inside 4-frame windows, so their gradient is a sum of
    # differences of nearly equal neighbouring terms (p_i (delta_ij - p_j)); the 2^-9 rounding of the bf16
    # encoder output that feeds the energies is amplified in it.  Stated gate 0.97 (measured 0.974-0.984; the
    # decoder recurrence itself runs in fp32, and the fp32-mode test holds these tensors to 2e-3 of max).
    fam, fam_gate = BF16_COS_GATE.get(name, ((), 0.99))

    def gate(n):
        if '.score.chunk_energy.' in n:
            return 0.97
        return fam_gate if n.startswith(fam) and fam else 0.99
    bad = {n: c for n, c in cos.items() if c < gate(n)}
    print('[' + tag + ' %s] loss rel %.2e, min cosine %.5f over %d tensors' % (
        name, abs(loss - ref) / abs(ref), min(cos.values()), len(cos)))
    assert abs(loss - ref) / abs(ref) < BF16_LOSS_GATE.get(name, 1e-3), (loss, ref)
    assert not bad, bad
    return abs(loss - ref) / abs(ref), min(cos.values())


@pytest.mark.parametrize('name', CASES)
def test_golden_bf16(name):
    fix = _load(name)
    loss, obs, eout, elens, grads, _ = _run(fix, 'bf16')
    assert_bf16_gates(name, fix, loss, grads)
