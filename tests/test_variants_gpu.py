"""csrc/norm_subsample.hip on the device (the bodies of tests/variants_common.py with where='gpu'), plus shapes the
host emulator is too slow for, and the model-level parity of the variants against their reference-generated fixtures
(conformer_{bn,gn,drop,add,meanpool,concat,conv1d}_ctc_xs, transformer_glu_ctc_xs) through the functions of
tests/test_golden_gpu.py.

NOTE (round 2): written after the round's GPU minutes were (almost) spent.  The kernel tests of this file (window / flip /
group_norm / batch_norm: 32 cases) passed on an MI355X at first contact in the round's last call
(profiles/r02d_variants_kernels_gpu.log); the model-level tests (test_variant_golden_*, config 1 at full size, the BLSTM
layer, weight noise) had only run through the CPU shim / emulator when they were committed -- in BOTH compute modes
since the emulator learned the bf16 kernels (tests/test_e2e_emu_cpu.py, profiles/r02e_bf16_mode_emulated.log)."""
import pytest
import torch

from tests import test_golden_gpu as golden
from tests import variants_common as vc

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


@pytest.mark.parametrize('T,f,kind', vc.SUM_CASES + [(801, 3, 'drop'), (800, 2, 'add'), (799, 4, 'mean_pool')])
def test_window_sum_subsamplers(T, f, kind):
    vc.check_window_sum('gpu', T, f, kind)


@pytest.mark.parametrize('T,k,stride,pad', vc.GATHER_CASES + [(801, 3, 2, 1), (800, 4, 4, 0)])
def test_window_gather_is_im2col(T, k, stride, pad):
    vc.check_window_gather('gpu', T, k, stride, pad)


def test_conv1d_subsampler_as_gather_plus_gemm_weight_view():
    vc.check_conv1d_weight_view('gpu')


@pytest.mark.parametrize('M,C', vc.BN_CASES + [(51200, 512)])     # 64 utterances x 800 frames, Conformer-L width
def test_batch_norm_swish_training_and_eval(M, C):
    vc.check_batch_norm('gpu', M, C)


@pytest.mark.parametrize('M,C', vc.GN_CASES + [(51200, 512)])
def test_group_norm_pairs_swish(M, C):
    vc.check_group_norm('gpu', M, C)


def test_flip_mask_and_bidirectional_merge():
    vc.check_flip_mask_and_merge('gpu')


def test_three_channel_first_conv_as_im2col_gemm():
    vc.check_im2col_conv('gpu')


@pytest.mark.parametrize('rows,klen,w,no_denom,lam', vc.MOCHA_CASES)
def test_mocha_alpha_and_beta_scans(rows, klen, w, no_denom, lam):
    vc.check_mocha_scans('gpu', rows, klen, w, no_denom, lam)


@pytest.mark.parametrize('act,with_loc', [('tanh', True), ('relu', False)])
def test_decoder_step_kernels(act, with_loc):
    vc.check_decoder_step_kernels('gpu', act, with_loc)


def test_all_weight_shadows_refreshed_by_one_launch():
    vc.check_shadow_refresh('gpu')


def test_weight_copies_follow_updates_that_do_not_bump_the_version_counter():
    vc.check_weights_changed_without_version_bump('gpu')


def test_fused_adam_training_matches_plain_adam():
    """torch.optim.Adam(fused=True) leaves Parameter._version alone (measured here, asserted below so that a torch
    upgrade that changes it is noticed): three bf16-mode training steps with it must track the same steps with the
    foreach implementation -- they did not before round 3, when the bf16 weight shadows were keyed by the version
    counter alone and every step after the first multiplied with step 1's weights."""
    import copy
    from neural_sp_amd import ops
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    args = conformer_rnnt_args('XS', n_layers=2, vocab=40, ctc_weight=0.3, ctc_fc_list='32', dropout=0.0)
    batch = synthetic_batch(B=3, t_range=(40, 64), u_range=(2, 6), vocab=40, seed=0)
    torch.manual_seed(0)
    base = Speech2Text(args).to('cuda:0')
    losses = {}
    with ops.compute_mode('bf16'):
        for fused in (True, False):
            model = copy.deepcopy(base)
            opt = torch.optim.Adam(model.parameters(), lr=2e-3, fused=fused)
            v0 = next(model.parameters())._version
            ls = []
            for _ in range(4):
                loss, _ = model(batch, task='all')
                loss.backward()
                opt.step()
                opt.zero_grad(set_to_none=True)
                ls.append(loss.item())
            losses[fused] = ls
            if fused:
                assert next(model.parameters())._version == v0, 'fused Adam now bumps _version: fine, but re-read ops._wkey'
    print('[fused vs foreach Adam, bf16 mode] losses', losses)
    assert losses[True][0] == losses[False][0]
    assert abs(losses[True][3] - losses[True][0]) / abs(losses[True][0]) > 1e-3, 'the loss does not move: stale weights?'
    for a, b in zip(losses[True], losses[False]):
        assert abs(a - b) / abs(b) < 2e-3, losses


@pytest.mark.parametrize('bidir_sum', [False, True])
def test_blstm_layer_matches_packed_torch_lstm(bidir_sum):
    """RNNEncoder._lstm_layer (two left-to-right runs of the LSTM kernels + nsp_time_flip_mask) against
    pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence on the CPU (rnn.py:534-547), fp32 mode:
    outputs and every gradient."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from neural_sp_amd import ops
    from neural_sp_amd.rnn_encoder import RNNEncoder
    torch.manual_seed(11)
    B, T, I, H = 5, 37, 48, 64
    lens = [37, 30, 22, 9, 1]
    enc = RNNEncoder(I, 'blstm', H, 0, 0, 1, 0, 0, 0.0, 0.0, '1', 'drop', 1, 1, None, bidir_sum, False, 0.1, '0', '0', True, 0.0)
    with torch.no_grad():
        for p in enc.parameters():
            p.uniform_(-0.3, 0.3)
    ref = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    ref.load_state_dict(enc.rnn[0].state_dict())
    x = torch.randn(B, T, I)
    xr = x.clone().requires_grad_(True)
    yr = pad_packed_sequence(ref(pack_padded_sequence(xr, lens, batch_first=True))[0], batch_first=True)[0]
    if bidir_sum:
        yr = yr[..., :H] + yr[..., H:]
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    enc.to(_dev())
    with ops.compute_mode('f32'):
        xo = x.clone().to(_dev()).requires_grad_(True)
        yo = enc._lstm_layer(xo, torch.tensor(lens, dtype=torch.int32, device=_dev()), enc.rnn[0])
        (yo * w.to(_dev())).sum().backward()
    torch.testing.assert_close(yo.detach().cpu(), yr.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xo.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-5)
    for (n, p), (_, q) in zip(enc.rnn[0].named_parameters(), ref.named_parameters()):
        assert (p.grad.cpu() - q.grad).abs().max() < 1e-3 * q.grad.abs().max() + 1e-5, n


def _config1_full_size(mode):
    """BASELINE configs[0] at its real size (examples/timit/s5/conf/blstm_ctc.yaml): 5 x 256-unit
    BLSTM layers, CTC, 40-dim features, B=16, T~U[150,500], U~U[20,60], ~64 output symbols -- against
    oracle/model_ref.py (pinned to the reference by the blstm fixtures) in fp32 on the host.
    Gates: fp32 mode loss 1e-4, every gradient 5e-3 of its max; bf16 mode loss 1e-3 (the north-star bar), gradients
    cosine >= 0.99 / norm within 5 % -- wider than the 0.999 / 2 % of the Conformer configurations because the bf16
    hidden-state shadows feed back through up to 500 recurrent steps in each of 5 layers and 2 directions (a single
    layer over 23 steps is held to 3e-2 by test_lstm_vs_torch[bf16]); not yet measured on hardware.  On the emulated
    kernels (tools/emu_config1.py bf16 4 100 200: the same 5 x 256 model, B=4, T~U[100,200]; 20 min of host time) the bf16
    mode gives loss rel 6.4e-6, worst cosine 0.99998, worst norm ratio 0.9999 over 42 tensors."""
    from tests import test_fullsize_parity_gpu as fs
    from neural_sp_amd.configs import blstm_ctc_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(9)
    margs = blstm_ctc_args(n_layers=5, n_units=256, vocab=64)
    model = Speech2Text(margs)
    fs._randomise_biases(model, 7)
    model.cuda(0)
    batch = synthetic_batch(B=16, t_range=(150, 500), u_range=(20, 60), vocab=64, input_dim=40, seed=19)
    loss, obs, grads = fs._hip(model, batch, mode)
    ref, robs, rgrads = fs._oracle(model, margs, batch)
    print('[config 1 %s] loss hip %.6f oracle %.6f rel %.2e' % (mode, loss, ref, abs(loss - ref) / abs(ref)))
    assert set(rgrads) == set(grads), set(rgrads) ^ set(grads)
    if mode == 'f32':
        assert abs(loss - ref) / abs(ref) < 1e-4, (loss, ref)
        err = {n: ((grads[n] - g).abs().max() / g.abs().max().clamp(min=1e-12)).item() for n, g in rgrads.items()}
        bad = {n: e for n, e in err.items() if e > 5e-3}
        assert not bad, bad
    else:
        assert abs(loss - ref) / abs(ref) < 1e-3, (loss, ref)
        bad, worst, skipped, n = fs._compare_grads(grads, rgrads, 0.99, 0.05)
        print('[config 1 bf16] %d tensors, worst (cos, norm ratio) %s, outside the gate: %s' % (n, worst, bad))
        assert not bad, bad


def _lstm_state_case(mode, tol):
    """ops.lstm_state (nsp_lstm_*_range) vs torch's nn.LSTM started from (h0, c0): outputs, final state, and the
    gradients w.r.t. input, initial state and weights when the loss also reads the final state"""
    from neural_sp_amd import ops
    torch.manual_seed(5)
    B, n, I, H = 5, 23, 48, 64
    dev = _dev()
    ref = torch.nn.LSTM(I, H, 1, batch_first=True)
    x = torch.randn(B, n, I, requires_grad=True)
    h0, c0 = torch.randn(B, H, requires_grad=True), torch.randn(B, H, requires_grad=True)
    wy, wh, wc = torch.randn(B, n, H), torch.randn(B, H), torch.randn(B, H)
    yr, (hn, cn) = ref(x, (h0[None], c0[None]))
    params = [ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0]
    gr = torch.autograd.grad((yr * wy).sum() + (hn[0] * wh).sum() + (cn[0] * wc).sum(), [x, h0, c0] + params)
    ins = [t.detach().clone().to(dev).requires_grad_(True) for t in [x, h0, c0] + params]
    with ops.compute_mode(mode):
        y, h, c = ops.lstm_state(ins[0], *ins[3:], ins[1], ins[2])
        g = torch.autograd.grad((y * wy.to(dev)).sum() + (h * wh.to(dev)).sum() + (c * wc.to(dev)).sum(), ins)
    rel = lambda a, r: ((a.cpu() - r).abs().max() / r.abs().max()).item()
    assert rel(y, yr) < tol and rel(h, hn[0]) < tol and rel(c, cn[0]) < tol
    for a, r in zip(g, gr):
        assert rel(a, r) < tol


def test_lstm_with_initial_and_final_state_fp32():
    _lstm_state_case('f32', 1e-4)


def test_weight_noise_on_device():
    """one multi-tensor add on device parameters; bf16 weight shadows follow the version counters"""
    from neural_sp_amd import ops
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.speech2text import Speech2Text
    model = Speech2Text(conformer_rnnt_args('XS', n_layers=2, vocab=40, weight_noise_std=0.01)).cuda(0)
    w = model.enc.layers[0].feed_forward.w_1.weight
    shadow0 = ops.weight_bf16(w).clone()
    before = w.detach().clone()
    torch.manual_seed(3)
    model.add_weight_noise(0.5)
    delta = (w.detach() - before)
    assert delta.abs().max() > 0 and (delta - delta.flatten()[0]).abs().max() < 1e-6
    assert not torch.equal(ops.weight_bf16(w), shadow0)


# ---- model level, in the order "what the emulator has verified" -> "what rests on a simulation": a first-contact
# failure under `pytest -x` then hides as little as possible
@pytest.mark.parametrize('name', golden.VARIANT_CASES)
def test_variant_golden_fp32(name):
    """loss 1e-4, encoder output 2e-4 of max, every gradient 2e-3 of its max against the reference's fixture"""
    golden.test_golden_fp32(name)



def test_timit_blstm_ctc_config1_full_size_fp32():
    _config1_full_size('f32')


@pytest.mark.parametrize('name', golden.VARIANT_CASES)
def test_variant_golden_bf16(name):
    golden.test_golden_bf16(name)



def test_timit_blstm_ctc_config1_full_size_bf16():
    _config1_full_size('bf16')


def test_lstm_with_initial_and_final_state_bf16():
    _lstm_state_case('bf16', 3e-2)
