"""csrc/norm_subsample.hip on the device (the bodies of tests/variants_common.py with where='gpu'), plus shapes the
host emulator is too slow for, and the model-level parity of the variants against their reference-generated fixtures
(conformer_{bn,gn,drop,add,meanpool,concat,conv1d}_ctc_xs, transformer_glu_ctc_xs) through the functions of
tests/test_golden_gpu.py.

NOTE (round 2): written after the round's GPU minutes were spent -- these tests had only run on the emulator
(tests/test_variants_emu_cpu.py) when they were committed."""
import pytest
import torch

from tests import test_golden_gpu as golden
from tests import variants_common as vc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', golden.VARIANT_CASES)
def test_variant_golden_fp32(name):
    """loss 1e-4, encoder output 2e-4 of max, every gradient 2e-3 of its max against the reference's fixture"""
    golden.test_golden_fp32(name)


@pytest.mark.parametrize('name', golden.VARIANT_CASES)
def test_variant_golden_bf16(name):
    golden.test_golden_bf16(name)


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
