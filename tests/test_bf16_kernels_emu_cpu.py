"""The bf16 (throughput-mode) kernels on the host emulator: the BODIES of the device tests (tests/test_kernels_basic_gpu.py,
tests/test_flash_attn_gpu.py) run unchanged on CPU tensors, their `_dev()` pointed at the CPU and the C-ABI calls routed
to tests/hipemu's build of the same .hip sources.

What this executes on the CPU tier: gemm_bf16.hip (register-staged KC / RC loaders, the LDS-DMA ring and single-stage
kernels with `global_load_lds`, transposed LDS reads `ds_read_b64_tr_b16`, XOR-swizzled tiles, split-K slabs, the
every fused epilogue) and flash_attn.hip (forward, backward, relative
position table, causal / chunk masks, dropout) -- v_mfma_f32_16x16x32_bf16, the transposed read and the LDS-DMA load
are emulated as documented in tests/hipemu/include/hip/hip_runtime.h (an LDS-DMA load lands only at the s_waitcnt that
retires it, so a too-weak counted wait fails these tests as well as a wrong address or lane mapping).  Shapes are the small ones of the device tests
(the large ones take minutes of host time); tolerances are the device tests' own."""
import os

import pytest

_ALL = __import__('os').environ.get('NSP_EMU_ALL', '0') == '1'     # the slow shapes of the parametrisations below
import torch

from tests.hipemu import build_emu

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the HIP emulator')
SLOW = os.environ.get('NSP_EMU_ALL', '0') == '1'     # + 10 min: three more decoders and train.py's call sequence (all pass)


@pytest.fixture
def basic(monkeypatch):
    import tests.test_kernels_basic_gpu as mod
    from tests.hipemu.shim import emulated_kernels
    monkeypatch.setattr(mod, '_dev', lambda: torch.device('cpu'))
    with emulated_kernels():
        yield mod


@pytest.fixture
def flash(monkeypatch):
    import tests.test_flash_attn_gpu as mod
    from tests.hipemu.shim import emulated_kernels
    monkeypatch.setattr(mod, '_dev', lambda: torch.device('cpu'))
    with emulated_kernels():
        yield mod


@pytest.fixture
def convloss(monkeypatch):
    import tests.test_kernels_conv_loss_gpu as mod
    from tests.hipemu.shim import emulated_kernels
    monkeypatch.setattr(mod, '_dev', lambda: torch.device('cpu'))
    with emulated_kernels():
        yield mod


@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (77, 130, 68), (300, 1000, 512)])
def test_gemm_nt_bf16(basic, M, N, K):
    basic.test_gemm_nt('bf16', 2e-2, M, N, K)


def test_gemm_layout_asymmetric(basic):
    basic.test_gemm_layout_asymmetric()


@pytest.mark.parametrize('M,N,K', [(256, 128, 96), (531, 1000, 512)])
def test_gemm_dgrad_wgrad_bf16(basic, M, N, K):
    basic.test_gemm_dgrad_wgrad('bf16', 2e-2, M, N, K)


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (77, 136, 72), (300, 1000, 512)])
def test_gemm_bf16_operands_all_layouts(basic, M, N, K):
    basic.test_gemm_bf16_operands_all_layouts(M, N, K)


def test_splitk_slabs_with_an_empty_split(basic):
    basic.test_splitk_slabs_with_an_empty_split_are_fully_written('bf16')


def test_weight_gradient_gemm_on_the_lds_dma_ring(basic, monkeypatch):
    """(15 s; what weight gradients with an output extent <= 128 run on -- the XS models of the fixtures)"""
    basic.test_weight_gradient_gemm_on_the_lds_dma_ring(monkeypatch, shapes=[(1280, 1000, 264), (512, 128, 384)])


@pytest.mark.parametrize('rows,N,K', ([(1291, 1000, 264)] if _ALL else [(523, 520, 264)]) + [(333, 136, 1280), (260, 256, 256)])   # (57 s / 10 s)
def test_weight_gradient_gemm_on_the_phase_interleaved_kernel(basic, rows, N, K, monkeypatch):
    basic.test_weight_gradient_gemm_on_the_phase_interleaved_kernel(rows, N, K, monkeypatch)


@pytest.mark.parametrize('M,N,K,grid', [(1000, 384, 128, 0), (300, 4352, 256, 8) if _ALL else (300, 1280, 256, 8), (300, 2304, 128, 8)])   # (48 s / 14 s)
def test_phase_interleaved_256_tile_gemm(basic, M, N, K, grid, monkeypatch):
    """(grid 8: 34 resp. 18 tiles on 8 workgroups -- the load-unit stream runs across tile boundaries, with two resp. one
    loop iteration per tile; the emulator lands an LDS-DMA load only at the wait that retires it)"""
    basic.test_phase_interleaved_256_tile_gemm(M, N, K, grid, monkeypatch)


@pytest.mark.parametrize('M,N,K,grid', [(6144, 256, 256, 16)])   # 24 tiles, two workgroups per XCD share the middle one of three
def test_phase_interleaved_gemm_stream_k(basic, M, N, K, grid, monkeypatch):
    basic.test_phase_interleaved_gemm_stream_k(M, N, K, grid, monkeypatch)


def test_layernorm_bf16_only_output_refuses_an_fp32_consumer(basic):
    basic.test_layernorm_bf16_only_output_refuses_an_fp32_consumer()


def test_residual_gradient_prepared_by_the_layer_norm_backward(basic, monkeypatch):
    basic.test_residual_gradient_prepared_by_the_layer_norm_backward(monkeypatch)


@pytest.mark.parametrize('rows,d', [(70, 256)])
def test_layer_norm_swish_for_a_gemm_only_consumer(basic, rows, d):
    basic.test_layer_norm_swish_for_a_gemm_only_consumer(rows, d)


@pytest.mark.parametrize('M,C', [(300, 64), (37, 256)])
def test_linear_glu_on_the_bf16_image(basic, M, C, monkeypatch):
    basic.test_linear_glu_on_the_bf16_image(M, C, monkeypatch)


@pytest.mark.parametrize('rows,d,offer', [(70, 256, True), (33, 64, False)])
def test_layer_norm_pair_at_a_block_boundary(basic, rows, d, offer):
    basic.test_layer_norm_pair_at_a_block_boundary(rows, d, offer)


@pytest.mark.parametrize('B,T,C,k', [(3, 70, 64, 15), (2, 37, 256, 7)])
def test_linear_glu_depthwise_conv_as_one_node(basic, B, T, C, k, monkeypatch):
    basic.test_linear_glu_depthwise_conv_as_one_node(B, T, C, k, monkeypatch)


@pytest.mark.parametrize('T,with_pos,causal,nc', [(130, True, False, 0), (64, False, False, 0), (96, True, False, 16)])
def test_flash_attention_matches_reference(flash, T, with_pos, causal, nc):
    flash.test_flash_attention_matches_reference(T, with_pos, causal, nc)


def test_flash_attention_dropout_mask(flash):
    flash.test_flash_attention_dropout_mask_is_consistent_between_forward_and_backward()


@pytest.mark.parametrize('T,with_pos,pdrop', [(130, True, 0.1), (96, False, 0.0)])
def test_flash_backward_finishes_the_query_gradient(flash, T, with_pos, pdrop):
    flash.test_flash_backward_finishes_the_query_gradient(T, with_pos, pdrop)


@pytest.mark.parametrize('mode', ['bf16', 'bf16maps'])
@pytest.mark.parametrize('Ci', [1, 32])
def test_conv3x3_bf16(convloss, mode, Ci, monkeypatch):
    convloss.test_conv3x3(mode, Ci, monkeypatch)


def test_maxpool2d_incl_bf16_maps(convloss):
    convloss.test_maxpool2d()


@pytest.mark.parametrize('U,J,V', [(6, 32, 32), (40, 64, 40)])
def test_rnnt_joint_loss_bf16_fused_backward(convloss, U, J, V):
    convloss.test_rnnt_joint_loss_bf16_fused_backward(U, J, V)


@pytest.mark.parametrize('B,T,U,J,V', [(3, 21, 6, 32, 29), (4, 37, 40, 64, 43), (2, 9, 0, 64, 130), (3, 30, 9, 128, 130)])
def test_rnnt_joint_loss_fused_compact(convloss, B, T, U, J, V, monkeypatch):
    convloss.test_rnnt_joint_loss_fused_compact_no_logit_tensor(B, T, U, J, V, '0', monkeypatch)


@pytest.mark.parametrize('B,T,U,J,V', [(3, 30, 9, 128, 130), (3, 21, 6, 256, 29)])
def test_rnnt_joint_loss_node_stationary_kernel(convloss, B, T, U, J, V, monkeypatch):
    """nsp_rnnt_joint_rows on the emulator: 3 resp. 1 vocabulary slices (the two-slice ring wraps; V = 29: a slice that
    is mostly padding, lanes without a single valid column), 256-node workgroups with ragged ends"""
    convloss.test_rnnt_joint_loss_fused_compact_no_logit_tensor(B, T, U, J, V, '1', monkeypatch)


@pytest.mark.parametrize('M,J,V,blank', [(1, 128, 70, 69), (300, 256, 130, 0), (40, 512, 70, 3)])
def test_rnnt_joint_rows_matches_tiled_gemm_path(convloss, M, J, V, blank):
    convloss.test_rnnt_joint_rows_matches_tiled_gemm_path(M, J, V, blank)


def test_lstm_vs_torch_bf16(convloss):
    convloss.test_lstm_vs_torch('bf16', 3e-2)


@pytest.mark.parametrize('nl,p_drop,B,L', [(1, 0.0, 5, 9), (2, 0.3, 18, 11)])
def test_lstm_stack_wavefront_vs_torch(convloss, nl, p_drop, B, L):
    convloss.test_lstm_stack_wavefront_vs_torch(nl, p_drop, B, L)


@pytest.mark.parametrize('nl,p_drop,B,L,H', [(2, 0.25, 7, 6, 256)] + ([(3, 0.0, 3, 4, 256)] if _ALL else []))   # (21 s each)
def test_lstm_stack_layer_by_layer_matches_wavefront(convloss, nl, p_drop, B, L, H, monkeypatch):
    """the layer-by-layer orchestration (one recurrence per layer, input projections and their gradients as GEMMs,
    dropout masks by the elementwise kernel) against the (layer, time) wavefront: same masks, same results.  (On the
    emulator both run the per-stage kernels -- a grid barrier needs co-resident workgroups.)"""
    convloss.test_lstm_stack_persistent_matches_wavefront(nl, p_drop, B, L, H, monkeypatch)


@pytest.fixture
def variants(monkeypatch):
    import tests.test_variants_gpu as mod
    from tests.hipemu.shim import emulated_kernels
    monkeypatch.setattr(mod, '_dev', lambda: torch.device('cpu'))
    with emulated_kernels():
        yield mod


def test_lstm_with_initial_and_final_state_bf16(variants):
    """nsp_lstm_{fwd,bwd}_range with the bf16 hidden-state shadows (the chunked latency-controlled BLSTM's kernel path)"""
    variants._lstm_state_case('bf16', 3e-2)


@pytest.mark.parametrize('bidir_sum', [False, True])
def test_blstm_layer_matches_packed_torch_lstm(variants, bidir_sum):
    variants.test_blstm_layer_matches_packed_torch_lstm(bidir_sum)


@pytest.fixture
def dropin(monkeypatch):
    import tests.test_dropin_gpu as mod
    from tests.cpu_ops_shim import host_logic_on_cpu
    monkeypatch.setattr(mod, '_dev', lambda: torch.device('cpu'))
    with host_logic_on_cpu(real_kernels=True):      # = emulated kernels + the pinned-staging stand-in (needs a device)
        yield mod


@pytest.mark.parametrize('name', ['conformer_ctc_xs', 'conformer_rnnt_xs'] + (['conformer_ctc_las_xs', 'conformer_ctc_mocha_xs'] if SLOW else []))
def test_greedy_decode_matches_reference_hypotheses(dropin, name):
    """bit-exact token sequences from the decode kernels (decode.hip) on the emulated encoder"""
    dropin.test_greedy_decode_matches_reference_hypotheses(name)


@pytest.mark.skipif(not SLOW, reason='4 min of host time: NSP_EMU_ALL=1 runs it (passes)')
def test_train_py_call_sequence_in_bf16_mode(dropin):
    """bin/asr/train.py's call sequence (training steps with dropout / SpecAugment / accumulation, dev loss, plot
    hooks, greedy decode) in the bf16 throughput mode"""
    dropin.test_train_py_call_sequence_runs_unchanged()


@pytest.mark.parametrize('rows,d', [(5, 256), (33, 1024), (64, 144)])
def test_layernorm_with_bf16_shadow(basic, rows, d):
    basic.test_layernorm(rows, d)


@pytest.mark.parametrize('clamp,causal,nc', [(10, False, 0), (-1, False, 0), (10, True, 0), (4, False, 8)])
def test_attn_softmax(basic, clamp, causal, nc):
    basic.test_attn_softmax(clamp, causal, nc)


def test_elementwise_and_batched_gemm(basic):
    basic.test_elementwise()
    basic.test_gemm_batched_strided()


@pytest.mark.skipif(not SLOW, reason='minutes of host time (M = 16389 rows x ~20 GEMMs): NSP_EMU_ALL=1 runs it')
def test_specialised_epilogues_on_the_single_stage_kernel(basic):
    basic.test_specialised_epilogues_on_the_single_stage_kernel(16389, 768, 64)
