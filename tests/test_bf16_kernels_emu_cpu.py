"""The bf16 (throughput-mode) kernels on the host emulator: the BODIES of the device tests (tests/test_kernels_basic_gpu.py,
tests/test_flash_attn_gpu.py) run unchanged on CPU tensors, their `_dev()` pointed at the CPU and the C-ABI calls routed
to tests/hipemu's build of the same .hip sources.

What this executes on the CPU tier: gemm_bf16.hip (register-staged KC / RC loaders, the LDS-DMA ring and single-stage
kernels with `global_load_lds`, transposed LDS reads `ds_read_b64_tr_b16`, XOR-swizzled tiles, split-K slabs, the
persistent kernel with its deferred epilogue, every fused epilogue) and flash_attn.hip (forward, backward, relative
position table, causal / chunk masks, dropout) -- v_mfma_f32_16x16x32_bf16, the transposed read and the LDS-DMA load
are emulated as documented in tests/hipemu/include/hip/hip_runtime.h (the DMA executes synchronously: a missing
s_waitcnt cannot be detected here, a wrong address or lane mapping is).  Shapes are the small ones of the device tests
(the large ones take minutes of host time); tolerances are the device tests' own."""
import pytest
import torch

from tests.hipemu import build_emu

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the HIP emulator')


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


@pytest.mark.parametrize('stages', ['2', '3'])
def test_weight_gradient_gemm_on_the_lds_dma_ring(basic, stages, monkeypatch):
    basic.test_weight_gradient_gemm_on_the_lds_dma_ring(stages, monkeypatch, shapes=[(1280, 1000, 264)])


@pytest.mark.parametrize('M,N,K', [(1000, 384, 128), (700, 2048, 64)])
def test_persistent_gemm_with_deferred_epilogue(basic, M, N, K, monkeypatch):
    basic.test_persistent_gemm_with_deferred_epilogue(M, N, K, monkeypatch)


@pytest.mark.parametrize('T,with_pos,causal,nc', [(130, True, False, 0), (64, False, False, 0), (96, True, False, 16)])
def test_flash_attention_matches_reference(flash, T, with_pos, causal, nc):
    flash.test_flash_attention_matches_reference(T, with_pos, causal, nc)


def test_flash_attention_dropout_mask(flash):
    flash.test_flash_attention_dropout_mask_is_consistent_between_forward_and_backward()
