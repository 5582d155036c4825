"""Kernels of csrc/norm_subsample.hip executed on the HOST EMULATOR (tests/hipemu) through the real
ctypes glue / autograd Functions of neural_sp_amd.ops, against torch-CPU restatements of the reference
modules (encoders/subsampling.py, modules/conformer_convolution.py:58-66,119-124).

This is how the kernels were checked in a round that had no GPU time left; tests/test_variants_gpu.py
runs the same comparisons on the device."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.hipemu import build_emu

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the HIP emulator')


def _swish(x):
    return x * torch.sigmoid(x)


def _grads(fn, *inputs):
    y = fn(*inputs)
    w = (torch.linspace(-1.0, 1.0, y.numel()).view_as(y) * 0.7 + 0.1).to(y.dtype)
    (y * w).sum().backward()
    return y.detach(), [None if i.grad is None else i.grad.clone() for i in inputs]


@pytest.mark.parametrize('T,f', [(11, 2), (12, 2), (13, 3), (5, 4), (1, 2)])
@pytest.mark.parametrize('kind', ['drop', 'add', 'mean_pool'])
def test_window_sum_subsamplers(T, f, kind):
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    if kind == 'add' and f != 2:
        pytest.skip('AddSubsampler asserts factor <= 2')
    torch.manual_seed(T * 10 + f)
    B, C = 3, 8
    x = torch.randn(B, T, C)

    def ref(x):
        if kind == 'drop':       # subsampling.py:118-121
            return x[:, ::f]
        if kind == 'add':        # subsampling.py:153-167
            xe = x[:, ::2]
            xo = x[:, 1::2] if T % 2 == 0 else torch.cat([x, x.new_zeros(B, 1, C)], dim=1)[:, 1::2]
            return xo + xe
        return F.avg_pool1d(x.transpose(2, 1), f, f, 0, ceil_mode=True).transpose(2, 1)   # :239-240

    To = math.ceil(T / f)
    k = {'drop': 1, 'add': 2, 'mean_pool': f}[kind]

    def ours(x):
        return ops.time_window_sum(x, k, f, 0, To, mean=(kind == 'mean_pool'))

    xr = x.clone().requires_grad_(True)
    yr, (gr,) = _grads(ref, xr)
    with emulated_kernels():
        xo = x.clone().requires_grad_(True)
        yo, (go,) = _grads(ours, xo)
    assert yo.shape == yr.shape
    torch.testing.assert_close(yo, yr, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(go, gr, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('T,k,stride,pad', [(12, 2, 2, 0), (13, 3, 3, 0), (13, 3, 2, 1), (14, 3, 3, 1), (7, 3, 4, 1)])
def test_window_gather_is_im2col(T, k, stride, pad):
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    torch.manual_seed(T)
    B, C = 2, 12
    x = torch.randn(B, T, C)
    To = (T + 2 * pad - (k - 1) - 1) // stride + 1

    def ref(x):
        xp = F.pad(x, (0, 0, pad, pad))
        return torch.stack([xp[:, to * stride:to * stride + k].reshape(B, k * C) for to in range(To)], dim=1)

    xr = x.clone().requires_grad_(True)
    yr, (gr,) = _grads(ref, xr)
    with emulated_kernels():
        xo = x.clone().requires_grad_(True)
        yo, (go,) = _grads(lambda t: ops.time_window_gather(t, k, stride, pad, To), xo)
    torch.testing.assert_close(yo, yr, rtol=0, atol=0)
    torch.testing.assert_close(go, gr, rtol=1e-6, atol=1e-6)


def test_conv1d_subsampler_as_gather_plus_gemm_weight_view():
    """The [Co, k*Ci] view of the Conv1d weight used with the gathered windows equals F.conv1d
    (Conv1dSubsampler, subsampling.py:55-94) -- checked with a torch matmul in place of the GEMM."""
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    torch.manual_seed(5)
    B, T, C, f = 2, 17, 8, 3
    conv = nn.Conv1d(C, C, 3, stride=f, padding=1)
    x = torch.randn(B, T, C)
    ref = conv(x.transpose(2, 1)).transpose(2, 1)
    To = (T + 2 - 2 - 1) // f + 1
    with emulated_kernels():
        g = ops.time_window_gather(x, 3, f, 1, To)
    w2 = conv.weight.permute(0, 2, 1).contiguous().view(C, 3 * C)
    torch.testing.assert_close(F.linear(g, w2, conv.bias), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('M,C', [(5, 8), (150, 64), (700, 260)])
def test_batch_norm_swish_training_and_eval(M, C):
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    torch.manual_seed(M)
    x = torch.randn(M, C) * 1.7 + torch.linspace(-3, 3, C)       # non-zero channel means
    bn_r = nn.BatchNorm1d(C)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5)
        bn_r.bias.uniform_(-0.5, 0.5)
        bn_r.running_mean.uniform_(-1, 1)
        bn_r.running_var.uniform_(0.5, 2)
    bn_o = nn.BatchNorm1d(C)
    bn_o.load_state_dict(bn_r.state_dict())

    def ref(x, w, b):       # conformer_convolution.py:119-122: norm on [B*T, C, 1], then Swish
        return _swish(bn_r(x.view(M, C, 1))).view(M, C)

    for training in (True, False):
        bn_r.train(training)
        bn_o.train(training)
        for p in list(bn_r.parameters()) + list(bn_o.parameters()):
            p.grad = None
        xr = x.clone().requires_grad_(True)
        yr, _ = _grads(ref, xr, bn_r.weight, bn_r.bias)
        with emulated_kernels():
            xo = x.clone().requires_grad_(True)
            yo, _ = _grads(lambda t, w, b: ops.batch_norm_act(t, bn_o, bn_o.training, act='swish'), xo,
                           bn_o.weight, bn_o.bias)
        torch.testing.assert_close(yo, yr, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(xo.grad, xr.grad, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(bn_o.weight.grad, bn_r.weight.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bn_o.bias.grad, bn_r.bias.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bn_o.running_mean, bn_r.running_mean, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(bn_o.running_var, bn_r.running_var, rtol=1e-5, atol=1e-5)
        assert int(bn_o.num_batches_tracked) == int(bn_r.num_batches_tracked) == 1


@pytest.mark.parametrize('M,C', [(3, 4), (90, 64), (300, 132)])
def test_group_norm_pairs_swish(M, C):
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    torch.manual_seed(C)
    x = torch.randn(M, C)
    gn = nn.GroupNorm(max(1, C // 2), C)                          # conformer_convolution.py:61-63
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)

    def ref(x, w, b):
        return _swish(F.group_norm(x.view(M, C, 1), gn.num_groups, w, b, gn.eps)).view(M, C)

    # fp64 reference: with pairs whose two channels nearly coincide rstd approaches 1/sqrt(eps) = 316 and torch's
    # own fp32 GroupNorm backward is off by 7e-4 absolute here (the kernel, differencing the pair first, by 1.5e-5)
    xr, wr, br = x.double().requires_grad_(True), gn.weight.detach().double().requires_grad_(True), \
        gn.bias.detach().double().requires_grad_(True)
    yr, (gx, gw, gb) = _grads(ref, xr, wr, br)
    yr, gx, gw, gb = yr.float(), gx.float(), gw.float(), gb.float()
    with emulated_kernels():
        xo, wo, bo = x.clone().requires_grad_(True), gn.weight.detach().clone().requires_grad_(True), \
            gn.bias.detach().clone().requires_grad_(True)
        yo, (ox, ow, ob) = _grads(lambda t, w, b: ops.group_norm2_act(t, w, b, gn.eps, act='swish'), xo, wo, bo)
    torch.testing.assert_close(yo, yr, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ox, gx, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(ow, gw, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ob, gb, rtol=1e-4, atol=1e-4)
