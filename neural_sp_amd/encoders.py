"""Encoders of the Speech2Text hot path: CNN frontend + Transformer / Conformer stacks.

Mirrors (class names, constructor signatures, parameter names, forward contract):
  neural_sp/models/seq2seq/encoders/conv.py            ConvEncoder, Conv2dBlock
  neural_sp/models/seq2seq/encoders/subsampling.py     MaxPoolSubsampler
  neural_sp/models/seq2seq/encoders/transformer_block.py
  neural_sp/models/seq2seq/encoders/conformer_block.py, conformer_block_v2.py
  neural_sp/models/seq2seq/encoders/transformer.py     TransformerEncoder
  neural_sp/models/seq2seq/encoders/conformer.py       ConformerEncoder
  neural_sp/models/seq2seq/encoders/build.py           build_encoder
Activations stay channels-last on the device; sequence lengths stay on the host
(IntTensor on CPU, as in the reference) plus one int32 device copy per resolution for the
in-kernel attention masks.  Streaming inference (caches) is out of scope.
"""
import copy
import logging
import math
import random

import numpy as np
import torch
import torch.nn as nn

from neural_sp_amd import ops
from neural_sp_amd.modules import (
    AttnMask,
    ConformerConvBlock,
    MultiheadAttentionMechanism as MHA,
    PositionalEncoding,
    PositionwiseFeedForward as FFN,
    RelativeMultiheadAttentionMechanism as RelMHA,
    XLPositionalEmbedding,
    init_with_lecun_normal,
)

random.seed(1)  # conformer_block.py:15 / transformer_block.py -- LayerDrop stream

logger = logging.getLogger(__name__)


class EncoderBase(nn.Module):
    """encoder_base.py:20-51 (properties only; plotting is out of scope)."""

    @property
    def output_dim(self):
        return self._odim

    @property
    def output_dim_sub1(self):
        return getattr(self, '_odim_sub1', self._odim)

    @property
    def output_dim_sub2(self):
        return getattr(self, '_odim_sub2', self._odim)

    @property
    def subsampling_factor(self):
        return self._factor

    @property
    def device(self):
        return next(self.parameters()).device

    def reset_cache(self):
        raise NotImplementedError

    def _plot_attention(self, save_path=None, n_cols=2):
        pass  # train.py:484-487 calls it through Speech2Text.plot_attention(); nothing is drawn here


# --------------------------------------------------------------------------- CNN frontend
def parse_cnn_config(channels, kernel_sizes, strides, poolings):
    """conv.py:480-505"""
    _channels, _kernel_sizes, _strides, _poolings = [], [], [], []
    is_1dconv = '(' not in kernel_sizes
    if len(channels) > 0:
        _channels = [int(c) for c in channels.split('_')]
    if len(kernel_sizes) > 0:
        if is_1dconv:
            _kernel_sizes = [int(c) for c in kernel_sizes.split('_')]
        else:
            _kernel_sizes = [[int(c.split(',')[0].replace('(', '')),
                              int(c.split(',')[1].replace(')', ''))] for c in kernel_sizes.split('_')]
    if len(strides) > 0:
        if is_1dconv:
            _strides = [int(s) for s in strides.split('_')]
        else:
            _strides = [[int(s.split(',')[0].replace('(', '')),
                         int(s.split(',')[1].replace(')', ''))] for s in strides.split('_')]
    if len(poolings) > 0:
        if is_1dconv:
            _poolings = [int(p) for p in poolings.split('_')]
        else:
            _poolings = [[int(p.split(',')[0].replace('(', '')),
                          int(p.split(',')[1].replace(')', ''))] for p in poolings.split('_')]
    return (_channels, _kernel_sizes, _strides, _poolings), is_1dconv


def _conv_len(seq_len, k, pad, stride):
    """conv.py:471-477 (_update_2d, conv branch)"""
    return int(math.floor((seq_len + 2 * pad - (k - 1) - 1) // stride + 1))


def _pool_len_ceil(seq_len, k, stride):
    """conv.py:471-474 (_update_2d, MaxPool ceil_mode branch)"""
    return int(math.ceil((seq_len + 1 - (k - 1) - 1) // stride + 1))


class Conv2dBlock(EncoderBase):
    """conv.py:289-396: conv3x3 -> ReLU -> conv3x3 -> ReLU -> MaxPool2d(ceil), on
    channels-last `[B,T,F,C]`.  Built: 3x3 kernels, stride (1,1), no normalisation, no
    residual (the configuration of every *former recipe)."""

    def __init__(self, input_dim, in_channel, out_channel, kernel_size, stride, pooling,
                 dropout, normalization, residual):
        super().__init__()
        if tuple(kernel_size) != (3, 3) or tuple(stride) != (1, 1) or normalization or residual:
            raise NotImplementedError('Conv2dBlock: only 3x3 / stride 1 / no norm / no residual is built')
        self.dropout_p = dropout
        self.time_axis = 0
        self.conv1 = nn.Conv2d(in_channel, out_channel, kernel_size=tuple(kernel_size),
                               stride=(1, 1), padding=(1, 1))
        self._odim = _conv_len(input_dim, 3, 1, 1)
        self.conv2 = nn.Conv2d(out_channel, out_channel, kernel_size=tuple(kernel_size),
                               stride=tuple(stride), padding=(1, 1))
        self._odim = _conv_len(self._odim, 3, 1, 1)
        self.pool = None
        self.pooling = (1, 1)
        self._factor = 1
        if len(pooling) > 0 and np.prod(pooling) > 1:
            self.pooling = tuple(pooling)
            self.pool = nn.MaxPool2d(kernel_size=tuple(pooling), stride=tuple(pooling),
                                     padding=(0, 0), ceil_mode=True)
            self._odim = _pool_len_ceil(self._odim, pooling[1], pooling[1])
            if self._odim % 2 != 0:
                self._odim = (self._odim // 2) * 2
            self._factor *= pooling[0]

    def forward(self, xs, xlens, lookback=False, lookahead=False, last=False):
        """xs `[B,T,F,C_i]` -> `[B,T',F',C_o]` (or `[B,T',C_o,F']` if last)."""
        if lookback or lookahead:
            raise NotImplementedError('CNN lookback/lookahead belong to streaming inference')
        xs = ops.conv3x3_relu(xs, self.conv1.weight, self.conv1.bias)
        xs = ops.dropout(xs, self.dropout_p, self.training)
        xlens = torch.IntTensor([_conv_len(int(x), 3, 1, 1) for x in xlens])
        xs = ops.conv3x3_relu(xs, self.conv2.weight, self.conv2.bias)
        xs = ops.dropout(xs, self.dropout_p, self.training)
        xlens = torch.IntTensor([_conv_len(int(x), 3, 1, 1) for x in xlens])
        if self.pool is not None or last:
            xs = ops.maxpool2d(xs, self.pooling[0], self.pooling[1], to_btcf=last)
        if self.pool is not None:
            xlens = torch.IntTensor([_pool_len_ceil(int(x), self.pooling[0], self.pooling[0]) for x in xlens])
        return xs, xlens


class ConvEncoder(EncoderBase):
    """conv.py:18-195 (2-D CNN variant)."""

    def __init__(self, input_dim, in_channel, channels, kernel_sizes, strides, poolings,
                 dropout, normalization, residual, bottleneck_dim, param_init):
        super().__init__()
        assert channels
        (channels, kernel_sizes, strides, poolings), is_1dconv = parse_cnn_config(
            channels, kernel_sizes, strides, poolings)
        if is_1dconv:
            raise NotImplementedError('Conv1dBlock frontend is not on the benchmarked path')
        self.is_1dconv = False
        self.in_channel = in_channel
        assert input_dim % in_channel == 0
        self.input_freq = input_dim // in_channel
        self.residual = residual
        assert len(channels) > 0
        assert len(channels) == len(kernel_sizes) == len(strides) == len(poolings)
        self.layers = nn.ModuleList()
        C_i = in_channel
        in_freq = self.input_freq
        for lth in range(len(channels)):
            block = Conv2dBlock(input_dim=in_freq, in_channel=C_i, out_channel=channels[lth],
                                kernel_size=kernel_sizes[lth], stride=strides[lth],
                                pooling=poolings[lth], dropout=dropout,
                                normalization=normalization, residual=residual)
            self.layers += [block]
            in_freq = block.output_dim
            C_i = channels[lth]
        self._odim = int(C_i * in_freq)
        self.bridge = None
        if bottleneck_dim > 0 and bottleneck_dim != self._odim:
            self.bridge = nn.Linear(self._odim, bottleneck_dim)
            self._odim = bottleneck_dim
        self._factor = 1
        for s in strides:
            self._factor *= s[0]
        for p in poolings:
            self._factor *= p[0]
        for n, p in self.named_parameters():
            init_with_lecun_normal(n, p, param_init)

    def forward(self, xs, xlens, lookback=False, lookahead=False):
        """xs `[B,T,F]`, xlens IntTensor (CPU) -> (`[B,T',d]`, xlens)."""
        B, T, F = xs.size()
        if self.in_channel == 1:
            xs = xs.reshape(B, T, F, 1)  # channels-last view of [B,1,T,F]
        else:
            # conv.py:167-175: the feature vector holds the channels one after the other (static | delta | delta-delta)
            xs = xs.view(B, T, self.in_channel, F // self.in_channel).permute(0, 1, 3, 2).contiguous()
        for i, block in enumerate(self.layers):
            xs, xlens = block(xs, xlens, lookback=lookback, lookahead=lookahead,
                              last=(i == len(self.layers) - 1))
        B, To, Co, Fo = xs.size()
        xs = xs.reshape(B, To, Co * Fo)  # == transpose(2,1).view(B,T',C*F') of conv.py:189
        if self.bridge is not None:
            xs = ops.linear(xs, self.bridge.weight, self.bridge.bias)
        return xs, xlens


class MaxPoolSubsampler(nn.Module):
    """subsampling.py:175-209"""

    def __init__(self, subsampling_factor):
        super().__init__()
        self.factor = subsampling_factor

    def forward(self, xs, xlens, batch_first=True):
        if self.factor == 1:
            return xs, xlens
        assert batch_first
        xs = ops.maxpool1d_time(xs, self.factor)
        xlens = torch.IntTensor([_pool_len_ceil(int(x), self.factor, self.factor) for x in xlens])
        return xs, xlens


class DropSubsampler(nn.Module):
    """subsampling.py:97-128 (`--subsample_type drop`, the reference's default): keep frames 0, f, 2f, ..."""

    def __init__(self, subsampling_factor):
        super().__init__()
        self.factor = subsampling_factor

    def forward(self, xs, xlens, batch_first=True):
        if self.factor == 1:
            return xs, xlens
        assert batch_first
        f = self.factor
        xs = ops.time_window_sum(xs, 1, f, 0, math.ceil(xs.size(1) / f))
        xlens = torch.IntTensor([max(1, math.ceil(int(i) / f)) for i in xlens])
        return xs, xlens


class AddSubsampler(nn.Module):
    """subsampling.py:131-172: x[2t] + x[2t+1] (a zero frame completes an odd length)."""

    def __init__(self, subsampling_factor):
        super().__init__()
        self.factor = subsampling_factor
        assert subsampling_factor <= 2

    def forward(self, xs, xlens, batch_first=True):
        if self.factor == 1:
            return xs, xlens
        assert batch_first
        f = self.factor
        xs = ops.time_window_sum(xs, 2, 2, 0, math.ceil(xs.size(1) / 2))
        xlens = torch.IntTensor([max(1, math.ceil(int(i) / f)) for i in xlens])
        return xs, xlens


class MeanPoolSubsampler(nn.Module):
    """subsampling.py:212-246: AvgPool1d(k = s = f, ceil_mode=True); the clipped last window is averaged over
    the frames it covers.  Lengths follow update_lens_1d's non-max-pool branch (conv.py:446-448), i.e.
    floor((len - f) / f) + 1 -- NOT the ceil of the tensor's own length (the reference's behaviour)."""

    def __init__(self, subsampling_factor):
        super().__init__()
        self.factor = subsampling_factor

    def forward(self, xs, xlens, batch_first=True):
        if self.factor == 1:
            return xs, xlens
        assert batch_first
        f = self.factor
        xs = ops.time_window_sum(xs, f, f, 0, math.ceil(xs.size(1) / f), mean=True)
        xlens = torch.IntTensor([math.floor((int(i) - (f - 1) - 1) // f + 1) for i in xlens])
        return xs, xlens


class ConcatSubsampler(nn.Module):
    """subsampling.py:13-52: f successive frames side by side (oldest first; trailing frames that do not fill
    a group are dropped) -> Linear(f * d, d) -> ReLU.  Window gather + one GEMM with the ReLU in its epilogue."""

    def __init__(self, subsampling_factor, n_units):
        super().__init__()
        self.factor = subsampling_factor
        if subsampling_factor > 1:
            self.proj = nn.Linear(n_units * subsampling_factor, n_units)

    def forward(self, xs, xlens, batch_first=True):
        if self.factor == 1:
            return xs, xlens
        assert batch_first
        f = self.factor
        xs = ops.time_window_gather(xs, f, f, 0, xs.size(1) // f)
        xs = ops.linear(xs, self.proj.weight, self.proj.bias, act='relu')
        xlens = torch.IntTensor([max(1, int(i) // f) for i in xlens])
        return xs, xlens


class Conv1dSubsampler(nn.Module):
    """subsampling.py:55-94: Conv1d(d, d, k=3, stride=f, padding=1) -> ReLU, as im2col over time + one GEMM
    against the `[C_out, k * C_in]` view of the Conv1d weight (the parameter keeps the nn.Conv1d layout)."""

    def __init__(self, subsampling_factor, n_units, kernel_size=3):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size should be odd for 'same' conv."
        self.factor = subsampling_factor
        if subsampling_factor > 1:
            self.conv1d = nn.Conv1d(in_channels=n_units, out_channels=n_units, kernel_size=kernel_size,
                                    stride=subsampling_factor, padding=(kernel_size - 1) // 2)

    def _out_len(self, n):   # conv.py:446-448
        k, f, pad = self.conv1d.kernel_size[0], self.conv1d.stride[0], self.conv1d.padding[0]
        return math.floor((n + 2 * pad - (k - 1) - 1) // f + 1)

    def forward(self, xs, xlens, batch_first=True):
        if self.factor == 1:
            return xs, xlens
        assert batch_first
        k, f, pad = self.conv1d.kernel_size[0], self.conv1d.stride[0], self.conv1d.padding[0]
        co, ci = self.conv1d.weight.shape[:2]
        xs = ops.time_window_gather(xs, k, f, pad, self._out_len(xs.size(1)))
        w2 = self.conv1d.weight.permute(0, 2, 1).contiguous().view(co, k * ci)   # [Co, Ci, k] -> [Co, k*Ci]
        xs = ops.linear(xs, w2, self.conv1d.bias, act='relu')
        xlens = torch.IntTensor([self._out_len(int(i)) for i in xlens])
        return xs, xlens


# --------------------------------------------------------------------------- blocks
class TransformerEncoderBlock(nn.Module):
    """transformer_block.py:20-141.  The reference's typo 'relaive' (:46) is kept: only
    pe_type 'relative_xl' selects RelMHA here."""

    def __init__(self, d_model, d_ff, n_heads, dropout, dropout_att, dropout_layer,
                 layer_norm_eps, ffn_activation, param_init, pe_type, clamp_len, ffn_bottleneck_dim):
        super().__init__()
        self.n_heads = n_heads
        self.rel_attn = pe_type in ['relaive', 'relative_xl']
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        mha = RelMHA if self.rel_attn else MHA
        self.self_attn = mha(kdim=d_model, qdim=d_model, adim=d_model, odim=d_model, n_heads=n_heads,
                             dropout=dropout_att, param_init=param_init,
                             xl_like=pe_type == 'relative_xl', clamp_len=clamp_len)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.feed_forward = FFN(d_model, d_ff, dropout, ffn_activation, param_init, ffn_bottleneck_dim)
        self.dropout_p = dropout
        self.dropout_layer = dropout_layer
        self._xx_aws = None

    @property
    def xx_aws(self):
        return self._xx_aws

    def forward(self, xs, xx_mask=None, cache=None, pos_embs=None, rel_bias=(None, None)):
        if cache is not None:
            raise NotImplementedError('streaming cache')
        self._xx_aws = None
        u_bias, v_bias = rel_bias
        if self.dropout_layer > 0:
            if self.training and random.random() < self.dropout_layer:
                return xs, {}
            else:
                xs = ops.scale(xs, 1.0 / (1 - self.dropout_layer))
        xn, xs = ops.layer_norm_split(xs, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        residual = xs
        if self.rel_attn:
            xs, self._xx_aws = self.self_attn(xn, xn, pos_embs, xx_mask, u_bias, v_bias,
                                              residual=residual, out_dropout=self.dropout_p)
        else:
            xs, self._xx_aws = self.self_attn(xn, xn, xn, mask=xx_mask, residual=residual,
                                              out_dropout=self.dropout_p)[:2]
        xn, xs = ops.layer_norm_split(xs, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        residual = xs
        xs = self.feed_forward(xn, residual=residual, alpha=1.0, out_dropout=self.dropout_p)
        return xs, {}


class ConformerEncoderBlock(nn.Module):
    """conformer_block.py:20-182: x += 1/2 FFN(LN x); x += RelMHA(LN x); x += Conv(LN x);
    x += 1/2 FFN(LN x); x = LN x.  Every `alpha*dropout(.) + residual` is the epilogue of
    the sub-block's last GEMM."""

    def __init__(self, d_model, d_ff, n_heads, kernel_size, dropout, dropout_att, dropout_layer,
                 layer_norm_eps, ffn_activation, param_init, pe_type, clamp_len,
                 ffn_bottleneck_dim, unidirectional, normalization='layer_norm'):
        super().__init__()
        self.n_heads = n_heads
        self.fc_factor = 0.5
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.feed_forward_macaron = FFN(d_model, d_ff, dropout, ffn_activation, param_init, ffn_bottleneck_dim)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.self_attn = RelMHA(kdim=d_model, qdim=d_model, adim=d_model, odim=d_model, n_heads=n_heads,
                                dropout=dropout_att, param_init=param_init,
                                xl_like=pe_type == 'relative_xl', clamp_len=clamp_len)
        self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.conv = ConformerConvBlock(d_model, kernel_size, param_init, normalization, causal=unidirectional)
        self.conv_context = kernel_size
        self.norm4 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.feed_forward = FFN(d_model, d_ff, dropout, ffn_activation, param_init, ffn_bottleneck_dim)
        self.norm5 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.dropout_p = dropout
        self.dropout_layer = dropout_layer
        self._xx_aws = None

    @property
    def xx_aws(self):
        return self._xx_aws

    def forward(self, xs, xx_mask=None, cache=None, pos_embs=None, rel_bias=(None, None), next_norm=None):
        """next_norm (round 6): the first LayerNorm of the block that will consume this block's output UNCHANGED (no
        subsampling, no layer dropout in between): its normalisation is then produced together with this block's last one
        (ops.layer_norm_pair) and travels on the output as `_nsp_prenorm` = (that LayerNorm module, its output)."""
        if cache is not None:
            raise NotImplementedError('streaming cache')
        self._xx_aws = None
        u_bias, v_bias = rel_bias
        pre = getattr(xs, '_nsp_prenorm', None)
        if self.dropout_layer > 0:
            if self.training and random.random() < self.dropout_layer:
                return xs, {}
            else:
                xs = ops.scale(xs, 1.0 / (1 - self.dropout_layer))
                pre = None
        p = self.dropout_p
        if pre is not None and pre[0] is self.norm1:
            xn = pre[1]                        # LN_1(xs), made by the previous block's last kernel; xs is its residual output
        else:
            xn, xs = ops.layer_norm_split(xs, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        xs = self.feed_forward_macaron(xn, residual=xs, alpha=self.fc_factor, out_dropout=p)
        xn, xs = ops.layer_norm_split(xs, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        xs, self._xx_aws = self.self_attn(xn, xn, pos_embs, xx_mask, u_bias, v_bias,
                                          residual=xs, out_dropout=p)
        xn, xs = ops.layer_norm_split(xs, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        xs = self.conv(xn, residual=xs, out_dropout=p)
        xn, xs = ops.layer_norm_split(xs, self.norm4.weight, self.norm4.bias, self.norm4.eps)
        xs = self.feed_forward(xn, residual=xs, alpha=self.fc_factor, out_dropout=p)
        if next_norm is not None and ops.layer_norm_pair_ok(xs, xs.shape[-1]):
            xn, xs = ops.layer_norm_pair(xs, self.norm5.weight, self.norm5.bias, self.norm5.eps,
                                         next_norm.weight, next_norm.bias, next_norm.eps)
            xs._nsp_prenorm = (next_norm, xn)
        else:
            xs = ops.layer_norm(xs, self.norm5.weight, self.norm5.bias, self.norm5.eps)
        return xs, {}


class ConformerEncoderBlock_v2(nn.Module):
    """conformer_block_v2.py:20-183: FFN -> Conv -> plain MHA -> FFN -> LN."""

    def __init__(self, d_model, d_ff, n_heads, kernel_size, dropout, dropout_att, dropout_layer,
                 layer_norm_eps, ffn_activation, param_init, pe_type, clamp_len,
                 ffn_bottleneck_dim, unidirectional, normalization='layer_norm'):
        super().__init__()
        self.n_heads = n_heads
        self.fc_factor = 0.5
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.feed_forward_macaron = FFN(d_model, d_ff, dropout, ffn_activation, param_init, ffn_bottleneck_dim)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.conv = ConformerConvBlock(d_model, kernel_size, param_init, normalization, causal=unidirectional)
        self.conv_context = kernel_size
        self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.self_attn = MHA(kdim=d_model, qdim=d_model, adim=d_model, odim=d_model, n_heads=n_heads,
                             dropout=dropout_att, param_init=param_init)
        self.norm4 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.feed_forward = FFN(d_model, d_ff, dropout, ffn_activation, param_init, ffn_bottleneck_dim)
        self.norm5 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.dropout_p = dropout
        self.dropout_layer = dropout_layer
        self._xx_aws = None

    @property
    def xx_aws(self):
        return self._xx_aws

    def forward(self, xs, xx_mask=None, cache=None, pos_embs=None, rel_bias=(None, None)):
        if cache is not None:
            raise NotImplementedError('streaming cache')
        self._xx_aws = None
        assert rel_bias[0] is None and rel_bias[1] is None
        if self.dropout_layer > 0:
            if self.training and random.random() < self.dropout_layer:
                return xs, {}
            else:
                xs = ops.scale(xs, 1.0 / (1 - self.dropout_layer))
        p = self.dropout_p
        xn, xs = ops.layer_norm_split(xs, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        xs = self.feed_forward_macaron(xn, residual=xs, alpha=self.fc_factor, out_dropout=p)
        xn, xs = ops.layer_norm_split(xs, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        xs = self.conv(xn, residual=xs, out_dropout=p)
        xn, xs = ops.layer_norm_split(xs, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        xs, self._xx_aws = self.self_attn(xn, xn, xn, mask=xx_mask, residual=xs, out_dropout=p)[:2]
        xn, xs = ops.layer_norm_split(xs, self.norm4.weight, self.norm4.bias, self.norm4.eps)
        xs = self.feed_forward(xn, residual=xs, alpha=self.fc_factor, out_dropout=p)
        xs = ops.layer_norm(xs, self.norm5.weight, self.norm5.bias, self.norm5.eps)
        return xs, {}


# --------------------------------------------------------------------------- encoder stacks
def chunkwise(xs, N_l, N_c, N_r, padding=True):
    """utils.py:13-45: fold overlapping windows into the batch dim (pure data movement)."""
    bs, xmax, idim = xs.size()
    n_chunks = math.ceil(xmax / N_c) if padding else xmax // (N_l + N_c + N_r)
    xs_tmp = xs.new_zeros(bs, n_chunks, N_l + N_c + N_r, idim)
    if padding:
        xs = torch.cat([xs.new_zeros(bs, N_l, idim), xs, xs.new_zeros(bs, N_r, idim)], dim=1)
    t = N_l
    for chunk_idx in range(n_chunks):
        xs_chunk = xs[:, t - N_l:t + (N_c + N_r)]
        xs_tmp[:, chunk_idx, :xs_chunk.size(1), :] = xs_chunk
        t += N_c
    return xs_tmp.view(bs * n_chunks, N_l + N_c + N_r, idim)


class TransformerEncoder(EncoderBase):
    """transformer.py:40-617 (training / full-utterance evaluation path)."""

    def __init__(self, input_dim, enc_type, n_heads, n_layers, n_layers_sub1, n_layers_sub2,
                 d_model, d_ff, ffn_bottleneck_dim, ffn_activation, pe_type, layer_norm_eps,
                 last_proj_dim, dropout_in, dropout, dropout_att, dropout_layer,
                 subsample, subsample_type, n_stacks, n_splices, frontend_conv,
                 task_specific_layer, param_init, clamp_len, lookahead,
                 chunk_size_left, chunk_size_current, chunk_size_right, streaming_type):
        super().__init__()
        self.subsample_factors = [1] * n_layers
        for lth, s in enumerate(list(map(int, subsample.split('_')[:n_layers]))):
            self.subsample_factors[lth] = s
        lookaheads = [0] * n_layers
        for lth, s in enumerate(list(map(int, lookahead.split('_')[:n_layers]))):
            lookaheads[lth] = s
        if n_layers_sub1 < 0 or (n_layers_sub1 > 1 and n_layers < n_layers_sub1):
            raise Warning('Set n_layers_sub1 between 1 to n_layers.')
        if n_layers_sub2 < 0 or (n_layers_sub2 > 1 and n_layers_sub1 < n_layers_sub2):
            raise Warning('Set n_layers_sub2 between 1 to n_layers_sub1.')
        self.enc_type = enc_type
        self.d_model = d_model
        self.n_layers = n_layers
        self.n_heads = n_heads
        self.pe_type = pe_type
        self.scale = math.sqrt(d_model)
        chunk_size_left = str(chunk_size_left)
        chunk_size_current = str(chunk_size_current)
        chunk_size_right = str(chunk_size_right)
        self.unidir = 'uni' in enc_type
        self.lookaheads = lookaheads
        if sum(lookaheads) > 0:
            assert self.unidir
        self.N_l = int(chunk_size_left.split('_')[-1]) // n_stacks
        self.N_c = int(chunk_size_current.split('_')[-1]) // n_stacks
        self.N_r = int(chunk_size_right.split('_')[-1]) // n_stacks
        self.lc_bidir = self.N_c > 0 and enc_type != 'conv' and 'uni' not in enc_type
        self.cnn_lookahead = self.unidir or enc_type == 'conv'
        self.streaming_type = streaming_type if self.lc_bidir else ''
        self.causal = self.unidir or self.streaming_type == 'mask'
        if self.unidir:
            assert self.N_l == self.N_c == self.N_r == 0
        if self.streaming_type == 'mask':
            assert self.N_r == 0
            assert self.N_l % self.N_c == 0
        if self.lc_bidir:
            assert n_layers_sub1 == 0 and n_layers_sub2 == 0 and not self.unidir
        self.n_layers_sub1 = n_layers_sub1
        self.n_layers_sub2 = n_layers_sub2
        self.task_specific_layer = task_specific_layer
        self.bridge = None
        self.bridge_sub1 = None
        self.bridge_sub2 = None
        self.aws_dict = {}
        self.data_dict = {}
        self.conv = frontend_conv
        if self.conv is not None:
            self._odim = self.conv.output_dim
        else:
            self._odim = input_dim * n_splices * n_stacks
            self.embed = nn.Linear(self._odim, d_model)
        self._factor = 1
        self.conv_factor = self.conv.subsampling_factor if self.conv is not None else 1
        self._factor *= self.conv_factor
        self.subsample_layers = None
        if np.prod(self.subsample_factors) > 1:
            self._factor *= np.prod(self.subsample_factors)
            if subsample_type == 'max_pool':
                self.subsample_layers = nn.ModuleList([MaxPoolSubsampler(factor)
                                                       for factor in self.subsample_factors])
            elif subsample_type == 'mean_pool':
                self.subsample_layers = nn.ModuleList([MeanPoolSubsampler(factor)
                                                       for factor in self.subsample_factors])
            elif subsample_type == 'concat':
                self.subsample_layers = nn.ModuleList([ConcatSubsampler(factor, self._odim)
                                                       for factor in self.subsample_factors])
            elif subsample_type == 'drop':
                self.subsample_layers = nn.ModuleList([DropSubsampler(factor)
                                                       for factor in self.subsample_factors])
            elif subsample_type == 'conv1d':
                assert not self.causal
                self.subsample_layers = nn.ModuleList([Conv1dSubsampler(factor, self._odim)
                                                       for factor in self.subsample_factors])
            elif subsample_type == 'add':
                self.subsample_layers = nn.ModuleList([AddSubsampler(factor)
                                                       for factor in self.subsample_factors])
            else:
                raise NotImplementedError(subsample_type)
        assert self.N_l % self._factor == 0
        assert self.N_c % self._factor == 0
        assert self.N_r % self._factor == 0
        self.pos_enc, self.pos_emb = None, None
        self.u_bias, self.v_bias = None, None
        if pe_type in ['relative', 'relative_xl']:
            self.pos_emb = XLPositionalEmbedding(d_model, dropout)
            if pe_type == 'relative_xl':
                self.u_bias = nn.Parameter(torch.Tensor(n_heads, d_model // n_heads))
                self.v_bias = nn.Parameter(torch.Tensor(n_heads, d_model // n_heads))
        else:
            self.pos_enc = PositionalEncoding(d_model, dropout_in, pe_type, param_init)
        mk = (d_model, d_ff, n_heads, dropout, dropout_att, layer_norm_eps, ffn_activation, param_init, pe_type,
              clamp_len, ffn_bottleneck_dim)
        self.layers = nn.ModuleList([copy.deepcopy(self._make_block(dropout_layer * (lth + 1) / n_layers, *mk))
                                     for lth in range(n_layers)])
        self.norm_out = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self._odim = d_model
        # auxiliary-task outputs of hierarchical multi-task training (transformer.py:233-263, conformer.py:92-106):
        # taken after layer n_layers_sub{1,2}, optionally through a task-specific extra block, then bridge / LayerNorm
        for sub, n_sub in (('sub1', n_layers_sub1), ('sub2', n_layers_sub2)):
            if n_sub > 0:
                if task_specific_layer:
                    setattr(self, 'layer_' + sub, self._make_block(dropout_layer * n_sub / n_layers, *mk))
                odim_sub = d_model
                if last_proj_dim > 0 and last_proj_dim != self.output_dim:
                    setattr(self, 'bridge_' + sub, nn.Linear(self._odim, last_proj_dim))
                    odim_sub = last_proj_dim
                setattr(self, 'norm_out_' + sub,
                        None if n_sub == n_layers else nn.LayerNorm(odim_sub, eps=layer_norm_eps))
        if last_proj_dim > 0 and last_proj_dim != self.output_dim:
            self.bridge = nn.Linear(self._odim, last_proj_dim)
            self._odim = last_proj_dim
        self.reset_parameters(param_init)
        self.reset_cache()

    def _make_block(self, dropout_layer, d_model, d_ff, n_heads, dropout, dropout_att, layer_norm_eps,
                    ffn_activation, param_init, pe_type, clamp_len, ffn_bottleneck_dim):
        return TransformerEncoderBlock(d_model, d_ff, n_heads, dropout, dropout_att, dropout_layer,
                                       layer_norm_eps, ffn_activation, param_init, pe_type, clamp_len,
                                       ffn_bottleneck_dim)

    def sub_module(self, xs, xx_mask, lth, pos_embs=None, module='sub1'):
        """transformer.py:619-630 (the task-specific block is called without the global u/v biases, as there)."""
        xs_sub = xs
        if self.task_specific_layer:
            layer = getattr(self, 'layer_' + module)
            xs_sub, _ = layer(xs, xx_mask, pos_embs=pos_embs)
            if not self.training:
                self.aws_dict['xx_aws_%s_layer%d' % (module, lth)] = layer.xx_aws
        bridge = getattr(self, 'bridge_' + module)
        if bridge is not None:
            xs_sub = ops.linear(xs_sub, bridge.weight, bridge.bias)
        norm = getattr(self, 'norm_out_' + module)
        if norm is not None:
            xs_sub = ops.layer_norm(xs_sub, norm.weight, norm.bias, norm.eps)
        return xs_sub

    def reset_parameters(self, param_init):
        """transformer.py:346-364"""
        if param_init == 'xavier_uniform':
            if self.conv is None:
                nn.init.xavier_uniform_(self.embed.weight)
                nn.init.constant_(self.embed.bias, 0.)
            for bridge in (self.bridge, self.bridge_sub1, self.bridge_sub2):
                if bridge is not None:
                    nn.init.xavier_uniform_(bridge.weight)
                    nn.init.constant_(bridge.bias, 0.)
            if self.pe_type == 'relative_xl':
                nn.init.xavier_uniform_(self.u_bias)
                nn.init.xavier_uniform_(self.v_bias)

    def reset_cache(self):
        self.cache = [None] * self.n_layers
        self.offset = 0

    def _mask(self, xlens, lookahead, N_l=0, N_c=0):
        klens = ops.h2d(xlens, self.device, torch.int32)
        if self.streaming_type == 'mask':
            return AttnMask(klens, False, 0, N_l, N_c)
        return AttnMask(klens, self.unidir, lookahead)

    def forward(self, xs, xlens, task, streaming=False, lookback=False, lookahead=False):
        """xs `[B,T,input_dim]` fp32 on the device, xlens IntTensor on the CPU ->
        {'ys': {'xs': `[B,T'',d]`, 'xlens': IntTensor}, 'ys_sub1': ..., 'ys_sub2': ...}"""
        if streaming:
            raise NotImplementedError('streaming encoding is inference-only (out of scope)')
        eouts = {'ys': {'xs': None, 'xlens': None},
                 'ys_sub1': {'xs': None, 'xlens': None},
                 'ys_sub2': {'xs': None, 'xlens': None}}
        bs, xmax = xs.size()[:2]
        n_chunks = 0
        lc_bidir = self.lc_bidir
        N_l, N_c, N_r = self.N_l, self.N_c, self.N_r
        if lc_bidir:
            if self.streaming_type == 'mask':
                xs = chunkwise(xs, 0, N_c, 0, padding=True)
            elif self.streaming_type == 'reshape':
                xs = chunkwise(xs, N_l, N_c, N_r, padding=True)
            n_chunks = xs.size(0) // bs
            assert bs * n_chunks == xs.size(0)
        if self.conv is None:
            xs = ops.linear(xs, self.embed.weight, self.embed.bias)
        else:
            xs, xlens = self.conv(xs, xlens, lookback=False if lc_bidir else lookback,
                                  lookahead=False if lc_bidir else lookahead)
            N_l = max(0, N_l // self.conv_factor)
            N_c = N_c // self.conv_factor
            N_r = N_r // self.conv_factor
        cb = getattr(self, '_after_frontend', None)
        if cb is not None:
            cb()  # host-side hook: work that is independent of the encoder gets enqueued here
        if self.streaming_type == 'mask':
            xs = xs.contiguous().view(bs, -1, xs.size(2))[:, :int(xlens.max())].contiguous()
        if self.enc_type == 'conv':
            eouts['ys']['xs'] = xs
            eouts['ys']['xlens'] = xlens
            return eouts
        self.reset_cache()
        if 'relative' in self.pe_type:
            xs, rel_pos_embs = self.pos_emb(xs, scale=True)
        else:
            xs = self.pos_enc(xs, scale=True, offset=self.offset)
            rel_pos_embs = None
        rel_bias = (self.u_bias, self.v_bias)
        if lc_bidir:
            xx_mask = self._mask(xlens, 0, N_l, N_c) if self.streaming_type == 'mask' else None
            for lth, layer in enumerate(self.layers):
                xs, _ = layer(xs, xx_mask, cache=None, pos_embs=rel_pos_embs, rel_bias=rel_bias)
                if lth < len(self.layers) - 1 and self.subsample_factors[lth] > 1:
                    xs, xlens = self.subsample_layers[lth](xs, xlens)
                    N_l = max(0, N_l // self.subsample_factors[lth])
                    N_c //= self.subsample_factors[lth]
                    N_r //= self.subsample_factors[lth]
                    if 'relative' in self.pe_type:
                        xs, rel_pos_embs = self.pos_emb(xs)
                    if self.streaming_type == 'mask':
                        xx_mask = self._mask(xlens, 0, N_l, N_c)
            if self.streaming_type == 'reshape':
                xs = xs[:, N_l:N_l + N_c]
                xs = xs.contiguous().view(bs, -1, xs.size(2))
                xs = xs[:, :int(xlens.max())].contiguous()
        else:
            xx_mask = self._mask(xlens, self.lookaheads[0])
            for lth, layer in enumerate(self.layers):
                # the next block's first LayerNorm rides in this block's last kernel when the block output reaches it
                # unchanged: same time resolution (no subsampling layer in between), no layer dropout, training mode only
                # (evaluation keeps per-layer outputs for plotting and takes the plain path)
                nxt = None
                if (self.training and isinstance(layer, ConformerEncoderBlock) and lth < len(self.layers) - 1
                        and self.subsample_factors[lth] == 1 and isinstance(self.layers[lth + 1], ConformerEncoderBlock)
                        and self.layers[lth + 1].dropout_layer == 0 and lth != self.n_layers_sub1 - 1
                        and lth != self.n_layers_sub2 - 1):
                    nxt = self.layers[lth + 1].norm1
                if nxt is not None:
                    xs, _ = layer(xs, xx_mask, cache=None, pos_embs=rel_pos_embs, rel_bias=rel_bias, next_norm=nxt)
                else:
                    xs, _ = layer(xs, xx_mask, cache=None, pos_embs=rel_pos_embs, rel_bias=rel_bias)
                if not self.training:
                    self.aws_dict['xx_aws_layer%d' % lth] = layer.xx_aws  # device tensor (plot on demand)
                    self.data_dict['elens%d' % lth] = xlens.numpy()
                # outputs of the auxiliary tasks are picked up before the projection layer (transformer.py:568-580)
                if lth == self.n_layers_sub1 - 1:
                    xs_sub1 = self.sub_module(xs, xx_mask, lth, rel_pos_embs, 'sub1')
                    xlens_sub1 = xlens.clone()
                    if task == 'ys_sub1':
                        eouts[task]['xs'], eouts[task]['xlens'] = xs_sub1, xlens_sub1
                        return eouts
                if lth == self.n_layers_sub2 - 1:
                    xs_sub2 = self.sub_module(xs, xx_mask, lth, rel_pos_embs, 'sub2')
                    xlens_sub2 = xlens.clone()
                    if task == 'ys_sub2':
                        eouts[task]['xs'], eouts[task]['xlens'] = xs_sub2, xlens_sub2
                        return eouts
                if lth < len(self.layers) - 1:
                    if self.subsample_factors[lth] > 1:
                        xs, xlens = self.subsample_layers[lth](xs, xlens)
                        if 'relative' in self.pe_type:
                            xs, rel_pos_embs = self.pos_emb(xs)
                        xx_mask = self._mask(xlens, self.lookaheads[lth + 1])
                    elif self.lookaheads[lth] != self.lookaheads[lth + 1]:
                        xx_mask = self._mask(xlens, self.lookaheads[lth + 1])
        xs = ops.layer_norm(xs, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps)
        if self.bridge is not None:
            xs = ops.linear(xs, self.bridge.weight, self.bridge.bias)
        if task in ['all', 'ys']:
            eouts['ys']['xs'], eouts['ys']['xlens'] = xs, xlens
        if self.n_layers_sub1 >= 1 and task == 'all':
            eouts['ys_sub1']['xs'], eouts['ys_sub1']['xlens'] = xs_sub1, xlens_sub1
        if self.n_layers_sub2 >= 1 and task == 'all':
            eouts['ys_sub2']['xs'], eouts['ys_sub2']['xlens'] = xs_sub2, xlens_sub2
        return eouts


class ConformerEncoder(TransformerEncoder):
    """conformer.py:18-111"""

    def __init__(self, input_dim, enc_type, n_heads, kernel_size, normalization,
                 n_layers, n_layers_sub1, n_layers_sub2, d_model, d_ff, ffn_bottleneck_dim,
                 ffn_activation, pe_type, layer_norm_eps, last_proj_dim,
                 dropout_in, dropout, dropout_att, dropout_layer,
                 subsample, subsample_type, n_stacks, n_splices, frontend_conv,
                 task_specific_layer, param_init, clamp_len, lookahead,
                 chunk_size_left, chunk_size_current, chunk_size_right, streaming_type):
        self._conformer_cfg = (kernel_size, normalization, 'conformer_v2' in enc_type)
        if 'conformer_v2' not in enc_type:
            assert pe_type in ['relative', 'relative_xl']
        super().__init__(input_dim, enc_type, n_heads, n_layers, n_layers_sub1, n_layers_sub2,
                         d_model, d_ff, ffn_bottleneck_dim, ffn_activation, pe_type, layer_norm_eps,
                         last_proj_dim, dropout_in, dropout, dropout_att, dropout_layer,
                         subsample, subsample_type, n_stacks, n_splices, frontend_conv,
                         task_specific_layer, param_init, clamp_len, lookahead,
                         chunk_size_left, chunk_size_current, chunk_size_right, streaming_type)

    def _make_block(self, dropout_layer, d_model, d_ff, n_heads, dropout, dropout_att, layer_norm_eps,
                    ffn_activation, param_init, pe_type, clamp_len, ffn_bottleneck_dim):
        kernel_size, normalization, v2 = self._conformer_cfg
        causal = self.unidir or (self.streaming_type == 'mask')
        block = ConformerEncoderBlock_v2 if v2 else ConformerEncoderBlock
        return block(d_model, d_ff, n_heads, kernel_size, dropout, dropout_att, dropout_layer, layer_norm_eps,
                     ffn_activation, param_init, pe_type, clamp_len, ffn_bottleneck_dim, causal, normalization)


def build_encoder(args):
    """build.py:7-152 for the conv / transformer / conformer / (B)LSTM families."""
    if args.enc_type in ('tds', 'gated_conv') or 'gru' in args.enc_type:
        raise NotImplementedError('enc_type=%s: TDS / gated-conv / GRU encoders are not built' % args.enc_type)
    if 'conv' in args.enc_type:
        assert args.n_stacks == 1 and args.n_splices == 1
        conv = ConvEncoder(args.input_dim, in_channel=args.conv_in_channel, channels=args.conv_channels,
                           kernel_sizes=args.conv_kernel_sizes, strides=args.conv_strides,
                           poolings=args.conv_poolings, dropout=0., normalization=args.conv_normalization,
                           residual=False,
                           bottleneck_dim=args.transformer_enc_d_model if 'former' in args.enc_type else args.conv_bottleneck_dim,
                           param_init=args.param_init)
    else:
        conv = None
    if not hasattr(args, 'transformer_enc_d_model') and hasattr(args, 'transformer_d_model'):
        args.transformer_enc_d_model = args.transformer_d_model
        args.transformer_dec_d_model = args.transformer_d_model
    if not hasattr(args, 'transformer_enc_d_ff') and hasattr(args, 'transformer_d_ff'):
        args.transformer_enc_d_ff = args.transformer_d_ff
    if not hasattr(args, 'transformer_enc_n_heads') and hasattr(args, 'transformer_n_heads'):
        args.transformer_enc_n_heads = args.transformer_n_heads
    common = dict(
        input_dim=args.input_dim if args.input_type == 'speech' else args.emb_dim,
        enc_type=args.enc_type, n_heads=args.transformer_enc_n_heads if hasattr(args, 'transformer_enc_n_heads') else None,
        n_layers=args.enc_n_layers, n_layers_sub1=args.enc_n_layers_sub1, n_layers_sub2=args.enc_n_layers_sub2)
    if 'transformer' in args.enc_type:
        return TransformerEncoder(
            d_model=args.transformer_enc_d_model, d_ff=args.transformer_enc_d_ff,
            ffn_bottleneck_dim=args.transformer_ffn_bottleneck_dim,
            ffn_activation=args.transformer_ffn_activation, pe_type=args.transformer_enc_pe_type,
            layer_norm_eps=args.transformer_layer_norm_eps,
            last_proj_dim=args.transformer_dec_d_model if 'transformer' in args.dec_type else 0,
            dropout_in=args.dropout_in, dropout=args.dropout_enc, dropout_att=args.dropout_att,
            dropout_layer=args.dropout_enc_layer, subsample=args.subsample,
            subsample_type=args.subsample_type, n_stacks=args.n_stacks, n_splices=args.n_splices,
            frontend_conv=conv, task_specific_layer=args.task_specific_layer,
            param_init=args.transformer_param_init, clamp_len=args.transformer_enc_clamp_len,
            lookahead=args.transformer_enc_lookaheads, chunk_size_left=args.lc_chunk_size_left,
            chunk_size_current=args.lc_chunk_size_current, chunk_size_right=args.lc_chunk_size_right,
            streaming_type=args.lc_type, **common)
    elif 'conformer' in args.enc_type:
        return ConformerEncoder(
            kernel_size=args.conformer_kernel_size, normalization=args.conformer_normalization,
            d_model=args.transformer_enc_d_model, d_ff=args.transformer_enc_d_ff,
            ffn_bottleneck_dim=args.transformer_ffn_bottleneck_dim, ffn_activation='swish',
            pe_type=args.transformer_enc_pe_type, layer_norm_eps=args.transformer_layer_norm_eps,
            last_proj_dim=args.transformer_dec_d_model if 'transformer' in args.dec_type else 0,
            dropout_in=args.dropout_in, dropout=args.dropout_enc, dropout_att=args.dropout_att,
            dropout_layer=args.dropout_enc_layer, subsample=args.subsample,
            subsample_type=args.subsample_type, n_stacks=args.n_stacks, n_splices=args.n_splices,
            frontend_conv=conv, task_specific_layer=args.task_specific_layer,
            param_init=args.transformer_param_init, clamp_len=args.transformer_enc_clamp_len,
            lookahead=args.transformer_enc_lookaheads, chunk_size_left=args.lc_chunk_size_left,
            chunk_size_current=args.lc_chunk_size_current, chunk_size_right=args.lc_chunk_size_right,
            streaming_type=args.lc_type, **common)
    # build.py:127-150: everything else is the (B)LSTM encoder -- BASELINE configs[0] (TIMIT BLSTM-CTC)
    from neural_sp_amd.rnn_encoder import RNNEncoder
    return RNNEncoder(
        input_dim=args.input_dim if args.input_type == 'speech' else args.emb_dim, enc_type=args.enc_type,
        n_units=args.enc_n_units, n_projs=args.enc_n_projs,
        last_proj_dim=args.transformer_dec_d_model if 'transformer' in args.dec_type else 0,
        n_layers=args.enc_n_layers, n_layers_sub1=args.enc_n_layers_sub1, n_layers_sub2=args.enc_n_layers_sub2,
        dropout_in=args.dropout_in, dropout=args.dropout_enc, subsample=args.subsample,
        subsample_type=args.subsample_type, n_stacks=args.n_stacks, n_splices=args.n_splices, frontend_conv=conv,
        bidir_sum_fwd_bwd=args.bidirectional_sum_fwd_bwd, task_specific_layer=args.task_specific_layer,
        param_init=args.param_init, chunk_size_current=args.lc_chunk_size_left,  # (sic) build.py:146
        chunk_size_right=args.lc_chunk_size_right, cnn_lookahead=args.cnn_lookahead, rsp_prob=args.rsp_prob_enc)
