"""(B)LSTM encoder of neural_sp/models/seq2seq/encoders/rnn.py on the HIP kernels -- BASELINE.json configs[0]
(TIMIT BLSTM-CTC, examples/timit/s5/conf/blstm_ctc.yaml) and the `blstm` / `lstm` / `conv_blstm` / `conv_lstm`
encoder family of the reference's recipes, full-context training path (rnn.py:268-383).

What the reference does with `pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence` per layer
(rnn.py:534-541) is done here with left-to-right runs of the LSTM kernels over the padded batch:
  * forward direction: LSTM over `[B,T,.]`, frames past each utterance's end zeroed afterwards (the states
    computed there are never used and receive no gradient);
  * backward direction: every utterance is reversed inside ITS OWN length (`nsp_time_flip_mask`), run left to
    right with the `_reverse` parameters, and reversed back -- exactly a packed backward LSTM that starts from a
    zero state at each utterance's last frame;
  * both results are written into the halves of one `[B,T,2H]` buffer (`ops.bidir_merge`).
The reference sorts the batch by length for packing and un-sorts at the end; utterances never interact, so the
sort is not reproduced.  Parameter names are those of the reference (`enc.rnn.{l}.weight_ih_l0[_reverse]`, ...),
so checkpoints are interchangeable.

The reference's BLSTM recipes (`lc_chunk_size_right: 40`, `lc_chunk_size_left: -1`) select a second variant, the
"latency-controlled" encoder in full-context mode (rnn.py:104,385-425): separate `rnn` / `rnn_bwd` unidirectional
LSTMs over the whole padded batch, no packing, no masking -- built as `_lstm_layer_full_context`.

With `lc_chunk_size_left > 0` the same encoder trains in chunks (`_forward_latency_controlled`, rnn.py:427-510): the
forward LSTM carries its state from chunk to chunk (`ops.lstm_state`: the step kernels started from a given state,
differentiable through it), the backward LSTM sees the chunk plus N_r frames of right context.

Not built (NotImplementedError): GRU cells, streaming inference / random state passing, blockwise CNN
(`cnn_lookahead=False`), the NiN layers.
"""
import logging
import math

import numpy as np
import torch
import torch.nn as nn

from neural_sp_amd import ops
from neural_sp_amd.encoders import (AddSubsampler, ConcatSubsampler, Conv1dSubsampler, DropSubsampler, EncoderBase,
                                    MaxPoolSubsampler, MeanPoolSubsampler)
from neural_sp_amd.modules import init_with_uniform

logger = logging.getLogger(__name__)


class RNNEncoder(EncoderBase):
    """rnn.py:35-525 (full-context path)."""

    def __init__(self, input_dim, enc_type, n_units, n_projs, last_proj_dim, n_layers, n_layers_sub1,
                 n_layers_sub2, dropout_in, dropout, subsample, subsample_type, n_stacks, n_splices,
                 frontend_conv, bidir_sum_fwd_bwd, task_specific_layer, param_init, chunk_size_current,
                 chunk_size_right, cnn_lookahead, rsp_prob):
        super().__init__()
        subsamples = [1] * n_layers
        for lth, s in enumerate(list(map(int, subsample.split('_')[:n_layers]))):
            subsamples[lth] = s
        if n_layers_sub1 < 0 or (n_layers_sub1 > 1 and n_layers < n_layers_sub1):
            raise Warning('Set n_layers_sub1 between 1 to n_layers.')
        if n_layers_sub2 < 0 or (n_layers_sub2 > 1 and n_layers_sub1 < n_layers_sub2):
            raise Warning('Set n_layers_sub2 between 1 to n_layers_sub1.')
        if 'gru' in enc_type:
            raise NotImplementedError('GRU encoder cells')
        self.enc_type = enc_type
        self.bidirectional = 'blstm' in enc_type
        self.n_units = n_units
        self.n_dirs = 2 if self.bidirectional else 1
        self.n_layers = n_layers
        self.bidir_sum = bidir_sum_fwd_bwd
        self.N_c = int(str(chunk_size_current).split('_')[0]) // n_stacks
        self.N_r = int(str(chunk_size_right).split('_')[0]) // n_stacks
        # rnn.py:104: what the BLSTM recipes of the reference actually run -- `lc_chunk_size_right: 40` with the default
        # `lc_chunk_size_left: -1` selects the "latency-controlled" encoder in its FULL-CONTEXT mode (N_c <= 0,
        # rnn.py:385-425, "pre-training of the LC-BLSTM"): separate forward / backward unidirectional LSTMs
        # (`rnn`, `rnn_bwd`) run over the whole PADDED batch without packing -- the backward direction starts in the
        # padding, nothing is masked -- instead of one packed bidirectional nn.LSTM.  N_c > 0 is the chunked
        # (streaming) training of `_forward_latency_controlled`.
        self.lc_bidir = (self.N_c > 0 or self.N_r > 0) and self.bidirectional
        if self.lc_bidir:
            assert enc_type not in ['lstm', 'conv_lstm']
            assert n_layers_sub2 == 0
        if rsp_prob > 0:
            raise NotImplementedError('random state passing')
        self.n_layers_sub1 = n_layers_sub1
        self.n_layers_sub2 = n_layers_sub2
        self.task_specific_layer = task_specific_layer
        self.bridge = None
        self.bridge_sub1 = None
        self.bridge_sub2 = None
        self.dropout_in_p = dropout_in
        self.dropout_p = dropout
        self.conv = frontend_conv
        self._odim = self.conv.output_dim if self.conv is not None else input_dim * n_splices * n_stacks
        if not cnn_lookahead:
            raise NotImplementedError('cnn_lookahead=False belongs to the latency-controlled encoder')
        if enc_type != 'conv':
            self.rnn = nn.ModuleList()
            if self.lc_bidir:
                self.rnn_bwd = nn.ModuleList()
            self.proj = nn.ModuleList() if n_projs > 0 else None
            self.subsample = nn.ModuleList() if np.prod(subsamples) > 1 else None
            for lth in range(n_layers):
                if self.lc_bidir:
                    self.rnn += [nn.LSTM(self._odim, n_units, 1, batch_first=True)]
                    self.rnn_bwd += [nn.LSTM(self._odim, n_units, 1, batch_first=True)]
                else:
                    self.rnn += [nn.LSTM(self._odim, n_units, 1, batch_first=True, bidirectional=self.bidirectional)]
                self._odim = n_units if bidir_sum_fwd_bwd else n_units * self.n_dirs
                for sub, n_sub in (('sub1', n_layers_sub1), ('sub2', n_layers_sub2)):
                    if lth == n_sub - 1 and task_specific_layer:
                        setattr(self, 'layer_' + sub, nn.Linear(self._odim, n_units))
                        setattr(self, '_odim_' + sub, n_units)
                        if last_proj_dim > 0 and last_proj_dim != self.output_dim:
                            setattr(self, 'bridge_' + sub, nn.Linear(n_units, last_proj_dim))
                            setattr(self, '_odim_' + sub, last_proj_dim)
                if self.proj is not None and lth != n_layers - 1:
                    self.proj += [nn.Linear(self._odim, n_projs)]
                    self._odim = n_projs
                if np.prod(subsamples) > 1:
                    self.subsample += [{
                        'max_pool': lambda f, d: MaxPoolSubsampler(f), 'mean_pool': lambda f, d: MeanPoolSubsampler(f),
                        'concat': lambda f, d: ConcatSubsampler(f, d), 'drop': lambda f, d: DropSubsampler(f),
                        'conv1d': lambda f, d: Conv1dSubsampler(f, d), 'add': lambda f, d: AddSubsampler(f),
                    }[subsample_type](subsamples[lth], self._odim)]
            if last_proj_dim > 0 and last_proj_dim != self.output_dim:
                self.bridge = nn.Linear(self._odim, last_proj_dim)
                self._odim = last_proj_dim
        self.conv_factor = self.conv.subsampling_factor if self.conv is not None else 1
        self._factor = self.conv_factor * int(np.prod(subsamples))
        for n, p in self.named_parameters():          # rnn.py:256-262
            if 'conv' in n:
                continue
            init_with_uniform(n, p, param_init)
        self.reset_cache()

    def reset_cache(self):
        self.hx_fwd = [None] * self.n_layers

    def _lstm_layer_full_context(self, xs, full_dev, rnn, rnn_bwd):
        """rnn.py:404-411: flip(rnn_bwd(flip(xs))) and rnn(xs) over the padded batch; `full_dev` holds T for every
        utterance, so nsp_time_flip_mask reverses the whole time axis and masks nothing."""
        y_f = ops.lstm(xs, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)
        y_r = ops.lstm(ops.time_flip_mask(xs, full_dev, True), rnn_bwd.weight_ih_l0, rnn_bwd.weight_hh_l0,
                       rnn_bwd.bias_ih_l0, rnn_bwd.bias_hh_l0)
        if self.bidir_sum:
            return ops.add(y_f, ops.time_flip_mask(y_r, full_dev, True))
        return ops.bidir_merge(y_f, y_r, full_dev)

    def _forward_latency_controlled(self, xs, xlens, N_c, N_r):
        """rnn.py:427-510, training path: the utterance is cut into chunks of N_c frames, each seen with N_r frames of
        right context.  Per chunk and layer: the backward LSTM runs over chunk + context from a zero state; the forward
        LSTM continues from the state it reached at the end of the previous chunk's N_c frames (`ops.lstm_state`, the
        state is trained through), a second run from there covers the right-context frames (they only feed the next
        layer's backward direction).  Only the first N_c (subsampled) frames of every chunk are emitted."""
        bs, xmax, _ = xs.size()
        n_chunks = math.ceil(xmax / N_c)
        xlens_sub1 = xlens.clone() if self.n_layers_sub1 > 0 else None
        xs_chunks, xs_chunks_sub1 = [], []
        H = self.n_units
        states = [(xs.new_zeros(bs, H), xs.new_zeros(bs, H)) for _ in range(self.n_layers)]
        for chunk_idx, t in enumerate(range(0, N_c * n_chunks, N_c)):
            xs_chunk = xs[:, t:t + (N_c + N_r)].contiguous()
            _N_c = N_c
            for lth in range(self.n_layers):
                rnn, rnn_bwd = self.rnn[lth], self.rnn_bwd[lth]
                full = ops.h2d(torch.full((bs,), xs_chunk.size(1), dtype=torch.int32), xs.device, torch.int32)
                y_r = ops.lstm(ops.time_flip_mask(xs_chunk, full, True), rnn_bwd.weight_ih_l0, rnn_bwd.weight_hh_l0,
                               rnn_bwd.bias_ih_l0, rnn_bwd.bias_hh_l0)
                w = (rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)
                h0, c0 = states[lth]
                if xs_chunk.size(1) <= _N_c:                   # last chunk
                    y_f, h1, c1 = ops.lstm_state(xs_chunk, *w, h0, c0)
                    states[lth] = (h1, c1)
                else:
                    y_1, h1, c1 = ops.lstm_state(xs_chunk[:, :_N_c].contiguous(), *w, h0, c0)
                    states[lth] = (h1, c1)
                    y_2, _, _ = ops.lstm_state(xs_chunk[:, _N_c:].contiguous(), *w, h1, c1)
                    y_f = torch.cat([y_1, y_2], dim=1)
                if self.bidir_sum:
                    xs_chunk = ops.add(y_f, ops.time_flip_mask(y_r, full, True))
                else:
                    xs_chunk = ops.bidir_merge(y_f, y_r, full)
                xs_chunk = ops.dropout(xs_chunk, self.dropout_p, self.training)
                if lth == self.n_layers_sub1 - 1:
                    xs_chunks_sub1.append(xs_chunk[:, :_N_c])
                    if chunk_idx == 0:
                        xlens_sub1 = xlens.clone()
                if self.proj is not None and lth != self.n_layers - 1:
                    xs_chunk = ops.linear(xs_chunk, self.proj[lth].weight, self.proj[lth].bias, act='relu')
                if self.subsample is not None:
                    xs_chunk, xlens_tmp = self.subsample[lth](xs_chunk, xlens)
                    if chunk_idx == 0:
                        xlens = xlens_tmp
                    _N_c = _N_c // self.subsample[lth].factor
            xs_chunks.append(xs_chunk[:, :_N_c])
        xs = torch.cat(xs_chunks, dim=1)
        xs_sub1 = None
        if self.n_layers_sub1 > 0:
            xs_sub1 = self.sub_module(torch.cat(xs_chunks_sub1, dim=1), xlens_sub1, 'sub1')
        return xs, xlens, xs_sub1

    def _lstm_layer(self, xs, lens_dev, rnn):
        """Padding.forward (rnn.py:534-547) for one (bidirectional) layer."""
        y_f = ops.lstm(xs, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)
        if not self.bidirectional:
            return ops.time_flip_mask(y_f, lens_dev, False)
        y_r = ops.lstm(ops.time_flip_mask(xs, lens_dev, True), rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse,
                       rnn.bias_ih_l0_reverse, rnn.bias_hh_l0_reverse)
        if self.bidir_sum:
            return ops.add(ops.time_flip_mask(y_f, lens_dev, False), ops.time_flip_mask(y_r, lens_dev, True))
        return ops.bidir_merge(y_f, y_r, lens_dev)

    def sub_module(self, xs, xlens, module='sub1'):
        """rnn.py:512-524"""
        xs_sub = xs
        if self.task_specific_layer:
            layer = getattr(self, 'layer_' + module)
            xs_sub = ops.linear(xs, layer.weight, layer.bias, act='relu',
                                dropout_p=self.dropout_p if self.training else 0.0)
        bridge = getattr(self, 'bridge_' + module)
        if bridge is not None:
            xs_sub = ops.linear(xs_sub, bridge.weight, bridge.bias)
        return xs_sub, xlens.clone()

    def forward(self, xs, xlens, task, streaming=False, lookback=False, lookahead=False):
        """xs `[B,T,input_dim]` fp32 on the device, xlens (list / IntTensor, CPU) -> the eouts dict of rnn.py:268-383"""
        if streaming:
            raise NotImplementedError('streaming encoding is inference-only (out of scope)')
        eouts = {'ys': {'xs': None, 'xlens': None},
                 'ys_sub1': {'xs': None, 'xlens': None},
                 'ys_sub2': {'xs': None, 'xlens': None}}
        xlens = torch.IntTensor([int(x) for x in xlens])
        xs = ops.dropout(xs, self.dropout_in_p, self.training)
        if self.conv is not None:
            xs, xlens = self.conv(xs, xlens, lookback=lookback, lookahead=lookahead)
            if self.enc_type == 'conv':
                eouts['ys']['xs'], eouts['ys']['xlens'] = xs, xlens
                return eouts
        self.reset_cache()
        xs_sub = {}
        chunked = self.lc_bidir and self.N_c > 0
        if chunked:
            xs, xlens, sub1 = self._forward_latency_controlled(xs, xlens, self.N_c // self.conv_factor,
                                                               self.N_r // self.conv_factor)
            if sub1 is not None:
                xs_sub['sub1'] = sub1
                if task == 'ys_sub1':
                    eouts[task]['xs'], eouts[task]['xlens'] = sub1
                    return eouts
        for lth in ([] if chunked else range(self.n_layers)):      # full-context variants: layer-major loop
            if self.lc_bidir:
                full = torch.full((xs.size(0),), xs.size(1), dtype=torch.int32)
                xs = self._lstm_layer_full_context(xs, ops.h2d(full, xs.device, torch.int32), self.rnn[lth],
                                                   self.rnn_bwd[lth])
            else:
                if xs.size(1) > int(xlens.max()):
                    xs = xs[:, :int(xlens.max())].contiguous()      # pad_packed_sequence returns max(xlens) frames
                lens_dev = ops.h2d(xlens, xs.device, torch.int32)
                xs = self._lstm_layer(xs, lens_dev, self.rnn[lth])
            xs = ops.dropout(xs, self.dropout_p, self.training)
            for sub, n_sub in (('sub1', self.n_layers_sub1), ('sub2', self.n_layers_sub2)):
                if lth == n_sub - 1:
                    xs_sub[sub] = self.sub_module(xs, xlens, sub)
                    if task == 'ys_' + sub:
                        eouts[task]['xs'], eouts[task]['xlens'] = xs_sub[sub]
                        return eouts
            if self.proj is not None and lth != self.n_layers - 1:
                xs = ops.linear(xs, self.proj[lth].weight, self.proj[lth].bias, act='relu')
            if self.subsample is not None:
                xs, xlens = self.subsample[lth](xs, xlens)
        if self.bridge is not None:
            xs = ops.linear(xs, self.bridge.weight, self.bridge.bias)
        if xs.size(1) > int(xlens.max()):
            xs = xs[:, :int(xlens.max())].contiguous()
        if task in ['all', 'ys']:
            eouts['ys']['xs'], eouts['ys']['xlens'] = xs, xlens
        for sub in ('sub1', 'sub2'):
            if sub in xs_sub and task == 'all':
                eouts['ys_' + sub]['xs'], eouts['ys_' + sub]['xlens'] = xs_sub[sub]
        return eouts
