"""Attention-based RNN decoder (LAS) and MoChA, training side.

SURVEY.md section 8f: rank 1, second half -- `RNNDecoder.forward_att`
(neural_sp/models/seq2seq/decoders/las.py:618-776: BASELINE config 3's recipe uses the LSTM decoder
with location-aware attention) -- and rank 2 -- MoChA training
(modules/mocha/mocha.py:164-311, hma_train.py:12-67, mocha_train.py:13-58, quantity loss
las.py:735-745).  Parameter names and shapes are the reference's (checked against
reference-generated fixtures), so checkpoints are interchangeable.

How it is built.  The decoder is a teacher-forced loop over target positions (`for i in range(ymax)`,
las.py:667) whose step is a handful of [B, .]-sized operations: that loop stays a host loop.  Every piece of
arithmetic inside it is a kernel of libnsp_hip.so: `ops.linear` (key / query / location projections, LSTM gate GEMMs,
bottleneck, output layer), `ops.add_energy` (v . tanh / relu of the summed projections in one pass),
`ops.row_softmax` (masked, sharpened), `ops.head_context` (batched GEMM), `ops.lstm_cell`, and for MoChA the scan
kernels `ops.mono_alpha` / `ops.chunk_beta` (csrc/mocha.hip); the label-smoothed XE + accuracy is the fused
`ops.xe_lsm_loss` kernel and the CTC branch is `CTC` (ctc.hip).  What remains in torch is glue (concatenations,
the Gaussian noise on the monotonic energies, the scalar quantity / latency losses) and the test-time (hard) MoChA
path used by greedy decoding.  Everything that does not feed back into the recurrence is hoisted out of the loop:
the key projections are computed once, the bottleneck + tanh + output layer run once over all steps.

Not built (NotImplementedError): LM fusion / initialisation, MBR training,
multi-head / GMM / dot-family attention, MoChA with several heads, 1-d conv, DeCoT / latency losses,
StableEmit, streaming / linear-time / beam-search decoding (greedy search is built: `RNNDecoder.greedy`, with
MoChA's test-time hard attention).
"""
import math

import random

import numpy as np
import torch
import torch.nn as nn

from neural_sp_amd import ops
from neural_sp_amd.decoders import CTC, DecoderBase

NEG_INF = float(np.finfo(np.float32).min)


class AttentionMechanism(nn.Module):
    """modules/attention.py:11-181, atype 'add' | 'location', single head."""

    def __init__(self, kdim, qdim, adim, atype, sharpening_factor=1, sigmoid_smoothing=False,
                 conv_out_channels=10, conv_kernel_size=201, dropout=0., lookahead=2):
        super().__init__()
        assert conv_kernel_size % 2 == 1, "Kernel size should be odd for 'same' conv."
        if atype not in ('add', 'location', 'triggered_attention'):
            raise NotImplementedError('attn_type=%s (built: location, add, triggered_attention, mocha)' % atype)
        self.atype, self.adim, self.n_heads = atype, adim, 1
        self.lookahead = lookahead
        self.sharpening_factor = sharpening_factor
        self.sigmoid_smoothing = sigmoid_smoothing
        self.dropout = nn.Dropout(p=dropout)
        self.w_key = nn.Linear(kdim, adim)
        self.w_query = nn.Linear(qdim, adim, bias=False)
        if atype == 'location':
            self.w_conv = nn.Linear(conv_out_channels, adim, bias=False)
            self.conv = nn.Conv2d(1, conv_out_channels, kernel_size=(1, conv_kernel_size), stride=1,
                                  padding=(0, (conv_kernel_size - 1) // 2), bias=False)
        self.v = nn.Linear(adim, 1, bias=False)
        self.reset()

    def reset(self):
        self.key = None
        self.mask = None

    def forward(self, key, value, query, mask=None, aw_prev=None, cache=False, mode='', trigger_points=None,
                streaming=False):
        """key/value [B,T,kdim], query [B,1,qdim], mask [B,1,T] (True = valid), aw_prev [B,1,1,T] ->
        cv [B,1,vdim], aw [B,1,1,T], {} (attention.py:96-181)."""
        bs, klen = key.shape[:2]
        aw_prev = key.new_zeros(bs, 1, klen) if aw_prev is None else aw_prev.squeeze(1)
        if self.key is None or not cache:
            self.key = ops.linear(key, self.w_key.weight, self.w_key.bias)          # [B,T,adim], once per batch
            self.mask = mask
        q_proj = ops.linear(query, self.w_query.weight)                              # [B,1,adim]
        loc = None
        if self.atype == 'location':
            # Conv2d(1 -> ch, (1,k), 'same') over the previous attention weights as a GEMM on the k-wide
            # windows of the zero-padded signal: [B*T, k] x [k, ch] (attention.py:142-145)
            k = self.conv.weight.shape[-1]
            win = nn.functional.pad(aw_prev, ((k - 1) // 2, (k - 1) // 2)).unfold(-1, k, 1)   # [B,1,T,k]
            conv_feat = ops.linear(win.reshape(bs, klen, k), self.conv.weight.view(-1, k))     # [B,T,ch]
            loc = ops.linear(conv_feat, self.w_conv.weight)                           # [B,T,adim]
        # e = v . tanh(key + query (+ location)): one pass over the key projection (nsp_add_energy_fwd)
        e = ops.add_energy(self.key, q_proj, self.v.weight, 'tanh', loc).unsqueeze(1)  # [B,1,T]
        mask_ = self.mask
        if self.atype == 'triggered_attention':
            # attention.py:165-169: nothing beyond the token's CTC boundary + `lookahead` frames is attended to
            assert trigger_points is not None
            j = torch.arange(klen, device=key.device).view(1, 1, klen)
            reach = j <= (trigger_points.view(bs, 1, 1).long() + self.lookahead)
            mask_ = reach if mask_ is None else (reach & (mask_ != 0))
        if self.sigmoid_smoothing:
            if mask_ is not None:
                e = e.masked_fill(mask_ == 0, NEG_INF)
            s = torch.sigmoid(e)
            aw = s / s.sum(-1, keepdim=True)
        else:
            aw = ops.row_softmax(e, mask_, self.sharpening_factor)                    # masked, sharpened soft-max: one kernel
        aw = self.dropout(aw)
        cv = ops.head_context(aw.unsqueeze(1), value.unsqueeze(2)).view(bs, 1, value.shape[-1])
        return cv, aw.unsqueeze(1), {}


class _AddEnergy(nn.Module):
    """monotonic_energy.py / chunk_energy.py, atype 'add', one head: e = v(relu(W_k k + W_q q)) (+ r)."""

    def __init__(self, kdim, qdim, adim, monotonic, init_r=-4):
        super().__init__()
        self.w_key = nn.Linear(kdim, adim)
        self.w_query = nn.Linear(qdim, adim, bias=False)
        self.v = nn.Linear(adim, 1, bias=False)
        self.monotonic = monotonic
        if monotonic:
            self.r = nn.Parameter(torch.Tensor([init_r]))
            # nn.utils.weight_norm(self.v, name='weight', dim=0) with g = sqrt(1 / adim)
            # (monotonic_energy.py:70-72): parameters v.weight_g (shape [1] in reference checkpoints: :72 replaces
            # the [1,1] storage by a 1-element vector), v.weight_v [1,adim]
            w = self.v.weight.data
            del self.v._parameters['weight']
            self.v.register_parameter('weight_g', nn.Parameter(torch.Tensor([1.0 / adim]).sqrt()))
            self.v.register_parameter('weight_v', nn.Parameter(w.clone()))
        self.reset()

    def reset(self):
        self.key = None
        self.mask = None

    def v_weight(self):
        if not self.monotonic:
            return self.v.weight
        wv = self.v.weight_v
        return wv * (self.v.weight_g / wv.norm(dim=1, keepdim=True))

    def forward(self, key, query, mask, cache=False):
        """-> e [B,1,qlen=1,klen]"""
        if self.key is None or not cache:
            self.key = ops.linear(key, self.w_key.weight, self.w_key.bias)
            self.mask = mask
        e = ops.add_energy(self.key, ops.linear(query, self.w_query.weight), self.v_weight(), 'relu').unsqueeze(1)  # [B,1,T]
        if self.monotonic:
            e = e + self.r
        if self.mask is not None:
            e = e.masked_fill(self.mask == 0, NEG_INF)
        return e.unsqueeze(1)


class MoChA(nn.Module):
    """modules/mocha/mocha.py:20-311, training ('parallel') mode, one monotonic and one chunkwise head,
    additive energies: chunk_size 1 = hard monotonic attention, > 1 = MoChA, -1 = MILk."""

    def __init__(self, kdim, qdim, adim, odim, atype, chunk_size, n_heads_mono=1, n_heads_chunk=1, conv1d=False,
                 init_r=-4, eps=1e-6, noise_std=1.0, no_denominator=False, sharpening_factor=1.0, dropout=0.,
                 decot=False, decot_delta=2, stableemit_weight=0.0):
        super().__init__()
        if atype != 'add' or n_heads_mono != 1 or n_heads_chunk != 1 or conv1d:
            raise NotImplementedError('MoChA: built for additive energies, one head, no 1-d conv')
        self.decot, self.decot_delta = decot, decot_delta
        assert stableemit_weight >= 0
        self.stableemit_weight = stableemit_weight
        self._stableemit_weight = 0          # curriculum: trigger_stableemit() (mocha.py:97-99,157-159)
        self.w = chunk_size
        self.milk = chunk_size == -1
        self.n_heads, self.H_ma, self.H_ca, self.H_total = 1, 1, 1, 1
        self.eps, self.noise_std, self.no_denom = eps, noise_std, no_denominator
        self.sharpening_factor = sharpening_factor
        self.monotonic_energy = _AddEnergy(kdim, qdim, adim, True, init_r)
        self.chunk_energy = _AddEnergy(kdim, qdim, adim, False) if (chunk_size > 1 or self.milk) else None
        self.dropout_attn = nn.Dropout(p=dropout)

    def reset(self):
        self.monotonic_energy.reset()
        if self.chunk_energy is not None:
            self.chunk_energy.reset()

    def trigger_stableemit(self):
        self._stableemit_weight = self.stableemit_weight

    def forward(self, key, value, query, mask, aw_prev=None, cache=False, mode='parallel', trigger_points=None,
                streaming=False):
        if mode not in ('parallel', 'hard'):
            raise ValueError("mode must be 'parallel' or 'hard'.")
        bs, klen = key.shape[:2]
        if aw_prev is None:
            aw_prev = key.new_zeros(bs, 1, 1, klen)
            aw_prev[:, :, :, 0] = 1.0                                                 # [1, 0, 0, ...] (mocha.py:204-206)
        e_ma = self.monotonic_energy(key, query, mask, cache)                         # [B,1,1,T]
        if mode == 'hard':
            return self._forward_hard(key, value, query, mask, aw_prev, cache, e_ma)
        # parallel_monotonic_attention (hma_train.py:12-67) for qlen = 1
        if self.noise_std > 0:                                                        # (training AND eval, as the reference)
            e_ma = e_ma + torch.zeros_like(e_ma).normal_(std=self.noise_std)
        # p_choose (StableEmit scaling, hma_train.py:43-44) -> exclusive cumprod in log space -> alpha recurrence:
        # ONE scan kernel per direction (csrc/mocha.hip) instead of ~12 tensor ops
        alpha, p_choose = ops.mono_alpha(e_ma, aw_prev, self.eps, self.no_denom, self._stableemit_weight)
        if self.decot:
            # delay-constrained training (hma_train.py:59-63): nothing may be selected more than `decot_delta`
            # frames after the token's reference boundary
            assert trigger_points is not None
            j = torch.arange(klen, device=key.device).view(1, 1, 1, klen)
            alpha = alpha.masked_fill(j > (trigger_points.view(bs, 1, 1, 1).long() + self.decot_delta), 0)
        beta = None
        if self.chunk_energy is not None:
            # soft_chunkwise_attention (mocha_train.py:13-58)
            u = self.chunk_energy(key, query, mask, cache)
            beta = ops.chunk_beta(u, alpha, self.w, self.sharpening_factor)     # clamped exp, window sums, beta: one kernel
            beta = self.dropout_attn(beta)
        aw = alpha if self.w == 1 else beta                                      # [B,1,1,T]
        cv = ops.head_context(aw, value.unsqueeze(2)).view(bs, 1, value.shape[-1])  # batched MFMA GEMM (was torch.bmm)
        return cv, alpha, {'beta': beta, 'p_choose': p_choose}


def _mocha_forward_hard(self, key, value, query, mask, aw_prev, cache, e_ma):
    """Test-time MoChA (mocha.py:222-311 with linear_decoding / streaming off; hma_test.py:12-53,
    mocha_test.py:16-60): p_choose thresholded at 0.5, alpha = the first selected frame at or after the
    previous one, beta = softmax of the chunk energies over the w frames ending there.  Quirks kept:
    `is_boundary` is decided for the whole batch; an utterance without a boundary in a step where another one
    has one gets a softmax over an all-masked row, i.e. UNIFORM chunk attention (mocha_test.py:44-58)."""
    bs, klen = key.shape[:2]
    p_sel = (torch.sigmoid(e_ma) >= 0.5).to(e_ma.dtype) * torch.cumsum(aw_prev, dim=-1)          # hma_test.py:33-37
    excl = torch.cumprod(torch.cat([p_sel.new_ones(bs, 1, 1, 1), 1 - p_sel[..., :-1]], dim=-1), dim=-1)
    alpha = p_sel * excl
    is_boundary = bool((alpha.sum() > 0).item())
    beta = None
    if self.chunk_energy is not None:
        if not is_boundary:
            beta = alpha.new_zeros(bs, 1, 1, klen)
        else:
            u = self.chunk_energy(key, query, mask, cache)
            a = alpha[:, 0, 0]                                                                    # [B,T]
            has = a.sum(-1) > 0
            boundary = torch.argmax((a > 0).to(torch.int32), dim=-1)                              # first non-zero frame
            j = torch.arange(klen, device=key.device).unsqueeze(0)
            lo = torch.zeros_like(boundary) if self.milk else torch.clamp(boundary - self.w + 1, min=0)
            win = (j >= lo.unsqueeze(1)) & (j <= boundary.unsqueeze(1)) & has.unsqueeze(1)
            win = win | (a != 0)                                                                  # mask starts as alpha.byte()
            beta = torch.softmax(u.masked_fill(~win.view(bs, 1, 1, klen), NEG_INF), dim=-1)
            beta = self.dropout_attn(beta)
    cv = torch.bmm((alpha if self.w == 1 else beta).squeeze(1), value)
    return cv, alpha, {'beta': beta, 'p_choose': torch.sigmoid(e_ma)}


MoChA._forward_hard = _mocha_forward_hard


class RNNDecoder(DecoderBase):
    """decoders/las.py:36-776, training side (forward = CTC branch :465-479 + forward_att :618-776)."""

    def __init__(self, special_symbols, enc_n_units, attn_type, n_units, n_projs, n_layers, bottleneck_dim, emb_dim,
                 vocab, tie_embedding, attn_dim, attn_sharpening_factor, attn_sigmoid_smoothing,
                 attn_conv_out_channels, attn_conv_kernel_size, attn_n_heads, dropout, dropout_emb, dropout_att,
                 lsm_prob, ss_prob, ctc_weight, ctc_lsm_prob, ctc_fc_list, mbr_training, mbr_ce_weight, external_lm,
                 lm_fusion, lm_init, backward, global_weight, mtl_per_batch, param_init, mocha_chunk_size,
                 mocha_n_heads_mono, mocha_init_r, mocha_eps, mocha_std, mocha_no_denominator, mocha_1dconv,
                 mocha_decot_lookahead, quantity_loss_weight, latency_metric, latency_loss_weight,
                 mocha_stableemit_weight, gmm_attn_n_mixtures, replace_sos, distillation_weight, discourse_aware):
        super().__init__()
        for flag, what in ((mbr_training, 'MBR training'), (external_lm is not None or lm_fusion or lm_init, 'LM fusion / init'),
                           (attn_n_heads > 1, 'multi-head attention'),
                           (latency_metric == 'interval', "latency metric 'interval'"), (replace_sos, 'replace_sos'),
                           (bool(discourse_aware), 'discourse-aware training')):
            if flag:
                raise NotImplementedError('RNNDecoder: %s is not built' % what)
        self.eos, self.unk = special_symbols['eos'], special_symbols['unk']
        self.pad, self.blank = special_symbols['pad'], special_symbols['blank']
        self.vocab, self.attn_type, self.enc_n_units = vocab, attn_type, enc_n_units
        self.dec_n_units, self.n_projs, self.n_layers = n_units, n_projs, n_layers
        self.lsm_prob, self.ss_prob, self._ss_prob = lsm_prob, ss_prob, 0
        self.att_weight = global_weight - ctc_weight
        self.ctc_weight = ctc_weight
        self.bwd, self.mtl_per_batch = backward, mtl_per_batch
        self.quantity_loss_weight, self._quantity_loss_weight = quantity_loss_weight, 0
        self.latency_metric, self.latency_loss_weight, self._latency_loss_weight = latency_metric, latency_loss_weight, 0
        if 'ctc_sync' in latency_metric or attn_type == 'triggered_attention':
            assert 0 < self.ctc_weight < 1          # las.py:161-162
        self.aws_dict, self.data_dict = {}, {}
        if ctc_weight > 0:
            self.ctc = CTC(eos=self.eos, blank=self.blank, enc_n_units=enc_n_units, vocab=vocab, dropout=dropout,
                           lsm_prob=ctc_lsm_prob, fc_list=ctc_fc_list, param_init=param_init)
        if self.att_weight > 0:
            qdim = n_units if n_projs == 0 else n_projs
            if attn_type == 'mocha':
                self.score = MoChA(enc_n_units, qdim, attn_dim, enc_n_units, atype='add', chunk_size=mocha_chunk_size,
                                   n_heads_mono=mocha_n_heads_mono, init_r=mocha_init_r, eps=mocha_eps,
                                   noise_std=mocha_std, no_denominator=mocha_no_denominator, conv1d=mocha_1dconv,
                                   sharpening_factor=attn_sharpening_factor, decot='decot' in latency_metric,
                                   decot_delta=mocha_decot_lookahead, stableemit_weight=mocha_stableemit_weight)
            else:
                self.score = AttentionMechanism(enc_n_units, qdim, attn_dim, attn_type,
                                                sharpening_factor=attn_sharpening_factor,
                                                sigmoid_smoothing=attn_sigmoid_smoothing,
                                                conv_out_channels=attn_conv_out_channels,
                                                conv_kernel_size=attn_conv_kernel_size, dropout=dropout_att, lookahead=2)
            self.rnn = nn.ModuleList()
            dec_odim = enc_n_units + emb_dim
            self.proj = nn.ModuleList([nn.Linear(n_units, n_projs) for _ in range(n_layers)]) if n_projs > 0 else None
            self.dropout = nn.Dropout(p=dropout)
            for _ in range(n_layers):
                self.rnn.append(nn.LSTMCell(dec_odim, n_units))
                dec_odim = n_projs if n_projs > 0 else n_units
            self.output_bn = nn.Linear(dec_odim + enc_n_units, bottleneck_dim)
            self.embed = nn.Embedding(vocab, emb_dim, padding_idx=self.pad)
            self.dropout_emb = nn.Dropout(p=dropout_emb)
            assert bottleneck_dim > 0, 'bottleneck_dim must be larger than zero.'
            self.output = nn.Linear(bottleneck_dim, vocab)
            if tie_embedding:
                if emb_dim != bottleneck_dim:
                    raise ValueError('When using tied flag, n_units must be equal to emb_dim.')
                self.output.weight = self.embed.weight
        self.reset_parameters(param_init)

    def reset_parameters(self, param_init):
        """las.py:417-437: uniform everywhere except the weight-norm gain and the offset r of MoChA."""
        for n, p in self.named_parameters():
            if n.startswith('ctc.') or 'score.monotonic_energy.v.weight_g' in n or 'score.monotonic_energy.r' in n:
                continue
            if p.dim() == 1:
                nn.init.constant_(p, 0.)
            else:
                nn.init.uniform_(p, a=-param_init, b=param_init)

    def trigger_quantity_loss(self):
        self._quantity_loss_weight = self.quantity_loss_weight

    def trigger_scheduled_sampling(self):
        """las.py:351-353 (train.py switches it on at `ss_start_epoch`)"""
        self._ss_prob = self.ss_prob

    def forward(self, eouts, elens, ys, task='all', teacher_logits=None, recog_params={}, idx2token=None,
                trigger_points=None):
        observation = {'loss': None, 'loss_att': None, 'loss_ctc': None, 'loss_mbr': None,
                       'acc_att': None, 'ppl_att': None}
        loss = eouts.new_zeros((1,))
        ctc_trigger_points = None
        if self.ctc_weight > 0 and (task == 'all' or 'ctc' in task):
            # CTC-synchronous training (las.py:463-465): the reference boundaries of the latency loss are the
            # forced alignment of this very CTC branch (nsp_ctc_forced_align), recomputed every step
            loss_ctc, ctc_trigger_points = self.ctc(
                eouts, elens, ys, forced_align=('ctc_sync' in self.latency_metric and self.training)
                or self.attn_type == 'triggered_attention')
            observation['loss_ctc'] = loss_ctc.detach()
            loss = loss + (loss_ctc if self.mtl_per_batch else loss_ctc * self.ctc_weight)
        forced = None
        if (self.latency_metric in ['minlt', 'decot', 'decot_ctc_sync'] or self.attn_type == 'triggered_attention') \
                and trigger_points is not None:
            forced = ops.h2d(np.asarray(trigger_points, dtype=np.int32), eouts.device)      # batch['trigger_points'] (:471-472)
        if self.att_weight > 0 and (task == 'all' or 'ctc' not in task):
            loss_att, acc_att, ppl_att, loss_quantity, loss_latency = self.forward_att(
                eouts, elens, ys, ctc_trigger_points=ctc_trigger_points, forced_trigger_points=forced)
            observation['loss_att'] = loss_att.detach()
            observation['acc_att'] = acc_att
            observation['ppl_att'] = ppl_att
            if self.attn_type == 'mocha':
                if self._quantity_loss_weight > 0:
                    loss_att = loss_att + loss_quantity * self._quantity_loss_weight
                observation['loss_quantity'] = loss_quantity.detach()
            if self.latency_metric:
                if self._latency_loss_weight > 0:
                    loss_att = loss_att + loss_latency * self._latency_loss_weight
                observation['loss_latency'] = loss_latency.detach() if self.training else 0
            loss = loss + (loss_att if self.mtl_per_batch else loss_att * self.att_weight)
        observation['loss'] = loss.detach()
        return loss, observation

    # ---- one decoder step (las.py:778-857)
    def _recurrency(self, x, hxs, cxs):
        new_h, new_c = [], []
        dout = x
        dout_score = None
        for l, cell in enumerate(self.rnn):
            # (the sum of the two gate GEMMs is the residual epilogue of the second one, not a separate add)
            gates = ops.linear(hxs[l], cell.weight_hh, cell.bias_hh, res=ops.linear(dout, cell.weight_ih, cell.bias_ih))
            h, c = ops.lstm_cell(gates, cxs[l])        # gate non-linearities + cell update: one kernel per direction
            new_h.append(h)
            new_c.append(c)
            dout = self.dropout(h)
            if self.proj is not None:
                dout = torch.relu(ops.linear(dout, self.proj[l].weight, self.proj[l].bias))
            if l == 0:
                dout_score = dout                       # the FIRST layer's output scores the attention
        return new_h, new_c, dout_score, dout

    def forward_att(self, eouts, elens, ys, ctc_trigger_points=None, forced_trigger_points=None):
        """-> (loss [1], acc (device scalar, %), ppl (device scalar), quantity loss, latency loss (device scalars))"""
        dev = eouts.device
        B, T = eouts.shape[:2]
        ylens = [len(y) + 1 for y in ys]
        L = max(ylens)
        ys_in = np.full((B, L), self.pad, dtype=np.int64)      # append_sos_eos (torch_utils.py:97-126), sos = eos
        ys_out = np.full((B, L), self.pad, dtype=np.int32)
        for b, y in enumerate(ys):
            yy = y[::-1] if self.bwd else y
            ys_in[b, 0] = self.eos
            ys_in[b, 1:len(yy) + 1] = yy
            ys_out[b, :len(yy)] = yy
            ys_out[b, len(yy)] = self.eos
        if forced_trigger_points is not None:
            forced_trigger_points = forced_trigger_points.clone()
            for b in range(B):
                forced_trigger_points[b, ylens[b] - 1] = int(elens[b]) - 1     # the boundary of <eos> (las.py:647-649)
        ys_in_d = ops.h2d(ys_in, dev)
        ys_out_d = ops.h2d(ys_out.reshape(-1), dev)
        elens_d = ops.h2d(elens, dev, torch.int64)
        src_mask = (torch.arange(T, device=dev).unsqueeze(0) < elens_d.unsqueeze(1)).unsqueeze(1)   # [B,1,T]
        hxs = [eouts.new_zeros(B, self.dec_n_units) for _ in range(self.n_layers)]
        cxs = [eouts.new_zeros(B, self.dec_n_units) for _ in range(self.n_layers)]
        cv = eouts.new_zeros(B, 1, self.enc_n_units)
        self.score.reset()
        aw, aws = None, []
        ys_emb = self.dropout_emb(self.embed(ys_in_d))           # [B,L,emb]
        douts, cvs = [], []
        # The recurrence runs its (tiny: M = B rows) GEMMs in the exact-fp32 MFMA mode whatever the ambient
        # mode: bf16 operands buy nothing at this size (launch-bound) and their rounding accumulates through
        # the L-step feedback of cv / aw (chunk-energy gradients of the XS fixture: cosine 0.975 in bf16).
        # The one large GEMM of the decoder, the output layer over V, stays in the ambient mode below.
        with ops.compute_mode('f32'):
            for i in range(L):
                # scheduled sampling (las.py:668,675-676): with probability ss_prob the step is fed the arg-max of
                # the model's own previous output distribution instead of the reference token.  Python's global
                # `random` stream, drawn only for i > 0 and only once sampling has been triggered -- in eval-mode
                # forwards too -- exactly as the reference does (same seed => same steps sampled).
                is_sample = i > 0 and self._ss_prob > 0 and random.random() < self._ss_prob
                if is_sample:
                    with torch.no_grad():
                        feat = torch.cat([douts[-1], cvs[-1]], dim=-1)
                        prev = torch.tanh(ops.linear(feat, self.output_bn.weight, self.output_bn.bias))
                        y_prev = ops.argmax_rows(ops.linear(prev, self.output.weight, self.output.bias)).long()
                    y_emb = self.dropout_emb(self.embed(y_prev))
                else:
                    y_emb = ys_emb[:, i]
                x = torch.cat([y_emb, cv.squeeze(1)], dim=-1)
                hxs, cxs, dout_score, dout_gen = self._recurrency(x, hxs, cxs)
                cv, aw, _ = self.score(eouts, eouts, dout_score.unsqueeze(1), src_mask, aw, cache=True, mode='parallel',
                                       trigger_points=forced_trigger_points[:, i:i + 1] if forced_trigger_points is not None else None)
                douts.append(dout_gen)
                cvs.append(cv.squeeze(1))
                aws.append(aw)
        self.score.reset()
        # generate() of every step at once (las.py:859-891 does it inside the loop; nothing of it feeds back)
        feats = torch.cat([torch.stack(douts, dim=1), torch.stack(cvs, dim=1)], dim=-1)          # [B,L,dec+enc]
        attn_v = torch.tanh(ops.linear(feats, self.output_bn.weight, self.output_bn.bias))
        logits = ops.linear(attn_v, self.output.weight, self.output.bias)
        lsm = self.lsm_prob if self.training else 0.0
        loss, loss_rows, correct = ops.xe_lsm_loss(logits, ys_out_d, lsm, self.pad, B)
        n_tokens = float(sum(ylens))
        ppl = torch.exp(loss_rows.sum() / n_tokens)
        acc = correct.sum().float() * (100.0 / n_tokens)
        loss_quantity = eouts.new_zeros(())
        loss_latency = eouts.new_zeros(())
        have_points = ctc_trigger_points is not None or forced_trigger_points is not None
        if self.attn_type == 'mocha' or have_points:
            aws_t = torch.cat(aws, dim=2)                                                        # [B,1,L,T]
            tgt_mask = (ys_out_d.view(B, L) != self.pad)
            aws_t = aws_t.masked_fill(~tgt_mask.view(B, 1, L, 1), 0)                             # attention padding (:724-726)
        if self.attn_type == 'mocha':
            n_pred = aws_t.sum(3).sum(2).sum(1) / aws_t.shape[1]
            loss_quantity = torch.mean(torch.abs(n_pred - tgt_mask.sum(1).float()))              # :731-736
        if ctc_trigger_points is not None or ('ctc_sync' not in self.latency_metric and forced_trigger_points is not None):
            # CTC-synchronous / minimum-latency / delay-constrained training (las.py:757-769): distance between the
            # expected attended frame of every token and its reference boundary (pad positions: 0 against 0)
            points = ctc_trigger_points if 'ctc_sync' in self.latency_metric else forced_trigger_points
            js = torch.arange(T, dtype=torch.float32, device=dev).view(1, 1, 1, T)
            exp_points = (js * aws_t).sum(3)                                                     # [B,1,L]
            loss_latency = torch.abs(exp_points - points[:, :L].float().unsqueeze(1)).sum() / float(sum(ylens))
        return loss, acc, ppl, loss_quantity, loss_latency

    def _plot_attention(self, save_path=None, n_cols=1):
        pass

    def _plot_ctc(self, save_path=None, topk=10):
        if self.ctc_weight > 0:
            self.ctc._plot_ctc(save_path, topk)

    def greedy(self, eouts, elens, max_len_ratio, idx2token=None, exclude_eos=False, refs_id=None, utt_ids=None,
               speakers=None, trigger_points=None):
        """las.py:893-1007 (what validate() runs with recog_beam_width 1): the teacher-forced step of
        forward_att with the arg-max token fed back; stops when every utterance has emitted <eos> or after
        ceil(T * max_len_ratio) steps.  -> (hyps: list of int arrays, None); the attention-weight plots of the
        reference's second return value are not produced.  MoChA decodes with its test-time (hard) attention."""
        dev = eouts.device
        B, T = eouts.shape[:2]
        with torch.no_grad():
            elens_d = ops.h2d(elens, dev, torch.int64)
            src_mask = (torch.arange(T, device=dev).unsqueeze(0) < elens_d.unsqueeze(1)).unsqueeze(1)
            hxs = [eouts.new_zeros(B, self.dec_n_units) for _ in range(self.n_layers)]
            cxs = [eouts.new_zeros(B, self.dec_n_units) for _ in range(self.n_layers)]
            cv = eouts.new_zeros(B, 1, self.enc_n_units)
            self.score.reset()
            aw = None
            y = torch.full((B,), self.eos, dtype=torch.int64, device=dev)
            hyps_batch = []
            ylens = [0] * B
            eos_flags = [False] * B
            ymax = int(math.ceil(T * max_len_ratio))
            tp = None
            if self.attn_type == 'triggered_attention':
                assert trigger_points is not None          # las.py:920-921
                tp = ops.h2d(np.asarray(trigger_points, dtype=np.int32), dev)
            with ops.compute_mode('f32'):
                for i in range(ymax):
                    x = torch.cat([self.dropout_emb(self.embed(y)), cv.squeeze(1)], dim=-1)
                    hxs, cxs, dout_score, dout_gen = self._recurrency(x, hxs, cxs)
                    cv, aw, _ = self.score(eouts, eouts, dout_score.unsqueeze(1), src_mask, aw, cache=True, mode='hard',
                                           trigger_points=tp[:, i:i + 1] if tp is not None and i < tp.shape[1] else None)
                    attn_v = torch.tanh(ops.linear(torch.cat([dout_gen, cv.squeeze(1)], dim=-1),
                                                   self.output_bn.weight, self.output_bn.bias))
                    y = ops.argmax_rows(ops.linear(attn_v, self.output.weight, self.output.bias)).long()
                    hyps_batch.append(y)
                    yh = y.tolist()                         # the reference syncs here too (y[b].item())
                    for b in range(B):
                        if not eos_flags[b]:
                            if yh[b] == self.eos:
                                eos_flags[b] = True
                            ylens[b] += 1                   # include <eos>
                    if all(eos_flags) or i == ymax - 1:
                        break
            self.score.reset()
            hb = torch.stack(hyps_batch, dim=1).cpu().numpy()
        hyps = [hb[b, :ylens[b]][::-1] if self.bwd else hb[b, :ylens[b]] for b in range(B)]
        if exclude_eos:
            hyps = [(h[1:] if self.bwd else h[:-1]) if eos_flags[b] else h for b, h in enumerate(hyps)]
        return hyps, None

    def beam_search(self, *a, **k):
        raise NotImplementedError('attention-decoder decoding is inference-side and not built')
