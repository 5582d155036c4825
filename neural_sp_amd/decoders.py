"""Training-side decoders of the Speech2Text hot path: CTC and RNN-Transducer losses.

Mirrors neural_sp/models/seq2seq/decoders/ctc.py (CTC.__init__ :51-103, forward :105-137,
loss_fn :139-150, CTCForcedAligner :628-753), rnn_transducer.py (RNNTransducer.__init__
:60-132, forward :174-215, forward_transducer :217-260, joint :262-276, recurrency
:278-311) and build.py.  Beam search / greedy decoding (ctc.py:219-531,
rnn_transducer.py:330-819) is inference and out of scope.
"""
import logging
import os
from collections import OrderedDict

import math

import numpy as np
import torch
import torch.nn as nn

import copy
from itertools import groupby

from neural_sp_amd import ops

logger = logging.getLogger(__name__)


class DecoderBase(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def device_id(self):
        return torch.cuda.device_of(next(self.parameters())).idx

    def reset_session(self):
        pass

    def trigger_scheduled_sampling(self):
        pass

    def trigger_quantity_loss(self):
        pass

    def trigger_latency_loss(self):
        # decoder_base.py:40-43
        if getattr(self, 'attn_type', '') == 'mocha':
            self._latency_loss_weight = getattr(self, 'latency_loss_weight', 0)

    def trigger_stableemit(self):
        # decoder_base.py:45-50
        if getattr(self, 'attn_type', '') == 'mocha' and hasattr(self, 'score'):
            self.score.trigger_stableemit()


def _labels_to_device(ys, device, pad=0):
    """list of label lists -> (labels int32 [B,Lmax] on device, ylens int32 on device, ylens list)."""
    ylens = [len(y) for y in ys]
    Lmax = max(1, max(ylens) if ylens else 1)
    lab = np.full((len(ys), Lmax), pad, dtype=np.int32)
    for b, y in enumerate(ys):
        lab[b, :len(y)] = np.asarray(y, dtype=np.int32)
    return (ops.h2d(lab, device), ops.h2d(np.asarray(ylens, dtype=np.int32), device), ylens)


class CTC(DecoderBase):
    """ctc.py:35-150: output head (Linear or fc-stack) + CTC loss (+ label smoothing)."""

    def __init__(self, eos, blank, enc_n_units, vocab, dropout=0., lsm_prob=0., fc_list=None,
                 param_init=0.1, backward=False):
        super().__init__()
        self.eos = eos
        self.blank = blank
        self.vocab = vocab
        self.lsm_prob = lsm_prob
        self.bwd = backward
        self.dropout_p = dropout
        self.prob_dict = {}
        self.data_dict = {}
        if fc_list is not None and len(fc_list) > 0:
            _fc_list = [int(fc) for fc in fc_list.split('_')]
            fc_layers = OrderedDict()
            for i in range(len(_fc_list)):
                input_dim = enc_n_units if i == 0 else _fc_list[i - 1]
                fc_layers['fc' + str(i)] = nn.Linear(input_dim, _fc_list[i])
                fc_layers['dropout' + str(i)] = nn.Dropout(p=dropout)
            fc_layers['fc' + str(len(_fc_list))] = nn.Linear(_fc_list[-1], vocab)
            self.output = nn.Sequential(fc_layers)
        else:
            self.output = nn.Linear(enc_n_units, vocab)
        self.forced_aligner = CTCForcedAligner(blank=blank)

    def logits(self, eouts):
        """self.output(eouts) of ctc.py:124 with the Dropout layers fused into the GEMM epilogues."""
        if isinstance(self.output, nn.Linear):
            return ops.linear(eouts, self.output.weight, self.output.bias)
        xs = eouts
        fcs = [m for m in self.output if isinstance(m, nn.Linear)]
        for i, fc in enumerate(fcs):
            last = i == len(fcs) - 1
            xs = ops.linear(xs, fc.weight, fc.bias,
                            dropout_p=0.0 if (last or not self.training) else self.dropout_p)
        return xs

    def forward(self, eouts, elens, ys, forced_align=False):
        """eouts `[B,T,enc_n_units]`, elens IntTensor (CPU), ys list of label lists ->
        (loss `[1]`, trigger_points IntTensor `[B,L+1]` or None)."""
        ys_lab = [y[::-1] if self.bwd else y for y in ys]
        lab, ylens_dev, _ = _labels_to_device(ys_lab, eouts.device, pad=0)
        elens_dev = ops.h2d(elens, eouts.device, torch.int32)
        logits = self.logits(eouts)
        loss, _ = ops.ctc_loss(logits, lab, elens_dev, ylens_dev, self.lsm_prob,
                               int(elens.sum()), self.blank)
        trigger_points = None
        if forced_align:
            trigger_points = self.forced_aligner(logits.detach(), elens, ys, None)
        if not self.training:
            self.data_dict['elens'] = elens.numpy()
        return loss, trigger_points

    def greedy(self, eouts, elens):
        """ctc.py:219-243: best path -> collapse repeats -> drop blanks.  The per-frame arg-max
        runs on the device (nsp_argmax_rows) and comes back in ONE transfer; the reference reads
        every frame with .item().  Returns `[B]` lists holding one hypothesis (n-best format)."""
        B, T = eouts.shape[:2]
        logits = self.logits(eouts)
        best = ops.argmax_rows(logits.reshape(B * T, -1)).view(B, T).cpu().numpy()
        hyps = []
        for b in range(B):
            collapsed = [k for k, _ in groupby(best[b, :int(elens[b])].tolist())]
            hyps.append([[x for x in collapsed if x != self.blank]])
        return hyps

    def beam_search(self, *a, **k):
        raise NotImplementedError('CTC prefix beam search is inference-side and not built: decode with '
                                  'recog_beam_width=1 (greedy), or load the state_dict into the reference')

    def _plot_ctc(self, save_path=None, topk=10):
        pass  # plotting needs matplotlib figures of self.prob_dict; the training loop only needs it not to raise

    def probs(self, eouts, temperature=1.):
        return torch.softmax(self.logits(eouts) / temperature, dim=-1)

    def scores(self, eouts, temperature=1.):
        return torch.log_softmax(self.logits(eouts) / temperature, dim=-1)


class CTCForcedAligner(object):
    """ctc.py:628-753: leftmost frame of every label on the best path under the CTC
    forward-backward posterior (+ last frame for <eos>)."""

    def __init__(self, blank=0):
        self.blank = blank

    def __call__(self, logits, elens, ys, ylens=None):
        lab, ylens_dev, _ = _labels_to_device(ys, logits.device, pad=0)
        elens_dev = ops.h2d(elens, logits.device, torch.int32)
        return ops.ctc_forced_align(logits, lab, elens_dev, ylens_dev, self.blank)


class RNNTransducer(DecoderBase):
    """rnn_transducer.py:32-311 (training part).  The joint network, its log-softmax and
    the lattice loss run as HIP kernels (ops.rnnt_joint_loss); the prediction network's
    LSTM recurrence runs on per-step MFMA kernels (ops.lstm); nn.LSTM modules are kept only
    as parameter containers so that state_dict names match the reference."""

    def __init__(self, special_symbols, enc_n_units, n_units, n_projs, n_layers, bottleneck_dim,
                 emb_dim, vocab, dropout, dropout_emb, ctc_weight, ctc_lsm_prob, ctc_fc_list,
                 external_lm, global_weight, mtl_per_batch, param_init):
        super().__init__()
        self.eos = special_symbols['eos']
        self.unk = special_symbols['unk']
        self.pad = special_symbols['pad']
        self.blank = special_symbols['blank']
        self.vocab = vocab
        self.enc_n_units = enc_n_units
        self.dec_n_units = n_units
        self.n_projs = n_projs
        self.n_layers = n_layers
        self.rnnt_weight = global_weight - ctc_weight
        self.ctc_weight = ctc_weight
        self.mtl_per_batch = mtl_per_batch
        if external_lm is not None:
            raise NotImplementedError('prediction-network initialisation from an external LM')
        if ctc_weight > 0:
            self.ctc = CTC(eos=self.eos, blank=self.blank, enc_n_units=enc_n_units, vocab=vocab,
                           dropout=dropout, lsm_prob=ctc_lsm_prob, fc_list=ctc_fc_list, param_init=0.1)
        if self.rnnt_weight > 0:
            self.rnn = nn.ModuleList()
            dec_odim = emb_dim
            self.proj = nn.ModuleList([copy.deepcopy(nn.Linear(n_units, n_projs))
                                       for _ in range(n_layers)]) if n_projs > 0 else None
            self.dropout = nn.Dropout(p=dropout)
            for _ in range(n_layers):
                self.rnn += [nn.LSTM(dec_odim, n_units, 1, batch_first=True)]
                dec_odim = n_projs if n_projs > 0 else n_units
            self.embed = nn.Embedding(vocab, emb_dim, padding_idx=self.pad)
            self.dropout_emb = nn.Dropout(p=dropout_emb)
            self.w_enc = nn.Linear(enc_n_units, bottleneck_dim)
            self.w_dec = nn.Linear(dec_odim, bottleneck_dim, bias=False)
            self.output = nn.Linear(bottleneck_dim, vocab)
        self.reset_parameters(param_init)

    def reset_parameters(self, param_init):
        """rnn_transducer.py:161-172"""
        for n, p in self.named_parameters():
            if p.dim() == 1:
                nn.init.constant_(p, 0.)
            elif p.dim() in [2, 4]:
                nn.init.uniform_(p, a=-param_init, b=param_init)
            else:
                raise ValueError(n)

    def forward(self, eouts, elens, ys, task='all', teacher_logits=None,
                recog_params={}, idx2token=None, trigger_points=None):
        observation = {'loss': None, 'loss_transducer': None, 'loss_ctc': None, 'loss_mbr': None}
        loss = eouts.new_zeros((1,))
        do_ctc = self.ctc_weight > 0 and (task == 'all' or 'ctc' in task)
        do_rnnt = self.rnnt_weight > 0 and (task == 'all' or 'ctc' not in task)
        loss_ctc = None
        ctc_stream = None
        if do_ctc:
            if do_rnnt and eouts.is_cuda and os.environ.get('NSP_CTC_STREAM', '1') != '0':
                # The two loss branches are independent and each contains a long latency-bound
                # lattice kernel on B workgroups (CTC alpha/beta ~0.4 ms, RNN-T ~0.5 ms): the CTC
                # branch runs on its own stream beside the transducer branch, forward and backward.
                ctc_stream = self.ensure_streams()[1]        # (None in single-stream mode: stock DDP, Speech2Text._ddp_guard)
            if ctc_stream is not None:
                cur = torch.cuda.current_stream(eouts.device)
                ctc_stream.wait_stream(cur)
                with torch.cuda.stream(ctc_stream):
                    loss_ctc, _ = self.ctc(eouts, elens, ys)
                eouts.record_stream(ctc_stream)
            else:
                loss_ctc, _ = self.ctc(eouts, elens, ys)
        if do_rnnt:
            loss_transducer = self.forward_transducer(eouts, elens, ys)
            observation['loss_transducer'] = loss_transducer.detach()
            loss = loss + (loss_transducer if self.mtl_per_batch else loss_transducer * self.rnnt_weight)
        if do_ctc:
            if ctc_stream is not None:
                cur = torch.cuda.current_stream(eouts.device)
                cur.wait_stream(ctc_stream)
                loss_ctc.record_stream(cur)
            observation['loss_ctc'] = loss_ctc.detach()   # device scalar; Speech2Text syncs once
            loss = loss + (loss_ctc if self.mtl_per_batch else loss_ctc * self.ctc_weight)
        observation['loss'] = loss.detach()
        return loss, observation

    def _prediction_network(self, ys, dev, defer_tail=False):
        """ys -> w_dec(recurrency(embed([eos]+y)))  `[B, L+1, J]` (rnn_transducer.py:229-236,273).
        defer_tail: when the LSTM stack runs as one (persistent) launch, stop behind it and return ('deferred', y_top) --
        the caller runs `_prediction_network_tail` after `ops.lstm_forward_resolve()`, so that a launch that timed out at
        its grid barrier is re-run before anything has read its output (ops._LSTM_FWD_RESCUE)."""
        L = max(len(y) for y in ys) + 1
        ys_in_np = np.full((len(ys), L), self.pad, dtype=np.int64)
        for b, y in enumerate(ys):
            ys_in_np[b, 0] = self.eos
            ys_in_np[b, 1:len(y) + 1] = np.asarray(y, dtype=np.int64)
        ys_in = ops.h2d(ys_in_np, dev)
        dout, st = self.recurrency(self.embed_token_id(ys_in), None, need_state=False, defer_tail=defer_tail)
        if st == 'deferred':
            return ('deferred', dout)
        return ops.linear(dout, self.w_dec.weight, None)

    def _prediction_network_tail(self, out):
        """the dropout behind the last LSTM layer and the output projection (rnn_transducer.py:303,273)"""
        return ops.linear(ops.dropout(out, self.dropout.p, self.training), self.w_dec.weight, None)

    def ensure_streams(self):
        """(prediction-network side stream, CTC-branch stream) of the training step, created on
        first use; None where the corresponding overlap is disabled / not applicable."""
        if not torch.cuda.is_available() or not next(self.parameters()).is_cuda:
            return None, None
        if getattr(self, '_nsp_single_stream', False):
            # stock DistributedDataParallel (no multi-stream communication hook): the whole step stays on the
            # current stream, see Speech2Text._ddp_guard
            return None, None
        dev = self.device
        side = ctc = None
        if self.rnnt_weight > 0 and os.environ.get('NSP_PREDNET_STREAM', '1') != '0':
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.Stream(
                    device=dev, priority=int(os.environ.get('NSP_PREDNET_STREAM_PRIORITY', '-1')))
            side = self._side_stream
        if self.ctc_weight > 0 and self.rnnt_weight > 0 and os.environ.get('NSP_CTC_STREAM', '1') != '0':
            if getattr(self, '_ctc_stream', None) is None:
                self._ctc_stream = torch.cuda.Stream(device=dev)
            ctc = self._ctc_stream
        return side, ctc

    def prediction_network_parameters(self):
        """Parameters whose gradients are produced on the side stream (embed, LSTM stack, proj, w_dec)."""
        if self.rnnt_weight <= 0:
            return []
        mods = [self.embed, self.rnn, self.w_dec] + ([self.proj] if self.proj is not None else [])
        return [p for m in mods for p in m.parameters()]

    def mark_step_start(self):
        """Record the point on the current stream the prediction network has to wait for
        (the previous step's optimizer update); everything enqueued later is independent of it."""
        if self.rnnt_weight <= 0 or not torch.cuda.is_available():
            return
        self._step_start_event = torch.cuda.Event()
        self._step_start_event.record(torch.cuda.current_stream(self.device))

    def start_prediction_network(self, ys):
        """Launch the prediction network on a side HIP stream.  It depends only on the labels,
        so its ~400 small sequential LSTM-step kernels overlap with the encoder instead of
        serialising after it; forward_transducer joins the stream.  The autograd engine replays
        the backward of these nodes on the same side stream (see ops.replay_graph_first for how
        that backward is made to start before the encoder's, not after it)."""
        if self.rnnt_weight <= 0 or not torch.cuda.is_available() or os.environ.get('NSP_PREDNET_STREAM', '1') == '0':
            return
        if getattr(self, '_nsp_single_stream', False):
            return
        dev = self.device
        self.ensure_streams()
        ev = getattr(self, '_step_start_event', None)
        self._step_start_event = None
        if ev is not None:
            self._side_stream.wait_event(ev)
        else:
            self._side_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._side_stream):
            dec_proj = self._prediction_network(ys, dev, defer_tail=True)
            self._pred_done = torch.cuda.Event()
            self._pred_done.record(self._side_stream)
        self._pending_dec_proj = (dec_proj, id(ys))

    def forward_transducer(self, eouts, elens, ys):
        dev = eouts.device
        lab, ylens_dev, ylens = _labels_to_device(ys, dev, pad=self.blank)     # ys_out, blank-padded
        elens_dev = ops.h2d(elens, dev, torch.int32)
        pending = getattr(self, '_pending_dec_proj', None)
        self._pending_dec_proj = None
        if pending is not None and pending[1] == id(ys):
            dec_proj = pending[0]
            cur = torch.cuda.current_stream(dev)
            if isinstance(dec_proj, tuple):           # ('deferred', LSTM output): check the launch, then the network's tail
                ops.lstm_forward_resolve()
                with torch.cuda.stream(self._side_stream):
                    dec_proj = self._prediction_network_tail(dec_proj[1])
                    self._pred_done = torch.cuda.Event()
                    self._pred_done.record(self._side_stream)
            cur.wait_event(self._pred_done)
            dec_proj.record_stream(cur)
            if torch.is_grad_enabled() and dec_proj.requires_grad \
                    and os.environ.get('NSP_PREDNET_PRIORITY', '1') != '0':
                dec_proj = ops.replay_graph_first(
                    dec_proj, self._side_stream if os.environ.get('NSP_REPLAY_SIDE', '1') != '0' else None)
        else:
            dec_proj = self._prediction_network(ys, dev, defer_tail=True)       # `[B,L+1,J]`
            if isinstance(dec_proj, tuple):
                # (inline on the step's stream: no host wait for a launch that was enqueued a moment ago -- except under the
                # test hook, which has nothing to wait for)
                ops.lstm_forward_resolve(block=os.environ.get('NSP_LSTM_TEST_FAKE_TIMEOUT', '0') == '1')
                dec_proj = self._prediction_network_tail(dec_proj[1])
        enc_proj = ops.linear(eouts, self.w_enc.weight, self.w_enc.bias)       # `[B,T,J]`
        loss, _ = ops.rnnt_joint_loss(enc_proj, dec_proj, self.output.weight, self.output.bias,
                                      lab, elens_dev, ylens_dev, self.blank,
                                      elens_host=elens.tolist(), ylens_host=ylens)
        return loss

    # ---- greedy decoding (validate() -> evaluators -> Speech2Text.decode)
    def joint(self, eouts, douts):
        """rnn_transducer.py:262-276: `[B,T,enc]` x `[B,L,dec]` -> logits `[B,T,L,V]` (inference sizes)."""
        e = ops.linear(eouts, self.w_enc.weight, self.w_enc.bias).unsqueeze(2)
        g = ops.linear(douts, self.w_dec.weight, None).unsqueeze(1)
        h = ops.act_fwd((e + g).contiguous(), ops.ACT['tanh'])
        return ops.linear(h, self.output.weight, self.output.bias)

    def _step_state(self, y, state, update):
        """One prediction-network step for a batch of states: embed -> per layer
        (x W_ih^T + b_ih + h W_hh^T + b_hh -> cell) [-> proj+ReLU]; rows with update == 0 keep
        their state and output (rnn_transducer.py:278-311 with dstate, eval mode)."""
        x = self.embed(y)
        new_state = []
        for lth in range(self.n_layers):
            rnn = self.rnn[lth]
            h_prev, c_prev = state[lth]
            gi = ops.linear(x, rnn.weight_ih_l0, rnn.bias_ih_l0)
            gates = ops.linear(h_prev, rnn.weight_hh_l0, rnn.bias_hh_l0, res=gi)
            h, c = ops.lstm_cell_step(gates, h_prev, c_prev, update)
            new_state.append((h, c))
            x = h
            if self.proj is not None:
                x = ops.linear(x, self.proj[lth].weight, self.proj[lth].bias, act='relu')
        return x, new_state

    @torch.no_grad()
    def greedy(self, eouts, elens, max_len_ratio=None, idx2token=None, exclude_eos=False,
               refs_id=None, utt_ids=None, speakers=None, trigger_points=None, teacher_force=False):
        """rnn_transducer.py:330-382 for the whole batch in lock-step over frames: per frame every
        utterance takes the 1-best of joint(e_t, d); the prediction network advances only in the
        rows that emitted a non-blank label (at most one label per frame, as in the reference).
        No host sync inside the loop: labels are collected on the device and read once."""
        B, T = eouts.shape[:2]
        dev = eouts.device
        H = self.dec_n_units
        enc_proj = ops.linear(eouts, self.w_enc.weight, self.w_enc.bias)            # [B,T,J]
        state = [(eouts.new_zeros(B, H), eouts.new_zeros(B, H)) for _ in range(self.n_layers)]
        y = torch.full((B,), self.eos, dtype=torch.int64, device=dev)
        dout, state = self._step_state(y, state, None)
        elens_dev = ops.h2d(elens, dev, torch.int32)
        labels = torch.empty((T, B), device=dev, dtype=torch.int32)
        for t in range(T):
            g = ops.linear(dout, self.w_dec.weight, None)
            h = ops.act_fwd(ops.axpby(enc_proj[:, t].contiguous(), g, 1.0, 1.0), ops.ACT['tanh'])
            yt = ops.argmax_rows(ops.linear(h, self.output.weight, self.output.bias))
            labels[t] = yt
            update = ((yt != self.blank) & (elens_dev > t)).to(torch.int32)
            dout, state = self._step_state(yt.long(), state, update)
        lab = labels.cpu().numpy()
        hyps = [[int(v) for v in lab[:int(elens[b]), b] if v != self.blank] for b in range(B)]
        if idx2token is not None:
            for b in range(B):
                if utt_ids is not None:
                    logger.debug('Utt-id: %s' % utt_ids[b])
                logger.debug('Hyp: %s' % idx2token(hyps[b]))
        return hyps, None

    def beam_search(self, *a, **k):
        raise NotImplementedError('RNN-T beam search is inference-side and not built: decode with '
                                  'recog_beam_width=1 (greedy), or load the state_dict into the reference')

    def _plot_attention(self, save_path=None, n_cols=1):
        pass

    def _plot_ctc(self, save_path=None, topk=10):
        if self.ctc_weight > 0:
            self.ctc._plot_ctc(save_path, topk)

    def embed_token_id(self, indices):
        # embedding lookup = row gather of a [V, emb] table (host-side indexing glue)
        return ops.dropout(self.embed(indices), self.dropout_emb.p, self.training)

    def recurrency(self, ys_emb, dstate, need_state=True, defer_tail=False):
        if dstate is None:
            dstate = self.zero_state(ys_emb.size(0)) if need_state else None
        if (not need_state and self.proj is None and ops.on_kernel_device(ys_emb)
                and os.environ.get('NSP_LSTM_STACK', '1') != '0'):
            layers = [(r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0) for r in self.rnn]
            if ops.lstm_stack_supported(layers, ys_emb):
                # training path: all layers as one wavefront over (layer, time); the dropout between
                # layers is fused, the one after the last layer is the ordinary op
                p = self.dropout.p if self.training else 0.0
                out = ops.lstm_stack(ys_emb, layers, p)
                if defer_tail:
                    return out, 'deferred'
                return ops.dropout(out, self.dropout.p, self.training), None
        new_hxs, new_cxs = [], []
        for lth in range(self.n_layers):
            rnn = self.rnn[lth]
            # zero initial state (training path); the recurrence runs on the HIP step kernels
            ys_emb = ops.lstm(ys_emb, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)
            new_hxs.append(ys_emb[:, -1:].transpose(0, 1))
            new_cxs.append(ys_emb.new_zeros(1, ys_emb.size(0), self.dec_n_units))  # not tracked in training
            ys_emb = ops.dropout(ys_emb, self.dropout.p, self.training)
            if self.proj is not None:
                ys_emb = ops.linear(ys_emb, self.proj[lth].weight, self.proj[lth].bias, act='relu')
        return ys_emb, {'hxs': torch.cat(new_hxs, dim=0), 'cxs': torch.cat(new_cxs, dim=0)}

    def zero_state(self, batch_size):
        w = next(self.parameters())
        return {'hxs': w.new_zeros(self.n_layers, batch_size, self.dec_n_units),
                'cxs': w.new_zeros(self.n_layers, batch_size, self.dec_n_units)}


class CTCOnlyDecoder(DecoderBase):
    """The reference routes CTC-only models through RNNDecoder/TransformerDecoder with
    ctc_weight == global_weight (las.py:438-505 / decoders/transformer.py:314-371); only
    their CTC branch runs.  This shell exposes the same `.ctc` submodule name and the same
    (loss, observation) contract for that case."""

    def __init__(self, special_symbols, enc_n_units, vocab, dropout, ctc_weight, ctc_lsm_prob,
                 ctc_fc_list, global_weight, mtl_per_batch):
        super().__init__()
        self.ctc_weight = ctc_weight
        self.mtl_per_batch = mtl_per_batch
        self.ctc = CTC(eos=special_symbols['eos'], blank=special_symbols['blank'],
                       enc_n_units=enc_n_units, vocab=vocab, dropout=dropout,
                       lsm_prob=ctc_lsm_prob, fc_list=ctc_fc_list, param_init=0.1)
        for n, p in self.named_parameters():
            if p.dim() == 1:
                nn.init.constant_(p, 0.)
            else:
                nn.init.uniform_(p, a=-0.1, b=0.1)

    def forward(self, eouts, elens, ys, task='all', teacher_logits=None, recog_params={},
                idx2token=None, trigger_points=None):
        observation = {'loss': None, 'loss_att': None, 'loss_ctc': None, 'loss_mbr': None,
                       'acc_att': None, 'ppl_att': None}
        loss_ctc, _ = self.ctc(eouts, elens, ys)
        observation['loss_ctc'] = loss_ctc.detach()
        loss = loss_ctc if self.mtl_per_batch else loss_ctc * self.ctc_weight
        observation['loss'] = loss.detach()
        return loss, observation

    def _plot_attention(self, save_path=None, n_cols=1):
        pass

    def _plot_ctc(self, save_path=None, topk=10):
        self.ctc._plot_ctc(save_path, topk)


class TransformerDecoderBlock(nn.Module):
    """modules/transformer.py:21-260 (training path: scaled-dot source attention, no cache, no LM
    fusion): x += drop(SelfMHA(LN1 x)); x += drop(SrcMHA(LN2 x; enc)); x += drop(FFN(LN3 x)).
    Same sub-module / parameter names as the reference block."""

    def __init__(self, d_model, d_ff, n_heads, dropout, dropout_att, dropout_layer, layer_norm_eps,
                 ffn_activation, param_init, ffn_bottleneck_dim=0, atype='scaled_dot', src_tgt_attention=True,
                 mma=None, dropout_head=0.0):
        super().__init__()
        from neural_sp_amd.modules import MultiheadAttentionMechanism as MHA, PositionwiseFeedForward as FFN
        self.n_heads = n_heads
        self.atype = atype
        self.xy_aws = None
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.self_attn = MHA(kdim=d_model, qdim=d_model, adim=d_model, odim=d_model, n_heads=n_heads,
                             dropout=dropout_att, dropout_head=dropout_head, param_init=param_init)
        self.src_tgt_attention = src_tgt_attention
        if not src_tgt_attention:
            self.src_attn = None          # MMA decoders: the layers below `mocha_first_layer` have no source attention
        elif 'mocha' in atype:
            from neural_sp_amd.mma import MMA
            self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
            self.n_heads = mma['n_heads_mono']
            self.src_attn = MMA(kdim=d_model, qdim=d_model, adim=d_model, odim=d_model, dropout=dropout_att,
                                dropout_head=dropout_head, param_init=param_init, **mma)
        else:
            self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
            self.src_attn = MHA(kdim=d_model, qdim=d_model, adim=d_model, odim=d_model, n_heads=n_heads,
                                dropout=dropout_att, param_init=param_init)
        self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.feed_forward = FFN(d_model, d_ff, dropout, ffn_activation, param_init, ffn_bottleneck_dim)
        self.dropout_p = dropout
        self.dropout_layer = dropout_layer

    def forward(self, ys, yy_mask, xs, xy_mask):
        import random
        if self.dropout_layer > 0 and self.training and random.random() < self.dropout_layer:
            return ys
        p = self.dropout_p
        yn, ys = ops.layer_norm_split(ys, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        out = self.self_attn(yn, yn, yn, mask=yy_mask, residual=ys, out_dropout=p)[0]
        self.xy_aws = None
        if self.src_attn is not None:
            on, out = ops.layer_norm_split(out, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            if 'mocha' in self.atype:
                out, self.xy_aws, _ = self.src_attn(xs, xs, on, mask=xy_mask.dense(on.shape[1], xs.shape[1]),
                                                    residual=out, out_dropout=p)
            else:
                out = self.src_attn(xs, xs, on, mask=xy_mask, residual=out, out_dropout=p)[0]
        on, out = ops.layer_norm_split(out, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return self.feed_forward(on, residual=out, alpha=1.0, out_dropout=p)


class TransformerDecoder(DecoderBase):
    """decoders/transformer.py:28-458, training side: hybrid CTC / attention loss of a Transformer
    decoder (BASELINE config 3's loss head family).  forward() = CTC branch
    (:349-357) + forward_att (:373-458): teacher-forced decoder stack on the HIP kernels, output
    projection, and the fused label-smoothed XE / accuracy kernel (criterion.py:45-86).  Source attention is scaled-dot
    MHA or monotonic multi-head attention (`attn_type='mocha'`, neural_sp_amd/mma.py, from layer `mma_first_layer` on).
    LM fusion, beam search and decoding of MMA models are not built (NotImplementedError); greedy decoding is."""

    def __init__(self, special_symbols, enc_n_units, attn_type, n_heads, n_layers, d_model, d_ff,
                 ffn_bottleneck_dim, pe_type, layer_norm_eps, ffn_activation, vocab, tie_embedding,
                 dropout, dropout_emb, dropout_att, dropout_layer, lsm_prob, ctc_weight, ctc_lsm_prob,
                 ctc_fc_list, backward, global_weight, mtl_per_batch, param_init, dropout_head=0.0, mma=None,
                 mma_first_layer=1, mma_quantity_loss_weight=0.0):
        super().__init__()
        from neural_sp_amd.modules import PositionalEncoding
        if attn_type not in ('scaled_dot', 'mocha'):
            raise NotImplementedError('transformer_dec_attn_type=%s' % attn_type)
        self.quantity_loss_weight, self._quantity_loss_weight = mma_quantity_loss_weight, 0
        self.eos, self.unk = special_symbols['eos'], special_symbols['unk']
        self.pad, self.blank = special_symbols['pad'], special_symbols['blank']
        self.vocab, self.enc_n_units, self.d_model = vocab, enc_n_units, d_model
        self.n_layers, self.n_heads, self.pe_type = n_layers, n_heads, pe_type
        self.lsm_prob = lsm_prob
        self.att_weight = global_weight - ctc_weight
        self.ctc_weight = ctc_weight
        self.bwd = backward
        self.mtl_per_batch = mtl_per_batch
        self.attn_type = attn_type
        self.aws_dict, self.data_dict = {}, {}
        if ctc_weight > 0:
            self.ctc = CTC(eos=self.eos, blank=self.blank, enc_n_units=enc_n_units, vocab=vocab, dropout=dropout,
                           lsm_prob=ctc_lsm_prob, fc_list=ctc_fc_list, param_init=0.1, backward=backward)
        if self.att_weight > 0:
            self.embed = nn.Embedding(vocab, d_model, padding_idx=self.pad)
            self.pos_enc = PositionalEncoding(d_model, dropout_emb, pe_type, param_init)
            self.layers = nn.ModuleList([copy.deepcopy(TransformerDecoderBlock(
                d_model, d_ff, n_heads, dropout, dropout_att, dropout_layer, layer_norm_eps, ffn_activation,
                param_init, ffn_bottleneck_dim, atype=attn_type, src_tgt_attention=lth >= mma_first_layer - 1,
                mma=mma, dropout_head=dropout_head)) for lth in range(n_layers)])
            self.norm_out = nn.LayerNorm(d_model, eps=layer_norm_eps)
            self.output = nn.Linear(d_model, vocab)
            if tie_embedding:
                self.output.weight = self.embed.weight
            if param_init == 'xavier_uniform':       # transformer.py:304-316
                nn.init.normal_(self.embed.weight, mean=0., std=d_model ** -0.5)
                nn.init.constant_(self.embed.weight[self.pad], 0.)
                nn.init.xavier_uniform_(self.output.weight)
                nn.init.constant_(self.output.bias, 0.)

    def forward(self, eouts, elens, ys, task='all', teacher_logits=None, recog_params={}, idx2token=None,
                trigger_points=None):
        observation = {'loss': None, 'loss_att': None, 'loss_ctc': None, 'loss_mbr': None,
                       'acc_att': None, 'ppl_att': None}
        loss = eouts.new_zeros((1,))
        if self.ctc_weight > 0 and (task == 'all' or 'ctc' in task):
            loss_ctc, _ = self.ctc(eouts, elens, ys)
            observation['loss_ctc'] = loss_ctc.detach()
            loss = loss + (loss_ctc if self.mtl_per_batch else loss_ctc * self.ctc_weight)
        if self.att_weight > 0 and (task == 'all' or 'ctc' not in task):
            loss_att, acc_att, ppl_att, loss_quantity = self.forward_att(eouts, elens, ys)
            observation['loss_att'] = loss_att.detach()
            observation['acc_att'] = acc_att
            observation['ppl_att'] = ppl_att
            if self.attn_type == 'mocha':                 # transformer.py:362-366
                if self._quantity_loss_weight > 0:
                    loss_att = loss_att + loss_quantity * self._quantity_loss_weight
                observation['loss_quantity'] = loss_quantity.detach()
            loss = loss + (loss_att if self.mtl_per_batch else loss_att * self.att_weight)
        observation['loss'] = loss.detach()
        return loss, observation

    def trigger_quantity_loss(self):
        if self.attn_type == 'mocha':
            self._quantity_loss_weight = self.quantity_loss_weight

    def trigger_stableemit(self):
        pass          # decoder_base.py:45-50: "TODO: MMA" in the reference -- nothing is switched for Transformer decoders

    def forward_att(self, eouts, elens, ys, trigger_points=None):
        """transformer.py:373-458 -> (loss [1], acc (device scalar, %), ppl (device scalar), quantity loss)."""
        from neural_sp_amd.modules import AttnMask
        dev = eouts.device
        B = len(ys)
        ylens = [len(y) + 1 for y in ys]                      # +1 for <eos> (torch_utils.py:123)
        L = max(ylens)
        ys_in = np.full((B, L), self.pad, dtype=np.int64)     # append_sos_eos (torch_utils.py:97-126), sos = eos
        ys_out = np.full((B, L), self.pad, dtype=np.int32)
        for b, y in enumerate(ys):
            yy = y[::-1] if self.bwd else y
            ys_in[b, 0] = self.eos
            ys_in[b, 1:len(yy) + 1] = yy
            ys_out[b, :len(yy)] = yy
            ys_out[b, len(yy)] = self.eos
        ys_in_d = ops.h2d(ys_in, dev)
        ys_out_d = ops.h2d(ys_out.reshape(-1), dev)
        ylens_d = ops.h2d(np.asarray(ylens, dtype=np.int32), dev)
        elens_d = ops.h2d(elens, dev, torch.int32)
        # tgt mask = (key j is not <pad>) & causal (:392-396); src mask = key t < elens_b (:399)
        yy_mask = AttnMask(ylens_d, causal=True, lookahead=0)
        xy_mask = AttnMask(elens_d)
        out = self.pos_enc(self.embed(ys_in_d), scale=True)   # scaled + dropout
        xy_aws_layers = []
        for layer in self.layers:
            out = layer(out, yy_mask, eouts, xy_mask)
            if layer.xy_aws is not None and self.attn_type == 'mocha':
                # attention padding (:426-429): target positions past <eos> select nothing
                tgt_valid = (ys_out_d.view(B, L) != self.pad).view(B, 1, L, 1)
                xy_aws_layers.append(layer.xy_aws.masked_fill(~tgt_valid, 0))
        out = ops.layer_norm(out, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps)
        logits = ops.linear(out, self.output.weight, self.output.bias)
        lsm = self.lsm_prob if self.training else 0.0         # criterion.py:64
        loss, loss_rows, correct = ops.xe_lsm_loss(logits, ys_out_d, lsm, self.pad, B)
        n_tokens = float(sum(ylens))
        ppl = torch.exp(loss_rows.sum() / n_tokens)           # criterion.py:66 / :84
        acc = correct.sum().float() * (100.0 / n_tokens)      # torch_utils.py:140-145
        loss_quantity = eouts.new_zeros(())
        if self.attn_type == 'mocha':
            # :444-452: expected number of selected frames per head, averaged over heads and MMA layers, against the
            # number of target tokens (<eos> included)
            n_ref = (ys_out_d.view(B, L) != self.pad).sum(1).float()
            n_pred = sum(torch.abs(a.sum(3).sum(2).sum(1) / a.size(1)) for a in xy_aws_layers) / len(xy_aws_layers)
            loss_quantity = torch.mean(torch.abs(n_pred - n_ref))
        return loss, acc, ppl, loss_quantity

    def _plot_attention(self, save_path=None, n_cols=1):
        pass

    def _plot_ctc(self, save_path=None, topk=10):
        if self.ctc_weight > 0:
            self.ctc._plot_ctc(save_path, topk)

    def greedy(self, eouts, elens, max_len_ratio, idx2token=None, exclude_eos=False, refs_id=None, utt_ids=None,
               speakers=None, cache_states=True):
        """decoders/transformer.py:460-566 (validate() with recog_beam_width 1): arg-max decoding, the whole
        prefix re-run through the stack at every step (no state cache: L <= ceil(T * max_len_ratio) short
        steps).  As in the reference the target mask is purely causal and the SOURCE attention is unmasked
        (`layer(out, causal_mask, eouts, None, ...)`, :500: padded encoder frames are attended to).
        -> (hyps: list of int arrays, None); attention-weight plots are not produced."""
        if self.attn_type == 'mocha':
            raise NotImplementedError('decoding with monotonic multi-head attention (test-time hard attention) is not '
                                      'built; evaluate MMA models with the teacher-forced accuracy (--metric accuracy)')
        from neural_sp_amd.modules import AttnMask
        dev = eouts.device
        B, T = eouts.shape[:2]
        with torch.no_grad():
            full = ops.h2d(np.full((B,), T, dtype=np.int32), dev)
            xy_mask = AttnMask(full)
            ys = torch.full((B, 1), self.eos, dtype=torch.int64, device=dev)
            hyps_batch = []
            ylens = [0] * B
            eos_flags = [False] * B
            ymax = int(math.ceil(T * max_len_ratio))
            for i in range(ymax):
                yy_mask = AttnMask(ops.h2d(np.full((B,), i + 1, dtype=np.int32), dev), causal=True, lookahead=0)
                out = self.pos_enc(self.embed(ys), scale=True)
                for layer in self.layers:
                    out = layer(out, yy_mask, eouts, xy_mask)
                out = ops.layer_norm(out[:, -1:].contiguous(), self.norm_out.weight, self.norm_out.bias, self.norm_out.eps)
                y = ops.argmax_rows(ops.linear(out, self.output.weight, self.output.bias).reshape(B, -1)).long()
                hyps_batch.append(y)
                yh = y.tolist()
                for b in range(B):
                    if not eos_flags[b]:
                        if yh[b] == self.eos:
                            eos_flags[b] = True
                        ylens[b] += 1
                if all(eos_flags) or i == ymax - 1:
                    break
                ys = torch.cat([ys, y.view(B, 1)], dim=-1)
            hb = torch.stack(hyps_batch, dim=1).cpu().numpy()
        hyps = [hb[b, :ylens[b]][::-1] if self.bwd else hb[b, :ylens[b]] for b in range(B)]
        if exclude_eos:
            hyps = [(h[1:] if self.bwd else h[:-1]) if eos_flags[b] else h for b, h in enumerate(hyps)]
        return hyps, None

    def beam_search(self, *a, **k):
        raise NotImplementedError('attention-decoder decoding is inference-side and not built')


def build_decoder(args, special_symbols, enc_n_units, vocab, ctc_weight, global_weight, external_lm=None):
    """decoders/build.py:7-140 for the loss heads on the hot path."""
    if args.dec_type in ['lstm_transducer', 'gru_transducer']:
        if args.dec_type == 'gru_transducer':
            raise NotImplementedError('gru_transducer')
        return RNNTransducer(
            special_symbols=special_symbols, enc_n_units=enc_n_units, n_units=args.dec_n_units,
            n_projs=args.dec_n_projs, n_layers=args.dec_n_layers, bottleneck_dim=args.dec_bottleneck_dim,
            emb_dim=args.emb_dim, vocab=vocab, dropout=args.dropout_dec, dropout_emb=args.dropout_emb,
            ctc_weight=ctc_weight, ctc_lsm_prob=args.ctc_lsm_prob, ctc_fc_list=args.ctc_fc_list,
            external_lm=external_lm if args.lm_init else None, global_weight=global_weight,
            mtl_per_batch=args.mtl_per_batch, param_init=args.param_init)
    if ctc_weight > 0 and abs(global_weight - ctc_weight) < 1e-12:
        return CTCOnlyDecoder(special_symbols, enc_n_units, vocab, args.dropout_dec, ctc_weight,
                              args.ctc_lsm_prob, args.ctc_fc_list, global_weight, args.mtl_per_batch)
    if args.dec_type == 'transformer':
        if external_lm is not None or getattr(args, 'lm_fusion', ''):
            raise NotImplementedError('LM fusion in the Transformer decoder')
        return TransformerDecoder(
            special_symbols=special_symbols, enc_n_units=enc_n_units, attn_type=args.transformer_dec_attn_type,
            n_heads=args.transformer_dec_n_heads, n_layers=args.dec_n_layers, d_model=args.transformer_dec_d_model,
            d_ff=args.transformer_dec_d_ff, ffn_bottleneck_dim=args.transformer_ffn_bottleneck_dim,
            pe_type=args.transformer_dec_pe_type, layer_norm_eps=args.transformer_layer_norm_eps,
            ffn_activation=args.transformer_ffn_activation, vocab=vocab, tie_embedding=args.tie_embedding,
            dropout=args.dropout_dec, dropout_emb=args.dropout_emb, dropout_att=args.dropout_att,
            dropout_layer=args.dropout_dec_layer, lsm_prob=args.lsm_prob, ctc_weight=ctc_weight,
            ctc_lsm_prob=args.ctc_lsm_prob, ctc_fc_list=args.ctc_fc_list, backward=False,
            global_weight=global_weight, mtl_per_batch=args.mtl_per_batch, param_init=args.transformer_param_init,
            dropout_head=getattr(args, 'dropout_head', 0.0),
            mma=dict(chunk_size=args.mocha_chunk_size, n_heads_mono=args.mocha_n_heads_mono,
                     n_heads_chunk=args.mocha_n_heads_chunk, init_r=args.mocha_init_r, eps=args.mocha_eps,
                     noise_std=args.mocha_std, no_denominator=args.mocha_no_denominator, conv1d=args.mocha_1dconv,
                     share_chunkwise_attention=getattr(args, 'share_chunkwise_attention', False)),
            mma_first_layer=getattr(args, 'mocha_first_layer', 1),
            mma_quantity_loss_weight=args.mocha_quantity_loss_weight)
    if args.dec_type in ('lstm', 'gru'):
        from neural_sp_amd.las import RNNDecoder      # decoders/build.py:88-140 (las.py uses LSTMCell for both)
        return RNNDecoder(
            special_symbols=special_symbols, enc_n_units=enc_n_units, n_units=args.dec_n_units,
            n_projs=args.dec_n_projs, n_layers=args.dec_n_layers, bottleneck_dim=args.dec_bottleneck_dim,
            emb_dim=args.emb_dim, vocab=vocab, tie_embedding=args.tie_embedding, attn_type=args.attn_type,
            attn_dim=args.attn_dim, attn_sharpening_factor=args.attn_sharpening_factor,
            attn_sigmoid_smoothing=args.attn_sigmoid, attn_conv_out_channels=args.attn_conv_n_channels,
            attn_conv_kernel_size=args.attn_conv_width, attn_n_heads=args.attn_n_heads, dropout=args.dropout_dec,
            dropout_emb=args.dropout_emb, dropout_att=args.dropout_att, lsm_prob=args.lsm_prob, ss_prob=args.ss_prob,
            ctc_weight=ctc_weight, ctc_lsm_prob=args.ctc_lsm_prob, ctc_fc_list=args.ctc_fc_list,
            mbr_training=args.mbr_training, mbr_ce_weight=args.mbr_ce_weight, external_lm=external_lm,
            lm_fusion=args.lm_fusion, lm_init=args.lm_init, backward=False, global_weight=global_weight,
            mtl_per_batch=args.mtl_per_batch, param_init=args.param_init, mocha_chunk_size=args.mocha_chunk_size,
            mocha_n_heads_mono=args.mocha_n_heads_mono, mocha_init_r=args.mocha_init_r, mocha_eps=args.mocha_eps,
            mocha_std=args.mocha_std, mocha_no_denominator=args.mocha_no_denominator, mocha_1dconv=args.mocha_1dconv,
            mocha_decot_lookahead=args.mocha_decot_lookahead, quantity_loss_weight=args.mocha_quantity_loss_weight,
            latency_metric=str(args.mocha_latency_metric), latency_loss_weight=args.mocha_latency_loss_weight,
            mocha_stableemit_weight=args.mocha_stableemit_weight, gmm_attn_n_mixtures=args.gmm_attn_n_mixtures,
            replace_sos=args.replace_sos, distillation_weight=args.distillation_weight,
            discourse_aware=args.discourse_aware)
    raise NotImplementedError('dec_type=%s' % args.dec_type)
