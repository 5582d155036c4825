"""Speech2Text: the drop-in boundary of the hot path.

Mirrors neural_sp/models/seq2seq/speech2text.py: `__init__(args, save_path, idx2token)`
(:45-204), `forward(batch, task, is_eval, teacher, teacher_lm) -> (loss[1], observation)`
(:239-269), `_forward` (:271-345), `encode` (:369-431), with the same submodule /
parameter names (`enc.*`, `dec_fwd.*`) so `state_dict`s are interchangeable and
neural_sp/bin/asr/train.py drives it unchanged.  Everything below `encode()` runs on the
HIP kernels (neural_sp_amd.ops); there is no CPU path.  Greedy decoding (what validate() runs)
is built; beam search / streaming raise NotImplementedError; the plot hooks are no-ops.
"""
import logging
import random

import copy
import inspect
import os

import numpy as np
import torch
import torch.nn as nn

from neural_sp_amd import ops
from neural_sp_amd.decoders import RNNTransducer as RNNT
from neural_sp_amd.decoders import build_decoder
from neural_sp_amd.encoders import build_encoder

random.seed(1)  # speech2text.py:37

logger = logging.getLogger(__name__)


class SpecAugment(object):
    """spec_augment.py:11-140: frequency / time band zeroing with ONE mask set per batch,
    drawn from np.random in the reference's call order; the zeroing is a HIP kernel."""

    def __init__(self, F, T, n_freq_masks, n_time_masks, p=1.0, W=40,
                 adaptive_number_ratio=0, adaptive_size_ratio=0, max_n_time_masks=20):
        self.W, self.F, self.T = W, F, T
        self.n_freq_masks, self.n_time_masks, self.p = n_freq_masks, n_time_masks, p
        self.adaptive_number_ratio = adaptive_number_ratio
        self.adaptive_size_ratio = adaptive_size_ratio
        self.max_n_time_masks = max_n_time_masks
        if adaptive_number_ratio > 0:
            self.n_time_masks = 0
        if adaptive_size_ratio > 0:
            self.T = 0
        self._freq_mask = None
        self._time_mask = None

    def draw(self, n_frames, n_bins):
        fb, tb = [], []
        for _ in range(self.n_freq_masks):
            f = int(np.random.uniform(low=0, high=self.F))
            f_0 = int(np.random.uniform(low=0, high=n_bins - f))
            fb.append((f_0, f_0 + f))
            self._freq_mask = (f_0, f_0 + f)
        if self.adaptive_number_ratio > 0:
            n_masks = min(int(n_frames * self.adaptive_number_ratio), self.max_n_time_masks)
        else:
            n_masks = self.n_time_masks
        T = self.adaptive_size_ratio * n_frames if self.adaptive_size_ratio > 0 else self.T
        for _ in range(n_masks):
            t = int(np.random.uniform(low=0, high=T))
            t = min(t, int(n_frames * self.p))
            t_0 = int(np.random.uniform(low=0, high=n_frames - t))
            tb.append((t_0, t_0 + t))
            self._time_mask = (t_0, t_0 + t)
        return fb, tb

    def __call__(self, xs):
        fb, tb = self.draw(xs.size(1), xs.size(2))
        return ops.specaug_apply_(xs, fb, tb)


class LazyObservation(dict):
    """The observation dict of Speech2Text.forward (python floats, speech2text.py:262-293).  The
    device scalars are copied to a pinned host block asynchronously inside forward(); reading the
    dict waits on the event recorded right behind that copy -- i.e. only for the forward that
    produced the values, not for whatever has been enqueued on the stream since (a plain
    .tolist() is a stream-synchronous copy and would drain the whole queue)."""

    def __init__(self, static, keys, stacked):
        super().__init__(static)
        if stacked.is_cuda:
            host = torch.empty(stacked.shape, dtype=stacked.dtype, pin_memory=True)
            host.copy_(stacked, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(stacked.device))
        else:
            host, ev = stacked, None
        self._pending = (keys, host, ev)

    def materialize(self):
        if self._pending is not None:
            keys, host, ev = self._pending
            self._pending = None
            if ev is not None:
                ev.synchronize()
                # the step's loss values are on the host: the forward's persistent-LSTM launches have finished too (the
                # loss depends on them), so a grid-barrier time-out is reported with THIS step's values, not one
                # launch later (the backward's launches are checked at the next step's first read)
                ops._lstm_poll_dead()
            for k, v in zip(keys, host.tolist()):
                dict.__setitem__(self, k, v)
        return self

    def __getitem__(self, k):
        return dict.__getitem__(self.materialize(), k)

    def get(self, k, default=None):
        return dict.get(self.materialize(), k, default)

    def items(self):
        return dict.items(self.materialize())

    def keys(self):
        return dict.keys(self.materialize())

    def __iter__(self):
        # overriding __iter__ also takes dict(obs) / {**obs} / dict.update(obs) off CPython's
        # raw-storage fast path (it is only used when tp_iter is dict's own), so every way of
        # reading the mapping goes through materialize()
        return dict.__iter__(self.materialize())

    def setdefault(self, k, default=None):
        return dict.setdefault(self.materialize(), k, default)

    def popitem(self):
        return dict.popitem(self.materialize())

    def values(self):
        return dict.values(self.materialize())

    def pop(self, *a):
        return dict.pop(self.materialize(), *a)

    def copy(self):
        return dict(self.materialize())

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        return dict.__repr__(self.materialize())

    def __reduce__(self):
        return (dict, (dict(self.materialize()),))


class Speech2Text(nn.Module):
    """Speech to text sequence-to-sequence model (training hot path on MI355X)."""

    def __init__(self, args, save_path=None, idx2token=None):
        super().__init__()
        self.save_path = save_path
        self.input_type = args.input_type
        self.input_dim = args.input_dim
        self.enc_type = args.enc_type
        self.dec_type = args.dec_type
        self.enc_n_layers = args.enc_n_layers
        self.enc_n_layers_sub1 = args.enc_n_layers_sub1
        self.subsample = [int(s) for s in args.subsample.split('_')]
        self.vocab = args.vocab
        self.vocab_sub1 = args.vocab_sub1
        self.vocab_sub2 = args.vocab_sub2
        self.blank, self.unk, self.eos, self.pad = 0, 1, 2, 3
        self.main_weight = args.total_weight - args.sub1_weight - args.sub2_weight
        self.sub1_weight = args.sub1_weight
        self.sub2_weight = args.sub2_weight
        self.mtl_per_batch = args.mtl_per_batch
        self.task_specific_layer = args.task_specific_layer
        self.ctc_weight = min(args.ctc_weight, self.main_weight)
        self.ctc_weight_sub1 = min(args.ctc_weight_sub1, self.sub1_weight)
        self.ctc_weight_sub2 = min(args.ctc_weight_sub2, self.sub2_weight)
        self.bwd_weight = min(args.bwd_weight, self.main_weight)
        self.fwd_weight = self.main_weight - self.bwd_weight - self.ctc_weight
        self.fwd_weight_sub1 = self.sub1_weight - self.ctc_weight_sub1
        self.fwd_weight_sub2 = self.sub2_weight - self.ctc_weight_sub2
        self.mbr_training = args.mbr_training
        self.recog_params = vars(args) if not isinstance(args, dict) else args
        self.idx2token = idx2token
        self.utt_id_prev = None
        if self.input_type != 'speech':
            raise NotImplementedError("input_type='text'")
        if self.bwd_weight > 0 or self.mbr_training:
            raise NotImplementedError('backward / MBR decoders are outside the hot path')
        self.input_noise_std = args.input_noise_std
        self.n_stacks = args.n_stacks
        self.n_skips = args.n_skips
        self.n_splices = args.n_splices
        self.weight_noise_std = args.weight_noise_std
        if self.n_stacks > 1 or self.n_splices > 1:
            raise NotImplementedError('frame stacking / splicing (numpy frontends) are out of scope')
        self.specaug = None
        if args.n_freq_masks > 0 or args.n_time_masks > 0:
            assert args.n_stacks == 1 and args.n_skips == 1
            assert args.n_splices == 1
            self.specaug = SpecAugment(F=args.freq_width, T=args.time_width,
                                       n_freq_masks=args.n_freq_masks, n_time_masks=args.n_time_masks,
                                       p=args.time_width_upper,
                                       adaptive_number_ratio=args.adaptive_number_ratio,
                                       adaptive_size_ratio=args.adaptive_size_ratio,
                                       max_n_time_masks=args.max_n_time_masks)
        self.ssn = None
        if args.sequence_summary_network:
            raise NotImplementedError('sequence summary network')
        self.enc = build_encoder(args)
        if args.freeze_encoder:
            for n, p in self.enc.named_parameters():
                if 'bridge' in n or 'sub1' in n:
                    continue
                p.requires_grad = False
        special_symbols = {'blank': self.blank, 'unk': self.unk, 'eos': self.eos, 'pad': self.pad}
        if args.external_lm:
            raise NotImplementedError('external LM fusion / initialisation')
        directions = []
        if self.fwd_weight > 0 or (self.bwd_weight == 0 and self.ctc_weight > 0):
            directions.append('fwd')
        for dir in directions:
            dec = build_decoder(args, special_symbols, self.enc.output_dim, args.vocab,
                                self.ctc_weight, self.main_weight - self.bwd_weight, None)
            setattr(self, 'dec_' + dir, dec)
        # auxiliary tasks of hierarchical multi-task training (speech2text.py:170-184): their own decoder on the
        # encoder's intermediate output, configured like the main one except for `dec_config_sub{1,2}`
        for sub in ['sub1', 'sub2']:
            if getattr(self, sub + '_weight') > 0:
                args_sub = copy.deepcopy(args)
                if hasattr(args, 'dec_config_' + sub):
                    for k, v in getattr(args, 'dec_config_' + sub).items():
                        setattr(args_sub, k, v)
                dec_sub = build_decoder(args_sub, special_symbols, getattr(self.enc, 'output_dim_' + sub),
                                        getattr(self, 'vocab_' + sub), getattr(self, 'ctc_weight_' + sub),
                                        getattr(self, sub + '_weight'), None)
                setattr(self, 'dec_fwd_' + sub, dec_sub)
        # Constant tables (the XL position embedding's `inv_freq`, ...) are the same on every rank by construction: keep them
        # out of stock DistributedDataParallel's per-forward buffer broadcast (train.py:263 leaves broadcast_buffers=True) --
        # one collective, and one cross-rank synchronisation at the top of every step, saved.  BatchNorm statistics are
        # state, not tables: they stay in.  (They remain in state_dict either way.)
        # An ALLOW-list of known constant tables: any other buffer -- BatchNorm statistics today, whatever
        # stateful buffer a later change adds -- keeps torch's default and is synchronised.
        self._ddp_params_and_buffers_to_ignore = [
            n for n, _ in self.named_buffers() if n.rsplit('.', 1)[-1] in ('inv_freq', 'pe')]

    # ---- bookkeeping members touched by neural_sp/bin/asr/train.py (speech2text.py:206-237, base.py)
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def device_id(self):
        return torch.cuda.device_of(next(self.parameters())).idx

    @property
    def use_cuda(self):
        return torch.cuda.is_available()

    @property
    def num_params_dict(self):
        if not hasattr(self, '_nparams_dict'):
            self._nparams_dict = {n: p.view(-1).size(0) for n, p in self.named_parameters()}
        return self._nparams_dict

    @property
    def total_parameters(self):
        if not hasattr(self, '_nparams'):
            self._nparams = sum(p.view(-1).size(0) for p in self.parameters())
        return self._nparams

    def cudnn_setting(self, deterministic=False, benchmark=True):
        pass  # no cuDNN/MIOpen autotuning on the hand-written path

    def trigger_scheduled_sampling(self):
        # speech2text.py:206-215: main and auxiliary-task decoders
        for name in ('dec_fwd', 'dec_bwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
            if hasattr(self, name):
                getattr(self, name).trigger_scheduled_sampling()

    def trigger_quantity_loss(self):
        # speech2text.py:217-221 (main task only)
        if hasattr(self, 'dec_fwd'):
            self.dec_fwd.trigger_quantity_loss()
            self.dec_fwd.trigger_latency_loss()

    def trigger_stableemit(self):
        if hasattr(self, 'dec_fwd'):
            self.dec_fwd.trigger_stableemit()

    def reset_session(self):
        pass

    # evaluators/*.py read these after every decoded batch (speech2text.py:700-707): streaming statistics of a MoChA /
    # triggered-attention decoder, the defaults for everything else
    def streamable(self):
        return getattr(self.dec_fwd, 'streamable', False)

    def quantity_rate(self):
        return getattr(self.dec_fwd, 'quantity_rate', 1.0)

    def last_success_frame_ratio(self):
        return getattr(self.dec_fwd, 'last_success_frame_ratio', 0)

    def plot_attention(self):
        """speech2text.py:494-503, called by train.py:484-487 every 10*print_step steps on rank 0.
        Drawing the attention maps needs the probabilities the fused kernels never materialise (and
        matplotlib); the training loop only needs the call to succeed: warn once, do nothing."""
        self._warn_once('plot_attention')

    def plot_ctc(self):
        """speech2text.py:505-511 (same call site)."""
        self._warn_once('plot_ctc')

    def _warn_once(self, what):
        done = self.__dict__.setdefault('_warned', set())
        if what not in done:
            done.add(what)
            logger.warning('%s() is a no-op in neural_sp_amd (attention / CTC posteriors are not '
                           'kept on the HIP path); training continues' % what)

    @staticmethod
    def _param(params, key, default=None):
        """recog_* hyper-parameters arrive as a dict, an argparse.Namespace or an OmegaConf DictConfig."""
        if isinstance(params, dict):
            return params.get(key, default)
        if hasattr(params, 'get') and not isinstance(params, dict):
            try:
                v = params.get(key)
                return default if v is None else v
            except Exception:
                pass
        return getattr(params, key, default)

    @torch.no_grad()
    def decode(self, xs, params, idx2token=None, exclude_eos=False, refs_id=None, refs=None,
               utt_ids=None, speakers=None, task='ys', ensemble_models=[], trigger_points=None,
               teacher_force=False):
        """speech2text.py:709-800 for GREEDY decoding (recog_beam_width == 1), which is what
        validate() (train.py:341,513-557 -> evaluators/*.py) runs during training with the default
        recog_* arguments: CTC best path (ctc.py:219-243) when the model is CTC-only or
        recog_ctc_weight == 1, else the RNN-T frame-synchronous 1-best (rnn_transducer.py:330-382).
        Returns (nbest_hyps_id `[B][1][L]`, aws None).  Beam search / LM fusion / streaming /
        ensembles are inference-side and raise NotImplementedError."""
        self.eval()
        ops.refresh_weight_shadows(force=not getattr(self, '_nsp_eval_epoch_started', False))   # (once per evaluation phase, see _forward)
        self._nsp_eval_epoch_started = True
        base = task.split('.')[0]
        if base not in ('ys', 'ys_sub1', 'ys_sub2'):
            raise ValueError(task)
        dir = {'ys': 'fwd', 'ys_sub1': 'fwd_sub1', 'ys_sub2': 'fwd_sub2'}[base]     # speech2text.py:734-741
        P = self._param
        if P(params, 'recog_streaming_encoding', False) or P(params, 'recog_block_sync', False):
            raise NotImplementedError('streaming encoding / block-synchronous decoding')
        if len(ensemble_models) > 0:
            raise NotImplementedError('ensemble decoding')
        beam = P(params, 'recog_beam_width', 1)
        eout_dict = self.encode(xs, task)
        eouts, elens = eout_dict[base]['xs'], eout_dict[base]['xlens']
        dec = getattr(self, 'dec_' + dir)
        if (self.fwd_weight == 0 and self.bwd_weight == 0) or \
                (self.ctc_weight > 0 and P(params, 'recog_ctc_weight', 0) == 1):
            if beam != 1:
                dec.ctc.beam_search()
            return dec.ctc.greedy(eouts, elens), None
        if beam != 1 or P(params, 'recog_fwd_bwd_attention', False):
            dec.beam_search()
        kw = {}
        if 'trigger_points' in inspect.signature(dec.greedy).parameters:
            kw['trigger_points'] = trigger_points        # triggered attention decodes from the CTC alignment
        best_hyps_id, aws = dec.greedy(eouts, elens, P(params, 'recog_max_len_ratio', 1.0), idx2token,
                                       exclude_eos, refs_id, utt_ids, speakers, **kw)
        return [[hyp] for hyp in best_hyps_id], aws

    # ---- hot path
    def forward(self, batch, task, is_eval=False, teacher=None, teacher_lm=None):
        if teacher is not None or teacher_lm is not None:
            raise NotImplementedError('knowledge distillation')
        # (speech2text.py:253-262 calls self.eval() / self.train() on every forward: a walk over ~400 modules, 1.2 ms of
        # host time per step at 16 utterances per GPU -- only when the mode actually changes.  The reference's call also
        # RE-PROPAGATES the mode, i.e. resets a sub-module someone put into the other mode on its own (a frozen encoder
        # in eval, say): the cheap check over the direct children keeps that behaviour for the top-level sub-modules;
        # a flag flipped deeper than that survives until the next mode change -- documented divergence)
        want = not is_eval
        if self.training != want or any(m.training != want for m in self.children()):
            self.train(want)
        if is_eval:
            with torch.no_grad():
                loss, observation = self._forward(batch, task)
        else:
            loss, observation = self._forward(batch, task)
        return loss, observation

    def _ddp_guard(self):
        """train.py:263 wraps the model in stock `DistributedDataParallel(model, device_ids=...)`.  Its reducer orders
        a bucket's all-reduce only after the stream of the gradient hook that completes the bucket; this step produces
        gradients on up to three HIP streams (main, prediction network, CTC branch), so with the stock wrapper a
        collective could read a bucket before another stream has written its part -- silently wrong gradients.
        Unless the model went through `parallel.wrap_ddp` (or the DDP class `neural_sp_amd.install()` puts in its
        place), which registers the multi-stream communication hook, a training forward inside an initialised process
        group of more than one rank therefore keeps the WHOLE step on the current stream (correct under any wrapper,
        a few per cent slower) and says so once."""
        import torch.distributed as dist
        unsafe = (not getattr(self, '_nsp_ddp_hooked', False) and dist.is_available() and dist.is_initialized()
                  and dist.get_world_size() > 1)
        for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
            dec = getattr(self, name, None)
            if dec is not None and hasattr(dec, 'ensure_streams'):
                dec._nsp_single_stream = unsafe
        if unsafe and not getattr(Speech2Text, '_warned_stock_ddp', False):
            Speech2Text._warned_stock_ddp = True
            logger.warning('neural_sp_amd: process group of %d ranks without the multi-stream DDP hook -- the step runs on '
                           'ONE stream (correct with stock DistributedDataParallel).  Call neural_sp_amd.install() before '
                           'train.py imports DistributedDataParallel, or wrap with neural_sp_amd.parallel.wrap_ddp, to '
                           'overlap the prediction network and the CTC branch again.', dist.get_world_size())
        return unsafe

    def _forward(self, batch, task):
        if torch.is_grad_enabled():
            self._ddp_guard()
        # every bf16 weight shadow that the last optimizer step made stale: one launch (before the side stream's
        # step-start event, so that the prediction network reads the refreshed images)
        # (forced in evaluation too: the last optimizer step of a fused optimizer leaves no trace in the version counters)
        # (in evaluation the epoch is forced on the FIRST batch after a training forward only -- the last optimizer step of a
        # fused optimizer leaves no trace in the version counters -- not once per batch, which rebuilt every unregistered
        # cache for each evaluation batch)
        training = torch.is_grad_enabled() and self.training
        ops.refresh_weight_shadows(force=training or not getattr(self, '_nsp_eval_epoch_started', False))
        self._nsp_eval_epoch_started = not training
        if isinstance(getattr(self, 'dec_fwd', None), RNNT) and task in ('all', 'ys'):
            # the prediction network overlaps with the encoder on a side stream; it is enqueued
            # right after the encoder's front-end so that neither stream starts the step idle
            dec, ys = self.dec_fwd, batch['ys']
            dec.mark_step_start()
            self.enc._after_frontend = lambda: dec.start_prediction_network(ys)
        try:
            eout_dict = self.encode(batch['xs'], task if self.mtl_per_batch else 'all')
        finally:
            self.enc._after_frontend = None
        observation = {}
        loss = torch.zeros((1,), dtype=torch.float32, device=self.device)
        if (self.fwd_weight > 0 or (self.bwd_weight == 0 and self.ctc_weight > 0)) \
                and task in ['all', 'ys', 'ys.ctc', 'ys.mbr']:
            loss_fwd, obs_fwd = self.dec_fwd(eout_dict['ys']['xs'], eout_dict['ys']['xlens'],
                                             batch['ys'], task, None, self.recog_params,
                                             self.idx2token, batch['trigger_points'])
            loss = loss + loss_fwd
            if isinstance(self.dec_fwd, RNNT):
                observation['loss.transducer'] = obs_fwd['loss_transducer']
            else:
                observation['acc.att'] = obs_fwd['acc_att']
                observation['ppl.att'] = obs_fwd['ppl_att']
                observation['loss.att'] = obs_fwd['loss_att']
                observation['loss.mbr'] = obs_fwd['loss_mbr']
                observation['loss.quantity'] = obs_fwd.get('loss_quantity')
                observation['loss.latency'] = obs_fwd.get('loss_latency')
            observation['loss.ctc'] = obs_fwd['loss_ctc']
        # only forward decoders for the auxiliary tasks (speech2text.py:326-343)
        for sub in ['sub1', 'sub2']:
            if (getattr(self, 'fwd_weight_' + sub) > 0 or getattr(self, 'ctc_weight_' + sub) > 0) \
                    and task in ['all', 'ys_' + sub, 'ys_' + sub + '.ctc']:
                if len(batch['ys_' + sub]) == 0:
                    continue  # evaluation sets without the auxiliary transcripts
                dec_sub = getattr(self, 'dec_fwd_' + sub)
                loss_sub, obs_sub = dec_sub(eout_dict['ys_' + sub]['xs'], eout_dict['ys_' + sub]['xlens'],
                                            batch['ys_' + sub], task)
                loss = loss + loss_sub
                if isinstance(dec_sub, RNNT):
                    observation['loss.transducer-' + sub] = obs_sub['loss_transducer']
                else:
                    observation['loss.att-' + sub] = obs_sub['loss_att']
                    observation['acc.att-' + sub] = obs_sub['acc_att']
                    observation['ppl.att-' + sub] = obs_sub['ppl_att']
                observation['loss.ctc-' + sub] = obs_sub['loss_ctc']
        return loss, self._finalize_observation(observation)

    @staticmethod
    def _finalize_observation(observation):
        """The reference calls .item() three times per step inside the decoders
        (rnn_transducer.py:199,208,214), each a full device sync in the middle of the step.  The
        decoders here hand back device scalars; they become the python floats the Reporter
        expects with ONE transfer, made when the dict is first read (LazyObservation): a
        training loop that reads it after loss.backward() keeps the host ahead of the GPU for
        the whole step instead of draining the queue between forward and backward."""
        keys = [k for k, v in observation.items() if torch.is_tensor(v)]
        if not keys:
            return observation
        stacked = torch.stack([observation[k].detach().reshape(()).float() for k in keys])
        obs = LazyObservation(observation, keys, stacked)
        return obs.materialize() if os.environ.get('NSP_EAGER_OBSERVATION', '0') == '1' else obs

    def add_weight_noise(self, std):
        """models/base.py:77-91, as the reference EXECUTES it: `Normal([0.],[std]).sample([N])` has shape
        [N,1] and `param_vector.add_(noise[0])` broadcasts its first row, so ONE N(0, std) scalar is added to
        every parameter of the model.  That scalar is reproduced bit for bit from the CPU generator (the
        first of a >= 16-element `normal_` fill depends on the first 16 uniforms only); the reference then
        goes on to draw the N - 16 values it never uses, so the CPU generator state afterwards differs.
        One multi-tensor add; the version counters move, so the bf16 weight shadows are rebuilt."""
        with torch.no_grad():
            eps = (torch.empty(16).normal_()[0] * torch.tensor(float(std))).item()   # fp32 product, as the reference
            torch._foreach_add_([p for p in self.parameters()], eps)

    def encode(self, xs, task='all', streaming=False, cnn_lookback=False, cnn_lookahead=False,
               xlen_block=-1):
        """xs: list of np.float32 `[T_i, input_dim]`.  One packed H2D copy, padding on the
        device (pad_list + np2tensor of speech2text.py:397 without the per-utterance loop)."""
        if streaming:
            raise NotImplementedError('streaming encoding')
        xlens = torch.IntTensor([len(x) for x in xs])
        dev = self.device
        B, Tmax, F = len(xs), int(xlens.max()), self.input_dim
        offs = torch.zeros(B, dtype=torch.int64)
        offs[1:] = torch.cumsum(xlens[:-1].long() * F, 0)
        xs = ops.pad_batch(ops.h2d_packed(xs, dev), ops.h2d(offs, dev), ops.h2d(xlens, dev), B, Tmax, F, 0.)
        if self.specaug is not None and self.training:
            xs = self.specaug(xs)
        if self.weight_noise_std > 0 and self.training:
            self.add_weight_noise(std=self.weight_noise_std)
            # the prediction network's side stream waits for the step-start event only: move that event behind
            # the in-place noise add, or it would read (and shadow) weights while they change
            ops.refresh_weight_shadows(force=True)          # the noise moved every parameter: one launch again
            dec = getattr(self, 'dec_fwd', None)
            if getattr(dec, '_step_start_event', None) is not None:
                dec.mark_step_start()
        if self.input_noise_std > 0 and self.training:
            noise = torch.normal(xs.new_zeros(xs.shape[-1]), self.input_noise_std)  # input_noise.py
            xs = ops.scale_add_bcast(xs, noise, 1.0)
        return self.enc(xs, xlens, task.split('.')[0], streaming, cnn_lookback, cnn_lookahead)

    def ctc_forced_align(self, xs, ys, task='ys'):
        """speech2text.py:470-492: CTC forced alignment -> trigger points `[B,L+1]` (np.int32)."""
        self.eval()
        ops.refresh_weight_shadows(force=not getattr(self, '_nsp_eval_epoch_started', False))   # (once per evaluation phase, see _forward)
        self._nsp_eval_epoch_started = True
        with torch.no_grad():
            eout_dict = self.encode(xs, 'ys')
            ctc = self.dec_fwd.ctc
            logits = ctc.logits(eout_dict[task]['xs'])
            tp = ctc.forced_aligner(logits, eout_dict[task]['xlens'], ys)
        return tp.cpu().numpy()
