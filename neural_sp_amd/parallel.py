"""Data-parallel plumbing: one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference's only parallelism is DistributedDataParallel(Speech2Text)
(neural_sp/bin/asr/train.py:263) plus the `.module` wrappers of models/data_parallel.py.
Utterances are independent, so the path shards with no data-path collective other than the
gradient reduction.  `wrap_ddp` keeps torch's DDP (so train.py runs unchanged) and adapts it to
this step's shape:

* **multi-stream backward.**  The step runs three streams (main; prediction network on a side
  stream; CTC branch on its own stream).  DDP's reducer launches a bucket's all-reduce from
  whichever gradient hook completes the bucket and orders it only after THAT hook's current
  stream -- gradients written into the same bucket from another stream would race.  The comm hook
  below launches every bucket from a small "gather" stream that first waits for all streams of the
  step, so the collective sees every write and no compute stream is ever blocked by it.
* **gradient accumulators pinned to their streams.**  DDP keeps every parameter's AccumulateGrad
  node alive from construction on, and a node runs on the stream that was current when it was
  created.  Created on the main stream, the prediction-network accumulators would make the main
  stream wait for the whole LSTM backward (the stall ops.replay_graph_first exists to avoid);
  `pin_grad_streams` creates them under the stream their gradients are produced on.
* **buckets sized for xGMI.**  7 point-to-point links per GPU: few large collectives beat many
  25 MB ones, but the LAST bucket (conv front-end + first blocks, ready only when backward ends)
  cannot overlap with anything, so the cap is 48 MB (362 MB of fp32 gradients -> 8 buckets, DDP
  re-orders them by gradient arrival after the first step); optional bf16 compression halves the
  bytes on the links (NSP_DDP_COMPRESS=bf16 or compress='bf16').
* no per-forward buffer broadcast: the buffers are constant tables -- and, with `conformer_normalization:
  batch_norm`, BatchNorm running statistics, which torch's default (broadcast_buffers=True) would overwrite on every
  rank with rank 0's before each forward; rank 0's own statistics (the ones checkpoints and rank-0 evaluation see)
  evolve identically either way, the other ranks simply keep theirs; `no_sync()` on
  accumulation micro-steps is torch DDP's own context manager (train.py reduces every micro-step,
  train.py:414-452; `accumulate(model, is_boundary)` below skips the collective until the boundary).
"""
import contextlib
import os

import torch
import torch.nn as nn


class CPUWrapperASR(nn.Module):
    """models/data_parallel.py:54-60 -- adds `.module` so train.py's `model.module.*` works."""

    def __init__(self, model):
        super().__init__()
        self.module = model

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def step_streams(model):
    """The non-default streams the training step of `model` uses (created on demand)."""
    out = []
    for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):      # auxiliary-task decoders may be transducers too
        dec = getattr(model, name, None)
        if dec is not None and hasattr(dec, 'ensure_streams'):
            out += [s for s in dec.ensure_streams() if s is not None]
    return out


def mark_hooked(model):
    """the model is (about to be) under a wrapper with the multi-stream communication hook: Speech2Text._ddp_guard keeps
    the multi-stream step, and a single-stream fallback chosen by an earlier un-wrapped forward is lifted"""
    model._nsp_ddp_hooked = True
    for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
        dec = getattr(model, name, None)
        if dec is not None and hasattr(dec, 'ensure_streams'):
            dec._nsp_single_stream = False


def pin_grad_streams(model):
    """Create (and keep alive on the model) the AccumulateGrad nodes of the parameters whose
    gradients are produced on a side stream, with that stream current.  Must run before DDP is
    constructed: DDP then adopts these nodes instead of creating its own on the main stream."""
    dec = getattr(model, 'dec_fwd', None)
    if dec is None or not hasattr(dec, 'ensure_streams') or not next(model.parameters()).is_cuda:
        return []
    side, ctc = dec.ensure_streams()
    keep = []

    def pin(params, stream):
        if stream is None:
            return
        with torch.cuda.stream(stream):
            for p in params:
                if p.requires_grad:
                    keep.append(p.view_as(p).grad_fn.next_functions[0][0])
    if side is not None:
        pin(dec.prediction_network_parameters(), side)
    if ctc is not None and getattr(dec, 'ctc', None) is not None and side is not None:
        pin(dec.ctc.parameters(), ctc)
    model._nsp_grad_accumulators = keep
    return keep


class HostEvent(object):
    """torch.cuda.Event's record / elapsed_time on the host clock: what CommStats and bench.py use where there is no device
    stream to record on (the CPU tier: gloo ranks on the host-emulated kernels)."""

    def __init__(self):
        self.t = None

    def record(self, stream=None):
        import time
        self.t = time.perf_counter()
        return self

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class CommStats(object):
    """What bench.py reports about the collective under DDP: per backward pass the number of buckets, their bytes,
    and the event recorded on the hook's stream when the LAST bucket became ready (= backward's compute is fully
    enqueued); `exposed_ms(after)` = time from that point to `after` (an event recorded once backward() has returned,
    i.e. behind DDP's wait for every collective) = communication that backward could not hide."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.buckets, self.bytes, self.last_ready, self.waited_side = 0, 0, None, 0
        self.wire_bytes = 0          # what the collective moves per rank incl. rs_ag padding and bf16 compression

    def exposed_ms(self, after):
        return None if self.last_ready is None else self.last_ready.elapsed_time(after)


def param_streams(model):
    """{id(parameter): stream} for the parameters whose gradients are produced on a side stream of the step
    (prediction network -> its side stream, CTC head -> the CTC stream); everything else is the main stream's."""
    out = {}
    for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
        dec = getattr(model, name, None)
        if dec is None or not hasattr(dec, 'ensure_streams') or not next(model.parameters()).is_cuda:
            continue
        side, ctc = dec.ensure_streams()
        if side is not None:
            for p in dec.prediction_network_parameters():
                out[id(p)] = side
            if ctc is not None and getattr(dec, 'ctc', None) is not None:
                for p in dec.ctc.parameters():
                    out[id(p)] = ctc
    return out


def _reduce_scatter_all_gather(buf, group, world, on_device):
    """mean-reduce `buf` (already divided by world) as reduce-scatter + all-gather: every rank reduces
    1/world of the bucket, then the reduced shards are gathered.  On 8 fully connected GPUs both halves are one-hop
    exchanges over all 7 xGMI links at once, and the shard boundary is where a sharded optimizer step would sit.  The same
    bytes per rank as a ring all-reduce; which of the two RCCL runs faster on a given node is a measurement this build has
    not been able to make (no multi-GPU box).  Returns a future of `buf`."""
    import torch.distributed as dist
    n = buf.numel()
    pad = (-n) % world
    src = buf if pad == 0 else torch.cat([buf, buf.new_zeros(pad)])
    shard = torch.empty((src.numel() // world,), device=buf.device, dtype=buf.dtype)
    if on_device and dist.get_backend(group) == 'nccl':
        # same communicator, same (gather) stream: RCCL runs the two collectives in issue order
        dist.reduce_scatter_tensor(shard, src, group=group, async_op=True)
        fut = dist.all_gather_into_tensor(src, shard, group=group, async_op=True).get_future()
    else:
        dist.reduce_scatter_tensor(shard, src, group=group)          # (gloo runs work items on a thread pool: order by waiting)
        fut = dist.all_gather_into_tensor(src, shard, group=group, async_op=True).get_future()

    def done(f):
        # The callback of a device future runs on a stream of torch's callback pool, not on the gather stream the
        # temporaries were allocated on: without record_stream the caching allocator may hand `src` / `shard` to the next
        # bucket's torch.cat on the gather stream while this copy is still in flight.
        if on_device:
            cb = torch.cuda.current_stream(buf.device)
            shard.record_stream(cb)
            if pad:
                src.record_stream(cb)
            buf.record_stream(cb)
        if pad:
            buf.copy_(src[:n])
        return buf
    return fut.then(done)


def make_comm_hook(streams, compress=None, pstreams=None, stats=None, main_stream=None, algorithm=None):
    """DDP communication hook: all-reduce(mean) of one bucket, launched from a gather stream.  `algorithm`: 'all_reduce'
    (default; NSP_DDP_ALGO) or 'rs_ag' = reduce-scatter + all-gather (_reduce_scatter_all_gather).  The gather stream
    waits for the hook's (main) stream and for the side streams that produced gradients OF THIS BUCKET
    (`pstreams` = param_streams(model)): a bucket of encoder parameters no longer waits for the prediction
    network's LSTM backward (8 ms per 64 utterances on its side stream), so the early buckets' collectives start
    while it is still running.  Without `pstreams` every bucket waits for every stream of the step.
    `compress='bf16'` sends bf16; `stats` (CommStats) collects what bench.py prints."""
    import torch.distributed as dist
    gather = {}
    algorithm = algorithm or os.environ.get('NSP_DDP_ALGO') or 'all_reduce'
    if algorithm not in ('all_reduce', 'rs_ag'):
        raise ValueError('NSP_DDP_ALGO / algorithm must be all_reduce or rs_ag, not %r' % (algorithm,))

    def hook(state, bucket):
        buf = bucket.buffer()
        group = state if state is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        if not buf.is_cuda:
            if stats is not None:
                stats.buckets += 1
                stats.bytes += buf.numel() * buf.element_size()
                stats.wire_bytes += ((buf.numel() + ((-buf.numel()) % world if algorithm == 'rs_ag' else 0))
                                     * (2 if compress == 'bf16' else buf.element_size()))
                if bucket.is_last():
                    stats.last_ready = HostEvent().record()
            if compress == 'bf16':        # (the CPU tier runs the same algorithm x compression matrix as the device branch)
                send = buf.to(torch.bfloat16).div_(world)
                if algorithm == 'rs_ag':
                    fut = _reduce_scatter_all_gather(send, group, world, False)
                else:
                    fut = dist.all_reduce(send, group=group, async_op=True).get_future().then(lambda f: f.value()[0])
                return fut.then(lambda f: buf.copy_(f.value()))
            if algorithm == 'rs_ag':
                return _reduce_scatter_all_gather(buf.div_(world), group, world, False)
            return dist.all_reduce(buf.div_(world), group=group, async_op=True).get_future().then(lambda f: f.value()[0])
        dev = buf.device
        g = gather.get(dev)
        if g is None:
            g = gather[dev] = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)      # the stream of the gradient that completed the bucket
        g.wait_stream(cur)
        if main_stream is not None and main_stream != cur:
            g.wait_stream(main_stream)             # (the other gradients of the bucket may be the main stream's)
        if pstreams is None:
            need = list(streams)
        else:
            need = []
            for p in bucket.parameters():
                s = pstreams.get(id(p))
                if s is not None and s not in need:
                    need.append(s)
        for s in need:
            g.wait_stream(s)
        if stats is not None:
            stats.buckets += 1
            stats.bytes += buf.numel() * buf.element_size()
            n_pad = buf.numel() + ((-buf.numel()) % world if algorithm == 'rs_ag' else 0)
            stats.wire_bytes += n_pad * (2 if compress == 'bf16' else buf.element_size())
            stats.waited_side += len(need)
            if bucket.is_last():
                stats.last_ready = torch.cuda.Event(enable_timing=True)
                stats.last_ready.record(cur)
        buf.record_stream(g)
        with torch.cuda.stream(g):
            if compress == 'bf16':
                send = buf.to(torch.bfloat16).div_(world)
                if algorithm == 'rs_ag':
                    fut = _reduce_scatter_all_gather(send, group, world, True)
                else:
                    fut = dist.all_reduce(send, group=group, async_op=True).get_future().then(lambda f: f.value()[0])

                def decompress(f):
                    v = f.value()
                    v = v[0] if isinstance(v, (list, tuple)) else v
                    cb = torch.cuda.current_stream(dev)          # (callback stream: see _reduce_scatter_all_gather)
                    v.record_stream(cb)
                    send.record_stream(cb)
                    buf.record_stream(cb)
                    buf.copy_(v)
                    return buf
                return fut.then(decompress)
            buf.div_(world)
            if algorithm == 'rs_ag':
                return _reduce_scatter_all_gather(buf, group, world, True)
            fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
        return fut.then(lambda f: f.value()[0])
    return hook


def wrap_ddp(model, local_rank=None, bucket_cap_mb=48, compress=None, cpu_hook=False):
    """DDP(model) with the settings above; local_rank=None wraps a CPU module (gloo tests; cpu_hook: with the package's
    communication hook and its CommStats, as bench.py --emulate wants them)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    # constant tables need no per-forward broadcast; BatchNorm running statistics do (torch DDP's default, which
    # the reference relies on: every rank evaluates / checkpoints rank 0's statistics)
    has_bn = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) or type(m).__name__.startswith('BatchNorm')
                 for m in model.modules())
    kw = dict(broadcast_buffers=has_bn, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    mark_hooked(model)
    if local_rank is None:
        ddp = DDP(model, **kw)            # (CPU module: no streams to order)
        if cpu_hook:
            ddp.comm_stats = CommStats()
            ddp.register_comm_hook(None, make_comm_hook([], compress if compress is not None else
                                                        (os.environ.get('NSP_DDP_COMPRESS') or None), None, ddp.comm_stats, None))
        return ddp
    pin_grad_streams(model)
    ddp = DDP(model, device_ids=[local_rank], **kw)
    compress = compress if compress is not None else (os.environ.get('NSP_DDP_COMPRESS') or None)
    ddp.comm_stats = CommStats()
    pstreams = param_streams(model) if os.environ.get('NSP_DDP_WAIT_ALL_STREAMS', '0') != '1' else None
    ddp.register_comm_hook(None, make_comm_hook(step_streams(model), compress, pstreams, ddp.comm_stats,
                                                torch.cuda.current_stream(torch.device('cuda', local_rank))))
    return ddp


def make_ddp_class():
    """The class `neural_sp_amd.install()` puts where train.py finds `DistributedDataParallel` (train.py:20,263):
    torch's DDP, constructed exactly as the caller asks, plus -- when the wrapped module is this package's
    Speech2Text on a HIP device -- what `wrap_ddp` adds: gradient accumulators pinned to their streams BEFORE the
    reducer is built, and the multi-stream communication hook after.  Anything else is plain torch DDP."""
    from torch.nn.parallel import DistributedDataParallel as TorchDDP
    if getattr(TorchDDP, '_nsp_patched', False):
        return TorchDDP

    class DistributedDataParallel(TorchDDP):
        _nsp_patched = True

        def __init__(self, module, device_ids=None, *args, **kwargs):
            from .speech2text import Speech2Text
            ours = isinstance(module, Speech2Text) and device_ids and next(module.parameters()).is_cuda
            if ours:
                mark_hooked(module)
                pin_grad_streams(module)
                has_bn = any(type(m).__name__.startswith('BatchNorm') for m in module.modules())
                kwargs.setdefault('broadcast_buffers', has_bn)
                kwargs.setdefault('bucket_cap_mb', 48)
                kwargs.setdefault('gradient_as_bucket_view', True)
            super().__init__(module, device_ids, *args, **kwargs)
            if ours:
                dev = device_ids[0]
                dev = dev if isinstance(dev, torch.device) else torch.device('cuda', int(dev))
                self.comm_stats = CommStats()
                pstreams = param_streams(module) if os.environ.get('NSP_DDP_WAIT_ALL_STREAMS', '0') != '1' else None
                self.register_comm_hook(None, make_comm_hook(step_streams(module), os.environ.get('NSP_DDP_COMPRESS') or None,
                                                             pstreams, self.comm_stats, torch.cuda.current_stream(dev)))
    DistributedDataParallel.__name__ = 'DistributedDataParallel'
    return DistributedDataParallel


@contextlib.contextmanager
def accumulate(ddp_model, is_boundary):
    """Gradient accumulation: skip the collective on non-boundary micro-steps (DDP.no_sync)."""
    if is_boundary or not hasattr(ddp_model, 'no_sync'):
        yield
    else:
        with ddp_model.no_sync():
            yield


def shard_batch(indices, rank, world):
    """Rank-strided split of a length-sorted bucket (datasets/asr/sampler.py:96)."""
    return indices[rank::world]


def scale_loss_for_ddp(loss, world):
    """train.py:423-424: DDP averages gradients while every rank normalises by its own
    batch, so the loss is pre-multiplied by the number of replicas."""
    return loss * world if world > 1 else loss


def aggregate_timing(dt, units, device=None):
    """(max over ranks of dt, sum over ranks of units) -- bench.py's whole-job numbers."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dt, units
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return t.item(), u.item()


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ (train.py:442) without its per-parameter Python loop
    (380 `.to()` calls = ~5 ms of host time per step on this model): two foreach launches and
    one stacked norm, no host sync.  Returns the total norm as a device scalar."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    norms = torch._foreach_norm(grads)
    total = torch.linalg.vector_norm(torch.stack(norms))
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
    torch._foreach_mul_(grads, coef)
    return total
