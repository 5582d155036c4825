"""Data-parallel plumbing: one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference's only parallelism is DistributedDataParallel(Speech2Text)
(neural_sp/bin/asr/train.py:263) plus the `.module` wrappers of models/data_parallel.py.
Utterances are independent, so the path shards with no data-path collective other than
the gradient reduction.  `wrap_ddp` keeps torch's DDP (so train.py runs unchanged) but
(a) drops the per-forward buffer broadcast (the only buffers are constant tables) and
(b) uses large buckets: xGMI is 7 point-to-point links per GPU, few large collectives
beat many 25 MB ones.
"""
import torch
import torch.nn as nn


class CPUWrapperASR(nn.Module):
    """models/data_parallel.py:54-60 -- adds `.module` so train.py's `model.module.*` works."""

    def __init__(self, model):
        super().__init__()
        self.module = model

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def wrap_ddp(model, local_rank=None, bucket_cap_mb=128):
    """DDP(model) with the settings above; local_rank=None wraps a CPU module (gloo tests)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    kw = dict(broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    if local_rank is None:
        return DDP(model, **kw)
    return DDP(model, device_ids=[local_rank], **kw)


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
