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


def wrap_ddp(model, local_rank, bucket_cap_mb=128):
    from torch.nn.parallel import DistributedDataParallel as DDP
    return DDP(model, device_ids=[local_rank], broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb,
               gradient_as_bucket_view=True)
