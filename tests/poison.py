"""Poison mode -- TEST INFRASTRUCTURE ONLY.  `NSP_POISON=1` makes every `torch.empty` / `torch.empty_like` /
`Tensor.new_empty` hand out memory filled with NaN (floating point) or a large bit pattern (integers) instead of whatever
the caching allocator recycled.  A kernel that reads a workspace / output element it was supposed to have written first then
produces NaN (or an index far out of range) on EVERY run, instead of the right answer on fresh (zero) hipMalloc blocks and
a slightly wrong one on recycled blocks (the round-4 stock-DDP failure: rank 1's second iteration).  Both tiers: the device
suite (`NSP_POISON=1 pytest -m gpu`) and the host emulator (`NSP_POISON=1 pytest -m "not gpu"`).  The spawned DDP workers
call `enable_from_env()` themselves (a spawned interpreter does not run conftest)."""
import os

_STATE = {'on': False, 'count': 0}


def enable():
    if _STATE['on']:
        return
    import torch
    _STATE['on'] = True
    real_empty, real_empty_like, real_new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def _fill(t):
        if t.numel() == 0:
            return t
        _STATE['count'] += 1
        with torch.no_grad():
            if t.dtype.is_floating_point or t.dtype.is_complex:
                t.fill_(float('nan'))
            elif t.dtype == torch.bool:
                t.fill_(True)
            elif t.dtype == torch.uint8:
                t.fill_(0xA5)
            elif t.dtype == torch.int8:
                t.fill_(0x5A)
            elif t.dtype == torch.int16:
                t.fill_(0x5A5A)
            elif t.dtype == torch.int32:
                t.fill_(0x5A5A5A5A)
            else:
                t.fill_(0x5A5A5A5A5A5A5A5A)
        return t

    def empty(*a, **k):
        return _fill(real_empty(*a, **k))

    def empty_like(*a, **k):
        return _fill(real_empty_like(*a, **k))

    def new_empty(self, *a, **k):
        return _fill(real_new_empty(self, *a, **k))

    torch.empty = empty
    torch.empty_like = empty_like
    torch.Tensor.new_empty = new_empty


def enable_from_env():
    if os.environ.get('NSP_POISON', '0') not in ('', '0'):
        enable()
        return True
    return False


def count():
    return _STATE['count']
