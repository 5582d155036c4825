"""Host glue mirrored from the reference (neural_sp/models/torch_utils.py:15-94).

These are list/array <-> tensor conversions on the host; no model arithmetic."""
import torch


def tensor2np(x):
    if x is None:
        return x
    return x.cpu().detach().numpy()


def tensor2scalar(x):
    if isinstance(x, float):
        return x
    return x.cpu().detach().item()


def np2tensor(array, device=None):
    return torch.from_numpy(array).to(device)


def pad_list(xs, pad_value=0., pad_left=False):
    """list of `[T_i, ...]` tensors -> `[B, T_max, ...]` (torch_utils.py:56-77)."""
    bs = len(xs)
    max_time = max(x.size(0) for x in xs)
    xs_pad = xs[0].new_zeros(bs, max_time, *xs[0].size()[1:]).fill_(pad_value)
    for b in range(bs):
        if len(xs[b]) == 0:
            continue
        if pad_left:
            xs_pad[b, -xs[b].size(0):] = xs[b]
        else:
            xs_pad[b, :xs[b].size(0)] = xs[b]
    return xs_pad


def make_pad_mask(seq_lens):
    bs = seq_lens.size(0)
    max_time = seq_lens.max()
    seq_range = torch.arange(0, max_time, dtype=torch.int32, device=seq_lens.device)
    seq_range = seq_range.unsqueeze(0).expand(bs, max_time)
    return seq_range < seq_lens.unsqueeze(-1)


def repeat(module, n_layers):
    import copy
    return torch.nn.ModuleList([copy.deepcopy(module) for _ in range(n_layers)])
