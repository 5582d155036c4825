"""Memory hygiene of the whole training step on the device (round 5; tools: tests/poison.py).

* guard bands: every device tensor the step allocates through torch.empty / zeros (and ops.zeros_small) sits between two
  4-KB bands of a byte pattern; after the step every band must be intact.  The host emulator's AddressSanitizer build
  bounds-checks the kernels it can run; the persistent (grid-barrier) LSTM kernels are not among them -- both launch modes
  are checked here.
* NaN poison: every `torch.empty*` hands out NaN-filled memory (recycled blocks hold the previous step's data, fresh ones
  zeros: a kernel that reads before writing otherwise looks right); the loss and every gradient must come out as in the
  unpoisoned step.
XS transducer (the small-tile kernels, 2 x 256 persistent LSTM, flash attention) and Conformer-L widths (8-phase GEMMs,
node-stationary joint, 2 x 1024 LSTM).
"""
import os

import pytest
import torch

from tests import poison

pytestmark = pytest.mark.gpu


def _models():
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    from tests import ddp_hip_worker as W

    def xs():
        args = W.model_args(small=False)
        torch.manual_seed(7)
        return Speech2Text(args).cuda(0), W.sub_batch(W.global_batch(args.vocab), [1, 3])

    def wide():
        args = conformer_rnnt_args('L', n_layers=2, vocab=1000, subsample='2_1')
        torch.manual_seed(3)
        return Speech2Text(args).cuda(0), synthetic_batch(B=8, t_range=(300, 420), u_range=(15, 40), vocab=1000, seed=11)
    return {'xs': xs, 'conformer_l_widths': wide}


def _step(model, batch):
    from neural_sp_amd import ops
    model.zero_grad(set_to_none=True)
    loss, _ = model(batch, task='all')
    loss.backward()
    torch.cuda.synchronize()
    ops.lstm_check()
    return loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('persistent', ['1', '0'])
@pytest.mark.parametrize('which', ['xs', 'conformer_l_widths'])
def test_no_kernel_writes_outside_its_tensors(which, persistent, monkeypatch):
    from neural_sp_amd import ops
    monkeypatch.setenv('NSP_LSTM_PERSISTENT', persistent)
    with ops.compute_mode('bf16'):
        model, batch = _models()[which]()
        l0, _ = _step(model, batch)
        with poison.guards() as g:
            l1, _ = _step(model, batch)
            bad, n = g.check()
        assert n > 100, n
        assert not bad, 'kernels wrote into the guard bands of: %r' % (bad[:8],)
        assert l1 == l0, (l0, l1)
    print('[guard bands, %s, persistent LSTM %s] %d allocations, all bands intact' % (which, persistent, n))


@pytest.mark.parametrize('which', ['xs', 'conformer_l_widths'])
def test_step_does_not_depend_on_what_the_allocator_hands_out(which):
    from neural_sp_amd import ops
    real = (torch.empty, torch.empty_like, torch.Tensor.new_empty)
    with ops.compute_mode('bf16'):
        model, batch = _models()[which]()
        l0, g0 = _step(model, batch)
        l0b, g0b = _step(model, batch)
        try:
            poison._STATE['on'] = False
            poison.enable()
            before = poison.count()
            l1, g1 = _step(model, batch)
            filled = poison.count() - before
        finally:
            torch.empty, torch.empty_like, torch.Tensor.new_empty = real
            poison._STATE['on'] = False
    assert filled > 100, filled
    assert l1 == l0 == l0b, (l0, l0b, l1)
    for n in g0:
        assert torch.isfinite(g1[n]).all(), n
        rerun = (g0b[n] - g0[n]).abs().max().item()
        assert (g1[n] - g0[n]).abs().max().item() <= max(4 * rerun, 2e-6 * g0[n].abs().max().item()), n
    print('[NaN-poisoned allocations, %s] %d allocations filled, loss and gradients unchanged' % (which, filled))
