"""CPU tests of the oracle (runs in the build container and on the GPU box alike):
  * RNN-T lattice restatement vs exhaustive path enumeration and finite differences;
  * the functional model restatement (oracle/model_ref.py) vs fixtures produced by the
    REFERENCE itself (tests/golden/*.pt, oracle/gen_golden.py);
  * when /root/reference is present: the reference is re-run live against the fixtures.
"""
import argparse
import glob
import os

import pytest
import torch

from oracle import model_ref
from oracle.rnnt_ref import rnnt_loss_bruteforce, rnnt_loss_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*_xs.pt')))


@pytest.mark.parametrize('T,U', [(1, 0), (1, 1), (2, 2), (3, 1), (4, 3), (3, 4)])
def test_rnnt_lattice_vs_bruteforce(T, U):
    torch.manual_seed(T * 10 + U)
    V = 5
    lp = torch.log_softmax(torch.randn(1, T, U + 1, V, dtype=torch.float64), -1)
    lab = torch.randint(1, V, (1, max(U, 1)))
    ref = rnnt_loss_ref(lp, lab, torch.tensor([T]), torch.tensor([U]))[0]
    bf = rnnt_loss_bruteforce(lp[0], lab[0], T, U)
    assert abs(ref.item() - bf.item()) < 1e-10


def test_rnnt_lattice_gradient_finite_difference():
    torch.manual_seed(0)
    B, T, U, V = 2, 4, 3, 6
    logits = torch.randn(B, T, U + 1, V, dtype=torch.float64, requires_grad=True)
    lab = torch.randint(1, V, (B, U))
    elens, ylens = torch.tensor([4, 3]), torch.tensor([3, 2])

    def f(z):
        return rnnt_loss_ref(torch.log_softmax(z, -1), lab, elens, ylens).mean()
    assert torch.autograd.gradcheck(f, (logits,), eps=1e-6, atol=1e-6)
    # padded lattice nodes receive no gradient
    g, = torch.autograd.grad(f(logits), logits)
    assert g[1, 3:].abs().max() == 0 and g[1, :, 3:].abs().max() == 0


@pytest.mark.parametrize('name', CASES)
def test_model_restatement_matches_reference_fixture(name):
    fix = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    args = argparse.Namespace(**fix['args'])
    sd = {k: v.clone().double().requires_grad_(v.is_floating_point() and 'inv_freq' not in k and k != 'enc.pos_enc.pe')
          if v.is_floating_point() else v for k, v in fix['state_dict'].items()}
    qw = args.mocha_quantity_loss_weight if fix['meta'].get('trigger_quantity_loss') else 0.0
    bn_out = {}
    ss_seed = fix['meta'].get('scheduled_sampling_seed')
    if ss_seed is not None:
        import random
        random.seed(ss_seed)      # scheduled sampling: the same Python random stream as the reference's training forward
    loss, obs, eouts, elens = model_ref.speech2text_loss(sd, args, fix['batch'], torch.float64, quantity_weight=qw,
                                                         bn_out=bn_out, scheduled_sampling=ss_seed is not None,
                                                         stableemit=bool(fix['meta'].get('trigger_stableemit')),
                                                         ctc_trigger_points=fix.get('ctc_trigger_points'),
                                                         latency_weight=(getattr(args, 'mocha_latency_loss_weight', 0.0)
                                                                         if fix['meta'].get('trigger_quantity_loss') else 0.0))
    ref = fix['loss'].item()
    assert abs(loss.item() - ref) / abs(ref) < 2e-5, (loss.item(), ref)
    if bn_out:
        # batch_norm convolution modules: the fixture's encoder output / eval loss are the EVAL-mode ones, computed
        # with the running statistics the training step above leaves behind (gen_golden: train step, then eval)
        sd_eval = dict(sd)
        sd_eval.update(bn_out)
        with torch.no_grad():
            loss_eval, _, eouts, _ = model_ref.speech2text_loss(sd_eval, args, fix['batch'], torch.float64,
                                                                training=False, quantity_weight=qw)
        assert abs(loss_eval.item() - fix['loss_eval'].item()) / abs(ref) < 2e-5
    assert list(elens) == fix['elens'].tolist()
    assert (eouts.float() - fix['eout']).abs().max() / fix['eout'].abs().max() < 1e-4
    for k, v in fix['observation'].items():
        if v is not None:
            assert abs(obs[k] - v) <= 2e-5 * abs(v) + 1e-6, (k, obs[k], v)
    names = [n for n in fix['grads']]
    grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    gmax = max(r.abs().max().item() for r in fix['grads'].values())
    for n, g in zip(names, grads):
        r = fix['grads'][n]
        assert g is not None, n
        if getattr(args, 'conformer_normalization', '') == 'batch_norm' and n.endswith('.conv.depthwise_conv.bias'):
            # BatchNorm removes a per-channel shift: the true gradient is zero, the reference holds fp32 noise
            assert g.abs().max() < 1e-5 * gmax and r.abs().max() < 1e-5 * gmax, n
            continue
        # zero-gradient tensors (w_key.bias) hold only fp32 noise in the reference
        assert (g.float() - r).abs().max() / max(r.abs().max().item(), 1e-5 * gmax) < 2e-3, n


def test_reference_live_matches_fixture():
    """Re-run the reference here (build container only) and compare with the stored fixture."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip('reference not present on this machine')
    from oracle import gen_golden
    ref_import.import_reference()
    gen_golden.install_rnnt_stub()
    from neural_sp.models.seq2seq.speech2text import Speech2Text
    name = 'transformer_ctc_xs'
    fix = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    model = Speech2Text(argparse.Namespace(**fix['args']))
    model.load_state_dict(fix['state_dict'])
    batch = dict(fix['batch'])
    batch.update(xlens=[len(x) for x in batch['xs']], ys_sub1=[], ys_sub2=[], trigger_points=None)
    loss, _ = model(batch, task='all')
    assert abs(loss.item() - fix['loss'].item()) < 1e-4 * abs(fix['loss'].item())


@pytest.mark.parametrize('ka', __import__('tests.rnnt_known_answers', fromlist=['CASES']).CASES, ids=lambda k: k['name'].split(' (')[0].replace(' ', '_'))
def test_rnnt_oracle_reproduces_the_published_warp_transducer_answers(ka):
    """oracle/rnnt_ref.py (Graves 2012 restated) against the known-answer cases of HawkAaron/warp-transducer's own unit tests
    -- the library behind the reference's CPU path (rnn_transducer.py:254-256): costs to 1e-6, the 30 published gradient
    entries of `small_test` to 5e-7.  This pins the RNN-T row of the oracle to a third-party vector."""
    from tests import rnnt_known_answers as K
    acts, labels, elens, ylens = K.tensors(ka, torch.float64)
    acts.requires_grad_(True)
    nll = rnnt_loss_ref(torch.log_softmax(acts, -1), labels.long(), elens.long(), ylens.long(), blank=0)
    (g,) = torch.autograd.grad(nll.sum(), [acts])
    K.check(ka, nll, g, tol_cost=1e-6, tol_grad=5e-7)     # (the published gradients are float32 prints)
