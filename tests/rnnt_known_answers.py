"""Third-party known answers for the RNN-T lattice loss -- TEST DATA + helpers (tests/ only).

The reference computes this loss in HawkAaron/warp-transducer (CPU: `warprnnt_pytorch.RNNTLoss()`, rnn_transducer.py:254-256)
or 1ytic/warp-rnnt (GPU, rnn_transducer.py:248-252); neither is vendored under /root/reference nor installable here (SURVEY
section 8c).  warp-transducer's own unit tests publish known-answer cases, which torchaudio's RNN-T loss tests reuse verbatim:

  * `small_test` (warp-transducer tests/test_cpu.cpp, test_gpu.cpp; torchaudio `get_B1_T2_U3_D5_data`): B = 1, T = 2,
    U + 1 = 3, V = 5 activations (log-softmax is applied by the loss), labels {1, 2}, blank 0: cost 4.495666 and the
    gradient with respect to the activations below;
  * `options_test` (same files; torchaudio `get_B2_T4_U3_D3_data`): B = 2, T = 4, U + 1 = 3, V = 3, labels {1, 2} / {1, 1}:
    costs 4.2806528590890736 and 3.9384369822503591.

They are restated here from the published tests (no file of either project is available in this container); that the
transcription is right is shown by the independent fp64 restatement of Graves 2012 in oracle/rnnt_ref.py reproducing all
33 numbers to their printed precision (tests/test_oracle_cpu.py) -- an accidental agreement of two wrong things on 33
numbers is not a possibility worth considering.  The kernels are then checked against these numbers directly, not via the
oracle (tests/test_kernels_emu_cpu.py on the host emulator, tests/test_kernels_conv_loss_gpu.py on the device).
"""
import torch

SMALL = dict(
    name='warp-transducer small_test (B1 T2 U3 V5)', shape=(1, 2, 3, 5), labels=[[1, 2]], elens=[2], ylens=[2],
    acts=[0.1, 0.6, 0.1, 0.1, 0.1, 0.1, 0.1, 0.6, 0.1, 0.1, 0.1, 0.1, 0.2, 0.8, 0.1,
          0.1, 0.6, 0.1, 0.1, 0.1, 0.1, 0.1, 0.2, 0.1, 0.1, 0.7, 0.1, 0.2, 0.1, 0.1],
    costs=[4.495666],
    grads=[-0.13116688, -0.3999269, 0.17703125, 0.17703125, 0.17703125,
           -0.18572757, 0.12247056, -0.18168412, 0.12247056, 0.12247056,
           -0.32091254, 0.06269141, 0.06928472, 0.12624499, 0.06269141,
           0.05456069, -0.21824276, 0.05456069, 0.05456069, 0.05456069,
           0.12073959, 0.12073959, -0.48295835, 0.12073959, 0.12073959,
           -0.6925882, 0.16871116, 0.18645467, 0.16871116, 0.16871116])
OPTIONS = dict(
    name='warp-transducer options_test (B2 T4 U3 V3)', shape=(2, 4, 3, 3), labels=[[1, 2], [1, 1]], elens=[4, 4], ylens=[2, 2],
    acts=[0.065357, 0.787530, 0.081592, 0.529716, 0.750675, 0.754135, 0.609764, 0.868140, 0.622532, 0.668522, 0.858039,
          0.164539, 0.989780, 0.944298, 0.603168, 0.946783, 0.666203, 0.286882, 0.094184, 0.366674, 0.736168, 0.166680,
          0.714154, 0.399400, 0.535982, 0.291821, 0.612642, 0.324241, 0.800764, 0.524106, 0.779195, 0.183314, 0.113745,
          0.240222, 0.339470, 0.134160, 0.505562, 0.051597, 0.640290, 0.430733, 0.829473, 0.177467, 0.320700, 0.042883,
          0.302803, 0.675178, 0.569537, 0.558474, 0.083132, 0.060165, 0.107958, 0.748615, 0.943918, 0.486356, 0.418199,
          0.652408, 0.024243, 0.134582, 0.366342, 0.295830, 0.923670, 0.689929, 0.741898, 0.250005, 0.603430, 0.987289,
          0.592606, 0.884672, 0.543450, 0.660770, 0.377128, 0.358021],
    costs=[4.2806528590890736, 3.9384369822503591], grads=None)
CASES = [SMALL, OPTIONS]


def tensors(ka, dtype=torch.float32):
    acts = torch.tensor(ka['acts'], dtype=dtype).view(ka['shape'])
    return (acts, torch.tensor(ka['labels'], dtype=torch.int32), torch.tensor(ka['elens'], dtype=torch.int32),
            torch.tensor(ka['ylens'], dtype=torch.int32))


def through_padded_kernels(L, ka, to_dev=lambda t: t, stream=0):
    """nsp_rnnt_logsoftmax_gather -> nsp_rnnt_lattice -> nsp_rnnt_grad_logits (csrc/rnnt.hip): (nll [B], d sum(nll) / d acts)"""
    acts, labels, elens, ylens = tensors(ka)
    B, T, U1, V = acts.shape
    work, labels, elens, ylens = (to_dev(t.contiguous()) for t in (acts.clone(), labels, elens, ylens))
    f = lambda *s: to_dev(torch.zeros(*s, dtype=torch.float32))
    lse, lpb, lpl, alpha, beta, gb, gl = (f(B, T, U1) for _ in range(7))
    nll = f(B)
    p = lambda t: t.data_ptr()
    assert L.nsp_rnnt_logsoftmax_gather(p(work), p(labels), p(elens), p(ylens), p(lse), p(lpb), p(lpl), B, T, U1, V, 0, stream) == 0
    assert L.nsp_rnnt_lattice(p(lpb), p(lpl), p(elens), p(ylens), p(alpha), p(beta), p(nll), p(gb), p(gl), B, T, U1, stream) == 0
    assert L.nsp_rnnt_grad_logits(p(work), p(lse), p(labels), p(gb), p(gl), p(elens), p(ylens), 1.0, None,
                                  B, T, U1, V, 0, None, 0, None, stream) == 0
    return nll.cpu(), work.cpu()


def through_compact_lattice(L, ka, to_dev=lambda t: t, stream=0):
    """nsp_rnnt_lattice_compact (csrc/rnnt_fused.hip: the lattice of the throughput path, nodes stored utterance by utterance
    as [T_b][U_b + 1]) on lp_blank / lp_label formed here in fp64 from the published activations; the kernel's occupancies
    g_blank / g_label = d nll / d lp_* are pulled back to the activations through the same log-softmax by autograd."""
    acts, labels, elens, ylens = tensors(ka, torch.float64)
    B, T, U1, V = acts.shape
    acts.requires_grad_(True)
    lp = torch.log_softmax(acts, -1)
    rows_b, rows_l, roff = [], [], [0]
    for b in range(B):
        Tb, Ub = int(elens[b]), int(ylens[b])
        for t in range(Tb):
            for u in range(Ub + 1):
                rows_b.append(lp[b, t, u, 0])
                rows_l.append(lp[b, t, u, int(labels[b, u])] if u < Ub else lp.new_tensor(float('-inf')))
        roff.append(len(rows_b))
    lpb64, lpl64 = torch.stack(rows_b), torch.stack(rows_l)
    M = roff[-1]
    lpb, lpl = to_dev(lpb64.detach().float().contiguous()), to_dev(lpl64.detach().float().contiguous())
    f = lambda *s: to_dev(torch.zeros(*s, dtype=torch.float32))
    alpha, beta, gb, gl, nll = f(M), f(M), f(M), f(M), f(B)
    el, yl, ro = to_dev(elens), to_dev(ylens), to_dev(torch.tensor(roff, dtype=torch.int64))
    p = lambda t: t.data_ptr()
    assert L.nsp_rnnt_lattice_compact(p(lpb), p(lpl), p(el), p(yl), p(ro), p(alpha), p(beta), p(nll), p(gb), p(gl), B, U1, stream) == 0
    gbc, glc = gb.cpu().double(), gl.cpu().double()
    glc = torch.where(torch.isinf(lpl64.detach()), torch.zeros_like(glc), glc)      # (no label arc out of the last column)
    lpl_safe = torch.where(torch.isinf(lpl64), torch.zeros_like(lpl64), lpl64)
    (g,) = torch.autograd.grad([lpb64, lpl_safe], [acts], [gbc, glc])
    return nll.cpu(), g.float()


def check(ka, nll, grad, tol_cost=2e-6, tol_grad=2e-6):
    for got, want in zip(nll.tolist(), ka['costs']):
        assert abs(got - want) <= tol_cost * max(1.0, abs(want)), (ka['name'], nll.tolist(), ka['costs'])
    if ka['grads'] is not None:
        want = torch.tensor(ka['grads']).view(ka['shape'])
        err = (grad.reshape(ka['shape']).double() - want.double()).abs().max().item()
        assert err <= tol_grad, (ka['name'], err)
