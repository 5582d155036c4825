"""RNN-Transducer lattice loss: fp64-capable CPU restatement (test oracle only).

PARITY STATUS: pinned to the third-party library's own published known answers (round 5), not to
an execution of it.  The arithmetic of the reference's RNN-T loss lives in un-vendored
third-party libraries that are absent from /root/reference and cannot be installed here:
  * GPU: warp_rnnt==0.3 (tools/Makefile:144-146), call site
    neural_sp/models/seq2seq/decoders/rnn_transducer.py:248-252
    (rnnt_loss(log_probs, ys_out.int(), elens, ylens, average_frames=False,
     reduction='mean', gather=False));
  * CPU: warprnnt_pytorch, HawkAaron/warp-transducer at unpinned HEAD
    (tools/Makefile:133-142), call site rnn_transducer.py:254-256.
The reference's own test (test/decoders/test_rnn_transducer_decoder.py:62-84)
asserts only shape and loss >= 0.  This file restates the published algorithm
(Graves 2012, "Sequence Transduction with Recurrent Neural Networks", eq. 16-18)
with the call-site semantics: blank = 0, labels padded with blank
(rnn_transducer.py:233), per-utterance -log P reduced by the mean over the batch.
It is pinned by (a) the known-answer cases of warp-transducer's own unit tests (`small_test`: cost
4.495666 + 30 gradient entries; `options_test`: costs 4.28065285908907 / 3.93843698225036; restated in
tests/rnnt_known_answers.py, reproduced by this file to 1e-6 / 5e-7 and by the HIP lattice kernels --
padded and compact -- on the emulator and on the device), (b) exhaustive path enumeration on tiny
lattices and (c) autograd vs finite differences (tests/test_oracle_cpu.py (test_rnnt_*)).  The reference
itself could not be executed for this head (its rnnt_loss import fails here), so "pinned by the
library's published vectors" is the strongest statement available.
"""
import itertools

import torch


def rnnt_loss_ref(log_probs, labels, elens, ylens, blank=0):
    """Per-utterance -log P(y|x).

    log_probs: [B,T,U+1,V] (already log-softmaxed, rnn_transducer.py:242)
    labels:    [B,U] long (padded with blank), elens/ylens: [B] long
    Differentiable through torch autograd (any float dtype; use fp64 for an oracle).
    """
    B = log_probs.size(0)
    out = []
    for b in range(B):
        T, U = int(elens[b]), int(ylens[b])
        lp = log_probs[b]
        neg_inf = lp.new_full((), float('-inf'))
        alpha = [[None] * (U + 1) for _ in range(T)]
        for t in range(T):
            for u in range(U + 1):
                if t == 0 and u == 0:
                    alpha[t][u] = lp.new_zeros(())
                    continue
                x = alpha[t - 1][u] + lp[t - 1, u, blank] if t > 0 else neg_inf
                y = alpha[t][u - 1] + lp[t, u - 1, labels[b, u - 1]] if u > 0 else neg_inf
                alpha[t][u] = torch.logaddexp(x, y)
        out.append(-(alpha[T - 1][U] + lp[T - 1, U, blank]))
    return torch.stack(out)


def rnnt_loss_bruteforce(log_probs, labels, elen, ylen, blank=0):
    """-log of the sum over ALL alignments (T blanks interleaved with U labels), one utterance.

    An alignment is a sequence of T+U moves; the last move must be the final blank.
    Exponential -- only for T,U <= 4.
    """
    T, U = int(elen), int(ylen)
    total = None
    # choose positions of the U label emissions among the first T+U-1 moves
    for pos in itertools.combinations(range(T + U - 1), U):
        t = u = 0
        lp = log_probs.new_zeros(())
        for step in range(T + U):
            if step in pos:
                lp = lp + log_probs[t, u, labels[u]]
                u += 1
            else:
                lp = lp + log_probs[t, u, blank]
                t += 1
        assert t == T and u == U
        total = lp if total is None else torch.logaddexp(total, lp)
    return -total


def rnnt_loss_ref_diag(log_probs, labels, elens, ylens, blank=0):
    """Same recursion as rnnt_loss_ref, vectorised over anti-diagonals (t+u = const) so that a
    T=200,U=200 lattice costs ~400 tensor ops instead of 40k Python iterations.  Used as the
    CPU-baseline implementation of the loss in bench.py; validated against rnnt_loss_ref in
    tests/test_oracle_cpu.py."""
    B = log_probs.size(0)
    out = []
    for b in range(B):
        T, U = int(elens[b]), int(ylens[b])
        lp = log_probs[b, :T, :U + 1]
        lpb = lp[:, :, blank]                                            # [T,U+1]
        if U > 0:
            lpl = lp[:, :U].gather(2, labels[b, :U].view(1, U, 1).expand(T, U, 1)).squeeze(2)  # [T,U]
        neg = lp.new_full((1,), float('-inf'))
        prev = lp.new_zeros(1)          # diagonal 0: alpha(0,0)
        prev_u0 = 0                     # u index of prev[0]
        for d in range(1, T + U):
            u_lo, u_hi = max(0, d - (T - 1)), min(U, d)
            us = torch.arange(u_lo, u_hi + 1)
            ts = d - us
            # from (t-1,u): valid when t>0 ; index in prev = u - prev_u0
            x = torch.cat([neg.expand(1), prev, neg.expand(1)])  # padded: prev index i -> x[i+1]
            ix = (us - prev_u0 + 1).clamp(0, x.numel() - 1)
            a_t = torch.where(ts > 0, x[ix] + lpb[(ts - 1).clamp(min=0), us], neg.expand(len(us)))
            iy = (us - 1 - prev_u0 + 1).clamp(0, x.numel() - 1)
            if U > 0:
                a_u = torch.where(us > 0, x[iy] + lpl[ts, (us - 1).clamp(min=0)], neg.expand(len(us)))
            else:
                a_u = neg.expand(len(us))
            prev = torch.logaddexp(a_t, a_u)
            prev_u0 = u_lo
        out.append(-(prev[-1] + lpb[T - 1, U]))
    return torch.stack(out)
