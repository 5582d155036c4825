"""The training step beside a HOSTILE NEIGHBOUR on the same device (round 5).

Round 4's device suite failed on the driver's box with a loss that CHANGED between two identical iterations of one rank
while a second rank shared the device -- and never in a process that had the device to itself.  What a second tenant
changes for a kernel: (1) the LDS bytes and registers it finds are no longer what the previous kernel of its own stream
left there (deterministic leftovers -> a deterministic result that only LOOKS initialised); (2) LDS, issue slots and HBM
are contended, so every asynchronous load lands later (a hand-counted `s_waitcnt` that is one too lenient starts to
matter); (3) workgroups become resident later and in another order.

These tests run the step alone, then again while `nsp_debug_scribble` workgroups (NaN / huge-integer patterns into LDS,
VGPRs, AGPRs; short-lived, so CU slots keep changing hands) and big device-to-device copies run on side streams, and demand
the SAME loss bit for bit (dropout 0, no optimizer step: every forward kernel is deterministic) and gradients equal up to
the rounding of atomically accumulated sums.  Both the one-stream step (stock DistributedDataParallel, train.py:263) and the
three-stream step; XS shapes (the small-tile kernels) and Conformer-L shapes (8-phase GEMMs, node-stationary joint).

`test_*_beside_a_busy_second_process` is the case that found the cause: the neighbour is a second
PROCESS running library GEMMs (tests/gpu_neighbour.py).  The -O3 build of the first conv layer with packed fp32 VALU
instructions (v_pk_fma_f32 ...) failed one launch in three there; the library is now built without them (_lib.CFLAGS).
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

NAN_BITS = 0x7FC00000        # quiet NaN as fp32, two NaNs as bf16 pairs (0x7FC0 / 0x0000 -> the high half is NaN)
BIG_BITS = 0x7F7F7F7F        # 3.4e38 as fp32, 3.4e38 as bf16 halves, a huge index as an integer


class Neighbour(object):
    """scribble launches + HBM copies enqueued on side streams; `cover(ms)` enqueues about `ms` milliseconds of them"""

    def __init__(self, dev, pattern, lds_bytes=16384, rounds=8):
        from neural_sp_amd import _lib
        self.lib = _lib.lib()
        self.dev, self.pattern, self.lds_bytes, self.rounds = dev, pattern, lds_bytes, rounds
        self.s1 = torch.cuda.Stream(device=dev)
        self.s2 = torch.cuda.Stream(device=dev)
        self.a = torch.empty((64 << 20,), device=dev, dtype=torch.float32)      # 256 MB
        self.b = torch.empty_like(self.a)
        self.n_wg = 32768
        # calibrate one scribble launch / one copy on an idle device
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(self.s1):
            e0.record()
            self._scribble()
            e1.record()
            self.b.copy_(self.a)
            e2.record()
        torch.cuda.synchronize()
        self.ms_scribble, self.ms_copy = max(e0.elapsed_time(e1), 1e-3), max(e1.elapsed_time(e2), 1e-3)

    def _scribble(self):
        assert self.lib.nsp_debug_scribble(self.n_wg, self.lds_bytes, self.pattern, self.rounds,
                                           torch.cuda.current_stream(self.dev).cuda_stream) == 0

    def cover(self, ms):
        with torch.cuda.stream(self.s1):
            for _ in range(min(4000, int(ms / self.ms_scribble) + 1)):
                self._scribble()
        with torch.cuda.stream(self.s2):
            for _ in range(min(2000, int(0.5 * ms / self.ms_copy) + 1)):
                self.b.copy_(self.a)


def _step(model, batch):
    model.zero_grad(set_to_none=True)
    loss, _ = model(batch, task='all')
    loss.backward()
    return loss


def _run_case(model, batch, tag, single_stream):
    from neural_sp_amd import ops
    dev = model.device
    if single_stream:
        def guard(self=model):
            for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
                dec = getattr(self, name, None)
                if dec is not None and hasattr(dec, 'ensure_streams'):
                    dec._nsp_single_stream = True
            return True
        model._ddp_guard = guard
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    quiet = []
    for it in range(3):
        e0.record()
        loss = _step(model, batch)
        e1.record()
        torch.cuda.synchronize()
        ops.lstm_check()
        quiet.append((loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    ms_step = e0.elapsed_time(e1)
    assert quiet[0][0] == quiet[1][0] == quiet[2][0], ('undisturbed step is not deterministic', [q[0] for q in quiet])
    rerun = {n: max((quiet[i][1][n] - quiet[0][1][n]).abs().max().item() for i in (1, 2)) / max(quiet[0][1][n].abs().max().item(), 1e-30)
             for n in quiet[0][1]}
    worst_all = 0.0
    for pattern in (NAN_BITS, BIG_BITS):
        nb = Neighbour(dev, pattern)
        for it in range(3):
            nb.cover(6.0 * ms_step + 20.0)
            e0.record()
            loss = _step(model, batch)
            e1.record()
            torch.cuda.synchronize()
            ops.lstm_check()
            noisy_ms = e0.elapsed_time(e1)
            assert loss.item() == quiet[0][0], ('%s: loss moved beside the neighbour (pattern %#x, iteration %d): %r vs %r'
                                                % (tag, pattern, it, loss.item(), quiet[0][0]))
            worst, wn = 0.0, ''
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                assert torch.isfinite(p.grad).all(), (tag, n)
                d = (p.grad - quiet[0][1][n]).abs().max().item() / max(quiet[0][1][n].abs().max().item(), 1e-30)
                if d > max(4 * rerun[n], 2e-6):
                    raise AssertionError('%s: gradient of %s moved by %.2e beside the neighbour (pattern %#x; run-to-run %.1e)'
                                         % (tag, n, d, pattern, rerun[n]))
                if d > worst:
                    worst, wn = d, n
            worst_all = max(worst_all, worst)
        print('[%s, pattern %#x] step %.1f ms alone, %.1f ms beside the neighbour (scribble launch %.2f ms, 256 MB copy %.2f ms); '
              'loss bit-equal, worst gradient difference %.1e (%s)' % (tag, pattern, ms_step, noisy_ms, nb.ms_scribble, nb.ms_copy, worst, wn))
        del nb
    return worst_all


def _xs_model():
    from neural_sp_amd.speech2text import Speech2Text
    from tests import ddp_hip_worker as W
    args = W.model_args(small=False)
    torch.manual_seed(7)
    model = Speech2Text(args).cuda(0)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(torch.empty_like(p).uniform_(-0.1, 0.1))
    full = W.global_batch(args.vocab)
    return model, W.sub_batch(full, [1, 3])


@pytest.mark.parametrize('single_stream', [True, False])
def test_xs_transducer_step_beside_a_hostile_neighbour(single_stream):
    """the model, batch and mode of tests/test_ddp_hip_gpu.py's rank 1 (the rank whose loss moved in round 4)"""
    from neural_sp_amd import ops
    with ops.compute_mode('bf16'):
        model, batch = _xs_model()
        _run_case(model, batch, 'XS rnnt, %s' % ('one stream' if single_stream else 'three streams'), single_stream)


@pytest.mark.parametrize('mode', ['bf16', 'f32'])
def test_conformer_l_shapes_beside_a_hostile_neighbour(mode):
    """Conformer-L widths (d 512, d_ff 2048, 8 heads, 2 x 1024 LSTM, J 512, V 1000), 3 layers with one subsampling stage,
    24 utterances of up to 640 frames: the 8-phase GEMMs, the flash kernels at 3 waves, the node-stationary joint and the
    persistent LSTM as the bench runs them."""
    from neural_sp_amd import ops
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    args = conformer_rnnt_args('L', n_layers=3, vocab=1000, subsample='1_2_1')
    torch.manual_seed(3)
    with ops.compute_mode(mode):
        model = Speech2Text(args).cuda(0)
        batch = synthetic_batch(B=24 if mode == 'bf16' else 8, t_range=(400, 640), u_range=(20, 60), vocab=1000, seed=11)
        _run_case(model, batch, 'Conformer-L shapes, %s' % mode, single_stream=False)


class SecondProcess(object):
    """`with SecondProcess(seconds):` -- tests/gpu_neighbour.py on cuda:0, entered once it says READY"""

    def __init__(self, seconds=60):
        self.seconds = seconds

    def __enter__(self):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        self.p = subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'gpu_neighbour.py'), str(self.seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = self.p.stdout.readline()
        assert 'READY' in line, 'the neighbour process did not start: %r' % line
        return self

    def __exit__(self, *a):
        self.p.kill()          # (this very process, by its handle)
        self.p.wait()


def test_first_conv_layer_beside_a_busy_second_process():
    """nsp_conv2d3x3_fwd (C_in = 1, bf16 maps) 6000 times on NaN-prefilled outputs while a second process runs GEMMs: every
    launch must give the bits of the undisturbed launch (round-4 build: 36 % of the launches wrong beside this neighbour)"""
    from neural_sp_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device('cuda', 0)
    B, T, F = 2, 180, 80
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(B, T, F, 1, generator=g).to(dev)
    w = (torch.randn(32, 3, 3, 1, generator=g) * 0.3).to(dev)
    bias = (torch.randn(32, generator=g) * 0.1).to(dev)
    st = torch.cuda.current_stream().cuda_stream

    def conv(y):
        assert L.nsp_conv2d3x3_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, T, F, 1, 32, 1, None, 0, 1, st) == 0

    y_ref = torch.empty((B, T, F, 32), device=dev, dtype=torch.bfloat16)
    conv(y_ref)
    # the undisturbed result against plain torch arithmetic (fp32 taps, one bf16 rounding)
    xp = torch.nn.functional.pad(x[..., 0], (1, 1, 1, 1))
    want = bias.view(1, 1, 1, 32).expand(B, T, F, 32).clone()
    for dt in range(3):
        for df in range(3):
            want += xp[:, dt:dt + T, df:df + F, None] * w[:, dt, df, 0].view(1, 1, 1, 32)
    want = torch.relu(want)
    assert (y_ref.float() - want).abs().max().item() <= 1e-2 * want.abs().max().item()
    pool = [torch.empty_like(y_ref) for _ in range(6)]
    bad = torch.zeros(2, device=dev, dtype=torch.int64)
    with SecondProcess(60):
        for i in range(6000):
            y = pool[i % 6]
            y.fill_(float('nan'))
            conv(y)
            d = (y != y_ref)
            bad += torch.stack([d.any().long(), d.sum()])
        torch.cuda.synchronize()
    print('[first conv layer beside a second process] %d of 6000 launches differ (%d elements)' % (bad[0].item(), bad[1].item()))
    assert bad[0].item() == 0


@pytest.mark.parametrize('single_stream', [True, False])
def test_xs_transducer_step_beside_a_busy_second_process(single_stream):
    """the whole XS transducer step (model / batch of the DDP test's rank 1) 40 times beside the GEMM neighbour: the loss of
    every step equals the undisturbed loss bit for bit, the gradients up to the rounding of their atomic sums"""
    from neural_sp_amd import ops
    with ops.compute_mode('bf16'):
        model, batch = _xs_model()
        if single_stream:
            def guard(self=model):
                for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
                    dec = getattr(self, name, None)
                    if dec is not None and hasattr(dec, 'ensure_streams'):
                        dec._nsp_single_stream = True
                return True
            model._ddp_guard = guard
        quiet = []
        for it in range(3):
            loss = _step(model, batch)
            torch.cuda.synchronize()
            quiet.append((loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        assert quiet[0][0] == quiet[1][0] == quiet[2][0]
        rerun = {n: max((quiet[i][1][n] - quiet[0][1][n]).abs().max().item() for i in (1, 2)) / max(quiet[0][1][n].abs().max().item(), 1e-30)
                 for n in quiet[0][1]}
        moved, worst = [], 0.0
        with SecondProcess(60):
            for it in range(40):
                loss = _step(model, batch)
                torch.cuda.synchronize()
                ops.lstm_check()
                if loss.item() != quiet[0][0]:
                    moved.append((it, loss.item()))
                for n, p in model.named_parameters():
                    if p.grad is None:
                        continue
                    d = (p.grad - quiet[0][1][n]).abs().max().item() / max(quiet[0][1][n].abs().max().item(), 1e-30)
                    worst = max(worst, d)
                    assert d <= max(4 * rerun[n], 2e-6), (it, n, d, rerun[n])
        print('[XS rnnt step beside a second process, %s] 40 steps, loss moved in %d of them, worst gradient difference %.1e'
              % ('one stream' if single_stream else 'three streams', len(moved), worst))
        assert not moved, (quiet[0][0], moved)
