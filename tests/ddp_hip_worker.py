"""Worker of tests/test_ddp_hip_gpu.py (importable under the `spawn` start method): one of two
ranks sharing cuda:0, gloo rendezvous on 127.0.0.1 (RCCL refuses two ranks on one device)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def global_batch(vocab, t_range=(120, 180)):
    """4 utterances; ranks take [0,2] / [1,3] (rank-strided, sampler.py:96).  Utterances 0 and 1
    both have the maximum length, so each rank pads to the same T_max as the single-process run on
    all four (padding is live data in this model)."""
    import numpy as np
    from neural_sp_amd.configs import synthetic_batch
    b = synthetic_batch(B=4, t_range=t_range, u_range=(5, 12), vocab=vocab, seed=31)
    rng = np.random.RandomState(5)
    b['xs'][1] = rng.randn(len(b['xs'][0]), 80).astype('float32')
    b['xlens'][1] = len(b['xs'][1])
    return b


def sub_batch(b, idx):
    out = dict(b)
    for k in ('xs', 'xlens', 'ys', 'utt_ids', 'speakers', 'sessions', 'text', 'feat_path'):
        out[k] = [b[k][i] for i in idx]
    return out


def model_args(small=False):
    from neural_sp_amd.configs import conformer_rnnt_args
    if small:       # the CPU tier (emulated kernels): same structure -- d_k = 64 flash attention, 2-layer LSTM stack, CTC + RNN-T
        return conformer_rnnt_args('XS', n_layers=2, vocab=40, ctc_weight=0.3, ctc_fc_list='', ctc_lsm_prob=0.0,
                                   transformer_enc_d_model=64, transformer_enc_n_heads=1, transformer_enc_d_ff=128,
                                   conformer_kernel_size=7, dec_n_units=64, dec_n_layers=2, emb_dim=32,
                                   dec_bottleneck_dim=32)
    # d_k = 64 -> flash attention; 2x256 LSTM -> persistent stack; CTC + RNN-T -> all three streams
    return conformer_rnnt_args('XS', n_layers=2, vocab=40, ctc_weight=0.3, ctc_fc_list='', ctc_lsm_prob=0.0,
                               transformer_enc_d_model=128, transformer_enc_n_heads=2, transformer_enc_d_ff=256,
                               conformer_kernel_size=7, dec_n_units=256, dec_n_layers=2, emb_dim=64,
                               dec_bottleneck_dim=32)


def run(rank, world, port, out_path, compress, device='cuda', mode='wrap'):
    """device='cpu': the same two-rank run on the host-emulated kernels (tests/test_ddp_gloo_cpu.py) -- CPU tensors, gloo
    all-reduce through the CPU branch of the comm hook, no streams"""
    if device == 'cpu':
        import torch
        torch.set_num_threads(2)
        from tests.cpu_ops_shim import host_logic_on_cpu
        with host_logic_on_cpu(real_kernels=True, real_conv=False, mode='bf16'):
            return _run(rank, world, port, out_path, compress, device, mode)
    return _run(rank, world, port, out_path, compress, device, mode)


def _run(rank, world, port, out_path, compress, device, mode='wrap'):
    """mode: 'wrap' = parallel.wrap_ddp; 'stock' = train.py:263 as written, `DistributedDataParallel(model, device_ids)`
    of an unpatched torch (the model's own guard must keep the step on one stream); 'install' = the same line after
    `neural_sp_amd.install()` (the patched class adds the multi-stream hook)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from tests import poison
    poison.enable_from_env()
    from neural_sp_amd import ops, parallel
    from neural_sp_amd.speech2text import Speech2Text
    on_gpu = device == 'cuda'
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if on_gpu:
        torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    args = model_args(small=not on_gpu)
    torch.manual_seed(7)
    model = Speech2Text(args)
    if on_gpu:
        model.cuda(0)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(torch.empty_like(p).uniform_(-0.1, 0.1))
    full = global_batch(args.vocab) if on_gpu else global_batch(args.vocab, t_range=(60, 90))
    ops.set_compute_mode('bf16')
    result = {}
    if rank == 0:
        model.zero_grad(set_to_none=True)
        loss, _ = model(full, task='all')
        loss.backward()
        sync()
        result['single'] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
        result['single_loss'] = loss.item()
        model.zero_grad(set_to_none=True)
    if on_gpu and mode == 'stock':
        from torch.nn.parallel import DistributedDataParallel as DDP
        assert not getattr(DDP, '_nsp_patched', False)
        ddp = DDP(model, device_ids=[0], bucket_cap_mb=1,                       # train.py:263 (small buckets: several at XS size)
                  broadcast_buffers=os.environ.get('NSP_DDP_NO_BCAST_BUFFERS', '0') != '1')
    elif on_gpu and mode == 'install':
        import neural_sp_amd
        assert 'torch.nn.parallel.DistributedDataParallel' in neural_sp_amd.install()
        from torch.nn.parallel import DistributedDataParallel as DDP            # what train.py:20 now imports
        ddp = DDP(model, device_ids=[0], bucket_cap_mb=1)
        assert model._nsp_ddp_hooked and len(model._nsp_grad_accumulators) > 0
    elif on_gpu:
        if mode == 'wrap_rs_ag':          # the hook's reduce-scatter + all-gather form (odd bucket sizes: the padded path)
            os.environ['NSP_DDP_ALGO'] = 'rs_ag'
        ddp = parallel.wrap_ddp(model, 0, bucket_cap_mb=1, compress=compress)   # 1 MB: several buckets even at XS size
        assert len(model._nsp_grad_accumulators) > 0
    else:
        ddp = parallel.wrap_ddp(model, None, bucket_cap_mb=0.05)               # 50 kB buckets: several at this size
        ddp.register_comm_hook(None, parallel.make_comm_hook(parallel.step_streams(model), compress))
    local = sub_batch(full, list(range(rank, 4, world)))
    losses = []
    # NSP_DDP_ITERS / NSP_DDP_DIAG (round 5's loop script, profiles/r05_ddp_loop_variants*.log): more iterations per process, and a checksum of every
    # sub-module's output per iteration (device scalars, read after the loop: no extra synchronisation inside a step) --
    # the first module whose checksum differs from iteration 0 is where a nondeterministic forward starts
    n_iters = int(os.environ.get('NSP_DDP_ITERS', '2'))
    diag = os.environ.get('NSP_DDP_DIAG', '0') == '1'
    sums, cur = [], {}
    if diag:
        def hook(name):
            def f(mod, inp, out):
                outs = out if isinstance(out, (tuple, list)) else (out,)
                tot = []
                for o in outs:
                    if torch.is_tensor(o) and o.is_floating_point() and o.numel() > 1:
                        tot.append(o.detach().double().sum())
                        o16 = getattr(o, '_nsp16', None)
                        if torch.is_tensor(o16):
                            tot.append(o16.detach().double().sum())
                if tot:
                    cur[name] = torch.stack(tot)
            return f
        for n, m in model.named_modules():
            if n:
                m.register_forward_hook(hook(n))
        real_pad = ops.pad_batch

        def pad_batch(packed, offs, lens, *a, **k):
            out = real_pad(packed, offs, lens, *a, **k)
            if os.environ.get('NSP_DIAG_SYNC_AFTER_PAD', '0') == '1':
                torch.cuda.synchronize()
            cur['<pad_batch inputs: packed, offs, lens; output>'] = torch.stack(
                [packed.double().sum(), offs.double().sum(), lens.double().sum(), out.double().sum()])
            return out
        ops.pad_batch = pad_batch
        real_conv = ops._conv3x3_fwd
        ncall = [0]
        keepy = {}

        def conv_fwd(x, w_cl, bias, relu, mask_src=None, out16=False):
            y = real_conv(x, w_cl, bias, relu, mask_src=mask_src, out16=out16)
            if relu and ncall[0] == 0:
                # where does the first conv's output differ from iteration 0's?  [B,T,F,C] bounding box + count
                if 'y0' not in keepy:
                    keepy['y0'] = y.detach().clone()
                else:
                    bad = (y != keepy['y0']).any(dim=3)                     # [B,T,F]
                    idx = bad.nonzero()
                    if 'bad' not in cur:
                        big = torch.full((1, 3), 1 << 30, device=y.device, dtype=idx.dtype)
                        lo = torch.cat([idx, big]).min(dim=0).values
                        hi = torch.cat([idx, -big]).max(dim=0).values
                        cur['<conv#0 wrong pixels: count, b/t/f min, b/t/f max, sum|x| there>'] = torch.cat(
                            [bad.sum().view(1), lo, hi, (x.detach().reshape(bad.shape).abs() * bad).sum().view(1).long()]).double()
            if relu:
                z = torch.zeros((), device=y.device, dtype=torch.float64)
                cur['<conv3x3 fwd #%d: x, w, bias, y>' % ncall[0]] = torch.stack(
                    [x.double().sum(), w_cl.double().sum(), bias.double().sum() if bias is not None else z, y.double().sum()])
                ncall[0] += 1
            return y
        ops._conv3x3_fwd = conv_fwd
        real_pos = ops.xl_pos_table

        def pos_table(inv_freq, L):
            out = real_pos(inv_freq, L)
            cur['<xl_pos_table: inv_freq, out>'] = torch.stack([inv_freq.double().sum(), out.double().sum()])
            return out
        ops.xl_pos_table = pos_table
    grad_sums = []
    keep_alive = torch.zeros(64, device='cuda') if (on_gpu and os.environ.get('NSP_DDP_EXTRA_BCAST', '0') == '1') else None
    for it in range(n_iters):                # 2nd iteration runs on DDP's rebuilt (arrival-ordered) buckets
        cur.clear()
        if diag:
            ncall[0] = 0
        if keep_alive is not None:           # (diagnosis: the timing of DDP's per-forward buffer broadcast without its temporaries)
            dist.broadcast(keep_alive, 0)
        ddp.zero_grad(set_to_none=True)
        loss, obs = ddp(local, task='all')
        loss = loss * world                  # train.py:423-424
        loss.backward()
        losses.append(loss.item())
        if diag:
            sums.append(dict(cur))
            grad_sums.append({n: p.grad.detach().double().abs().sum() for n, p in model.named_parameters() if p.grad is not None})
    sync()
    moved = []
    if diag:
        for it in range(1, n_iters):
            mods = [n for n in sums[0] if n in sums[it] and not torch.equal(sums[it][n], sums[0][n])]
            mods = [n + (' delta %s' % (sums[it][n] - sums[0][n]).tolist() if n.startswith('<') else '') for n in mods]
            for n in sums[it]:
                if n.startswith('<conv#0 wrong') and sums[it][n][0].item() > 0:
                    mods.insert(0, n + ' = %s' % sums[it][n].tolist())
            gr = [(n, abs(grad_sums[it][n].item() / max(grad_sums[0][n].item(), 1e-30) - 1.0)) for n in grad_sums[0]]
            gr = sorted([g for g in gr if g[1] > 1e-4], key=lambda g: -g[1])
            if mods or gr or losses[it] != losses[0]:
                moved.append((it, losses[it], mods[:6], len(mods), gr[:6]))
    ops.lstm_check()
    if on_gpu:
        single_stream = bool(getattr(model.dec_fwd, '_nsp_single_stream', False))
        assert single_stream == (mode == 'stock'), (mode, single_stream)       # only the stock wrapper falls back to one stream
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    # every rank must hold the same averaged gradient
    flat = torch.cat([g.flatten() for g in grads.values()])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    result_same = bool(torch.equal(ref, flat))
    gathered = [None] * world
    dist.all_gather_object(gathered, (result_same, losses, moved))
    if rank == 0:
        result.update(ddp=grads, same=[g[0] for g in gathered], losses=[g[1] for g in gathered], moved=[g[2] for g in gathered])
        torch.save(result, out_path)
    dist.barrier()
    dist.destroy_process_group()
