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
        ddp = DDP(model, device_ids=[0], bucket_cap_mb=1)                       # train.py:263 (small buckets: several at XS size)
    elif on_gpu and mode == 'install':
        import neural_sp_amd
        assert 'torch.nn.parallel.DistributedDataParallel' in neural_sp_amd.install()
        from torch.nn.parallel import DistributedDataParallel as DDP            # what train.py:20 now imports
        ddp = DDP(model, device_ids=[0], bucket_cap_mb=1)
        assert model._nsp_ddp_hooked and len(model._nsp_grad_accumulators) > 0
    elif on_gpu:
        ddp = parallel.wrap_ddp(model, 0, bucket_cap_mb=1, compress=compress)   # 1 MB: several buckets even at XS size
        assert len(model._nsp_grad_accumulators) > 0
    else:
        ddp = parallel.wrap_ddp(model, None, bucket_cap_mb=0.05)               # 50 kB buckets: several at this size
        ddp.register_comm_hook(None, parallel.make_comm_hook(parallel.step_streams(model), compress))
    local = sub_batch(full, list(range(rank, 4, world)))
    losses = []
    for it in range(2):                      # 2nd iteration runs on DDP's rebuilt (arrival-ordered) buckets
        ddp.zero_grad(set_to_none=True)
        loss, obs = ddp(local, task='all')
        loss = loss * world                  # train.py:423-424
        loss.backward()
        losses.append(loss.item())
    sync()
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
    dist.all_gather_object(gathered, (result_same, losses))
    if rank == 0:
        result.update(ddp=grads, same=[g[0] for g in gathered], losses=[g[1] for g in gathered])
        torch.save(result, out_path)
    dist.barrier()
    dist.destroy_process_group()
