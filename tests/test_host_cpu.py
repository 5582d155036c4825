"""Host-side logic that needs no GPU: constructor parity of names/shapes with the fixtures
(= the reference's state_dict), length bookkeeping, SpecAugment RNG stream, configs."""
import argparse
import glob
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*_xs.pt')))


@pytest.mark.parametrize('name', CASES)
def test_state_dict_names_and_shapes_match_reference(name):
    from neural_sp_amd.speech2text import Speech2Text
    fix = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    model = Speech2Text(argparse.Namespace(**fix['args']))
    sd = model.state_dict()
    assert set(sd) == set(fix['state_dict'])
    for k, v in fix['state_dict'].items():
        assert sd[k].shape == v.shape, k
    model.load_state_dict(fix['state_dict'], strict=True)


def test_conformer_large_parameter_count():
    """90,572,656 parameters for Conformer-L + lstm_transducer (V=1000)."""
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.speech2text import Speech2Text
    m = Speech2Text(conformer_rnnt_args('L'))
    assert m.total_parameters == 90572656
    assert sum(p.numel() for n, p in m.named_parameters() if n.startswith('enc.')) == 73288608


def test_length_bookkeeping_matches_torch_pooling():
    from neural_sp_amd.encoders import _conv_len, _pool_len_ceil
    for n in range(1, 70):
        x = torch.zeros(1, 1, n)
        assert _pool_len_ceil(n, 2, 2) == torch.nn.functional.max_pool1d(x, 2, 2, ceil_mode=True).shape[-1]
        assert _conv_len(n, 3, 1, 1) == n


def test_specaugment_draws_follow_reference_rng_stream():
    """Same np.random call order as spec_augment.py:112-140 -> same bands for the same seed."""
    from neural_sp_amd.speech2text import SpecAugment
    sa = SpecAugment(F=27, T=100, n_freq_masks=2, n_time_masks=2, p=1.0)
    np.random.seed(5)
    fb, tb = sa.draw(n_frames=700, n_bins=80)
    np.random.seed(5)
    exp_f, exp_t = [], []
    for _ in range(2):
        f = int(np.random.uniform(0, 27))
        f0 = int(np.random.uniform(0, 80 - f))
        exp_f.append((f0, f0 + f))
    for _ in range(2):
        t = min(int(np.random.uniform(0, 100)), 700)
        t0 = int(np.random.uniform(0, 700 - t))
        exp_t.append((t0, t0 + t))
    assert fb == exp_f and tb == exp_t


def test_unsupported_configurations_fail_loudly():
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.speech2text import Speech2Text
    with pytest.raises(NotImplementedError):
        Speech2Text(conformer_rnnt_args('XS', n_layers=2, enc_type='tds'))
    with pytest.raises(NotImplementedError):
        Speech2Text(conformer_rnnt_args('XS', n_layers=2, subsample='2_1', subsample_type='no_such_type'))
    with pytest.raises(NotImplementedError):
        Speech2Text(conformer_rnnt_args('XS', n_layers=2, dec_type='gru_transducer'))


def test_weight_noise_adds_the_reference_scalar():
    """models/base.py:77-91 as executed: Normal([0.],[std]).sample([N]) is [N,1] and `add_(noise[0])` broadcasts
    ONE scalar to every parameter.  With the CPU generator in the same state the increment is the reference's,
    bit for bit (checked live against the reference when it is present, against its algorithm otherwise)."""
    import torch
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.speech2text import Speech2Text
    from oracle import ref_import
    args = conformer_rnnt_args('XS', n_layers=2, vocab=40, weight_noise_std=0.01)
    model = Speech2Text(args)
    before = [p.detach().clone() for p in model.parameters()]
    versions = [p._version for p in model.parameters()]
    torch.manual_seed(77)
    model.add_weight_noise(0.01)
    deltas = [(p.detach() - b) for p, b in zip(model.parameters(), before)]
    torch.manual_seed(77)
    n_total = sum(p.numel() for p in model.parameters())
    expect = (torch.empty(n_total, 1).normal_() * torch.tensor([0.01]))[0]        # what sample([N]) draws, row 0
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b + expect), 'same scalar on every element'
    assert all(p._version > v for p, v in zip(model.parameters(), versions)), 'weight shadows get invalidated'
    assert abs(deltas[0].flatten()[0].item()) > 0
    if ref_import.available():
        ref_import.import_reference()
        from neural_sp.models.seq2seq.speech2text import Speech2Text as RefS2T
        ref = RefS2T(args)
        ref.load_state_dict({k: v.clone() for k, v in zip(model.state_dict().keys(), [t.clone() for t in model.state_dict().values()])})
        rb = [p.detach().clone() for p in ref.parameters()]
        torch.manual_seed(77)
        ref.add_weight_noise(0.01)
        for p, b in zip(ref.parameters(), rb):
            assert torch.equal(p.detach(), b + expect)


def test_lazy_observation_behaves_like_the_reference_dict_of_floats():
    """speech2text.py:262-293 returns python floats; LazyObservation defers their transfer but
    must read like that dict in every access pattern train.py / Reporter use."""
    import pickle
    import torch
    from neural_sp_amd.speech2text import LazyObservation, Speech2Text
    vals = torch.tensor([1.5, 2.25])
    obs = LazyObservation({'loss.ctc': None, 'loss.transducer': None, 'acc.att': None}, ['loss.ctc', 'loss.transducer'], vals)
    assert obs['loss.ctc'] == 1.5 and isinstance(obs['loss.ctc'], float)
    assert obs.get('loss.transducer') == 2.25 and obs.get('missing', 7) == 7
    assert dict(obs.items()) == {'loss.ctc': 1.5, 'loss.transducer': 2.25, 'acc.att': None}
    assert obs == {'loss.ctc': 1.5, 'loss.transducer': 2.25, 'acc.att': None}
    assert pickle.loads(pickle.dumps(obs)) == dict(obs)
    assert 'loss.ctc' in obs and len(obs) == 3
    # _finalize_observation: tensors become lazily fetched floats, everything else passes through
    out = Speech2Text._finalize_observation({'loss.ctc': torch.tensor(3.0), 'loss.mbr': None})
    assert out['loss.ctc'] == 3.0 and out['loss.mbr'] is None
    assert Speech2Text._finalize_observation({'a': None}) == {'a': None}


def test_bench_cpu_core_detection_and_h2d_on_cpu():
    """bench._usable_cpus never exceeds the affinity mask; ops.h2d is a plain conversion when the
    target is the CPU (no pinned staging, no device)."""
    import importlib.util
    import os
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(__file__), '..', 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench._usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    from neural_sp_amd import ops
    t = ops.h2d(np.arange(6, dtype=np.int64).reshape(2, 3), 'cpu', torch.int32)
    assert t.dtype == torch.int32 and t.tolist() == [[0, 1, 2], [3, 4, 5]]


def test_lazy_observation_has_no_raw_storage_bypass():
    """dict(obs), {**obs}, dict.update(obs), keys()+lookup and json all see python floats, never the
    pending device tensors (CPython's dict fast paths skip __getitem__ for subclasses
    unless __iter__/keys are overridden)."""
    import json
    import torch
    from neural_sp_amd.speech2text import LazyObservation

    def make():
        return LazyObservation({'loss.ctc': torch.tensor(9.0), 'loss.transducer': torch.tensor(8.0), 'acc.att': None},
                               ['loss.ctc', 'loss.transducer'], torch.tensor([1.5, 2.25]))
    want = {'loss.ctc': 1.5, 'loss.transducer': 2.25, 'acc.att': None}
    assert dict(make()) == want and all(not torch.is_tensor(v) for v in dict(make()).values())
    assert {**make()} == want
    d = {}
    d.update(make())
    assert d == want
    o = make()
    assert {k: o[k] for k in o.keys()} == want
    assert json.loads(json.dumps(make())) == want
    assert list(make()) == list(want) and [v for v in make().values()] == list(want.values())


def test_train_py_member_surface_exists_on_cpu():
    """Every member neural_sp/bin/asr/train.py touches through `model.module` (train.py:154-157,238,
    260,316-320,402-403,442-443,486-487; lr_scheduler.py:218) exists and the no-op hooks return."""
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.parallel import CPUWrapperASR
    from neural_sp_amd.speech2text import Speech2Text
    m = CPUWrapperASR(Speech2Text(conformer_rnnt_args('XS', n_layers=4, vocab=40, ctc_weight=0.3, ctc_fc_list='32')))
    mod = m.module
    for hook in ('trigger_scheduled_sampling', 'trigger_quantity_loss', 'trigger_stableemit', 'reset_session',
                 'plot_attention', 'plot_ctc'):
        assert getattr(mod, hook)() is None
    mod.plot_attention()   # second call: the warn-once path
    mod.cudnn_setting(deterministic=True, benchmark=False)
    assert mod.total_parameters == sum(mod.num_params_dict.values()) == sum(p.numel() for p in mod.parameters())
    assert callable(mod.decode) and callable(mod.dec_fwd.greedy) and callable(mod.dec_fwd.ctc.greedy)
    assert mod.enc.subsampling_factor == 8 and mod.enc.output_dim == 64
    assert set(k.split('.')[0] for k in mod.state_dict()) == {'enc', 'dec_fwd'}


def test_dropout_seed_follows_torch_seed_and_rank(monkeypatch):
    """Dropout streams derive from torch's seed (train.py:57-58) and the data-parallel rank; sites
    get independent 64-bit stream seeds (no 2^24-site overflow of a shifted offset)."""
    import torch
    from neural_sp_amd import ops

    def first_sites(seed, rank):
        torch.manual_seed(seed)
        monkeypatch.setenv('RANK', str(rank))
        ops._DROPOUT_STATE.update(seed=None, counter=0)
        return [ops.next_dropout_seed() for _ in range(3)]
    a, b, c, d = first_sites(1, 0), first_sites(1, 0), first_sites(1, 1), first_sites(2, 0)
    assert a == b and a != c and a != d and c != d
    assert len({s for s, _ in a}) == 3 and all(o == 0 and 0 <= s < 2 ** 64 for s, o in a)
    ops._DROPOUT_STATE['counter'] = 1 << 30          # far beyond the old 2^24-site limit
    s, o = ops.next_dropout_seed()
    assert 0 <= s < 2 ** 64 and o == 0
    ops.manual_dropout_seed(5)
    assert ops.next_dropout_seed() == (ops._mix64(5 ^ (0xD1342543DE82EF95 & ops._M64)), 0)
    ops._DROPOUT_STATE.update(seed=None, counter=0)


def test_ctc_alignment_files_round_trip(tmp_path):
    """ctc_forced_align.py:72-83 writes `<token> <frame>` lines + `<eos> <frame>`; datasets/alignment.py:98-112
    reads the frame column back as np.int32 [L+1]; collate pads to [B, L_max+1]."""
    import numpy as np
    from neural_sp_amd import alignment
    tp = np.array([[3, 9, 17, 40], [5, 12, 0, 0]], dtype=np.int32)
    ys = [[11, 12, 13], [7]]
    toks = [['▁he', 'll', 'o'], ['▁a']]
    for b in range(2):
        path = alignment.write_ctc_alignment(str(tmp_path), 'spk%d' % b, 'utt%d' % b, toks[b], tp[b])
        assert path.endswith('spk%d/utt%d.txt' % (b, b))
    assert open(str(tmp_path / 'spk0' / 'utt0.txt'), encoding='utf-8').read() == '▁he 3\nll 9\no 17\n<eos> 40\n'
    assert alignment.load_ctc_alignment(str(tmp_path), 'spk1', 'utt1').tolist() == [5, 12]
    assert alignment.load_ctc_alignment(str(tmp_path), 'spk1', 'missing') is None
    got = alignment.collate_trigger_points(str(tmp_path), ['spk0', 'spk1'], ['utt0', 'utt1'], ys)
    assert got.dtype == np.int32 and got.tolist() == [[3, 9, 17, 40], [5, 12, 0, 0]]
    assert alignment.collate_trigger_points(str(tmp_path), ['spk0', 'nope'], ['utt0', 'x'], ys) is None

    class FakeModel(object):
        def ctc_forced_align(self, xs, ys_):
            return tp[:len(xs)]
    n = alignment.align_batches(FakeModel(), [{'xs': [0, 1], 'ys': ys, 'speakers': ['s', 's'], 'utt_ids': ['a', 'b']}],
                                str(tmp_path / 'out'), lambda ids, return_list=True: ['t%d' % i for i in ids])
    assert n == 2 and alignment.load_ctc_alignment(str(tmp_path / 'out'), 's', 'a').tolist() == [3, 9, 17, 40]


def test_bench_cpu_baseline_leg_runs_with_dropout_as_configured():
    """bench.py's cpu_baseline() -- the oracle port timed on a bounded sample -- at the XS size: the leg the driver's
    round-end bench runs on the GPU box's host cores (dropout 0.1 as bench configures it, full step incl. Adam)"""
    import argparse
    import bench
    from neural_sp_amd.configs import conformer_rnnt_args
    a = argparse.Namespace(size='XS', tmin=60, tmax=90, umin=5, umax=9, cpu_batch=2)
    margs = conformer_rnnt_args('XS', n_layers=2, vocab=40, dropout=0.1, ctc_weight=0.3)
    threads = torch.get_num_threads()
    try:
        out = bench.cpu_baseline(a, margs, 4)
    finally:
        torch.set_num_threads(threads)
    assert out['kind'] == 'port' and out['unit'] == 'frames/s' and out['value'] > 0 and out['cores'] == 4
    assert 'full training step' in out['sample'] or 'probe' in out['sample']


def test_install_substitutes_ddp_and_is_idempotent(monkeypatch):
    """neural_sp_amd.install(): what train.py:20 imports afterwards is a torch-DDP subclass that adds the multi-stream
    hook for this package's Speech2Text; a second call changes nothing."""
    import torch.nn.parallel as tnp
    import neural_sp_amd
    stock = tnp.DistributedDataParallel
    monkeypatch.setattr(tnp, 'DistributedDataParallel', stock)                 # restored after the test
    monkeypatch.setattr(tnp.distributed, 'DistributedDataParallel', stock)
    try:        # (an earlier test of this process may have imported the reference: its class name is restored as well)
        import neural_sp.models.seq2seq.speech2text as ref_mod
        monkeypatch.setattr(ref_mod, 'Speech2Text', ref_mod.Speech2Text)
    except ImportError:
        pass
    assert not getattr(stock, '_nsp_patched', False)
    done = neural_sp_amd.install()
    assert 'torch.nn.parallel.DistributedDataParallel' in done
    from torch.nn.parallel import DistributedDataParallel as DDP
    assert DDP is not stock and issubclass(DDP, stock) and DDP._nsp_patched and DDP.__name__ == 'DistributedDataParallel'
    assert neural_sp_amd.install() == []


def test_training_forward_without_the_ddp_hook_goes_single_stream(monkeypatch):
    """Speech2Text._ddp_guard: inside a process group of more than one rank, a model that did not go through wrap_ddp /
    the installed DDP class keeps its step on one stream (decoders' ensure_streams() -> (None, None))."""
    import torch.distributed as dist
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.speech2text import Speech2Text
    model = Speech2Text(conformer_rnnt_args('XS', n_layers=2, vocab=40))
    assert model._ddp_guard() is False and not model.dec_fwd._nsp_single_stream
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda *a: 2)
    assert model._ddp_guard() is True and model.dec_fwd._nsp_single_stream
    assert model.dec_fwd.ensure_streams() == (None, None)
    model._nsp_ddp_hooked = True                                               # what wrap_ddp / the installed class set
    assert model._ddp_guard() is False and not model.dec_fwd._nsp_single_stream


def test_kernel_event_classes_bookkeeping(monkeypatch):
    """ops._kev_class (bench.py's roofline.classes): off -> one shared no-op context; on -> an event pair and the work
    of every use under the class name, '<name>@side' when the current stream is not the default one (events and streams
    mocked: the bookkeeping is host logic)."""
    from neural_sp_amd import ops

    class Ev(object):
        n = 0

        def __init__(self, enable_timing=False):
            assert enable_timing
            self.recorded = False

        def record(self):
            Ev.n += 1
            self.recorded = True
    cur = ['main']
    monkeypatch.setattr(torch.cuda, 'Event', Ev)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a: cur[0])
    monkeypatch.setattr(torch.cuda, 'default_stream', lambda *a: 'main')
    monkeypatch.setitem(ops._KEV, 'on', False)
    a, b = ops._kev_class('ln', 10.0, 'byte'), ops._kev_class('flash', 3.0, 'flop')
    assert a is b
    with a:
        pass
    assert Ev.n == 0
    monkeypatch.setitem(ops._KEV, 'on', True)
    monkeypatch.setitem(ops._KEV, 'classes', {})
    for w in (10.0, 5.0):
        with ops._kev_class('ln', w, 'byte'):
            pass
    cur[0] = 'side'
    with ops._kev_class('ln', 1.0, 'byte'):
        pass
    cl = ops._KEV['classes']
    assert set(cl) == {'ln', 'ln@side'} and cl['ln']['work'] == 15.0 and cl['ln']['unit'] == 'byte'
    assert len(cl['ln']['events']) == 2 and len(cl['ln@side']['events']) == 1 and Ev.n == 6
    assert all(e0.recorded and e1.recorded for e0, e1 in cl['ln']['events'])


def test_bench_rank_to_device_binding_and_refusals():
    """bench.py --gpus N: device = LOCAL_RANK, and every way the launcher's environment can disagree with the job that was
    asked for ends in SystemExit before anything is measured (the N > 1 path has never met real devices)"""
    import importlib.util
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    env8 = lambda r: {'WORLD_SIZE': '8', 'RANK': str(r), 'LOCAL_RANK': str(r), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '1'}
    for r in range(8):
        assert bench.rank_binding(8, False, env8(r), 8) == (8, r, r, r)
    assert bench.rank_binding(1, False, {}, 1) == (1, 0, 0, 0)
    assert bench.rank_binding(2, True, {'WORLD_SIZE': '2', 'RANK': '1', 'LOCAL_RANK': '1'}, 1) == (2, 1, 1, 0)     # --same-device
    for args in [(4, False, env8(0), 8),                                       # --gpus 4 under an 8-rank launcher
                 (8, False, env8(5), 4),                                       # 8 ranks, 4 devices visible
                 (1, False, {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'}, 8),
                 (2, False, {'WORLD_SIZE': '2', 'RANK': '1'}, 8),              # no LOCAL_RANK
                 (2, False, {'WORLD_SIZE': '2', 'RANK': '1', 'LOCAL_RANK': '0'}, 8),     # two ranks -> one device without --same-device
                 (2, False, {'WORLD_SIZE': '2', 'RANK': '2', 'LOCAL_RANK': '2'}, 8)]:
        with pytest.raises(SystemExit):
            bench.rank_binding(*args)


def test_constant_buffers_are_not_broadcast_by_stock_ddp(tmp_path):
    """train.py:263 `DDP(model)` with torch's defaults broadcasts every buffer from rank 0 before every forward; the
    model's buffers that are constant tables (inv_freq) opt out through torch DDP's own `_ddp_params_and_buffers_to_ignore`,
    BatchNorm statistics (conformer_normalization: batch_norm) stay in; state_dict is unchanged"""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from neural_sp_amd.configs import conformer_rnnt_args
    from neural_sp_amd.speech2text import Speech2Text
    model = Speech2Text(conformer_rnnt_args('XS', n_layers=2, vocab=40, conformer_normalization='batch_norm'))
    names = [n for n, _ in model.named_buffers()]
    assert any(n.endswith('inv_freq') for n in names) and any(n.endswith('running_mean') for n in names)
    assert 'enc.pos_emb.inv_freq' in model.state_dict()
    dist.init_process_group('gloo', init_method='file://' + str(tmp_path / 'store'), rank=0, world_size=1)
    try:
        ddp = DDP(model)
        synced = set(id(b) for b in ddp.modules_buffers)
        by_name = dict(model.named_buffers())
        assert id(by_name['enc.pos_emb.inv_freq']) not in synced
        assert all(id(b) in synced for n, b in by_name.items() if n.endswith('running_mean') or n.endswith('running_var'))
    finally:
        dist.destroy_process_group()
