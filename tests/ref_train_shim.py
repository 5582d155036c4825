"""Run the REFERENCE's own `neural_sp/bin/asr/train.py` -- unmodified, `main(args)` as its `__main__` block calls it --
around this package's Speech2Text.  TEST INFRASTRUCTURE ONLY.

What this file provides and nothing more:
  * stand-ins for third-party modules the reference imports that are not installed in this image (no network):
    `configargparse` (argparse + values from `--config` YAML files), `omegaconf` (attribute-style dict + YAML load / save /
    merge), `kaldiio.load_mat` (-> numpy.load), `tensorboardX.SummaryWriter`, `wandb`, `setproctitle`, `nltk.translate.bleu_score`.  Each is the
    smallest object that satisfies the calls train.py and the modules it imports make;
  * a tiny synthetic dataset in the reference's on-disk format (TSV with utt_id / speaker / feat_path / xlen / xdim / text /
    token_id / ylen / ydim, a `dict.txt`, one feature matrix per utterance), written to a temporary directory;
  * `run_train(argv, mode)`: mode 'substitute' rebinds `neural_sp.models.seq2seq.speech2text.Speech2Text` (INTEGRATION.md
    section 1, the three lines a maintainer adds); mode 'install' calls `neural_sp_amd.install()` instead.

The reference lives under /root/reference in the build container only: tests that use this module skip elsewhere.
"""
import argparse
import copy
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get('NSP_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isfile(os.path.join(REF_ROOT, 'neural_sp', 'bin', 'asr', 'train.py'))


# --------------------------------------------------------------------------------------------------------------------
# third-party stand-ins
# --------------------------------------------------------------------------------------------------------------------
class DictConfig(dict):
    """omegaconf.DictConfig as train.py uses it: attribute and item access, `in`, .get, .items(), setattr, deepcopy;
    nested mappings become DictConfig too."""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, DictConfig):
            v = DictConfig(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return DictConfig({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


class _OmegaConf(object):
    @staticmethod
    def load(path):
        import yaml
        with open(path) as fh:
            return DictConfig(yaml.safe_load(fh) or {})

    @staticmethod
    def save(config, path):
        import yaml
        with open(path, 'w') as fh:
            yaml.safe_dump(_plain(dict(config)), fh)

    @staticmethod
    def merge(*configs):
        out = DictConfig()
        for c in configs:
            for k, v in c.items():
                out[k] = v
        return out


class _ConfigArgParser(argparse.ArgumentParser):
    """configargparse.ArgumentParser: options marked `is_config_file=True` name YAML files whose values act as defaults
    (below the command line, above the parser's own defaults)."""

    def __init__(self, *a, config_file_parser_class=None, **k):
        super().__init__(*a, **k)
        self._cfg_dests = []

    def add_argument(self, *a, is_config_file=False, **k):
        act = super().add_argument(*a, **k)
        if is_config_file:
            self._cfg_dests.append(act.dest)
        return act

    add = add_argument

    def parse_known_args(self, args=None, namespace=None):
        import yaml
        if args is None:
            args = sys.argv[1:]
        first, _ = super().parse_known_args(list(args), None)
        from_files = {}
        for d in self._cfg_dests:
            path = getattr(first, d, None)
            if path:
                with open(path) as fh:
                    from_files.update(yaml.safe_load(fh) or {})
        known = {a.dest for a in self._actions}
        mine = {k: v for k, v in from_files.items() if k in known and not isinstance(v, dict)}
        saved = {k: self.get_default(k) for k in mine}
        self.set_defaults(**mine)
        try:
            return super().parse_known_args(list(args), namespace)
        finally:
            self.set_defaults(**saved)


class _SummaryWriter(object):
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, name, value, step=None):
        self.scalars.append((name, value, step))

    def close(self):
        pass


def install_stubs():
    """sys.modules entries for the absent third-party imports (idempotent; a really installed module is left alone)"""
    def absent(name):
        m = sys.modules.get(name)
        if m is not None:
            # a file-less module that is not one of ours is somebody else's minimal stand-in (oracle/ref_import.py's omegaconf
            # raises from load / save): replace it -- ours is a superset
            return getattr(m, '__file__', None) is None and not getattr(m, '_nsp_train_shim', False)
        try:
            __import__(name)
            return False
        except ImportError:
            return True

    if absent('configargparse'):
        m = types.ModuleType('configargparse')
        m.ArgumentParser = _ConfigArgParser
        m.YAMLConfigFileParser = object
        m.ArgumentDefaultsHelpFormatter = argparse.ArgumentDefaultsHelpFormatter
        sys.modules['configargparse'] = m
    if absent('omegaconf'):
        m = types.ModuleType('omegaconf')
        m.OmegaConf = _OmegaConf
        m.DictConfig = DictConfig
        sys.modules['omegaconf'] = m
    if absent('kaldiio'):
        m = types.ModuleType('kaldiio')
        m.load_mat = lambda path: np.load(path)
        sys.modules['kaldiio'] = m
    if absent('tensorboardX'):
        m = types.ModuleType('tensorboardX')
        m.SummaryWriter = _SummaryWriter
        sys.modules['tensorboardX'] = m
    if absent('wandb'):
        sys.modules['wandb'] = types.ModuleType('wandb')
    if absent('setproctitle'):
        m = types.ModuleType('setproctitle')
        m.setproctitle = lambda name: None
        sys.modules['setproctitle'] = m
    if absent('nltk'):
        # evaluators/wordpiece_bleu.py imports two functions at module level; BLEU is not a metric of these tests
        pkg, tr, bs = types.ModuleType('nltk'), types.ModuleType('nltk.translate'), types.ModuleType('nltk.translate.bleu_score')
        bs.corpus_bleu = bs.sentence_bleu = lambda *a, **k: 0.0
        pkg.translate, tr.bleu_score = tr, bs
        sys.modules.update({'nltk': pkg, 'nltk.translate': tr, 'nltk.translate.bleu_score': bs})
    for name in ('configargparse', 'omegaconf', 'kaldiio', 'tensorboardX', 'wandb', 'setproctitle', 'nltk'):
        if getattr(sys.modules.get(name), '__file__', None) is None:
            sys.modules[name]._nsp_train_shim = True
    # reference modules imported BEFORE this call (by another test of the same process) hold the objects of the stand-ins
    # that were in place then: rebind them
    fresh = {'OmegaConf': sys.modules['omegaconf'].OmegaConf, 'configargparse': sys.modules['configargparse'],
             'kaldiio': sys.modules['kaldiio'], 'SummaryWriter': sys.modules['tensorboardX'].SummaryWriter,
             'wandb': sys.modules['wandb'], 'setproctitle': sys.modules['setproctitle'].setproctitle}
    for modname, mod in list(sys.modules.items()):
        if modname.startswith('neural_sp.') and mod is not None:
            for attr, obj in fresh.items():
                if attr in getattr(mod, '__dict__', {}) and mod.__dict__[attr] is not obj:
                    setattr(mod, attr, obj)
    if absent('warprnnt_pytorch') and absent('warp_rnnt'):
        # the reference's decoders import one of them at module import time on the RNN-T path; the class is never called
        # here (this package's decoder computes the loss)
        m = types.ModuleType('warprnnt_pytorch')
        m.RNNTLoss = lambda *a, **k: (lambda *x, **y: (_ for _ in ()).throw(RuntimeError('reference RNN-T loss is absent')))
        sys.modules['warprnnt_pytorch'] = m


# --------------------------------------------------------------------------------------------------------------------
# a dataset in the reference's on-disk format
# --------------------------------------------------------------------------------------------------------------------
CHARS = list('abcdefghijklmnopqrst')


def write_dataset(root, n_train=6, n_dev=2, t_range=(44, 64), u_range=(3, 6), seed=0, input_dim=80):
    """-> dict(train_tsv, dev_tsv, dict): character-unit utterances with random features"""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'feat'), exist_ok=True)
    dict_path = os.path.join(root, 'dict.txt')
    # datasets/token_converter/character.py: `token id` per line, ids from 1 (0 = blank); <unk> <eos> <pad> as in the recipes
    tokens = ['<unk>', '<eos>', '<pad>'] + CHARS
    with open(dict_path, 'w') as fh:
        for i, t in enumerate(tokens):
            fh.write('%s %d\n' % (t, i + 1))
    out = {'dict': dict_path}
    for name, n in (('train', n_train), ('dev', n_dev)):
        rows = ['\t'.join(['utt_id', 'speaker', 'feat_path', 'xlen', 'xdim', 'text', 'token_id', 'ylen', 'ydim'])]
        for i in range(n):
            T = int(rng.randint(t_range[0], t_range[1] + 1))
            U = int(rng.randint(u_range[0], u_range[1] + 1))
            ids = [int(v) for v in rng.randint(4, len(tokens) + 1, size=U)]
            text = ''.join(tokens[v - 1] for v in ids)
            path = os.path.join(root, 'feat', '%s_%d.npy' % (name, i))
            np.save(path, rng.randn(T, input_dim).astype(np.float32))
            rows.append('\t'.join(['%s-utt%d' % (name, i), 'spk%d' % (i % 2), path, str(T), str(input_dim), text,
                                   ' '.join(map(str, ids)), str(U), str(len(tokens) + 1)]))
        tsv = os.path.join(root, '%s_char.tsv' % name)
        with open(tsv, 'w') as fh:
            fh.write('\n'.join(rows) + '\n')
        out[name + '_tsv'] = tsv
    return out


CONFIG = dict(
    # topology: the two-rank DDP test's CPU-tier model (flash attention at d_k = 64, 2-layer prediction network, CTC + RNN-T)
    n_stacks=1, n_skips=1, max_n_frames=1600, conv_in_channel=1, conv_channels='32_32', conv_kernel_sizes='(3,3)_(3,3)',
    conv_strides='(1,1)_(1,1)', conv_poolings='(1,1)_(2,2)', subsample='1_1', subsample_type='max_pool',
    enc_type='conv_conformer', enc_n_layers=2, transformer_enc_pe_type='relative', transformer_enc_d_model=64,
    transformer_enc_d_ff=128, transformer_enc_n_heads=1, transformer_enc_clamp_len=10, conformer_kernel_size=7,
    dec_type='lstm_transducer', dec_n_units=64, dec_n_projs=0, dec_n_layers=2, dec_bottleneck_dim=32, emb_dim=32,
    tie_embedding=False, ctc_fc_list='32',
    # optimisation: 2 epochs of 3 steps, dev loss every step, checkpoint per epoch, greedy validation in epoch 2
    batch_size=2, optimizer='adam', n_epochs=2, convert_to_sgd_epoch=100, print_step=1, metric='edit_distance', lr=1e-3,
    lr_decay_type='always', lr_decay_start_epoch=2, lr_decay_rate=0.85, lr_decay_patient_n_epochs=0,
    early_stop_patient_n_epochs=5, sort_stop_epoch=100, eval_start_epoch=2, warmup_start_lr=1e-4, warmup_n_steps=2,
    accum_grad_n_steps=1, param_init=0.1, clip_grad_norm=5.0,
    dropout_in=0.0, dropout_enc=0.1, dropout_dec=0.1, dropout_emb=0.1, dropout_att=0.0, weight_decay=1e-6,
    ctc_weight=0.3, ctc_lsm_prob=0.1, mtl_per_batch=False, task_specific_layer=False)


def write_config(root, **overrides):
    import yaml
    cfg = dict(CONFIG)
    cfg.update(overrides)
    path = os.path.join(root, 'conf.yaml')
    with open(path, 'w') as fh:
        yaml.safe_dump(cfg, fh)
    return path


# --------------------------------------------------------------------------------------------------------------------
# the run
# --------------------------------------------------------------------------------------------------------------------
def import_train(mode):
    """the reference's train module with this package's Speech2Text bound in: 'substitute' = the three lines of
    INTEGRATION.md section 1; 'install' = `neural_sp_amd.install()` before the script's imports"""
    assert available(), 'the reference is not present'
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    install_stubs()
    for name in [n for n in sys.modules if n == 'neural_sp.bin.asr.train']:
        del sys.modules[name]
    import neural_sp_amd
    import neural_sp_amd.speech2text as amd
    import neural_sp.models.seq2seq.speech2text as ref_mod0
    if not hasattr(ref_mod0, '_nsp_reference_class') and ref_mod0.Speech2Text is not amd.Speech2Text:
        ref_mod0._nsp_reference_class = ref_mod0.Speech2Text          # (run_train puts it back: other tests build the reference)
    if mode == 'install':
        done = neural_sp_amd.install()
        assert 'neural_sp.models.seq2seq.speech2text.Speech2Text' in done or _ref_class() is amd.Speech2Text, done
    else:
        import neural_sp.models.seq2seq.speech2text as ref_mod
        ref_mod.Speech2Text = amd.Speech2Text
    import neural_sp.bin.asr.train as train
    assert train.Speech2Text is amd.Speech2Text, 'train.py did not pick up the substituted class'
    return train


def _ref_class():
    import neural_sp.models.seq2seq.speech2text as ref_mod
    return ref_mod.Speech2Text


def run_train(data, conf, save_dir, mode='substitute', n_gpus=0, resume=None, extra=()):
    """parse_args_train(argv) + main(args), exactly as train.py's `__main__` block does for a single process"""
    train = import_train(mode)
    argv = ['--corpus', 'ci_test', '--config', conf, '--n_gpus', str(n_gpus), '--train_set', data['train_tsv'],
            '--dev_set', data['dev_tsv'], '--eval_sets', data['dev_tsv'], '--unit', 'char', '--dict', data['dict'],
            '--model_save_dir', save_dir, '--stdout', 'false', '--remove_old_checkpoints', 'false', '--workers', '0'] + list(extra)
    if resume:
        argv += ['--resume', resume]
    import logging
    old = sys.argv
    sys.argv = ['train.py'] + argv          # (parse_args_train ends with parser.parse_args(), which reads sys.argv)
    # train.py's set_logger is logging.basicConfig(filename=train.log): a no-op when the root logger already has handlers
    # (pytest's capture handlers, an earlier run's file) -- give it the clean root logger a fresh interpreter has
    root_logger = logging.getLogger()
    saved_handlers, saved_level = root_logger.handlers[:], root_logger.level
    root_logger.handlers = []
    try:
        args = train.parse_args_train(argv)
        args.distributed = args.n_gpus > 1 and args.local_world_size > 1      # train.py:563
        assert not args.distributed
        return train.main(args), train
    finally:
        sys.argv = old
        import neural_sp.models.seq2seq.speech2text as ref_mod0
        if hasattr(ref_mod0, '_nsp_reference_class'):
            ref_mod0.Speech2Text = ref_mod0._nsp_reference_class
        for h in root_logger.handlers:
            h.close()
        root_logger.handlers = saved_handlers
        root_logger.setLevel(saved_level)
