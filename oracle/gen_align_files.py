#!/usr/bin/env python
"""TEST INFRASTRUCTURE (build container only): the alignment files the REFERENCE writes.

Runs the reference's own `neural_sp/bin/asr/ctc_forced_align.py:main()` -- its model (Speech2Text with the weights of the
fixture tests/golden/conformer_ctc_mocha_decot_xs.pt), its CTCForcedAligner, its file-writing loop (:66-85) and its
Idx2char vocabulary object -- on CPU.  The modules that loop imports for argument parsing, checkpoint averaging, logging and
the Kaldi data loader (configargparse / kaldiio are not in this image) are replaced by small stand-ins that hand it
the fixture's arguments, weights and ONE batch; nothing of the path under test is stubbed.
Output: tests/golden/ctc_align_files.pt = {'files': {relative path: bytes}, 'speakers', 'utt_ids', 'dict': [tokens],
'source': the fixture}.  tests/test_alignment_e2e_gpu.py makes the HIP path produce the same bytes; tests/test_oracle_cpu.py
re-runs this script's function live when /root/reference is present."""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SOURCE = 'conformer_ctc_mocha_decot_xs'


def batch_ids(n):
    return ['spk%d' % (b % 2) for b in range(n)], ['utt-%03d' % b for b in range(n)]


def vocabulary(vocab):
    """dictionary file lines `<token> <id>` as the reference's Idx2char reads them (ids 0..vocab-1; 0 = blank, 1 = unk,
    2 = eos, 3 = pad as in its recipes), tokens without spaces"""
    toks = ['<blank>', '<unk>', '<eos>', '<pad>'] + ['t%02d' % i for i in range(4, vocab)]
    return toks


def reference_files():
    from oracle.ref_import import import_reference
    import_reference()
    fix = torch.load(os.path.join(GOLDEN, SOURCE + '.pt'), weights_only=False)
    args = argparse.Namespace(**fix['args'])
    batch = dict(fix['batch'])
    speakers, utt_ids = batch_ids(len(batch['xs']))
    batch.update(speakers=speakers, utt_ids=utt_ids)
    toks = vocabulary(args.vocab)
    td = tempfile.mkdtemp()
    dict_path = os.path.join(td, 'dict.txt')
    with open(dict_path, 'w') as f:
        for i, t in enumerate(toks):
            f.write('%s %d\n' % (t, i))
    from neural_sp.datasets.token_converter.character import Idx2char

    class Loader(list):                       # `for batch in dataloader`, `len(dataloader)`, `dataloader.idx2token[0]`
        idx2token = [Idx2char(dict_path)]

    args.recog_dir = os.path.join(td, 'out')
    os.makedirs(args.recog_dir)
    args.recog_stdout, args.recog_model, args.recog_n_average = False, ['fixture'], 1
    args.recog_unit, args.recog_batch_size, args.recog_n_gpus, args.recog_sets = 'char', 4, 0, ['fixture.tsv']
    # stand-ins for the imports of the CLI that need configargparse / kaldiio (+ its logger set-up)
    def fake(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    names = ('neural_sp.bin.args_asr', 'neural_sp.bin.eval_utils', 'neural_sp.datasets.asr.build')
    saved = {k: sys.modules.get(k) for k in names}
    fake('neural_sp.bin.args_asr', parse_args_eval=lambda argv: (args, td))
    fake('neural_sp.bin.eval_utils', average_checkpoints=lambda model, path, n_average: model.load_state_dict(fix['state_dict']))
    fake('neural_sp.datasets.asr.build', build_dataloader=lambda args, tsv_path, batch_size: Loader([batch]))
    try:
        sys.modules.pop('neural_sp.bin.asr.ctc_forced_align', None)
        import neural_sp.bin.asr.ctc_forced_align as cli       # the reference's file, unmodified
        cli.set_logger = lambda *a, **k: None                  # (train_utils itself is the reference's: the model imports it)
        cli.main()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    out = {}
    base = os.path.join(args.recog_dir, 'ctc_forced_alignments')
    for dp, _, fns in os.walk(base):
        for fn in fns:
            p = os.path.join(dp, fn)
            out[os.path.relpath(p, base)] = open(p, 'rb').read()
    return {'files': out, 'speakers': speakers, 'utt_ids': utt_ids, 'dict': toks, 'source': SOURCE}


if __name__ == '__main__':
    res = reference_files()
    path = os.path.join(GOLDEN, 'ctc_align_files.pt')
    torch.save(res, path)
    for k in sorted(res['files']):
        print(k, res['files'][k].decode().replace('\n', ' | '))
    print('->', path)
