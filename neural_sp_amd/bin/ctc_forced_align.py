#!/usr/bin/env python3
"""CTC forced alignment with a trained model on the HIP path (mirror of
neural_sp/bin/asr/ctc_forced_align.py:26-88).

    python -m neural_sp_amd.bin.ctc_forced_align --recog_model <ckpt> --recog_sets <tsv> --recog_dir <out> ...

Argument parsing, checkpoint averaging and the TSV/Kaldi data loader are the reference's own
(neural_sp.bin.args_asr / eval_utils / datasets.asr.build: Kaldi I/O, sentencepiece vocabularies) and are
imported from an installed `neural_sp`; what this module replaces is the model (neural_sp_amd.Speech2Text)
and the aligner kernel.  Programmatic use without the reference: neural_sp_amd.alignment.align_batches."""
import logging
import os
import shutil
import sys

logger = logging.getLogger(__name__)


def main(argv=None):
    try:
        from neural_sp.bin.args_asr import parse_args_eval
        from neural_sp.bin.eval_utils import average_checkpoints
        from neural_sp.bin.train_utils import set_logger
        from neural_sp.datasets.asr.build import build_dataloader
    except ImportError as e:
        raise SystemExit('this CLI drives the reference\'s argument parser / data loader (pip install neural_sp and its '
                         'Kaldi I/O dependencies): %s.  Use neural_sp_amd.alignment.align_batches(model, batches, dir, '
                         'idx2token) to align batches you load yourself.' % e)
    from neural_sp_amd.alignment import align_batches
    from neural_sp_amd.speech2text import Speech2Text
    args, dir_name = parse_args_eval(sys.argv[1:] if argv is None else argv)
    if os.path.isfile(os.path.join(args.recog_dir, 'align.log')):
        os.remove(os.path.join(args.recog_dir, 'align.log'))
    set_logger(os.path.join(args.recog_dir, 'align.log'), stdout=args.recog_stdout)
    model = Speech2Text(args, dir_name)
    average_checkpoints(model, args.recog_model[0], n_average=args.recog_n_average)
    model.cuda()
    for s in args.recog_sets:
        args.min_n_frames = 0
        args.max_n_frames = 1e5
        dataloader = build_dataloader(args=args, tsv_path=s, batch_size=args.recog_batch_size)
        save_path = os.path.join(args.recog_dir, 'ctc_forced_alignments')
        if os.path.isdir(save_path):
            shutil.rmtree(save_path)
        os.makedirs(save_path, exist_ok=True)
        n = align_batches(model, dataloader, save_path, dataloader.idx2token[0])
        logger.info('%d utterances aligned -> %s' % (n, save_path))


if __name__ == '__main__':
    main()
