#!/usr/bin/env python3
"""CTC forced alignment with a trained model on the HIP path (mirror of
neural_sp/bin/asr/ctc_forced_align.py:26-88).

    python -m neural_sp_amd.bin.ctc_forced_align --recog_model <ckpt> --recog_sets <tsv> --recog_dir <out> ...

Argument parsing, checkpoint averaging and the TSV/Kaldi data loader are the reference's own
(neural_sp.bin.args_asr / eval_utils / datasets.asr.build: Kaldi I/O, sentencepiece vocabularies); by default they are
imported from an installed `neural_sp`, and every one of them can be handed to `main()` instead (that is how
tests/test_alignment_cpu.py runs this function in an image without configargparse / kaldiio).  What this module replaces is
the model (neural_sp_amd.Speech2Text) and the aligner kernel.  Programmatic use: neural_sp_amd.alignment.align_batches."""
import logging
import os
import shutil
import sys

logger = logging.getLogger(__name__)


def main(argv=None, parse_args=None, average_checkpoints=None, set_logger=None, build_dataloader=None, model_cls=None,
         device='cuda'):
    """The reference's main() (ctc_forced_align.py:26-88) call for call; returns {recog set: utterances aligned}.
    parse_args(argv) -> (args, dir_name); average_checkpoints(model, path, n_average=); set_logger(path, stdout=);
    build_dataloader(args=, tsv_path=, batch_size=) -> iterable of batches with `.idx2token[0]`."""
    if None in (parse_args, average_checkpoints, set_logger, build_dataloader):
        try:
            from neural_sp.bin.args_asr import parse_args_eval
            from neural_sp.bin.eval_utils import average_checkpoints as ref_average
            from neural_sp.bin.train_utils import set_logger as ref_set_logger
            from neural_sp.datasets.asr.build import build_dataloader as ref_build
        except ImportError as e:
            raise SystemExit('this CLI drives the reference\'s argument parser / data loader (pip install neural_sp and its '
                             'Kaldi I/O dependencies): %s.  Use neural_sp_amd.alignment.align_batches(model, batches, dir, '
                             'idx2token) to align batches you load yourself.' % e)
        parse_args = parse_args or parse_args_eval
        average_checkpoints = average_checkpoints or ref_average
        set_logger = set_logger or ref_set_logger
        build_dataloader = build_dataloader or ref_build
    from neural_sp_amd.alignment import align_batches
    if model_cls is None:
        from neural_sp_amd.speech2text import Speech2Text as model_cls
    args, dir_name = parse_args(sys.argv[1:] if argv is None else argv)
    if os.path.isfile(os.path.join(args.recog_dir, 'align.log')):
        os.remove(os.path.join(args.recog_dir, 'align.log'))
    set_logger(os.path.join(args.recog_dir, 'align.log'), stdout=args.recog_stdout)
    model = model_cls(args, dir_name)
    average_checkpoints(model, args.recog_model[0], n_average=args.recog_n_average)
    if not args.recog_unit:
        args.recog_unit = args.unit
    if device == 'cuda':
        model.cuda()
    done = {}
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
        done[s] = n
    return done


if __name__ == '__main__':
    main()
