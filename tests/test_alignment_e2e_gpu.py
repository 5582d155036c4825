"""SURVEY 8 row f3 END TO END on the device: a real Speech2Text with a CTC head ->
bin/ctc_forced_align.main() (HIP encoder, CTC head and aligner kernel) -> alignment files -> the dataset-side loader
(alignment.collate_trigger_points = datasets/alignment.py:98-112 + datasets/asr/dataset.py:319-322) ->
`batch['trigger_points']` -> a DeCoT latency-training step (las.py:463-472,647-649).
Checked against the reference at three points: the FILE BYTES equal what the reference's own CLI wrote for the same model
and batch (tests/golden/ctc_align_files.pt, oracle/gen_align_files.py); the collated trigger points equal the ones the
reference's aligner produced (the fixture's `batch['trigger_points']`); loss and every gradient of the training step fed
from the files equal the reference's (fixture) within the golden gates."""
import argparse

import numpy as np
import pytest
import torch

from tests import alignment_common as ac

pytestmark = pytest.mark.gpu


def test_alignment_files_feed_latency_training_end_to_end(tmp_path):
    from neural_sp_amd import alignment, ops
    from neural_sp_amd.speech2text import Speech2Text
    ref, fix = ac.load()
    with ops.compute_mode('f32'):
        got, base = ac.run_cli(tmp_path, device='cuda')
        # (1) the files, byte for byte
        assert sorted(got) == sorted(ref['files'])
        for k in ref['files']:
            assert got[k] == ref['files'][k], (k, got[k], ref['files'][k])
        # (2) what the dataset would put into the batch
        ys = fix['batch']['ys']
        tp = alignment.collate_trigger_points(base, ref['speakers'], ref['utt_ids'], ys)
        want = np.asarray(fix['batch']['trigger_points'])
        assert tp.dtype == np.int32 and tp.shape == want.shape and np.array_equal(tp, want)
        # (3) the training step fed from the files = the reference's step
        model = Speech2Text(argparse.Namespace(**fix['args']))
        model.load_state_dict(fix['state_dict'], strict=True)
        model.cuda()
        if fix['meta'].get('trigger_quantity_loss'):
            model.trigger_quantity_loss()
        in_memory = model.ctc_forced_align(fix['batch']['xs'], ys)               # the aligner's output, no files
        assert np.array_equal(in_memory[:, :tp.shape[1]], tp)
        batch = dict(fix['batch'], trigger_points=tp, xlens=[len(x) for x in fix['batch']['xs']], ys_sub1=[], ys_sub2=[])
        model.zero_grad(set_to_none=True)
        loss, obs = model(batch, task='all')
        loss.backward()
        torch.cuda.synchronize()
    assert abs(loss.item() - fix['loss']) / abs(fix['loss']) < 1e-4, (loss.item(), fix['loss'])
    worst = 0.0
    for n, p in model.named_parameters():
        g = fix['grads'].get(n)
        if g is None:
            continue
        e = ((p.grad.detach().cpu().reshape(g.shape) - g).abs().max() / max(g.abs().max().item(), 1e-8)).item()
        worst = max(worst, e)
    print('[f3 end to end] %d files byte-identical to the reference CLI; loss %.6f (reference %.6f); worst gradient error %.2e'
          % (len(got), loss.item(), fix['loss'], worst))
    assert worst < 2e-3, worst
