"""On-disk CTC forced alignments.

File format of neural_sp/bin/asr/ctc_forced_align.py:72-83: one text file per utterance at
`<dir>/<speaker>/<utt_id>.txt`, one line per token `"<token> <frame>"` (frame = leftmost encoder frame of
the token on the best CTC path) and a final `"<eos> <frame>"` line; read back by
neural_sp/datasets/alignment.py:98-112 (`load_ctc_alignment`) into the np.int32 `[L+1]` rows that the
dataset hands to the model as `batch['trigger_points']` (datasets/asr/dataset.py:308-324).
The trigger points themselves come from the HIP aligner (Speech2Text.ctc_forced_align ->
nsp_ctc_forced_align, bit-exact against the reference's CTCForcedAligner)."""
import codecs
import os

import numpy as np


def write_ctc_alignment(alignment_dir, speaker, utt_id, tokens, trigger_points):
    """tokens: list[str] (length L); trigger_points: int array with at least L+1 entries."""
    spk_dir = os.path.join(alignment_dir, str(speaker))
    os.makedirs(spk_dir, exist_ok=True)
    path = os.path.join(spk_dir, str(utt_id) + '.txt')
    with codecs.open(path, 'w', encoding='utf-8') as f:
        for i, tok in enumerate(tokens):
            f.write('%s %d\n' % (tok, int(trigger_points[i])))
        f.write('%s %d\n' % ('<eos>', int(trigger_points[len(tokens)])))
    return path


def load_ctc_alignment(alignment_dir, speaker, utt_id):
    """datasets/alignment.py:98-112 -> np.int32 `[L+1]` or None if the file does not exist."""
    path = os.path.join(alignment_dir, str(speaker), str(utt_id) + '.txt')
    if not os.path.isfile(path):
        return None
    with codecs.open(path, 'r', encoding='utf-8') as f:
        boundaries = [int(line.strip().split(' ')[1]) for line in f]
    return np.array(boundaries, dtype=np.int32)


def align_batches(model, batches, alignment_dir, idx2token):
    """The loop of ctc_forced_align.py:66-85 over an iterable of reference-style batches
    (`xs`, `ys`, `speakers`, `utt_ids`); idx2token(ids, return_list=True) -> list[str].
    Returns the number of utterances written."""
    n = 0
    for batch in batches:
        tp = model.ctc_forced_align(batch['xs'], batch['ys'])          # `[B, L+1]` np.int32
        for b in range(len(batch['xs'])):
            tokens = idx2token(batch['ys'][b], return_list=True)
            write_ctc_alignment(alignment_dir, batch['speakers'][b], batch['utt_ids'][b], tokens, tp[b])
            n += 1
    return n


def collate_trigger_points(alignment_dir, speakers, utt_ids, ys):
    """`batch['trigger_points']` as the reference dataset builds it: np.int32 `[B, L_max+1]`, zero
    padded, None if any utterance has no alignment file."""
    rows = [load_ctc_alignment(alignment_dir, s, u) for s, u in zip(speakers, utt_ids)]
    if any(r is None for r in rows):
        return None
    lmax = max(len(y) for y in ys) + 1
    out = np.zeros((len(rows), lmax), dtype=np.int32)
    for b, r in enumerate(rows):
        out[b, :len(r)] = r
    return out
