"""The reference's recipes, as its own argument parser hands them to Speech2Text (oracle/recipes.py), construct on
the HIP path with exactly the reference model's state_dict names and shapes -- so `train.py --config <recipe>` can be
pointed at neural_sp_amd.Speech2Text and reference checkpoints load.  Build container only (needs /root/reference);
the full sweep over all 119 recipes is tools/recipe_coverage.py -> RECIPES.md."""
import os

import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='reference not present on this machine')

# the recipes behind BASELINE.json's five configurations and their closest relatives
RECIPES = [
    'timit/s5/conf/blstm_ctc.yaml',                                                                        # config 1
    'librispeech/s5/conf/asr/transformer/transformer.yaml',                                                # config 2
    'librispeech/s5/conf/asr/transformer/conformer_kernel15_clamp10_hie_subsample8_las_long_ln.yaml',      # config 3
    'librispeech/s5/conf/asr/transformer/conformer_kernel15_clamp10_hie_subsample8_las_long_ln_large.yaml',   # config 4 (encoder)
    'librispeech/s5/conf/asr/transducer/blstm_transducer_bpe1k.yaml',                                      # config 4 (decoder)
    'librispeech/s5/conf/asr/mocha/uni_conformer_kernel7_clamp10_hie_subsample8_mocha_ln_stableemit0.2_qua0.2.yaml',   # config 5
    'aishell/s5/conf/asr/conformer_kernel15_clamp10_hie_subsample8_las_ln_2mtl.yaml',                      # multi-task
    'csj/s5/conf/asr/las/blstm_las.yaml',                                                                  # the BLSTM-LAS family
    'librispeech/s5/conf/asr/mocha/lcblstm_mocha_chunk4040_ctc_sync.yaml',                                 # streaming LC-BLSTM + MoChA
    'librispeech/s5/conf/asr/mma/streaming/lc_transformer_mma_subsample8_ma4H_ca4H_w16_from4L_512dmodel_8H_64_128_64.yaml',   # MMA
]


@pytest.mark.parametrize('recipe', RECIPES)
def test_recipe_constructs_with_the_reference_state_dict(recipe):
    from oracle.recipes import recipe_args
    from neural_sp_amd.speech2text import Speech2Text
    path = os.path.join(ref_import.REFERENCE_ROOT, 'examples', recipe)
    args = recipe_args(path)
    torch.manual_seed(0)
    ours = Speech2Text(args)
    ref_import.import_reference()
    from neural_sp.models.seq2seq.speech2text import Speech2Text as RefS2T
    ref = RefS2T(args)
    sd, rd = ours.state_dict(), ref.state_dict()
    assert set(sd) == set(rd), (sorted(set(sd) ^ set(rd))[:10])
    for k, v in rd.items():
        assert sd[k].shape == v.shape, k
    ours.load_state_dict(rd, strict=True)


def test_committed_recipe_table_is_current_enough():
    """RECIPES.md (tools/recipe_coverage.py) must list every recipe above as constructing"""
    table = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'RECIPES.md')).read()
    for r in RECIPES:
        line = [ln for ln in table.splitlines() if '`%s`' % r in ln]
        assert line and '| constructs |' in line[0], r
