"""Argument bags for the BASELINE.json configurations.

`base_args()` carries every attribute that neural_sp's Speech2Text.__init__ reads on the
conv+(trans|con)former + CTC / RNN-T path, with the defaults of neural_sp/bin/args_asr.py,
args_common.py, encoders/transformer.py:275-322, encoders/conformer.py:113-164,
encoders/conv.py:107-125 and decoders/rnn_transducer.py:134-148.  The same Namespace
constructs both the reference model (oracle) and neural_sp_amd.Speech2Text.
"""
import argparse


def base_args(**kw):
    a = dict(
        input_type='speech', input_dim=80, enc_type='conv_conformer', dec_type='lstm_transducer',
        enc_n_layers=12, enc_n_layers_sub1=0, enc_n_layers_sub2=0, enc_n_units=512, enc_n_projs=0,
        subsample='1_1_1_1_1_1_1_1_1_1_1_1', subsample_type='max_pool',
        vocab=1000, vocab_sub1=-1, vocab_sub2=-1,
        total_weight=1.0, sub1_weight=0.0, sub2_weight=0.0, mtl_per_batch=False,
        task_specific_layer=False, ctc_weight=0.0, ctc_weight_sub1=0.0, ctc_weight_sub2=0.0,
        bwd_weight=0.0, mbr_training=False, mbr_ce_weight=0.0,
        input_noise_std=0, weight_noise_std=0, n_stacks=1, n_skips=1, n_splices=1,
        n_freq_masks=0, n_time_masks=0, freq_width=27, time_width=100, time_width_upper=1.0,
        adaptive_number_ratio=0.0, adaptive_size_ratio=0.0, max_n_time_masks=20,
        sequence_summary_network=False, param_init=0.1, freeze_encoder=False,
        external_lm=False, lm_fusion='', lm_init=False,
        emb_dim=512, dropout_emb=0.0, tie_embedding=False,
        conv_in_channel=1, conv_channels='32_32', conv_kernel_sizes='(3,3)_(3,3)',
        conv_strides='(1,1)_(1,1)', conv_poolings='(1,1)_(2,2)', conv_normalization='',
        conv_bottleneck_dim=0, cnn_lookahead=True, bidirectional_sum_fwd_bwd=False, rsp_prob_enc=0.0,
        transformer_enc_n_heads=8, transformer_enc_d_model=512, transformer_enc_d_ff=2048,
        transformer_enc_pe_type='relative', transformer_enc_clamp_len=10,
        transformer_enc_lookaheads='0_0_0_0_0_0_0_0_0_0_0_0',
        transformer_ffn_bottleneck_dim=0, transformer_ffn_activation='swish',
        transformer_layer_norm_eps=1e-12, transformer_param_init='xavier_uniform',
        transformer_dec_d_model=256, transformer_dec_d_ff=2048, transformer_dec_n_heads=4,
        conformer_kernel_size=15, conformer_normalization='layer_norm',
        dropout_in=0.0, dropout_enc=0.0, dropout_att=0.0, dropout_enc_layer=0.0, dropout_dec=0.0,
        lc_chunk_size_left='0', lc_chunk_size_current='0', lc_chunk_size_right='0', lc_type='reshape',
        dec_n_units=1024, dec_n_projs=0, dec_n_layers=2, dec_bottleneck_dim=512,
        ctc_lsm_prob=0.0, ctc_fc_list='', lsm_prob=0.0, ss_prob=0.0,
        attn_type='location', attn_dim=128, attn_sharpening_factor=1.0, attn_sigmoid=False,
        attn_conv_n_channels=10, attn_conv_width=201, attn_n_heads=1,
        mocha_chunk_size=1, mocha_n_heads_mono=1, mocha_n_heads_chunk=1, mocha_init_r=-4,
        mocha_eps=1e-6, mocha_std=1.0, mocha_no_denominator=False, mocha_1dconv=False,
        mocha_decot_lookahead=0, mocha_quantity_loss_weight=0.0, mocha_latency_metric='',
        mocha_latency_loss_weight=0.0, mocha_stableemit_weight=0.0, mocha_first_layer=1,
        mocha_head_divergence_loss_weight=0.0, share_chunkwise_attention=False,
        gmm_attn_n_mixtures=5, replace_sos=False, distillation_weight=0.1, discourse_aware=False,
        transformer_dec_attn_type='scaled_dot', transformer_dec_pe_type='add',
        dropout_dec_layer=0.0, dropout_head=0.0,
    )
    a.update(kw)
    return argparse.Namespace(**a)


def conformer_rnnt_args(size='L', n_layers=12, vocab=1000, dropout=0.0, ctc_weight=0.3, **kw):
    """BASELINE config 4: Conformer-{M,L} (examples/librispeech/s5/conf/asr/transformer/
    conformer_kernel15_clamp10_hie_subsample8_las_long_ln{,_large}.yaml) + RNN-T head
    (conf/asr/transducer/blstm_transducer_bpe1k.yaml:19-26,54-55)."""
    d, ff, h = {'L': (512, 2048, 8), 'M': (256, 1024, 4), 'S': (144, 576, 4), 'XS': (64, 128, 4)}[size]
    sub = ['1'] * n_layers
    if n_layers >= 8:
        sub[3] = '2'
        sub[7] = '2'
    elif n_layers >= 4:
        sub[1] = '2'
        sub[3] = '2'
    a = dict(enc_type='conv_conformer', dec_type='lstm_transducer', enc_n_layers=n_layers,
             subsample='_'.join(sub), subsample_type='max_pool', vocab=vocab,
             transformer_enc_d_model=d, transformer_enc_d_ff=ff, transformer_enc_n_heads=h,
             transformer_enc_pe_type='relative', transformer_enc_clamp_len=10,
             conformer_kernel_size=15, conformer_normalization='layer_norm',
             conv_poolings='(1,1)_(2,2)', ctc_weight=ctc_weight, ctc_lsm_prob=0.1 if ctc_weight > 0 else 0.0,
             ctc_fc_list='512' if size in ('L', 'M') else '32',
             dec_n_units=1024 if size in ('L', 'M') else 64, dec_n_layers=2,
             dec_bottleneck_dim=512 if size in ('L', 'M') else 32,
             emb_dim=512 if size in ('L', 'M') else 32,
             dropout_in=0.0, dropout_enc=dropout, dropout_att=dropout, dropout_dec=dropout,
             dropout_emb=dropout, dropout_enc_layer=0.0)
    a.update(kw)
    return base_args(**a)


def transformer_ctc_args(n_layers=12, d_model=256, d_ff=2048, n_heads=4, vocab=1000, dropout=0.0, **kw):
    """BASELINE config 2: Transformer-small encoder + CTC (librispeech transformer.yaml encoder
    half, CTC-only as in examples/ci_test/conf/asr/transformer_ctc.yaml)."""
    a = dict(enc_type='conv_transformer', dec_type='transformer', enc_n_layers=n_layers,
             subsample='_'.join(['1'] * n_layers), vocab=vocab, conv_poolings='(2,2)_(2,2)',
             transformer_enc_d_model=d_model, transformer_enc_d_ff=d_ff, transformer_enc_n_heads=n_heads,
             transformer_enc_pe_type='none', transformer_enc_clamp_len=-1,
             transformer_ffn_activation='relu', transformer_dec_d_model=d_model,
             ctc_weight=1.0, ctc_lsm_prob=0.0, ctc_fc_list='512' if d_model >= 128 else '16',
             dropout_in=dropout, dropout_enc=dropout, dropout_att=dropout, dropout_dec=dropout)
    a.update(kw)
    return base_args(**a)


def conformer_ctc_att_args(size='M', n_layers=12, vocab=10000, dropout=0.0, ctc_weight=0.3, dec_n_layers=6, **kw):
    """BASELINE config 3 family: Conformer-M encoder + hybrid CTC / attention loss with a Transformer
    decoder (decoders/transformer.py; LibriSpeech recipe shape: ctc_weight 0.3, lsm_prob 0.1, V = 10k)."""
    a = vars(conformer_rnnt_args(size, n_layers=n_layers, vocab=vocab, dropout=dropout, ctc_weight=ctc_weight))
    d = a['transformer_enc_d_model']
    a.update(dec_type='transformer', dec_n_layers=dec_n_layers, transformer_dec_d_model=d,
             transformer_dec_d_ff=a['transformer_enc_d_ff'], transformer_dec_n_heads=a['transformer_enc_n_heads'],
             transformer_dec_attn_type='scaled_dot', transformer_dec_pe_type='add',
             transformer_ffn_activation='relu', lsm_prob=0.1, dropout_dec=dropout, dropout_dec_layer=0.0)
    a.update(kw)
    return argparse.Namespace(**a)


def conformer_ctc_las_args(size='M', n_layers=12, vocab=10000, dropout=0.0, ctc_weight=0.3, attn_type='location', **kw):
    """BASELINE config 3 as its recipe writes it: Conformer-M encoder + hybrid CTC / attention loss with the
    LSTM decoder (decoders/las.py; 1 x 1024 units, location-aware attention, lsm_prob 0.1, V = 10k).
    attn_type='mocha' gives the streaming decoder of config 5 (chunk size via mocha_chunk_size)."""
    a = vars(conformer_rnnt_args(size, n_layers=n_layers, vocab=vocab, dropout=dropout, ctc_weight=ctc_weight))
    big = size in ('L', 'M')
    a.update(dec_type='lstm', dec_n_layers=1, dec_n_units=1024 if big else 64, dec_n_projs=0,
             dec_bottleneck_dim=1024 if big else 48, emb_dim=512 if big else 32, attn_type=attn_type,
             attn_dim=512 if big else 40, attn_conv_n_channels=10, attn_conv_width=201 if big else 21,
             lsm_prob=0.1, dropout_dec=dropout, dropout_att=0.0, param_init=0.1)
    a.update(kw)
    return argparse.Namespace(**a)


def blstm_ctc_args(n_layers=5, n_units=256, vocab=64, dropout=0.0, **kw):
    """BASELINE config 1: TIMIT BLSTM-CTC (examples/timit/s5/conf/blstm_ctc.yaml: 5 x 256-unit BLSTM layers, no CNN,
    no subsampling, CTC only; SURVEY 8d: 40-dim features, ~64 output symbols)."""
    a = dict(enc_type='blstm', dec_type='lstm', enc_n_layers=n_layers, enc_n_units=n_units, enc_n_projs=0,
             subsample='_'.join(['1'] * n_layers), subsample_type='drop', input_dim=40, vocab=vocab,
             conv_poolings='(1,1)_(1,1)', ctc_weight=1.0, ctc_lsm_prob=0.0, ctc_fc_list='',
             dropout_in=dropout, dropout_enc=dropout, dropout_dec=dropout, dropout_emb=dropout, param_init=0.1,
             lc_chunk_size_left='0', lc_chunk_size_current='0', lc_chunk_size_right='0')
    a.update(kw)
    return base_args(**a)


def synthetic_batch(B, t_range, u_range, vocab, input_dim=80, seed=0, vocab_sub1=0, vocab_sub2=0):
    """The batch dict of datasets/asr/build.py:73-105 filled with synthetic data of the shapes in
    SURVEY.md section 8d: features ~ N(0,1), lengths uniform in the given ranges, labels ~ U[4,V).
    vocab_sub{1,2} > 0 add the transcripts of the auxiliary tasks (their own generator: the main stream of
    random numbers, hence every existing batch, is unchanged)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    tl = rng.randint(t_range[0], t_range[1] + 1, size=B)
    tl[0] = t_range[1]
    ul = rng.randint(u_range[0], u_range[1] + 1, size=B)
    xs = [rng.randn(int(t), input_dim).astype(np.float32) for t in tl]
    ys = [rng.randint(4, vocab, size=int(u)).tolist() for u in ul]
    subs = {}
    for k, (name, v) in enumerate((('ys_sub1', vocab_sub1), ('ys_sub2', vocab_sub2))):
        rs = np.random.RandomState(seed + 7919 * (k + 1))
        subs[name] = [rs.randint(4, v, size=int(u)).tolist()
                      for u in rs.randint(u_range[0], u_range[1] + 1, size=B)] if v > 0 else []
    return {'xs': xs, 'xlens': [int(t) for t in tl], 'ys': ys, 'ys_sub1': subs['ys_sub1'], 'ys_sub2': subs['ys_sub2'],
            'utt_ids': ['utt%d' % i for i in range(B)], 'speakers': ['spk'] * B,
            'sessions': ['sess'] * B, 'text': [''] * B, 'feat_path': [''] * B,
            'trigger_points': None}
