"""The reference's recipe configurations as the Namespace its own argument parser produces (TEST INFRASTRUCTURE).

`recipe_args(yaml_path, **overrides)` = what `neural_sp/bin/args_asr.py:parse_args_train` hands to
`Speech2Text(args, ...)` (bin/asr/train.py:138) for `--config <yaml>`: the parser defaults of args_common.py /
args_asr.py plus the `add_args` of the selected encoder / decoder classes, overlaid with the YAML file.
`configargparse` and `omegaconf` are absent from this container; both are used by that code only as an argparse
with a config-file option and as a YAML loader, and are stubbed as such.  Build container only (needs /root/reference).
"""
import argparse
import sys
import types

import yaml

from oracle.ref_import import import_reference

# set by train.py from the data set, not by the parser (train.py:84-131)
DATASET_FIELDS = dict(vocab=10000, vocab_sub1=-1, vocab_sub2=-1, input_dim=80)


def _install_configargparse_stub():
    if 'configargparse' in sys.modules:
        return
    m = types.ModuleType('configargparse')

    class ArgumentParser(argparse.ArgumentParser):
        def __init__(self, *a, config_file_parser_class=None, **k):
            super().__init__(*a, **k)

        def add_argument(self, *a, is_config_file=False, **k):
            if is_config_file:
                k.pop('required', None)
            return super().add_argument(*a, **k)

        add = add_argument

    m.ArgumentParser = ArgumentParser
    m.YAMLConfigFileParser = object
    m.ArgumentDefaultsHelpFormatter = argparse.ArgumentDefaultsHelpFormatter
    sys.modules['configargparse'] = m


class _Attr(dict):
    """the attribute access of an OmegaConf node, for `dec_config_sub1.dec_type`-style reads"""

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return self.get(k)


def recipe_args(yaml_path, **overrides):
    import_reference()
    _install_configargparse_stub()
    from neural_sp.bin import args_asr
    with open(yaml_path) as fh:
        config = yaml.safe_load(fh) or {}
    cli = []
    for k in ('enc_type', 'dec_type'):
        if k in config:
            cli += ['--' + k, str(config[k])]
    parser = args_asr.build_parser()
    for act in parser._actions:
        act.required = False
    user = parser.parse_known_args(cli)[0]
    parser = args_asr.register_args_encoder(parser, user, user.enc_type)
    user = parser.parse_known_args(cli)[0]
    parser = args_asr.register_args_decoder(parser, user, user.dec_type)
    sub1 = config.get('dec_config_sub1') or {}
    if sub1.get('dec_type') and sub1['dec_type'] != user.dec_type:
        user = parser.parse_known_args(cli)[0]
        try:
            parser = args_asr.register_args_decoder(parser, user, sub1['dec_type'])
        except argparse.ArgumentError:
            pass
    for act in parser._actions:
        act.required = False
    args = vars(parser.parse_known_args(cli)[0])
    args.update(config)                         # parse_args_train: YAML wins, parser fills what it lacks
    for k, v in DATASET_FIELDS.items():
        args.setdefault(k, v)
        if args[k] is None or args[k] is False:
            args[k] = v
    for k in ('dec_config_sub1', 'dec_config_sub2'):
        if isinstance(args.get(k), dict):
            args[k] = _Attr(args[k])
        elif args.get(k) is None:
            args.pop(k, None)                  # speech2text.py:174 tests hasattr(args, 'dec_config_sub*')
    for sub in ('sub1', 'sub2'):               # auxiliary-task vocabularies also come from the data set
        if float(args.get(sub + '_weight', 0) or 0) > 0 and int(args['vocab_' + sub]) <= 0:
            args['vocab_' + sub] = 300
    if int(args.get('conv_in_channel', 1) or 1) > 1 and args['input_dim'] % int(args['conv_in_channel']):
        args['input_dim'] = 41 * int(args['conv_in_channel'])   # TIMIT / WSJ: (40 + energy) x (static, delta, delta-delta)
    args.update(overrides)
    return argparse.Namespace(**args)
