"""neural_sp_amd -- MI355X-native Speech2Text training hot path."""
__version__ = '0.1.0'


def install():
    """The ONE line a reference maintainer adds (top of neural_sp/bin/asr/train.py, before its imports of the model and of
    DistributedDataParallel -- i.e. outside the step loop): afterwards the unmodified script builds this package's
    Speech2Text (train.py:46,138) and its `DDP(model, device_ids=device_ids)` (train.py:20,263) carries the multi-stream
    communication hook.  Idempotent.  Returns the names it replaced.

    Without it nothing breaks: Speech2Text notices a process group without the hook and keeps the whole training step on
    one stream (Speech2Text._ddp_guard)."""
    import sys
    import torch.nn.parallel as tnp
    from . import parallel
    from .speech2text import Speech2Text
    done = []
    ddp = parallel.make_ddp_class()
    if tnp.DistributedDataParallel is not ddp:
        tnp.DistributedDataParallel = ddp
        tnp.distributed.DistributedDataParallel = ddp
        done.append('torch.nn.parallel.DistributedDataParallel')
    try:
        import neural_sp.models.seq2seq.speech2text as ref_mod      # the reference, when it is installed
        if ref_mod.Speech2Text is not Speech2Text:
            ref_mod.Speech2Text = Speech2Text
            done.append('neural_sp.models.seq2seq.speech2text.Speech2Text')
    except ImportError:
        pass
    # a script that had already imported the names (install() called late): rebind them in its namespace
    main = sys.modules.get('__main__')
    if main is not None:
        if getattr(main, 'DDP', None) is not None and getattr(main.DDP, '__name__', '') == 'DistributedDataParallel' \
                and main.DDP is not ddp:
            main.DDP = ddp
            done.append('__main__.DDP')
        ref_cls = getattr(main, 'Speech2Text', None)
        if ref_cls is not None and ref_cls is not Speech2Text and getattr(ref_cls, '__module__', '').startswith('neural_sp.'):
            main.Speech2Text = Speech2Text
            done.append('__main__.Speech2Text')
    return done
