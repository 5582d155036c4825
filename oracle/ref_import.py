"""Import the read-only reference (hirofumi0810/neural_sp @ /root/reference) on CPU.

Only usable in the build container: /root/reference does not exist on the GPU box,
so this module is imported ONLY by oracle/gen_golden.py and by CPU tests that skip
when the reference is absent.  One missing third-party module (omegaconf, used by
neural_sp/bin/train_utils.py:69-86 for YAML I/O only) is stubbed.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('NEURAL_SP_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'neural_sp'))


def import_reference():
    """Return the reference `neural_sp` package (stubs omegaconf if needed)."""
    if not available():
        raise ImportError('reference not present at %s' % REFERENCE_ROOT)
    if 'omegaconf' not in sys.modules:
        try:
            import omegaconf  # noqa: F401
        except ImportError:
            m = types.ModuleType('omegaconf')

            class OmegaConf(object):
                @staticmethod
                def load(*a, **k):
                    raise RuntimeError('omegaconf stub')

                @staticmethod
                def save(*a, **k):
                    raise RuntimeError('omegaconf stub')

            m.OmegaConf = OmegaConf
            m.DictConfig = dict
            sys.modules['omegaconf'] = m
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import neural_sp  # noqa: F401
    return sys.modules['neural_sp']
