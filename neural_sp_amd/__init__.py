"""neural_sp_amd -- MI355X-native Speech2Text training hot path."""
__version__ = '0.1.0'
