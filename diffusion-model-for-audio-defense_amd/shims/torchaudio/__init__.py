"""Minimal stand-in for the `torchaudio` import of the reference's certification driver
(certified_robustness_eval.py:8,85-86): only `torchaudio.transforms.{MelSpectrogram, AmplitudeToDB}` with
the arguments that driver passes, backed by the MI355X engine."""
from . import transforms  # noqa: F401

__version__ = '0.11.0+dmad-shim'
