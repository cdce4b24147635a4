"""Host ingest transforms of the certification driver (reference transforms/__init__.py: `from transforms import *`).
Only the two that `certified_robustness_eval.py:66` composes are provided; the augmentation / STFT transforms of the
training code are outside the hot-path scope."""
from .transforms_wav import *  # noqa: F401,F403
