"""Minimal stand-in for `from torchvision.transforms import *` (certified_robustness_eval.py:7,66,87)."""
from . import transforms  # noqa: F401
