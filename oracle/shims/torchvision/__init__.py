"""Minimal stub of torchvision for `dust3r/utils/image.py:12,23` (ToTensor + Normalize only)."""
from . import transforms  # noqa: F401
