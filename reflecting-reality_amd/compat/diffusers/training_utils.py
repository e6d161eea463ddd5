"""`from diffusers.training_utils import compute_snr` (train_brushnet_mirror.py:43) on the MI355X path."""
from reflecting_reality_amd.training import compute_snr  # noqa: F401
