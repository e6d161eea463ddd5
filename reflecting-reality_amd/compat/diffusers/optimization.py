"""`from diffusers.optimization import get_scheduler` (train_brushnet_mirror.py:44, :1257) on the MI355X path."""
from reflecting_reality_amd.optimization import *  # noqa: F401,F403
from reflecting_reality_amd.optimization import get_scheduler  # noqa: F401
