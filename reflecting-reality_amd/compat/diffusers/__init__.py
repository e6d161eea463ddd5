"""Drop-in alias: put `<repo>/reflecting-reality_amd/compat` (and `<repo>`) on PYTHONPATH and the reference's
scripts' `from diffusers import BrushNetModel, UNet2DConditionModel, AutoencoderKL, DDIMScheduler, PNDMScheduler,
UniPCMultistepScheduler, StableDiffusionBrushNetPipeline` (examples/brushnet/test_brushnet.py:13, train_brushnet_mirror.py:35-42) resolve to
the MI355X implementations.  Only the names of the accelerated hot path exist here; anything else raises."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from reflecting_reality_amd.models import AutoencoderKL, BrushNetModel, UNet2DConditionModel  # noqa: E402,F401
from reflecting_reality_amd.pipeline import StableDiffusionBrushNetPipeline, StableDiffusionXLBrushNetPipeline  # noqa: E402,F401
from reflecting_reality_amd.schedulers import DDIMScheduler, DDPMScheduler, PNDMScheduler, UniPCMultistepScheduler  # noqa: E402,F401
from reflecting_reality_amd.attn_processor import MfhipAttnProcessor  # noqa: E402,F401

__version__ = "0.27.0.dev0+mi355x"


def __getattr__(name):
    raise AttributeError(f"diffusers.{name} is outside the MI355X hot path (see SURVEY.md §8 / DESIGN.md 'Out of scope')")
