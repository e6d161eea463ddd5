"""Inputs of the full-size single-layer fixtures (tests/golden/sd15_layers.npz, tools/make_golden.py::layers_full):
one generator, draws in the generator script's order; weights from synth.state_dict_for(shapes, seed)."""
import torch

from reflecting_reality_amd import synth
from util import keys

SEEDS = {"resnet_320_64": 61, "resnet_2560_1280_16": 62, "transformer_320_4096": 63, "attention_4096_40": 64}


def cases():
    shapes = keys("sd15_layers")
    g = torch.Generator().manual_seed(606)
    out = {}
    for name, cin, hw in (("resnet_320_64", 320, 64), ("resnet_2560_1280_16", 2560, 16)):
        x = torch.randn(1, cin, hw, hw, generator=g)
        temb = torch.randn(1, 1280, generator=g)
        out[name] = (synth.state_dict_for(shapes[name], SEEDS[name]), x, temb)
    x = torch.randn(1, 320, 64, 64, generator=g)
    ehs = torch.randn(1, 77, 768, generator=g)
    out["transformer_320_4096"] = (synth.state_dict_for(shapes["transformer_320_4096"], 63), x, ehs)
    tok = torch.randn(1, 4096, 320, generator=g)
    out["attention_4096_40"] = (synth.state_dict_for(shapes["attention_4096_40"], 64), tok, None)
    return out
