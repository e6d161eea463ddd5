"""The reference's own layer known-answer tests (tests/models/test_layers_utils.py:135-145,187-200,228-256): inputs and
weights are re-created from torch's default initialisers under the same seed and construction order as the reference
modules (ResnetBlock2D.__init__: conv1, time_emb_proj, conv2[, conv_shortcut]; Upsample2D / Downsample2D: conv), the
expected 3x3 output slices are copied from the reference tests."""
import torch
import torch.nn as nn

EXPECT = {
    "resnet_default": [-1.9010, -0.2974, -0.8245, -1.3533, 0.8742, -0.9645, -2.0584, 1.3387, -0.4746],
    "resnet_shortcut": [0.2226, -1.0791, -0.1629, 0.3659, -0.2889, -1.2376, 0.0582, 0.9206, 0.0044],
    "upsample_conv": [0.7145, 1.3773, 0.3492, 0.8448, 1.0839, -0.3341, 0.5956, 0.1250, -0.4841],
    "downsample_conv": [0.9267, 0.5878, 0.3337, 1.2321, -0.1191, -0.3984, -0.7532, -0.0715, -0.3913],
}


def resnet_case(shortcut: bool):
    torch.manual_seed(0)
    sample, temb = torch.randn(1, 32, 64, 64), torch.randn(1, 128)
    conv1, lin, conv2 = nn.Conv2d(32, 32, 3, padding=1), nn.Linear(128, 32), nn.Conv2d(32, 32, 3, padding=1)
    sd = {"norm1.weight": torch.ones(32), "norm1.bias": torch.zeros(32), "norm2.weight": torch.ones(32),
          "norm2.bias": torch.zeros(32), "conv1.weight": conv1.weight.data, "conv1.bias": conv1.bias.data,
          "time_emb_proj.weight": lin.weight.data, "time_emb_proj.bias": lin.bias.data, "conv2.weight": conv2.weight.data,
          "conv2.bias": conv2.bias.data}
    if shortcut:
        sc = nn.Conv2d(32, 32, 1)
        sd["conv_shortcut.weight"], sd["conv_shortcut.bias"] = sc.weight.data, sc.bias.data
    return sd, sample, temb


def sampler_case(kind: str):
    torch.manual_seed(0)
    sample = torch.randn(1, 32, 32, 32) if kind == "up" else torch.randn(1, 32, 64, 64)
    conv = nn.Conv2d(32, 32, 3, padding=1) if kind == "up" else nn.Conv2d(32, 32, 3, stride=2, padding=1)
    return {"conv.weight": conv.weight.data, "conv.bias": conv.bias.data}, sample


def check_slice(name, out):
    got = out[0, -1, -3:, -3:].flatten().float().cpu()
    exp = torch.tensor(EXPECT[name])
    assert torch.allclose(got, exp, atol=1e-3), f"{name}: {got.tolist()} vs reference {exp.tolist()}"

EXPECT["transformer_cross"] = [0.0143, -0.6909, -2.1547, -1.8893, 1.4097, 0.1359, -0.2521, -1.3359, 0.2598]


def transformer_case():
    """test_spatial_transformer_cross_attention_dim (test_layers_utils.py:343-364): Transformer2DModel(in_channels=64,
    heads 2 x 32, cross_attention_dim=64); construction order norm, proj_in, [norm1, attn1(q,k,v,out), norm2, attn2,
    norm3, ff(GEGLU proj, out)], proj_out; the context is drawn after the module is built."""
    torch.manual_seed(0)
    sample = torch.randn(1, 64, 64, 64)
    c = 64
    sd = {}

    def lin(name, i, o, bias=True):
        l = nn.Linear(i, o, bias=bias)
        sd[name + ".weight"] = l.weight.data
        if bias:
            sd[name + ".bias"] = l.bias.data

    def ln(name):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(c), torch.zeros(c)

    ln("norm")
    pin = nn.Conv2d(c, c, 1)
    sd["proj_in.weight"], sd["proj_in.bias"] = pin.weight.data, pin.bias.data
    b = "transformer_blocks.0."
    for norm, attn in (("norm1", "attn1"), ("norm2", "attn2")):
        ln(b + norm)
        lin(b + attn + ".to_q", c, c, False); lin(b + attn + ".to_k", c, c, False); lin(b + attn + ".to_v", c, c, False)
        lin(b + attn + ".to_out.0", c, c)
    ln(b + "norm3")
    lin(b + "ff.net.0.proj", c, 8 * c); lin(b + "ff.net.2", 4 * c, c)
    pout = nn.Conv2d(c, c, 1)
    sd["proj_out.weight"], sd["proj_out.bias"] = pout.weight.data, pout.bias.data
    context = torch.randn(1, 4, 64)
    return sd, sample, context

EXPECT["down_block"] = [-0.0232, -0.9869, 0.8054, -0.0637, -0.1688, -1.4264, 0.4470, -1.3394, 0.0904]
EXPECT["up_block"] = [-0.2041, -0.4165, -0.3022, 0.0041, -0.6628, -0.7053, 0.1928, -0.0325, 0.0523]


def _resnet_sd(sd, p, cin, cout):
    """ResnetBlock2D.__init__ order: conv1, time_emb_proj, conv2, conv_shortcut (if cin != cout)."""
    c1, lin, c2 = nn.Conv2d(cin, cout, 3, padding=1), nn.Linear(128, cout), nn.Conv2d(cout, cout, 3, padding=1)
    sd[p + "norm1.weight"], sd[p + "norm1.bias"] = torch.ones(cin), torch.zeros(cin)
    sd[p + "norm2.weight"], sd[p + "norm2.bias"] = torch.ones(cout), torch.zeros(cout)
    sd[p + "conv1.weight"], sd[p + "conv1.bias"] = c1.weight.data, c1.bias.data
    sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"] = lin.weight.data, lin.bias.data
    sd[p + "conv2.weight"], sd[p + "conv2.bias"] = c2.weight.data, c2.bias.data
    if cin != cout:
        sc = nn.Conv2d(cin, cout, 1)
        sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"] = sc.weight.data, sc.bias.data


def block_case(kind: str):
    """DownBlock2DTests / UpBlock2DTests (tests/models/unets/test_unet_2d_blocks.py:23-29,200-210 with the dummy inputs
    of test_unet_blocks_common.py:46-78): hidden [4,32,32,32] and temb [4,128] from seed 0; the up block's skip tensor
    re-seeds the global generator with 1 before the block is built."""
    torch.manual_seed(0)
    hidden, temb = torch.randn(4, 32, 32, 32), torch.randn(4, 128)
    sd = {}
    if kind == "down":
        _resnet_sd(sd, "resnets.0.", 32, 32)
        conv = nn.Conv2d(32, 32, 3, stride=2, padding=1)
        skip = None
    else:
        torch.manual_seed(1)
        skip = torch.randn(4, 32, 32, 32)
        _resnet_sd(sd, "resnets.0.", 64, 32)
        conv = nn.Conv2d(32, 32, 3, padding=1)
    sd["sampler.conv.weight"], sd["sampler.conv.bias"] = conv.weight.data, conv.bias.data
    return sd, hidden, temb, skip
