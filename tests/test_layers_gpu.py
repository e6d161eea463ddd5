"""F6: single layers at the production sizes of the SD1.5 path on the HIP kernels, against the outputs of the
reference's own modules (tests/golden/sd15_layers.npz, recorded by tools/make_golden.py::layers_full)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from reflecting_reality_amd import models as M, ops  # noqa: E402
from layer_cases import cases  # noqa: E402
from util import golden, report, strided_sample  # noqa: E402

DEV = "cuda"
# fp32 mode: measured <= 3e-6 against the reference (bar: 1e-3); bf16 mode: bf16 operands / activations, fp32 accumulation
# f16x3 mode (three fp16 MFMAs per product, 22-bit operands): the same bound as fp32
TOL = {"fp32": dict(atol=5e-5, rtol=5e-5), "f16x3": dict(atol=5e-5, rtol=5e-5), "bf16": dict(atol=5e-2, rtol=5e-2), "fp16": dict(atol=8e-3, rtol=8e-3)}


def bare_model(prec):
    m = M.UNet2DConditionModel.__new__(M.UNet2DConditionModel)
    m.prec, m.device = ops.Precision.get(prec), torch.device(DEV)
    m.config = {"norm_num_groups": 32, "norm_eps": 1e-5}
    m.P, m.tdepth, m._cross_kv, m._ehs_gen = {}, {}, {}, 0
    return m


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def compare(name, y, prec):
    G = golden("sd15_layers.npz")
    st = G[name + "_stats"]
    report(f"{name}[{prec}]", strided_sample(y, st[2]), G[name + "_sample"], **TOL[prec])
    if prec not in ("bf16", "fp16"):       # the whole tensor, through its sum: |sum error| <= 2e-5 of sum |y|
        assert abs(float(y.double().sum()) - st[0]) <= 2e-5 * st[1]


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16", "fp16"])
@pytest.mark.parametrize("name", ["resnet_320_64", "resnet_2560_1280_16"])
def test_resnet_block_full_size(name, prec):
    sd, x, temb = cases()[name]
    m = bare_model(prec)
    m._prepare_resnet(sd, "")
    f32 = ops.Precision.get("fp32")
    tw = ops.ConvWeight(sd["time_emb_proj.weight"], sd["time_emb_proj.bias"], f32, DEV)
    t = ops.linear(F.silu(temb).to(DEV), tw, out_dtype=torch.float32)           # resnet.py:369-376
    m.temb_slices = {"": (0, t.shape[1])}
    y = m._resnet("", nhwc(x, m.prec.act), t)
    compare(name, nchw(y), prec)


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16", "fp16"])
def test_transformer_2d_full_size(prec):
    sd, x, ehs = cases()["transformer_320_4096"]
    m = bare_model(prec)
    m._prepare_transformer(sd, "")
    y = m._transformer("", nhwc(x, m.prec.act), ehs.to(DEV, m.prec.act), 8)   # what _bind_prompt hands the blocks
    compare("transformer_320_4096", nchw(y), prec)


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16", "fp16"])
def test_self_attention_full_size(prec):
    sd, tok, _ = cases()["attention_4096_40"]
    m = bare_model(prec)
    for l in ("to_v", "to_out.0"):
        m.P[l] = ops.ConvWeight(sd[l + ".weight"], sd.get(l + ".bias"), m.prec, DEV, raw=(l == "to_v"))
    m.P["to_qk"] = ops.ConvWeight(torch.cat([sd["to_q.weight"], sd["to_k.weight"]], 0), None, m.prec, DEV)
    y = m._attention("", tok.to(DEV, m.prec.act), None, 8, None)
    compare("attention_4096_40", y.float().cpu(), prec)
