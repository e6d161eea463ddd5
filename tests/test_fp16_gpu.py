"""The fp16 storage mode (precision "fp16" = torch_dtype=torch.float16, MF_F16): the default precision of the reference's
inference script (examples/brushnet/test_brushnet.py:122-126) on the bf16 kernels' byte layout with the f16 MFMA forms.

Operator level here (every tile form that is instantiated for it, the fused epilogues, the LayerNorm fold and the transposed-V
projection, norms, GEGLU, the flash kernel); model / pipeline level in test_models_gpu.py, test_layers_gpu.py and
test_pipeline_gpu.py under the "fp16" parametrisation, against the REFERENCE's own fp16 deviation
(tests/golden/fp16_envelope.json, tools/make_bf16_envelope.py --dtype fp16).  Inputs are rounded to fp16 first, so the
float64-accumulated reference sees the operands the kernels see: what is left is the output rounding (2^-11) and fp32
accumulation order."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from reflecting_reality_amd import hip, ops  # noqa: E402

DEV = "cuda"
H = torch.float16
ATOL, RTOL = 2e-3, 2e-3


def rh(t):  # round through fp16
    return t.half().float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV, H)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def check(name, got, ref, atol=ATOL, rtol=RTOL):
    got = got.float().cpu()
    err = (got - ref).abs()
    bad = (err > atol + rtol * ref.abs()).sum().item()
    print(f"{name}: max_abs_err={err.max().item():.3e} ref_max={ref.abs().max().item():.3e} bad={bad}/{ref.numel()}")
    assert bad == 0, f"{name}: {bad} elements out of tolerance, max err {err.max().item():.3e}"


def prec():
    return ops.Precision.get("fp16")


def test_precision_surface():
    p = ops.Precision.get(torch.float16)
    assert p.name == "fp16" and p.compute == p.act == torch.float16 and p.code == hip.MF_F16 and p.vec == 8 and p.half and not p.split
    assert ops.Precision.get("fp16").code == hip.dt_code(torch.float16)


def test_fp32_activations_are_refused_by_the_fp16_kernels():
    """the fp16 instantiations have no fp32-converting load (only the bf16 ones do), so an fp32 activation with an
    fp16 weight must be refused by the C ABI — it used to be read as fp16 pairs and give garbage."""
    g = torch.Generator().manual_seed(3)
    w = rh(torch.randn(64, 32, 3, 3, generator=g) * 0.05)
    cw = ops.ConvWeight(w, torch.zeros(64), prec(), DEV)
    x32 = torch.randn(1, 8, 8, 32, generator=g).to(DEV)
    with pytest.raises(hip.MfhipError, match="fp16 activations"):
        ops.conv2d(x32, cw)
    lw = ops.ConvWeight(rh(torch.randn(64, 32, generator=g)).view(64, 32, 1, 1), torch.zeros(64), prec(), DEV)
    with pytest.raises(hip.MfhipError, match="fp16 activations"):
        ops.linear(torch.randn(16, 32, generator=g).to(DEV), lw)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 13, 14, 15, 25, 26, 29, 30, 31, 34, 36, 67, 41, 42, 43, 44, 45, 46, 48, 50, 52, 54, 56, 57, 58, 60, 62, 64, 66])
def test_conv3x3_tiles(tile):
    g = torch.Generator().manual_seed(1)
    x = rh(torch.randn(2, 40, 20, 12, generator=g))       # M = 480 (tails on every tile), Cin = 40
    w = rh(torch.randn(72, 40, 3, 3, generator=g) * 0.05)
    b = torch.randn(72, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    y = ops.conv2d(nhwc(x), ops.ConvWeight(w, b, prec(), DEV), tile=tile)
    assert y.dtype == H
    check(f"conv3x3[fp16,tile{tile}]", nchw(y), ref)


@pytest.mark.parametrize("tile", [20, 21, 22, 23, 24, 27, 28, 68, 37, 38, 39, 40, 47, 49, 51, 53, 55, 59, 61, 63, 65])
@pytest.mark.parametrize("case", ["plain", "cat", "epilogue", "big"])
def test_conv3x3_dx_reuse_tiles(tile, case):
    g = torch.Generator().manual_seed(11)
    b, h, w_, c0, c1, n = {"plain": (2, 16, 32, 64, 0, 160), "cat": (2, 16, 16, 64, 64, 128), "epilogue": (2, 16, 32, 128, 0, 320),
                           "big": (2, 64, 64, 320, 0, 320)}[case]
    x = rh(torch.randn(b, c0, h, w_, generator=g))
    x1 = rh(torch.randn(b, c1, h, w_, generator=g)) if c1 else None
    w = rh(torch.randn(n, c0 + c1, 3, 3, generator=g) * 0.03)
    bias = torch.randn(n, generator=g)
    ref = F.conv2d((torch.cat([x, x1], 1) if c1 else x).double(), w.double(), bias.double(), padding=1).float()
    kw = {}
    if case == "epilogue":
        temb = torch.randn(b, n, generator=g)
        r0 = rh(torch.randn(b, n, h, w_, generator=g))
        ref = F.silu(0.5 * (ref + temb[:, :, None, None]) + r0)
        kw = dict(temb=temb.to(DEV), res0=nhwc(r0), alpha=0.5, act=hip.ACT_SILU)
    y = ops.conv2d(nhwc(x), ops.ConvWeight(w, bias, prec(), DEV), x1=nhwc(x1) if c1 else None, tile=tile, splitk=1, **kw)
    check(f"conv3x3_dxr[fp16,tile{tile},{case}]", nchw(y), ref, 4e-3, 2e-3)


@pytest.mark.parametrize("tile", [41, 44, 48, 50, 52, 54, 57, 62, 14, 3])
@pytest.mark.parametrize("case", ["up", "s2", "1x1res", "splitk"])
def test_conv_variants(tile, case):
    g = torch.Generator().manual_seed(77 + tile)
    x = rh(torch.randn(2, 64, 16, 16, generator=g))
    r0 = None
    if case == "up":
        w = rh(torch.randn(96, 64, 3, 3, generator=g) * 0.05)
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest").double(), w.double(), padding=1).float()
        kw = dict(upsample=True, splitk=1)
    elif case == "s2":
        w = rh(torch.randn(96, 64, 3, 3, generator=g) * 0.05)
        ref = F.conv2d(x.double(), w.double(), stride=2, padding=1).float()
        kw = dict(stride=2, padding=1, splitk=2)
    elif case == "splitk":
        w = rh(torch.randn(96, 64, 3, 3, generator=g) * 0.05)
        ref = F.conv2d(x.double(), w.double(), padding=1).float()
        kw = dict(splitk=3)
    else:
        w = rh(torch.randn(96, 64, 1, 1, generator=g) * 0.1)
        r0 = rh(torch.randn(2, 96, 16, 16, generator=g))
        ref = F.silu(F.conv2d(x.double(), w.double()).float() + r0)
        kw = dict(padding=0, res0=nhwc(r0), act=hip.ACT_SILU, splitk=1)
    y = ops.conv2d(nhwc(x), ops.ConvWeight(w, None, prec(), DEV), tile=tile, **kw)
    check(f"conv[fp16,tile{tile},{case}]", nchw(y), ref, 4e-3, 2e-3)


@pytest.mark.parametrize("tile", [0, 41, 43, 44, 46, 48, 52, 54, 58, 60])
@pytest.mark.parametrize("mode", ["linear", "geglu", "qkv"])
def test_linear_with_folded_layernorm_and_transposed_v(tile, mode):
    g = torch.Generator().manual_seed(5 + tile)
    rows, c = 4 * 64, 320
    x = rh(torch.randn(rows, c, generator=g) * 2.0 + 0.3)
    gamma, beta = torch.randn(c, generator=g) * 0.2 + 1.0, torch.randn(c, generator=g) * 0.1
    xn = F.layer_norm(x.double(), (c,), gamma.double(), beta.double(), 1e-5)
    if mode == "linear":
        w, b = torch.randn(c, c, generator=g) * 0.05, torch.randn(c, generator=g)
        res = rh(torch.randn(rows, c, generator=g))
        y = ops.linear(x.to(DEV, H), ops.ConvWeight(w, b, prec(), DEV, ln=(gamma, beta, 1e-5)), res0=res.to(DEV, H), tile=tile)
        ref = (F.linear(xn, w.double(), b.double()) + res.double()).float()
        check(f"linear_ln[fp16,tile{tile}]", y, ref, 2e-2, 1e-2)
    elif mode == "geglu":
        w, b = torch.randn(8 * c, c, generator=g) * 0.05, torch.randn(8 * c, generator=g)
        y = ops.linear_geglu(x.to(DEV, H), ops.geglu_weight(w, b, prec(), DEV, ln=(gamma, beta, 1e-5)), tile=tile)
        hh = F.linear(xn, w.double(), b.double())
        ref = (hh[:, :4 * c] * F.gelu(hh[:, 4 * c:])).float()
        check(f"geglu_ln[fp16,tile{tile}]", y, ref, 3e-2, 1e-2)
    else:
        w = torch.randn(3 * c, c, generator=g) * 0.05
        qk, vt = ops.linear_qkv(x.view(4, 64, c).to(DEV, H), ops.ConvWeight(w, None, prec(), DEV, ln=(gamma, beta, 1e-5)), tile=tile)
        ref = F.linear(xn, w.double()).float().view(4, 64, 3 * c)
        assert qk.dtype == H and vt.dtype == H
        check(f"qkv.qk[fp16,tile{tile}]", qk, ref[..., :2 * c], 2e-2, 1e-2)
        check(f"qkv.vt[fp16,tile{tile}]", vt.transpose(1, 2), ref[..., 2 * c:], 2e-2, 1e-2)


@pytest.mark.parametrize("c0,c1,hw,silu", [(320, 0, 64 * 64, True), (1280, 640, 16 * 16, True), (40, 0, 30, False)])
def test_groupnorm(c0, c1, hw, silu):
    g = torch.Generator().manual_seed(3)
    x0 = rh(torch.randn(2, hw, c0, generator=g) * 2.0 + 0.5)
    x1 = rh(torch.randn(2, hw, c1, generator=g)) if c1 else None
    gamma, beta = torch.randn(c0 + c1, generator=g), torch.randn(c0 + c1, generator=g)
    groups = 32 if (c0 + c1) % 32 == 0 else 8
    y = hip.groupnorm(x0.to(DEV, H), gamma.to(DEV), beta.to(DEV), groups=groups, eps=1e-5, silu=silu, out_dtype=H,
                      x1=x1.to(DEV, H) if c1 else None)
    xc = torch.cat([x0, x1], -1) if c1 else x0
    ref = F.group_norm(xc.double().permute(0, 2, 1), groups, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1)
    ref = (F.silu(ref) if silu else ref).float()
    check(f"groupnorm[fp16,{c0}+{c1}]", y, ref, 4e-3, 2e-3)


def test_layernorm_and_geglu_kernels():
    g = torch.Generator().manual_seed(4)
    x = rh(torch.randn(300, 320, generator=g) * 3.0)
    gamma, beta = torch.randn(320, generator=g), torch.randn(320, generator=g)
    y = hip.layernorm(x.to(DEV, H), gamma.to(DEV), beta.to(DEV), 1e-5, H)
    check("layernorm[fp16]", y, F.layer_norm(x.double(), (320,), gamma.double(), beta.double(), 1e-5).float(), 4e-3, 2e-3)
    hsrc = rh(torch.randn(64, 2 * 160, generator=g))
    y = hip.geglu(hsrc.to(DEV, H), H)
    check("geglu[fp16]", y, (hsrc[:, :160].double() * F.gelu(hsrc[:, 160:].double())).float(), 4e-3, 2e-3)


@pytest.mark.parametrize("heads,d,sq,skv", [(8, 40, 1024, 1024), (8, 40, 4096, 77), (4, 80, 512, 512), (2, 160, 256, 256), (2, 8, 64, 77),
                                            (1, 64, 200, 200)])
def test_flash_attention_fp16(heads, d, sq, skv):
    """mf_attention_f16 (the flash kernel on the f16 MFMA forms: Q~ scaling, P, the exponent offset's pieces and the ones in fp16)
    against softmax(q k^T / sqrt(d)) v in float64 on the fp16-rounded operands — including a spiked key row that forces the
    deferred-max rescale branch (cdna_hip_programming.md rule 26)."""
    g = torch.Generator().manual_seed(9)
    c = heads * d
    b = 2
    q, k, v = (rh(torch.randn(b, s, c, generator=g)) for s in (sq, skv, skv))
    k[:, skv // 2, :] *= 6.0                      # one key dominates mid-stream: the running offset has to move
    ldv = (skv + 7) // 8 * 8
    vt = torch.zeros(b, c, ldv)
    vt[:, :, :skv] = v.transpose(1, 2)
    out = ops.attention(q.to(DEV, H), k.to(DEV, H), vt.to(DEV, H), heads, skv, d ** -0.5, prec())
    assert out.dtype == H
    qh, kh, vh = (t.double().view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).transpose(1, 2).reshape(b, sq, c).float()
    # P and the output are rounded to fp16 (2^-11 each), P <= 2^5 under the deferred maximum: a few 1e-3 on |out| <= 4 (bf16: 2e-2)
    check(f"attention[fp16,{heads}x{d},{sq}x{skv}]", out, ref, 6e-3, 4e-3)


def test_from_pretrained_accepts_the_reference_default_dtype(tmp_path):
    """test_brushnet.py:122-145: BrushNetModel.from_pretrained(path, torch_dtype=torch.float16)."""
    from oracle import mirrorfusion_ref as R
    from reflecting_reality_amd import models as M, synth
    from util import keys
    bn = M.BrushNetModel(dict(R.brushnet_config(R.TINY_UNET, 6)), precision="fp32", device=DEV)
    bn.load_state_dict(synth.state_dict_for(keys("tiny")["brushnet"], 1))
    bn.save_pretrained(str(tmp_path / "brushnet"))
    b16 = M.BrushNetModel.from_pretrained(str(tmp_path / "brushnet"), torch_dtype=torch.float16)
    assert b16.prec.name == "fp16"
