"""The split precision codes of mf_gemm_conv / mf_attention_f16x3 (fp32 operands, every value split into two 16-bit
halves, three MFMAs per product) against float64 references: this is the mode that must meet the reference's fp32
results to 1e-3 on the latents while running on the 16-bit matrix pipe, so it is held to fp32-class error bounds."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from reflecting_reality_amd import hip, ops  # noqa: E402

DEV = "cuda"
# max |err| relative to the scale of the result (the rms of the float64 reference): 22-bit operands (f16x3) sit within a
# small factor of fp32 MFMA, 16-bit operands (bf16x3) ~64x above
REL = {"fp32": 2e-6, "f16x3": 1e-5, "bf16x3": 3e-4}


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV, torch.float32)


def relerr(got, ref64):
    ref64 = ref64.double()
    return float((got.double().cpu() - ref64).abs().max() / ref64.pow(2).mean().sqrt())


@pytest.mark.parametrize("prec_name", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("m,k,n", [(300, 36, 72), (2048, 320, 320), (154, 768, 320), (64, 1280, 8), (4096, 1280, 320)])
def test_split_linear_all_tiles(prec_name, m, k, n):
    """Every tile instantiated for the split codes, pre-split weight (w_split=1) and raw fp32 weight (w_split=0):
    ragged M / N, K not a multiple of the 32-wide block (zero-padded packed rows)."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    ref = F.linear(x.double(), w.double(), b.double())
    xd = x.to(DEV)
    raw_tiles = (0, 1, 2, 3, 6, 14) if prec_name == "f16x3" else (0, 1, 2, 3, 6)     # f16x3 keeps the big-conv tile for raw weights (training)
    ws_tiles = (41, 44) if prec_name == "f16x3" else ()      # round 3: the warp-specialised ring tiles serve the parity mode too
    for raw, tiles in ((False, (0, 1, 2, 3, 6, 7, 14) + ws_tiles), (True, raw_tiles)):
        lw = ops.ConvWeight(w, b, prec, DEV, raw=raw)
        assert lw.w_split == (0 if raw else 1)
        for tile in tiles:
            y = ops.linear(xd, lw, tile=tile, splitk=1)
            e = relerr(y, ref)
            print(f"split linear[{prec_name}, w_split={lw.w_split}, tile {tile}, {m}x{k}x{n}]: rel err {e:.2e}")
            assert e < REL[prec_name]
    with pytest.raises(hip.MfhipError, match="not instantiated"):
        ops.linear(xd, ops.ConvWeight(w, b, prec, DEV, raw=True), tile=7 if prec_name == "f16x3" else 14)


def test_split_modes_rank_between_bf16_and_fp32():
    """The point of the mode: on the same product f16x3 is within a small factor of the exact fp32 MFMA and orders of
    magnitude tighter than bf16.  Its precision is 22 bits RELATIVE for |x| >= 2^-3 and 2^-25 ABSOLUTE below that (the low
    half becomes an fp16 subnormal — kept by v_cvt_pkrtz_f16_f32 and by the matrix pipe, tools/micro/f16_denorm.hip, but
    with the subnormal spacing 2^-24): operands scaled down to ~1e-3 lose relative precision, gracefully (no flush to the
    11 bits of the high half alone, which would be 5e-4), while bf16x3 (fp32 exponent range) does not care."""
    g = torch.Generator().manual_seed(32)
    x = torch.randn(1024, 640, generator=g)
    w = torch.randn(640, 640, generator=g) / 25.0
    errs = {}
    for scale in (1.0, 1e-3):            # 1e-3: every low half of x * scale is an fp16 subnormal
        ref = F.linear((x * scale).double(), w.double())
        for name in ("fp32", "f16x3", "bf16x3", "bf16"):
            prec = ops.Precision.get(name)
            y = ops.linear((x * scale).to(DEV, prec.act), ops.ConvWeight(w, None, prec, DEV), out_dtype=torch.float32)
            errs[(name, scale)] = relerr(y, ref)
        print({k: f"{v:.2e}" for k, v in errs.items() if k[1] == scale})
        if scale == 1.0:
            assert errs[("f16x3", scale)] < 8 * max(errs[("fp32", scale)], 2e-7)
            assert errs[("f16x3", scale)] < errs[("bf16x3", scale)]
        else:
            assert errs[("f16x3", scale)] < 3e-4            # 2^-25 / 1e-3 per operand, not the high half's 2^-11
        assert errs[("bf16x3", scale)] < errs[("bf16", scale)]
        assert errs[("bf16x3", scale)] < 1e-4 and errs[("bf16", scale)] > 1e-3


@pytest.mark.parametrize("prec_name", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("case", ["3x3", "s2p1", "s2asym", "up", "cat", "cat_up", "1x1", "splitk", "epilogue"])
def test_split_conv_variants(prec_name, case):
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(33)
    cin, cout = 24, 40
    x = torch.randn(2, cin, 10, 14, generator=g)
    x1 = torch.randn(2, 16, 10, 14, generator=g)
    if case in ("cat", "cat_up"):
        w = torch.randn(cout, cin + 16, 3, 3, generator=g) * 0.05
    elif case == "1x1":
        w = torch.randn(cout, cin, 1, 1, generator=g) * 0.2
    elif case == "splitk":
        cin = 256
        x = torch.randn(2, cin, 10, 14, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.02
    else:
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    cw = ops.ConvWeight(w, b, prec, DEV)
    xa, xd, wd, bd = nhwc(x), x.double(), w.double(), b.double()
    if case == "s2p1":
        ref, y = F.conv2d(xd, wd, bd, stride=2, padding=1), ops.conv2d(xa, cw, stride=2, padding=1)
    elif case == "s2asym":
        ref, y = F.conv2d(F.pad(xd, (0, 1, 0, 1)), wd, bd, stride=2), ops.conv2d(xa, cw, stride=2, padding=(0, 0, 1, 1))
    elif case == "up":
        ref = F.conv2d(F.interpolate(xd, scale_factor=2.0, mode="nearest"), wd, bd, padding=1)
        y = ops.conv2d(xa, cw, upsample=True)
    elif case == "cat":
        ref, y = F.conv2d(torch.cat([xd, x1.double()], 1), wd, bd, padding=1), ops.conv2d(xa, cw, x1=nhwc(x1))
    elif case == "cat_up":
        ref = F.conv2d(F.interpolate(torch.cat([xd, x1.double()], 1), scale_factor=2.0, mode="nearest"), wd, bd, padding=1)
        y = ops.conv2d(xa, cw, x1=nhwc(x1), upsample=True)
    elif case == "1x1":
        ref, y = F.conv2d(xd, wd, bd), ops.conv2d(xa, cw, padding=0)
    elif case == "splitk":
        ref, y = F.conv2d(xd, wd, bd, padding=1), ops.conv2d(xa, cw, splitk=5)
    elif case == "epilogue":
        temb = torch.randn(2, cout, generator=g)
        r0 = torch.randn(2, cout, 10, 14, generator=g)
        ref = F.silu(0.5 * (F.conv2d(xd, wd, bd, padding=1) + temb.double()[:, :, None, None]) + r0.double())
        y = ops.conv2d(xa, cw, temb=temb.to(DEV), res0=nhwc(r0), alpha=0.5, act=hip.ACT_SILU)
    else:
        ref, y = F.conv2d(xd, wd, bd, padding=1), ops.conv2d(xa, cw)
    e = relerr(y.permute(0, 3, 1, 2), ref)
    print(f"split conv {case}[{prec_name}]: rel err {e:.2e}")
    assert e < REL[prec_name]


@pytest.mark.parametrize("prec_name", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("tile", [20, 21, 22, 37, 38])
@pytest.mark.parametrize("case", ["ragged", "cat_big", "splitk"])
def test_split_conv3x3_dx_reuse_tiles(prec_name, tile, case):
    """dx-tap reuse tiles (one staged A window for the three kx taps) in split precision; 37 / 38: their warp-specialised forms
    (f16x3 with a pre-split weight only: refused for bf16x3, never rerouted)."""
    prec = ops.Precision.get(prec_name)
    if tile >= 37 and prec_name != "f16x3":
        x0 = torch.zeros(1, 7, 16, 64, device=DEV)
        with pytest.raises(hip.MfhipError, match="not instantiated|does not apply"):
            ops.conv2d(x0, ops.ConvWeight(torch.zeros(40, 64, 3, 3), None, prec, DEV), tile=tile)
        return
    g = torch.Generator().manual_seed(34)
    b, h, w_, c0, c1, n = {"ragged": (3, 7, 16, 64, 0, 40), "cat_big": (2, 32, 64, 64, 32, 160),
                           "splitk": (1, 16, 16, 256, 0, 200)}[case]
    x = torch.randn(b, c0, h, w_, generator=g)
    x1 = torch.randn(b, c1, h, w_, generator=g) if c1 else None
    w = torch.randn(n, c0 + c1, 3, 3, generator=g) * 0.03
    bias = torch.randn(n, generator=g)
    xin = torch.cat([x, x1], 1) if c1 else x
    ref = F.conv2d(xin.double(), w.double(), bias.double(), padding=1)
    y = ops.conv2d(nhwc(x), ops.ConvWeight(w, bias, prec, DEV), x1=nhwc(x1) if c1 else None, tile=tile,
                   splitk=3 if case == "splitk" else 1)
    e = relerr(y.permute(0, 3, 1, 2), ref)
    print(f"split dxr[{prec_name}, tile {tile}, {case}]: rel err {e:.2e}")
    assert e < REL[prec_name]


def sdpa64(q, k, v, heads):
    b, sq, c = q.shape
    d = c // heads
    qh, kh, vh = (t.double().view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (p @ vh).transpose(1, 2).reshape(b, sq, c)


@pytest.mark.parametrize("heads,d,sq,skv", [(8, 40, 4096, 4096), (8, 80, 1024, 1024), (8, 40, 4096, 77), (8, 80, 1024, 77),
                                            (4, 8, 200, 77), (3, 64, 130, 190), (8, 160, 256, 256), (8, 160, 64, 77)])
def test_attention_split_flash(heads, d, sq, skv):
    """The flash kernel in split precision (mf_attention_f16x3; d = 160 takes the unfused split GEMMs) at the SD1.5
    production sizes against a float64 softmax attention: this is the evidence that the <= 1e-3 parity mode runs the
    shipped attention kernel."""
    prec = ops.Precision.get("f16x3")
    g = torch.Generator().manual_seed(35)
    c = heads * d
    q = torch.randn(2, sq, c, generator=g)
    k = torch.randn(2, skv, c, generator=g)
    v = torch.randn(2, skv, c, generator=g)
    ref = sdpa64(q, k, v, heads)
    ld = (skv + 7) // 8 * 8
    vt = torch.zeros(2, c, ld, device=DEV)
    vt[:, :, :skv] = v.transpose(1, 2).to(DEV)
    o = ops.attention(q.to(DEV), k.to(DEV), vt, heads, skv, 1.0 / math.sqrt(d), prec)
    assert o.dtype == torch.float32
    e = float((o.double().cpu() - ref).abs().max())
    print(f"split attention[h{heads}, d{d}, {sq}x{skv}]: max abs err {e:.2e} (|ref| max {float(ref.abs().max()):.2f})")
    assert e < 5e-6
    if d != 160:     # the fused q|k projection layout of self-attention: q and k are column slices of one tensor
        qk = torch.cat([q, k[:, :sq] if skv >= sq else q], -1).to(DEV) if skv >= sq else None
        if qk is not None and skv == sq:
            o2 = ops.attention(qk[..., :c], qk[..., c:], vt, heads, skv, 1.0 / math.sqrt(d), prec, c=c)
            assert torch.equal(o2, o)


def test_a_f32_tile_resolution():
    """fp32 activations with bf16 compute only exist for tiles 1..12: larger tiles are refused (they used to be launched
    on a 128x128 kernel with the bigger tile's grid, leaving rows unwritten) and the heuristic stays inside 1..6."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(36)
    x = torch.randn(8, 64, 64, 64, generator=g)                     # M = 32768
    w = torch.randn(320, 64, 1, 1, generator=g) * 0.1
    ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float())
    cw = ops.ConvWeight(w, None, prec, DEV)
    xa = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    old = hip.AUTOTUNE
    try:
        for auto in (False, True):
            hip.AUTOTUNE = auto
            y = ops.conv2d(xa, cw, padding=0, out_dtype=torch.float32)
            err = float((y.permute(0, 3, 1, 2).cpu() - ref).abs().max())
            assert err < 2e-3, (auto, err)
    finally:
        hip.AUTOTUNE = old
    for tile in (13, 14, 15, 20):
        with pytest.raises(hip.MfhipError, match="does not apply"):
            ops.conv2d(xa, cw, padding=0, out_dtype=torch.float32, tile=tile)
    for tile in (7, 8, 12):                                          # 3-stage twins of 1..6 run the 2-stage kernel of the same shape
        y = ops.conv2d(xa, cw, padding=0, out_dtype=torch.float32, tile=tile)
        assert float((y.permute(0, 3, 1, 2).cpu() - ref).abs().max()) < 2e-3


def test_f16x3_range_guard_flags_saturated_operands():
    """v_cvt_pkrtz_f16_f32 saturates above 65504 (no inf, no NaN): the kernels that split operands track max |operand| and
    raise sticky flags that mf_split_overflow reads.  GEMM operand, attention operand, weight gradient
    operand, host-split weight; and a clean run raises nothing."""
    prec = ops.Precision.get("f16x3")
    hip.split_overflow(reset=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, 64, generator=g).to("cuda")
    lw = ops.ConvWeight(torch.randn(64, 64, generator=g) / 8, None, prec, "cuda")
    y = ops.linear(x, lw)
    assert hip.split_overflow() == 0 and torch.isfinite(y).all()
    xb = x.clone()
    xb[17, 5] = 7.0e4                                   # above the fp16 range, small weights: the OUTPUT stays tiny and finite
    yb = ops.linear(xb, lw)
    assert torch.isfinite(yb).all()
    assert hip.split_overflow() & 1, "an fp16-saturated GEMM operand went unnoticed"
    assert hip.split_overflow() == 0, "flags are cleared by a reset read"
    hip.split_halves(xb.contiguous())
    assert hip.split_overflow() & 2
    with pytest.raises(hip.SplitRangeError):
        ops.ConvWeight(torch.full((8, 32), 1.0e5), None, prec, "cuda")
    # bf16x3 has fp32's range: the same operand is fine there
    lw3 = ops.ConvWeight(torch.randn(64, 64, generator=g) / 8, None, ops.Precision.get("bf16x3"), "cuda")
    ops.linear(xb, lw3)
    assert hip.split_overflow() == 0
    # weight-gradient operands
    dy = torch.randn(256, 64, generator=g).to("cuda")
    dw = torch.empty(64, 64, device="cuda")
    hip.conv_wgrad(xb, dy, dw, code=hip.MF_F16X3, c0=64, batch=256, h_in=1, w_in=1, h_out=1, w_out=1, n=64, accumulate=False)
    assert hip.split_overflow() & 4
