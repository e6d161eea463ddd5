"""Per-operator parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 CPU ops.

fp32 mode is held to fp32-accumulation-order tolerances; bf16 mode is compared with the same fp32
reference evaluated on bf16-rounded operands (what the kernel multiplies), so only accumulation order
and the final bf16 store differ.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from reflecting_reality_amd import hip, ops  # noqa: E402

DEV = "cuda"


def rb(t):  # round through bf16
    return t.bfloat16().float()


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def check(name, got, ref, atol, rtol):
    got = got.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol).sum().item()
    print(f"{name}: max_abs_err={err.max().item():.3e} ref_max={ref.abs().max().item():.3e} bad={bad}/{ref.numel()}")
    assert bad == 0, f"{name}: {bad} elements out of tolerance, max err {err.max().item():.3e}"


PRECS = [("fp32", 2e-4, 2e-4), ("f16x3", 2e-4, 2e-4), ("bf16", 2e-2, 1e-2)]
SPLIT_TILES = (0, 1, 2, 3, 6, 7, 14, 41, 44)  # tiles instantiated for f16x3 with a pre-split weight (41 / 44: warp-specialised rings)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
@pytest.mark.parametrize("tile", list(range(16)) + [25, 26, 29, 30, 31, 32, 33, 34, 35, 36, 67, 41, 42, 43, 44, 45, 46, 48, 50, 52, 54, 56, 57, 58, 60, 62, 64, 66])
def test_conv3x3_tiles(prec_name, atol, rtol, tile):
    prec = ops.Precision.get(prec_name)
    if (prec.split and tile not in SPLIT_TILES) or ((25 <= tile <= 30 or tile >= 37) and prec_name == "fp32"):
        x = torch.zeros(1, 8, 8, 32, device=DEV)
        with pytest.raises(hip.MfhipError, match="not instantiated|does not apply"):     # refused, never rerouted
            ops.conv2d(x, ops.ConvWeight(torch.zeros(8, 32, 3, 3), None, prec, DEV), tile=tile)
        return
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 40, 20, 12, generator=g)       # M = 480 (tails on every tile), Cin = 40
    w = torch.randn(72, 40, 3, 3, generator=g) * 0.05  # N = 72 (tail)
    b = torch.randn(72, generator=g)
    if prec_name == "bf16":
        x, w = rb(x), rb(w)
    ref = F.conv2d(x, w, b, padding=1)
    cw = ops.ConvWeight(w, b, prec, DEV)
    y = ops.conv2d(nhwc(x, prec.act), cw, tile=tile)
    check(f"conv3x3[{prec_name},tile{tile}]", nchw(y), ref, atol, rtol)


@pytest.mark.parametrize("tile", [16, 17, 18, 19, 20, 21, 22, 23, 24, 27, 28, 68, 37, 38, 39, 40, 47, 49, 51, 53, 55, 59, 61, 63, 65])
@pytest.mark.parametrize("case", ["plain", "tailN", "cat", "splitk", "epilogue", "big"])
def test_conv3x3_halo_tiles(tile, case):
    """conv3x3_halo_kernel (input patch resident in LDS, weights streamed per tap) and the 8-wave ping-pong
    256x160 variant (tile 19: 16x16 pixel tiles, 64-channel chunks) against F.conv2d."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(11)
    b, h, w_, c0, c1, n = {"plain": (2, 16, 32, 64, 0, 160), "tailN": (1, 8, 16, 32, 0, 200), "cat": (2, 16, 16, 64, 32, 128),
                           "splitk": (1, 16, 16, 256, 0, 160), "epilogue": (2, 8, 32, 96, 0, 320),
                           "big": (2, 64, 64, 320, 0, 320)}[case]
    if tile >= 19:      # 64-channel chunks (tile 19 also needs 16-row tiles)
        b, h, w_, c0, c1, n = {"plain": (2, 16, 32, 64, 0, 160), "tailN": (1, 16, 16, 64, 0, 200), "cat": (2, 16, 16, 64, 64, 128),
                               "splitk": (1, 16, 16, 256, 0, 160), "epilogue": (2, 16, 32, 128, 0, 320),
                               "big": (2, 64, 64, 320, 0, 320)}[case]
    x = rb(torch.randn(b, c0, h, w_, generator=g))
    x1 = rb(torch.randn(b, c1, h, w_, generator=g)) if c1 else None
    w = rb(torch.randn(n, c0 + c1, 3, 3, generator=g) * 0.03)
    bias = torch.randn(n, generator=g)
    cw = ops.ConvWeight(w, bias, prec, DEV)
    xin = torch.cat([x, x1], 1) if c1 else x
    ref = F.conv2d(xin, w, bias, padding=1)
    kw = {}
    if case == "epilogue":
        temb = torch.randn(b, n, generator=g)
        r0 = rb(torch.randn(b, n, h, w_, generator=g))
        ref = F.silu(0.5 * (ref + temb[:, :, None, None]) + r0)
        kw = dict(temb=temb.to(DEV), res0=nhwc(r0, prec.act), alpha=0.5, act=hip.ACT_SILU)
    y = ops.conv2d(nhwc(x, prec.act), cw, x1=nhwc(x1, prec.act) if c1 else None, tile=tile,
                   splitk=3 if case == "splitk" else 1, **kw)
    check(f"conv3x3_halo[tile{tile},{case}]", nchw(y), ref, 2e-2, 1e-2)
    if 20 <= tile <= 24 and case in ("plain", "epilogue"):          # dx-tap reuse also runs in the fp32 parity mode, any image size
        p32 = ops.Precision.get("fp32")
        x32 = torch.randn(3, 64, 7, 16, generator=g)              # M = 336: a ragged last tile, image rows of 16 pixels
        w32 = torch.randn(40, 64, 3, 3, generator=g) * 0.05
        y32 = ops.conv2d(nhwc(x32, p32.act), ops.ConvWeight(w32, bias[:40], p32, DEV), tile=tile, splitk=2 if case == "epilogue" else 1)
        check(f"conv3x3_dxr_fp32[tile{tile}]", nchw(y32), F.conv2d(x32, w32, bias[:40], padding=1), 2e-4, 2e-4)
    if case == "plain":     # a call the halo kernel cannot serve is refused, not silently rerouted
        with pytest.raises(hip.MfhipError):
            ops.conv2d(nhwc(x, prec.act), cw, stride=2, tile=tile)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
def test_conv_epilogue_temb_res_alpha_silu(prec_name, atol, rtol):
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 32, 9, 7, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    temb = torch.randn(3, 100, generator=g)            # strided slice [3, 64] of a wider table
    r0 = torch.randn(3, 64, 9, 7, generator=g)
    r1 = torch.randn(3, 64, 9, 7, generator=g)
    if prec_name == "bf16":
        x, w, r0 = rb(x), rb(w), rb(r0)
    ref = 0.5 * (F.conv2d(x, w, b, padding=1) + temb[:, 20:84, None, None]) + r0 + r1
    ref = F.silu(ref)
    cw = ops.ConvWeight(w, b, prec, DEV)
    y = ops.conv2d(nhwc(x, prec.act), cw, temb=temb.to(DEV)[:, 20:84], res0=nhwc(r0, prec.act),
                   res1=nhwc(r1, torch.float32), alpha=0.5, act=hip.ACT_SILU)
    check(f"conv_epilogue[{prec_name}]", nchw(y), ref, atol, rtol)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
@pytest.mark.parametrize("splitk", [1, 2])
def test_conv_shared_residual_rows(prec_name, atol, rtol, splitk):
    """mf_gemm_desc.res1_rows: a residual with 1/r of the output's batch is added to every replica (vector epilogue with
    prefetched bf16 residuals, fp32 residuals, the split-K reduce, a linear call, and ops.add)."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(4, 32, 8, 8, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    r0 = torch.randn(4, 64, 8, 8, generator=g)
    r1 = torch.randn(2, 64, 8, 8, generator=g)
    if prec_name == "bf16":
        x, w, r0, r1 = rb(x), rb(w), rb(r0), rb(r1)
    ref = F.conv2d(x, w, b, padding=1) + r0 + torch.cat([r1, r1])
    cw = ops.ConvWeight(w, b, prec, DEV)
    for r1_dt in (prec.act, torch.float32):
        y = ops.conv2d(nhwc(x, prec.act), cw, res0=nhwc(r0, prec.act), res1=nhwc(r1, r1_dt), splitk=splitk)
        check(f"conv_shared_res1[{prec_name},sk{splitk},{r1_dt}]", nchw(y), ref, atol, rtol)
    xl = torch.randn(6, 10, 32, generator=g)
    wl = torch.randn(48, 32, generator=g) * 0.1
    rl = torch.randn(2, 10, 48, generator=g)
    if prec_name == "bf16":
        xl, wl, rl = rb(xl), rb(wl), rb(rl)
    yl = ops.linear(xl.to(DEV, prec.act), ops.ConvWeight(wl, None, prec, DEV), res1=rl.to(DEV, prec.act), splitk=splitk)
    check(f"linear_shared_res1[{prec_name}]", yl, xl @ wl.T + torch.cat([rl] * 3), atol, rtol)
    a = torch.randn(4, 5, 5, 8, generator=g)
    bb = torch.randn(2, 5, 5, 8, generator=g)
    check("add_shared", ops.add(a.to(DEV), bb.to(DEV), torch.float32), a + torch.cat([bb, bb]), 1e-6, 1e-6)
    with pytest.raises(hip.MfhipError):
        ops.conv2d(nhwc(x, prec.act), cw, res1=nhwc(torch.randn(3, 64, 8, 8), prec.act))


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
@pytest.mark.parametrize("case", ["s2p1", "s2asym", "up", "cat", "cat_up", "1x1", "splitk"])
def test_conv_variants(prec_name, atol, rtol, case):
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(3)
    cin, cout = 24, 40
    x = torch.randn(2, cin, 10, 14, generator=g)
    x1 = torch.randn(2, 16, 10, 14, generator=g)
    kw = {}
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
    if prec_name == "bf16":
        x, x1, w = rb(x), rb(x1), rb(w)
    cw = ops.ConvWeight(w, b, prec, DEV)
    xa = nhwc(x, prec.act)
    if case == "s2p1":
        ref = F.conv2d(x, w, b, stride=2, padding=1)
        y = ops.conv2d(xa, cw, stride=2, padding=1)
    elif case == "s2asym":   # downsampling.py:140-142: F.pad(x, (0, 1, 0, 1)) then stride-2 conv, padding 0
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        y = ops.conv2d(xa, cw, stride=2, padding=(0, 0, 1, 1))
    elif case == "up":
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
        y = ops.conv2d(xa, cw, upsample=True)
    elif case == "cat":
        ref = F.conv2d(torch.cat([x, x1], 1), w, b, padding=1)
        y = ops.conv2d(xa, cw, x1=nhwc(x1, prec.act))
    elif case == "cat_up":
        ref = F.conv2d(F.interpolate(torch.cat([x, x1], 1), scale_factor=2.0, mode="nearest"), w, b, padding=1)
        y = ops.conv2d(xa, cw, x1=nhwc(x1, prec.act), upsample=True)
    elif case == "1x1":
        ref = F.conv2d(x, w, b)
        y = ops.conv2d(xa, cw, padding=0)
    else:
        ref = F.conv2d(x, w, b, padding=1)
        y = ops.conv2d(xa, cw, splitk=5)
    check(f"conv_{case}[{prec_name}]", nchw(y), ref, atol, rtol)


@pytest.mark.parametrize("tile", [37, 38, 39, 40, 41, 43, 44, 45, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66])
@pytest.mark.parametrize("hw,batch", [(8, 5), (16, 3), (4, 9)])
def test_staged_epilogue_rows_on_warp_specialised_tiles(tile, hw, batch, monkeypatch):
    """GemmArgs::epb: bias / time-embedding rows fetched into LDS by the staging waves' first DMAs.  Tiles of 128 / 256 rows over
    images of 64 / 256 / 16 pixels touch 2-16 images each (ragged last tile): every output row must add the time-embedding row of
    ITS image (the 16-pixel images exceed the rows a tile reserves: those calls keep the global loads), and the result must be
    bit-identical to the same launch with the staging switched off (MFHIP_NO_EPB is read once per process: compared against the
    reduce path of a 2-way split-K instead, which never stages, at a tolerance, and against F.conv2d)."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(500 + tile + hw)
    cin, cout = 64, 200                                   # N tail of 40 columns
    x = rb(torch.randn(batch, cin, hw, hw, generator=g))
    w = rb(torch.randn(cout, cin, 3, 3, generator=g) * 0.05)
    bias = torch.randn(cout, generator=g)
    temb = torch.randn(batch, cout, generator=g) * 3.0    # large against the conv: a wrong image's row is far outside the tolerance
    r0 = rb(torch.randn(batch, cout, hw, hw, generator=g))
    ref = F.conv2d(x, w, bias, padding=1) + temb[:, :, None, None] + r0
    cw = ops.ConvWeight(w, bias, prec, DEV)
    dx_only = tile in (37, 38, 39, 40, 47, 49, 51, 53, 55, 59, 61, 63, 65)
    if dx_only and not ((hw <= 128 and 128 % hw == 0) or hw % 128 == 0):
        pytest.skip("dx-reuse tiles need image rows that tile the block")
    try:
        y = ops.conv2d(nhwc(x, prec.act), cw, temb=temb.to(DEV), res0=nhwc(r0, prec.act), tile=tile, splitk=1)
    except hip.MfhipError:
        pytest.skip("tile does not apply to this geometry")
    check(f"epb[tile{tile},{hw}x{hw}x{batch}]", nchw(y), ref, 3e-2, 2e-2)
    y2 = ops.conv2d(nhwc(x, prec.act), cw, temb=temb.to(DEV), res0=nhwc(r0, prec.act), tile=tile, splitk=2)     # reduce launch: never staged
    assert float((y.float() - y2.float()).abs().max()) <= 0.07, "staged rows differ from the global-load epilogue"


@pytest.mark.parametrize("tile", [41, 42, 43, 44, 45, 46, 48, 50, 52, 54, 56, 57, 58, 60, 62, 64, 66])
@pytest.mark.parametrize("case", ["up", "s2", "cat", "1x1res"])
def test_conv_warp_specialised_ring(tile, case):
    """The warp-specialised form of the plain ring (four staging waves + the compute waves, 3-deep LDS ring) on the fast
    staging path (channel counts that are multiples of 64): upsampled, strided, concatenated and 1x1 calls with a fused
    epilogue, ragged M and N."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(77 + tile)
    b, h, w_, c0, c1, n = {"up": (2, 9, 12, 64, 0, 200), "s2": (2, 18, 14, 128, 0, 72), "cat": (1, 13, 16, 64, 128, 330),
                           "1x1res": (3, 10, 10, 192, 0, 170)}[case]
    x = rb(torch.randn(b, c0, h, w_, generator=g))
    x1 = rb(torch.randn(b, c1, h, w_, generator=g)) if c1 else None
    k = 1 if case == "1x1res" else 3
    w = rb(torch.randn(n, c0 + c1, k, k, generator=g) * 0.04)
    bias = torch.randn(n, generator=g)
    cw = ops.ConvWeight(w, bias, prec, DEV)
    xin = torch.cat([x, x1], 1) if c1 else x
    if case == "up":
        ref = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), w, bias, padding=1)
        y = ops.conv2d(nhwc(x, prec.act), cw, upsample=True, tile=tile, splitk=1)
    elif case == "s2":
        ref = F.conv2d(xin, w, bias, stride=2, padding=1)
        y = ops.conv2d(nhwc(x, prec.act), cw, stride=2, padding=1, tile=tile, splitk=2)
    elif case == "cat":
        ref = F.conv2d(xin, w, bias, padding=1)
        y = ops.conv2d(nhwc(x, prec.act), cw, x1=nhwc(x1, prec.act), tile=tile, splitk=1)
    else:
        r0 = rb(torch.randn(b, n, h, w_, generator=g))
        ref = F.silu(F.conv2d(xin, w, bias) + r0)
        y = ops.conv2d(nhwc(x, prec.act), cw, padding=0, res0=nhwc(r0, prec.act), act=hip.ACT_SILU, tile=tile, splitk=1)
    check(f"conv_ws_ring[tile{tile},{case}]", nchw(y), ref, 2e-2, 1e-2)


def test_conv_f32_activations_bf16_compute():
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 32, 8, 8, generator=g)
    w = rb(torch.randn(48, 32, 3, 3, generator=g) * 0.05)
    ref = F.conv2d(rb(x), w, None, padding=1)
    cw = ops.ConvWeight(w, None, prec, DEV)
    y = ops.conv2d(nhwc(x, torch.float32), cw, out_dtype=torch.float32)
    check("conv_a_f32", nchw(y), ref, 2e-3, 2e-3)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
@pytest.mark.parametrize("m,k,n", [(154, 768, 320), (8, 320, 1280), (300, 40, 77), (4096, 320, 2560), (64, 1280, 4)])
def test_linear(prec_name, atol, rtol, m, k, n):
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    if prec_name == "bf16":
        x, w = rb(x), rb(w)
    ref = F.linear(x, w, b)
    lw = ops.ConvWeight(w, b, prec, DEV)
    y = ops.linear(x.to(DEV, prec.act), lw)
    check(f"linear[{prec_name},{m}x{k}x{n}]", y, ref, atol, rtol)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
@pytest.mark.parametrize("tile,splitk", [(1, 1), (2, 1), (6, 3), (14, 1), (3, 2), (41, 1), (42, 2), (45, 1), (50, 2), (52, 1), (54, 1), (60, 2), (57, 1)])
def test_linear_column_panels_and_flattened_splits(prec_name, atol, rtol, tile, splitk):
    """Wide 1x1 GEMMs run their column tiles in panels of 8 (the last panel narrower) and K splits are part of the 1-D
    block order: every (tile_m, tile_n, split) must be visited exactly once — N = 1448 gives 12 / 23 / 10 column tiles, M
    and N both ragged."""
    prec = ops.Precision.get(prec_name)
    if tile >= 37 and prec_name != "bf16":
        pytest.skip("the warp-specialised tiles are bf16 only")
    g = torch.Generator().manual_seed(55)
    m, k, n = 700, 192, 1448
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    if prec_name == "bf16":
        x, w = rb(x), rb(w)
    y = ops.linear(x.to(DEV, prec.act), ops.ConvWeight(w, b, prec, DEV), tile=tile, splitk=splitk)
    check(f"linear_panels[{prec_name},tile{tile},sk{splitk}]", y, F.linear(x, w, b), atol, rtol)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
def test_linear_t(prec_name, atol, rtol):
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 77, 96, generator=g)
    w = torch.randn(64, 96, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    if prec_name == "bf16":
        x, w = rb(x), rb(w)
    ref = F.linear(x, w, b).transpose(1, 2)             # [3, 64, 77]
    lw = ops.ConvWeight(w, b, prec, DEV, raw=True)      # the weight is the A operand: never pre-split
    vt = ops.linear_t(x.to(DEV, prec.act), lw, 80)
    assert vt.shape == (3, 64, 80)
    check(f"linear_t[{prec_name}]", vt[:, :, :77], ref, atol, rtol)
    assert vt[:, :, 77:].abs().max().item() == 0.0


def sdpa_ref(q, k, v, heads):
    b, sq, c = q.shape
    d = c // heads
    qh = q.view(b, sq, heads, d).transpose(1, 2)
    kh = k.view(b, -1, heads, d).transpose(1, 2)
    vh = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(b, sq, c)


def make_vt(v, ld, dtype):
    b, skv, c = v.shape
    vt = torch.zeros(b, c, ld, dtype=dtype, device=DEV)
    vt[:, :, :skv] = v.transpose(1, 2).to(DEV, dtype)
    return vt


@pytest.mark.parametrize("prec_name,atol,rtol", [("fp32", 2e-4, 2e-4), ("f16x3", 2e-4, 2e-4), ("bf16x3", 1e-3, 1e-3), ("bf16", 2e-2, 2e-2)])
@pytest.mark.parametrize("heads,d,sq,skv", [(2, 40, 200, 200), (2, 8, 64, 77), (1, 512, 96, 96)])
def test_attention_unfused(prec_name, atol, rtol, heads, d, sq, skv):
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(7)
    c = heads * d
    q = torch.randn(2, sq, c, generator=g)
    k = torch.randn(2, skv, c, generator=g)
    v = torch.randn(2, skv, c, generator=g)
    if prec_name == "bf16":
        q, k, v = rb(q), rb(k), rb(v)
    ref = sdpa_ref(q, k, v, heads)
    ld = (skv + 7) // 8 * 8
    o = ops.attention_unfused(q.to(DEV, prec.act), k.to(DEV, prec.act), make_vt(v, ld, prec.act), heads, skv,
                              1.0 / math.sqrt(d), prec)
    check(f"attn_unfused[{prec_name},h{heads},d{d},{sq}x{skv}]", o, ref, atol, rtol)


@pytest.mark.parametrize("heads,d,sq,skv", [(8, 40, 4096, 4096), (8, 80, 1024, 1024), (8, 160, 256, 256),
                                            (8, 160, 64, 64), (8, 40, 4096, 77), (8, 80, 1024, 77),
                                            (8, 160, 256, 77), (4, 8, 200, 77), (3, 64, 130, 190)])
def test_attention_flash_bf16(heads, d, sq, skv):
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(8)
    c = heads * d
    q = rb(torch.randn(2, sq, c, generator=g))
    k = rb(torch.randn(2, skv, c, generator=g))
    v = rb(torch.randn(2, skv, c, generator=g))
    ref = sdpa_ref(q, k, v, heads)
    ld = (skv + 7) // 8 * 8
    o = ops.attention(q.to(DEV, torch.bfloat16), k.to(DEV, torch.bfloat16), make_vt(v, ld, torch.bfloat16), heads,
                      skv, 1.0 / math.sqrt(d), prec)
    check(f"attn_flash[h{heads},d{d},{sq}x{skv}]", o, ref, 2e-2, 2e-2)


@pytest.mark.parametrize("in_dt,out_dt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                          (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("c0,c1,hw,groups,silu", [(320, 0, 64 * 64, 32, True), (1280, 640, 16 * 16, 32, True),
                                                  (640, 320, 32 * 32, 32, False), (32, 0, 9 * 7, 32, True),
                                                  (64, 32, 5 * 5, 32, False), (512, 0, 1000, 32, True),
                                                  (1280, 1280, 8 * 8, 32, True), (1280, 0, 8 * 8, 32, False),
                                                  (640, 0, 32 * 32, 32, True), (1280, 640, 32 * 32, 32, True),
                                                  (320, 0, 32 * 32, 32, True), (320, 320, 33 * 31, 32, True)])
def test_groupnorm(in_dt, out_dt, c0, c1, hw, groups, silu):
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(2, c0, hw, 1, generator=g) * 2 + 0.7
    x1 = torch.randn(2, c1, hw, 1, generator=g) - 1.0 if c1 else None
    gamma = torch.randn(c0 + c1, generator=g)
    beta = torch.randn(c0 + c1, generator=g)
    if in_dt == torch.bfloat16:
        x0 = rb(x0)
        x1 = rb(x1) if x1 is not None else None
    xc = torch.cat([x0, x1], 1) if x1 is not None else x0
    ref = F.group_norm(xc, groups, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    y = hip.groupnorm(nhwc(x0, in_dt), gamma.to(DEV), beta.to(DEV), groups=groups, eps=1e-5, silu=silu,
                      out_dtype=out_dt, x1=nhwc(x1, in_dt) if x1 is not None else None)
    tol = 2e-5 if out_dt == torch.float32 else 2e-2
    check(f"groupnorm[{in_dt},{out_dt},{c0}+{c1},{hw}]", nchw(y), ref, tol * 5, tol)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,c", [(4096 * 2, 320), (77, 1280), (5, 32), (1000, 640)])
def test_layernorm(dt, rows, c):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(rows, c, generator=g) * 3 + 1
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    if dt == torch.bfloat16:
        x = rb(x)
    ref = F.layer_norm(x, (c,), gamma, beta, 1e-5)
    y = hip.layernorm(x.to(DEV, dt), gamma.to(DEV), beta.to(DEV), 1e-5, dt)
    tol = 1e-5 if dt == torch.float32 else 2e-2
    check(f"layernorm[{dt},{rows}x{c}]", y, ref, tol * 5, tol)


def test_softmax_rows():
    g = torch.Generator().manual_seed(11)
    s = torch.randn(300, 80, generator=g) * 4
    ref = torch.softmax(s[:, :77], -1)
    p = hip.softmax_rows(s.to(DEV), 77, torch.float32)
    check("softmax", p[:, :77], ref, 1e-6, 1e-5)
    assert p[:, 77:].abs().max().item() == 0.0


def test_elementwise_and_layout():
    g = torch.Generator().manual_seed(12)
    a = torch.randn(2, 4, 8, 8, generator=g)
    b = torch.randn(2, 6, 8, 8, generator=g)
    packed = hip.pack_nhwc(a.to(DEV), b.to(DEV), 16, torch.float32)
    ref = torch.zeros(2, 8, 8, 16)
    ref[..., :4] = a.permute(0, 2, 3, 1)
    ref[..., 4:10] = b.permute(0, 2, 3, 1)
    check("pack", packed, ref, 0, 0)
    back = hip.unpack_nchw(packed, 10)
    check("unpack", back, torch.cat([a, b], 1), 0, 0)
    x = torch.randn(1000, generator=g)
    y = torch.randn(1000, generator=g)
    check("add", hip.add(x.to(DEV), y.to(DEV).bfloat16(), torch.float32), x + rb(y), 1e-7, 1e-7)
    h = torch.randn(50, 2 * 96, generator=g)
    check("geglu", hip.geglu(h.to(DEV), torch.float32), h[:, :96] * F.gelu(h[:, 96:]), 1e-6, 1e-5)
    check("silu", hip.silu_f32(x.to(DEV)), F.silu(x), 1e-6, 1e-6)
    # timestep embedding (embeddings.py:27-67, flip_sin_to_cos=True, shift 0)
    t = torch.tensor([981.0, 1.0, 500.0])
    half = 160
    e = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    emb = t[:, None] * e[None]
    ref = torch.cat([torch.cos(emb), torch.sin(emb)], -1)
    check("timestep_embedding", hip.timestep_embedding(t.to(DEV), 320, True, 0.0), ref, 2e-4, 0)
    # cfg + ddim
    eu, ec, xx = torch.randn(3, 512, generator=g).unbind(0)
    sa, s1, sp, dc = 0.3, 0.95, 0.35, 0.93
    eps = eu + 7.5 * (ec - eu)
    ref = sp * ((xx - s1 * eps) / sa) + dc * eps
    got = hip.cfg_ddim_step(eu.to(DEV), ec.to(DEV), 7.5, xx.to(DEV), sa, s1, sp, dc)
    check("cfg_ddim", got, ref, 1e-5, 1e-5)
    check("cfg_combine", hip.cfg_combine(eu.to(DEV), ec.to(DEV), 7.5), eps, 1e-5, 1e-5)
    check("axpby", hip.axpby_n([eu.to(DEV), ec.to(DEV), xx.to(DEV)], [0.5, -2.0, 3.0]), 0.5 * eu - 2 * ec + 3 * xx,
          1e-5, 1e-5)
    # vae sample
    mom = torch.randn(2, 8, 6, 6, generator=g)
    noise = torch.randn(2, 4, 6, 6, generator=g)
    ref = (mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * noise) * 0.18215
    check("vae_sample", hip.vae_sample(nhwc(mom, torch.float32), noise.to(DEV), 4, 0.18215), ref, 1e-6, 1e-5)
    # nearest resize
    src = torch.randn(2, 3, 64, 48, generator=g)
    check("nearest", hip.nearest_resize(src.to(DEV), 8, 6), F.interpolate(src, size=(8, 6)), 0, 0)
    check("nearest2", hip.nearest_resize(src.to(DEV), 20, 10), F.interpolate(src, size=(20, 10)), 0, 0)


@pytest.mark.parametrize("prec_name,atol,rtol", PRECS)
def test_linear_geglu_fused(prec_name, atol, rtol):
    """FeedForward's GEGLU (activations.py:100-103) computed in the GEMM epilogue on interleaved weight rows."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(300, 64, generator=g)
    w = torch.randn(2 * 256, 64, generator=g) / 8
    b = torch.randn(2 * 256, generator=g)
    if prec_name == "bf16":
        x, w = rb(x), rb(w)
    h, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    y = ops.linear_geglu(x.to(DEV, prec.act), ops.geglu_weight(w, b, prec, DEV))
    assert y.shape == (300, 256)
    check(f"linear_geglu[{prec_name}]", y, ref, atol, rtol)


def test_c_abi_error_paths_and_edge_cases():
    """The C ABI refuses what it cannot serve (error code + message, nothing launched) and handles degenerate sizes."""
    import ctypes as C
    lib = hip.load()
    prec = ops.Precision.get("bf16")
    x = torch.randn(2, 8, 8, 32, device=DEV).bfloat16()
    cw = ops.ConvWeight(torch.randn(16, 32, 3, 3) * 0.05, torch.randn(16), prec, DEV)
    # channel mismatch, non-contiguous input, misaligned operand, unknown attention head dim
    with pytest.raises(hip.MfhipError, match="channels"):
        ops.conv2d(x[..., :24].contiguous(), cw)
    with pytest.raises(hip.MfhipError, match="contiguous"):
        ops.conv2d(x.permute(0, 2, 1, 3), cw)
    out = torch.empty(2, 8, 8, 16, device=DEV, dtype=torch.bfloat16)
    big = torch.zeros(2 * 8 * 8 * 32 + 8, device=DEV, dtype=torch.bfloat16)
    mis = big[1:1 + 2 * 8 * 8 * 32].view(2, 8, 8, 32)                      # 2-byte offset: not 16-byte aligned
    with pytest.raises(hip.MfhipError, match="aligned"):
        hip.gemm_conv(mis, cw.w, out, dtype=torch.bfloat16, c0=32, lda0=32, batch=2, h_in=8, w_in=8, h_out=8, w_out=8,
                      kh=3, kw=3, pad_t=1, pad_l=1, n=16, bias=cw.bias)
    q = torch.randn(1, 16, 8 * 24, device=DEV).bfloat16()
    with pytest.raises(hip.MfhipError, match="head_dim"):
        hip.attention_bf16(q, q, torch.zeros(1, 8 * 24, 16, device=DEV).bfloat16(), torch.empty_like(q), ldq=192, ldk=192,
                           ldvt=16, ldo=192, batch=1, heads=8, sq=16, skv=16, head_dim=24, scale=0.2)
    with pytest.raises(hip.MfhipError, match="groups"):
        hip.groupnorm(x, torch.ones(32, device=DEV), torch.zeros(32, device=DEV), groups=5, eps=1e-5, silu=False,
                      out_dtype=torch.bfloat16)
    assert lib.mf_gemm_conv(None, None) != 0 and b"null" in lib.mf_last_error()
    # degenerate sizes: one pixel, one output channel group, K smaller than a tile; and a 1-row LayerNorm
    g = torch.Generator().manual_seed(9)
    x1 = rb(torch.randn(1, 8, 1, 1, generator=g))
    w1 = rb(torch.randn(8, 8, 3, 3, generator=g) * 0.1)
    y = ops.conv2d(nhwc(x1, prec.act), ops.ConvWeight(w1, None, prec, DEV))
    check("conv 1x1 image", nchw(y), F.conv2d(x1, w1, None, padding=1), 2e-2, 1e-2)
    ln = hip.layernorm(torch.ones(1, 8, device=DEV), torch.ones(8, device=DEV), torch.zeros(8, device=DEV), 1e-5, torch.float32)
    assert torch.allclose(ln.cpu(), torch.zeros(1, 8), atol=1e-3)
    # determinism: the same launch twice is bit-identical (no atomics anywhere, split-K slabs reduced in a fixed order)
    xb = torch.randn(2, 16, 16, 256, device=DEV).bfloat16()
    cwb = ops.ConvWeight(torch.randn(64, 256, 3, 3) * 0.02, torch.randn(64), prec, DEV)
    a = ops.conv2d(xb, cwb, splitk=4, tile=1)
    b = ops.conv2d(xb, cwb, splitk=4, tile=1)
    assert torch.equal(a, b)
    gn1 = hip.groupnorm(xb, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), groups=32, eps=1e-5, silu=True,
                        out_dtype=torch.bfloat16)
    gn2 = hip.groupnorm(xb, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), groups=32, eps=1e-5, silu=True,
                        out_dtype=torch.bfloat16)
    assert torch.equal(gn1, gn2)


def test_reference_layer_known_answers_on_the_hip_path():
    """The reference's own ResnetBlock2D / Upsample2D / Downsample2D KATs (tests/models/test_layers_utils.py) through
    the C ABI in the fp32 parity mode: GroupNorm+SiLU, conv with fused time-embedding / shortcut / residual epilogues,
    nearest-2x upsampling folded into the conv gather, stride-2 conv."""
    import kat_layers as K
    prec = ops.Precision.get("fp32")
    dev = DEV

    def cw(sd, name):
        return ops.ConvWeight(sd[name + ".weight"], sd[name + ".bias"], prec, dev)

    for name, shortcut in (("resnet_default", False), ("resnet_shortcut", True)):
        sd, x, temb = K.resnet_case(shortcut)
        xh = nhwc(x, prec.act)
        ones, zeros = torch.ones(32, device=dev), torch.zeros(32, device=dev)
        t = ops.linear(F.silu(temb).to(dev), cw(sd, "time_emb_proj"), out_dtype=torch.float32)
        h = hip.groupnorm(xh, ones, zeros, groups=32, eps=1e-6, silu=True, out_dtype=prec.act)
        h = ops.conv2d(h, cw(sd, "conv1"), temb=t)
        h = hip.groupnorm(h, ones, zeros, groups=32, eps=1e-6, silu=True, out_dtype=prec.act)
        sc = ops.conv2d(xh, cw(sd, "conv_shortcut"), padding=0) if shortcut else xh
        K.check_slice(name, nchw(ops.conv2d(h, cw(sd, "conv2"), res0=sc)))
    sd, x = K.sampler_case("up")
    K.check_slice("upsample_conv", nchw(ops.conv2d(nhwc(x, prec.act), cw(sd, "conv"), upsample=True)))
    sd, x = K.sampler_case("down")
    K.check_slice("downsample_conv", nchw(ops.conv2d(nhwc(x, prec.act), cw(sd, "conv"), stride=2, padding=1)))
    # DownBlock2D / UpBlock2D: the never-materialised cat([h, skip]) feeds GroupNorm, conv1 and the 1x1 shortcut
    for kind in ("down", "up"):
        sd, x, temb, skip = K.block_case(kind)
        xh, sk = nhwc(x, prec.act), (nhwc(skip, prec.act) if skip is not None else None)
        cin = 64 if kind == "up" else 32
        t = ops.linear(F.silu(temb).to(dev), cw(sd, "resnets.0.time_emb_proj"), out_dtype=torch.float32)
        h = hip.groupnorm(xh, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), groups=32, eps=1e-6, silu=True,
                          out_dtype=prec.act, x1=sk)
        h = ops.conv2d(h, cw(sd, "resnets.0.conv1"), temb=t)
        h = hip.groupnorm(h, torch.ones(32, device=dev), torch.zeros(32, device=dev), groups=32, eps=1e-6, silu=True,
                          out_dtype=prec.act)
        sc = ops.conv2d(xh, cw(sd, "resnets.0.conv_shortcut"), padding=0, x1=sk) if kind == "up" else xh
        h = ops.conv2d(h, cw(sd, "resnets.0.conv2"), res0=sc)
        out = ops.conv2d(h, cw(sd, "sampler.conv"), stride=2, padding=1) if kind == "down" else \
            ops.conv2d(h, cw(sd, "sampler.conv"), upsample=True)
        K.check_slice(f"{kind}_block", nchw(out))
    # Transformer2DModel with cross-attention through the model code (GroupNorm, proj_in, LN, attention, GEGLU, proj_out)
    from reflecting_reality_amd import models as M
    sd, x, ctx = K.transformer_case()
    m = M.UNet2DConditionModel.__new__(M.UNet2DConditionModel)
    m.prec, m.device, m.config = prec, torch.device(dev), {"norm_num_groups": 32}
    m.P, m.tdepth, m._cross_kv, m._ehs_gen = {}, {}, {}, 0
    m._prepare_transformer(sd, "")
    out = m._transformer("", nhwc(x, prec.act), ctx.to(dev), 2)
    K.check_slice("transformer_cross", nchw(out))


@pytest.mark.parametrize("dtype,heads,d,cross,tol", [(torch.bfloat16, 8, 40, False, 3e-2), (torch.bfloat16, 8, 40, True, 3e-2),
                                                     (torch.float32, 2, 24, True, 2e-4)])
def test_attn_processor_on_the_reference_operator_abi(dtype, heads, d, cross, tol):
    """MfhipAttnProcessor called the way `Attention.forward` calls its processor (attention_processor.py:490-531),
    against the reference processor's arithmetic (AttnProcessor2_0: torch SDPA) on the same torch layers."""
    from reflecting_reality_amd import MfhipAttnProcessor
    torch.manual_seed(3)
    c, cd = heads * d, 48

    class Attn(torch.nn.Module):                       # the attributes a processor reads (attention_processor.py:80-215)
        def __init__(self):
            super().__init__()
            self.heads, self.spatial_norm, self.group_norm, self.norm_cross = heads, None, None, None
            self.residual_connection, self.rescale_output_factor = False, 1.0
            self.to_q = torch.nn.Linear(c, c, bias=False)
            self.to_k = torch.nn.Linear(cd if cross else c, c, bias=False)
            self.to_v = torch.nn.Linear(cd if cross else c, c, bias=False)
            self.to_out = torch.nn.ModuleList([torch.nn.Linear(c, c), torch.nn.Dropout(0.0)])

    attn = Attn().to(DEV, dtype)
    x = torch.randn(2, 136, c, device=DEV, dtype=dtype)
    ctx = torch.randn(2, 77, cd, device=DEV, dtype=dtype) if cross else None
    got = MfhipAttnProcessor()(attn, x, encoder_hidden_states=ctx)
    src = ctx if cross else x
    q, k, v = attn.to_q(x), attn.to_k(src), attn.to_v(src)
    sp = lambda t: t.view(2, -1, heads, d).transpose(1, 2).float()
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(2, -1, c)
    ref = attn.to_out[0](ref.to(dtype)).float()
    check(f"attn processor[{dtype},{'cross' if cross else 'self'}]", got, ref.cpu(), tol, tol)
    with pytest.raises(NotImplementedError):
        MfhipAttnProcessor()(attn, x, attention_mask=torch.zeros(1, device=DEV))


def test_add_vector_and_scalar_paths():
    """mf_add: the 8-wide bf16 kernel (aligned, n % 8 == 0) and the generic one give the fp32 sum rounded once."""
    g = torch.Generator().manual_seed(11)
    for n in (8, 4096 * 320, 1000):                      # 1000 % 8 == 0 too; odd sizes below
        a = torch.randn(n, generator=g).bfloat16()
        b = torch.randn(n, generator=g).bfloat16()
        ref = (a.float() + b.float()).bfloat16()
        assert torch.equal(hip.add(a.to(DEV), b.to(DEV), torch.bfloat16).cpu(), ref)
    a = torch.randn(1003, generator=g).bfloat16()
    b = torch.randn(1003, generator=g)
    assert torch.equal(hip.add(a.to(DEV), b.to(DEV), torch.float32).cpu(), a.float() + b)      # mixed dtypes: generic kernel
    assert torch.equal(hip.add(a.to(DEV), a.to(DEV), torch.bfloat16).cpu(), (a.float() * 2).bfloat16())   # n % 8 != 0


WS_RING_TILES = (0, 41, 42, 43, 44, 45, 46, 48, 50, 52, 54, 56, 57, 58, 60, 62, 64, 66)     # the tiles that serve ln_colsum / vt_out (0 = autotuned among them)


@pytest.mark.parametrize("tile", WS_RING_TILES)
@pytest.mark.parametrize("rows,c,n,mode", [(2 * 1024, 640, 640, "linear"), (2 * 4096, 320, 320, "res"), (300, 320, 1280, "linear"),
                                           (2 * 256, 1280, 1280, "linear"), (2 * 1024, 640, 2 * 2560, "geglu"), (77, 64, 72, "linear")])
def test_linear_with_folded_layernorm(tile, rows, c, n, mode):
    """LayerNorm folded into the Linear that consumes it (attention.py:203,233,261 + the to_q / GEGLU projections): the GEMM
    reads the UN-normalised rows, its staging waves gather (mean, rstd) of every row, the epilogue applies
    rstd * (acc - mean * colsum(W gamma)) + (bias + W beta).  Reference: fp32 LayerNorm -> Linear on the bf16-rounded rows."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(31)
    x = rb(torch.randn(rows, c, generator=g) * 1.7 + 0.9)                  # mean != 0: the rank-1 correction is live
    gamma, beta = torch.randn(c, generator=g) * 0.5 + 1.0, torch.randn(c, generator=g) * 0.3
    w = torch.randn(n, c, generator=g) / math.sqrt(c)
    b = torch.randn(n, generator=g)
    res = rb(torch.randn(rows, n, generator=g)) if mode == "res" else None
    xn = F.layer_norm(x, (c,), gamma, beta, 1e-5)
    if mode == "geglu":
        hh, gate = F.linear(xn, w, b).chunk(2, dim=-1)
        ref = hh * F.gelu(gate)
        y = ops.linear_geglu(x.to(DEV, prec.act), ops.geglu_weight(w, b, prec, DEV, ln=(gamma, beta, 1e-5)), tile=tile)
    else:
        ref = F.linear(xn, w, b) + (res if res is not None else 0.0)
        lw = ops.ConvWeight(w, b, prec, DEV, ln=(gamma, beta, 1e-5))
        y = ops.linear(x.to(DEV, prec.act), lw, res0=res.to(DEV, prec.act) if res is not None else None, tile=tile)
    # W gamma is rounded to bf16 once more than in the unfused path: the usual bf16 GEMM tolerance
    check(f"linear_ln[{mode},{rows}x{c}->{n},tile{tile}]", y, ref, 3e-2, 2e-2)


@pytest.mark.parametrize("ptile", [69, 70])
@pytest.mark.parametrize("prec_name", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,c,n,mode,with_ln", [(2048, 320, 2 * 2560, "geglu", True), (8192, 320, 2 * 1280, "geglu", True), (1024, 640, 640, "res", False),
                                                   (16384, 128, 640, "linear", True), (4096, 192, 480, "silu", False), (2048, 1280, 1280, "res", True),
                                                   (32768, 320, 320, "res", False), (8192, 640, 2 * 2560, "geglu", False), (256, 320, 1600, "linear", True)])
def test_persistent_short_k_gemm(prec_name, rows, c, n, mode, with_ln, ptile):
    """Tile 70 (csrc/gemm_pers.hip, round 6): the same structure on 128 rows with EIGHT compute waves + four staging + four epilogue
    waves, a two-deep ring beside a whole 128 x 160 fp32 slab, and the epilogue's global operands fetched a tile ahead.
    Tile 69 (csrc/gemm_nloop.hip): a block keeps 64 rows of A and walks a range of 160-column output tiles, the epilogue of tile j
    running on its own waves under the main loop of tile j + 1.  Cases: two to sixteen output tiles per block, one (only the shared last
    slab), an odd count; two, three, five, ten and twenty K tiles per output tile (two: every chunk of a slab in ONE barrier interval);
    GEGLU, residual, SiLU, plain epilogues; with and without a folded LayerNorm; both 16-bit storage types.  Reference: fp32 on the
    rounded operands; also compared with the warp-specialised ring tile 48 on the same call."""
    prec = ops.Precision.get(prec_name)
    rnd = rb if prec_name == "bf16" else (lambda t: t.half().float())
    g = torch.Generator().manual_seed(69 + rows + n)
    x = rnd(torch.randn(rows, c, generator=g) * 1.4 + 0.6)
    gamma, beta = torch.randn(c, generator=g) * 0.5 + 1.0, torch.randn(c, generator=g) * 0.3
    w = torch.randn(n, c, generator=g) / math.sqrt(c)
    b = torch.randn(n, generator=g)
    ln = (gamma, beta, 1e-5) if with_ln else None
    xn = F.layer_norm(x, (c,), gamma, beta, 1e-5) if with_ln else x
    wr = w if with_ln else rnd(w)
    xd = x.to(DEV, prec.act)
    if mode == "geglu":
        hh, gate = F.linear(xn, wr, b).chunk(2, dim=-1)
        ref = hh * F.gelu(gate)
        gw = ops.geglu_weight(w, b, prec, DEV, ln=ln)
        y, y48 = ops.linear_geglu(xd, gw, tile=ptile), ops.linear_geglu(xd, gw, tile=48)
    else:
        res = rnd(torch.randn(rows, n, generator=g)) if mode == "res" else None
        ref = F.linear(xn, wr, b) + (res if res is not None else 0.0)
        if mode == "silu":
            ref = F.silu(ref)
        lw = ops.ConvWeight(w, b, prec, DEV, ln=ln)
        kw = dict(res0=res.to(DEV, prec.act) if res is not None else None)
        if mode == "silu":
            kw["act"] = hip.ACT_SILU
        y, y48 = ops.linear(xd, lw, tile=ptile, **kw), ops.linear(xd, lw, tile=48, **kw)
    tol = (3e-2, 2e-2) if prec_name == "bf16" else (6e-3, 4e-3)
    check(f"persistent tile {ptile}[{prec_name},{mode},{rows}x{c}->{n}]", y, ref, *tol)
    # against the ring tile: the same products summed in another order, then one rounding to 16 bits
    d = (y.float() - y48.float()).abs().max().item()
    assert d <= (2.0 ** -6 if prec_name == "bf16" else 2.0 ** -9) * max(1.0, ref.abs().max().item()), f"tile {ptile} vs tile 48: {d}"


def test_persistent_short_k_gemm_is_refused_where_it_cannot_run():
    prec = ops.Precision.get("bf16")
    lw = ops.ConvWeight(torch.randn(320, 320), None, prec, DEV)
    for rows, why in ((100, "M % 64"),):
        with pytest.raises(hip.MfhipError, match="persistent short-K"):
            ops.linear(torch.randn(rows, 320, device=DEV).bfloat16(), lw, tile=69)
    with pytest.raises(hip.MfhipError, match="persistent short-K"):              # N % 160 != 0
        ops.linear(torch.randn(128, 320, device=DEV).bfloat16(), ops.ConvWeight(torch.randn(200, 320), None, prec, DEV), tile=69)
    with pytest.raises(hip.MfhipError, match="persistent short-K"):              # one K tile
        ops.linear(torch.randn(128, 64, device=DEV).bfloat16(), ops.ConvWeight(torch.randn(160, 64), None, prec, DEV), tile=69)


def test_persistent_128_row_gemm_is_refused_where_it_cannot_run():
    prec = ops.Precision.get("bf16")
    lw = ops.ConvWeight(torch.randn(320, 320), None, prec, DEV)
    with pytest.raises(hip.MfhipError, match="persistent 128-row"):              # M % 128 != 0
        ops.linear(torch.randn(192, 320, device=DEV).bfloat16(), lw, tile=70)
    with pytest.raises(hip.MfhipError, match="persistent 128-row"):              # N % 160 != 0
        ops.linear(torch.randn(128, 320, device=DEV).bfloat16(), ops.ConvWeight(torch.randn(200, 320), None, prec, DEV), tile=70)
    with pytest.raises(hip.MfhipError, match="persistent 128-row"):              # a 3 x 3 convolution
        ops.conv2d(torch.randn(1, 16, 8, 64, device=DEV).bfloat16(), ops.ConvWeight(torch.randn(160, 64, 3, 3), None, prec, DEV), tile=70)
    with pytest.raises(hip.MfhipError, match="persistent 128-row|bf16"):         # fp32 mode
        ops.linear(torch.randn(128, 320, device=DEV), ops.ConvWeight(torch.randn(320, 320), None, ops.Precision.get("fp32"), DEV), tile=70)


def test_folded_layernorm_is_refused_where_it_cannot_run():
    prec = ops.Precision.get("bf16")
    lw = ops.ConvWeight(torch.randn(64, 64), None, prec, DEV, ln=(torch.ones(64), torch.zeros(64), 1e-5))
    x = torch.randn(128, 64, device=DEV).bfloat16()
    with pytest.raises(hip.MfhipError, match="does not serve"):
        ops.linear(x, lw, tile=14)                       # a tile without staging waves
    with pytest.raises(hip.MfhipError, match="plain bf16"):
        ops.ConvWeight(torch.randn(64, 64), None, ops.Precision.get("f16x3"), DEV, ln=(torch.ones(64), torch.zeros(64), 1e-5))


@pytest.mark.parametrize("tile", WS_RING_TILES + (70,))
@pytest.mark.parametrize("batch,tokens,c,with_ln", [(2, 4096, 320, True), (2, 1024, 640, True), (3, 256, 1280, True), (2, 64, 1280, True),
                                                    (2, 1024, 320, False), (8, 4096, 320, True), (1, 128, 320, False)])
def test_fused_qkv_projection_with_transposed_v(tile, batch, tokens, c, with_ln):
    """Self-attention's to_q | to_k | to_v as ONE GEMM (attention_processor.py:1246-1254): q | k leave as [B, S, 2C], the V third
    leaves the epilogue transposed as V^T [B, C, S] (what mf_attention_bf16 reads), norm1 folded in.  Tile 70 (round 6): the V tiles
    of the persistent GEMM go through a transposed slab; a block's range may mix q | k and V tiles, start inside V, or be one tile."""
    if tile == 70 and (batch * tokens) % 128:
        with pytest.raises(hip.MfhipError, match="persistent 128-row"):
            ops.linear_qkv(torch.zeros(batch, tokens, c, device=DEV).bfloat16(), ops.ConvWeight(torch.zeros(3 * c, c), None, ops.Precision.get("bf16"), DEV), tile=70)
        return
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(32)
    x = rb(torch.randn(batch, tokens, c, generator=g) * 1.3 - 0.4)
    gamma, beta = torch.randn(c, generator=g) * 0.5 + 1.0, torch.randn(c, generator=g) * 0.3
    w = torch.randn(3 * c, c, generator=g) / math.sqrt(c)
    xn = F.layer_norm(x, (c,), gamma, beta, 1e-5) if with_ln else x
    ref = F.linear(xn, w if with_ln else rb(w))
    lw = ops.ConvWeight(w, None, prec, DEV, ln=(gamma, beta, 1e-5) if with_ln else None)
    qk, vt = ops.linear_qkv(x.to(DEV, prec.act), lw, tile=tile)
    assert qk.shape == (batch, tokens, 2 * c) and vt.shape == (batch, c, tokens)
    check(f"qkv.qk[{batch}x{tokens}x{c},tile{tile}]", qk, ref[..., :2 * c], 3e-2, 2e-2)
    check(f"qkv.vt[{batch}x{tokens}x{c},tile{tile}]", vt.transpose(1, 2), ref[..., 2 * c:], 3e-2, 2e-2)


# ---- round 4: the in-launch split-K combine ---------------------------------------------------------------------------------
@pytest.mark.parametrize("prec_name,tile", [("bf16", t) for t in (1, 2, 3, 6, 41, 43, 44, 48)] + [("f16x3", t) for t in (1, 2, 3, 6, 41, 44)]
                         + [("bf16", 14), ("bf16", 45), ("fp32", 3)])
def test_in_launch_split_k_combine_equals_the_reduce_launch(prec_name, tile):
    """mf_gemm_desc.sk_tickets: the K-slice block that arrives last at its tile's ticket sums the slabs in slice order and runs the
    epilogue inside the GEMM launch.  Same slabs, same order, same epilogue as the separate reduce launch => BIT-identical output,
    for every tile that carries the tail (and for tiles / precisions that do not: they keep the reduce launch, (bf16, 14) and
    (fp32, 3) here).  Cases: a deep-K 3x3 conv at 8 x 8 with a ragged M (tile rows past M), residual + temb + SiLU epilogue;
    a Linear with an N tail and a shared residual; repeated launches (tickets re-armed) and two streams with their own tickets."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(100 + tile)
    b, cin, cout, h, w = 3, 128, 200, 8, 8                                  # M = 192 (a 128-row tile and a half), N tail of 40
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.03
    bias = torch.randn(cout, generator=g)
    temb = torch.randn(b, cout, generator=g).to(DEV)
    r0 = torch.randn(b, cout, h, w, generator=g)
    cw = ops.ConvWeight(wt, bias, prec, DEV)
    kw = dict(temb=temb, res0=nhwc(r0, prec.act), act=hip.ACT_SILU, tile=tile)
    hip.sk_tickets(DEV)
    table = lambda: hip._tickets[torch.cuda.current_device()][0]
    for sk in (2, 3, 6):
        ref = ops.conv2d(nhwc(x, prec.act), cw, splitk=sk, **kw)
        for rep in range(3):
            got = ops.conv2d(nhwc(x, prec.act), cw, splitk=sk, sk_fused=True, **kw)
            assert torch.equal(got, ref), f"tile {tile} splitk {sk} rep {rep}: max diff {float((got.float() - ref.float()).abs().max()):.3e}"
            assert int(table().abs().sum()) == 0, "tickets must be zero again after the launch"
    xl = torch.randn(70, 512, generator=g)                                   # M = 70 (ragged), K = 512, N = 72 (scalar tail path when N % 8)
    for n in (72, 77 if prec_name != "bf16" else 80):
        wl = torch.randn(n, 512, generator=g) * 0.05
        rl = torch.randn(35, n, generator=g)
        lw = ops.ConvWeight(wl, None, prec, DEV)
        ref = ops.linear(xl.to(DEV, prec.act), lw, res1=rl.to(DEV, prec.act), splitk=4, tile=tile)
        got = ops.linear(xl.to(DEV, prec.act), lw, res1=rl.to(DEV, prec.act), splitk=4, tile=tile, sk_fused=True)
        assert torch.equal(got, ref), f"linear n={n} tile {tile}"
    # two streams at once, each with its own slabs and tickets
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ref = ops.conv2d(nhwc(x, prec.act), cw, splitk=3, **kw)
    torch.cuda.synchronize()
    outs = []
    for st in (s1, s2):
        with torch.cuda.stream(st):
            xa = nhwc(x, prec.act)
            for _ in range(4):
                outs.append(ops.conv2d(xa, cw, splitk=3, sk_fused=True, **kw))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)
    assert int(table().abs().sum()) == 0


def test_in_launch_split_k_combine_under_uneven_load_and_graph_replay():
    """The hand-off must not depend on timing, placement or cold caches (cdna_hip_programming.md Guideline 16, Pitfall 3): a
    production-size split (512 x 1280 x 11520, 6 and 12 slices: 192 / 384 blocks over 8 XCDs) replayed 20 times from a hipGraph
    back to back with an unrelated streaming kernel in between, every output compared bit for bit with the reduce-launch form."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 1280, 8, 8, generator=g)
    wt = torch.randn(1280, 1280, 3, 3, generator=g) * 0.01
    cw = ops.ConvWeight(wt, torch.randn(1280, generator=g), prec, DEV)
    xa = nhwc(x, prec.act)
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    for tile, sk in ((41, 6), (48, 12), (44, 4), (6, 3)):
        ref = ops.conv2d(xa, cw, splitk=sk, tile=tile)
        hip.sk_tickets(DEV)
        out = torch.empty_like(ref)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(3):
                y = ops.conv2d(xa, cw, splitk=sk, tile=tile, sk_fused=True)
                big.mul_(1.0000001)
            out.copy_(y)
        for rep in range(20):
            gr.replay()
            assert torch.equal(out, ref), f"tile {tile} splitk {sk} replay {rep}"
        assert int(hip._tickets[torch.cuda.current_device()][0].abs().sum()) == 0


# ---- round 6: GroupNorm statistics from the producing GEMM's epilogue (mf_gemm_desc.gn_part -> mf_groupnorm_desc.part0 / part1) ----
GN_PART_TILES = [0, 1, 2, 3, 5, 6, 9, 13, 14, 15, 16, 20, 21, 25, 26, 27, 29, 31, 34, 37, 38, 40, 41, 42, 44, 45, 47, 48, 49, 50, 51, 52, 53, 54,
                 57, 58, 59, 60, 62, 63, 64, 66, 67, 68]


def _colsum_ref(y_nhwc: torch.Tensor, rows: int):
    """(sum, sum of squares) of every block of `rows` output rows, per channel, in float64."""
    n = y_nhwc.shape[-1]
    y = y_nhwc.double().reshape(-1, rows, n)
    return torch.stack([y.sum(1), (y * y).sum(1)], dim=-1)          # [blocks, n, 2]


@pytest.mark.parametrize("prec_name", ["bf16", "fp16", "f16x3", "fp32"])
@pytest.mark.parametrize("tile", GN_PART_TILES)
def test_conv_leaves_groupnorm_partial_sums_of_its_final_output(prec_name, tile):
    """A conv with temb + two residuals + alpha and gn_part=True: the partial sums attached to the output equal the per-channel
    (sum, sum of squares) of the STORED output over blocks of R rows, R reported by the library (the tile's rows when the
    epilogue produced them, 128 / 64 / 32 from the column-sum launch otherwise), and they are bit-identical from run to run."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(7)
    b, h, w, cin, n = 2, 32, 32, 64, 320            # hw = 1024: one image per block of rows for every tile (BM <= 256)
    x = torch.randn(b, h, w, cin, generator=g)
    wt = torch.randn(n, cin, 3, 3, generator=g) * 0.05
    bias = torch.randn(n, generator=g)
    temb = torch.randn(b, n, generator=g).to(DEV)
    r0 = torch.randn(b, h, w, n, generator=g)
    r1 = torch.randn(b, h, w, n, generator=g)
    cw = ops.ConvWeight(wt, bias, prec, DEV)
    args = dict(temb=temb, res0=r0.to(DEV, prec.act), res1=r1.to(DEV, prec.act), alpha=0.5, gn_part=32, tile=tile)
    try:
        y = ops.conv2d(x.to(DEV, prec.act), cw, **args)
    except hip.MfhipError as e:
        assert "not instantiated" in str(e) or "does not apply" in str(e), e     # a tile this precision does not have
        return
    part, rows, grouped = y._gn_part
    assert rows in (32, 64, 128, 192, 256) and (h * w) % rows == 0
    nb = b * h * w // rows
    got = part[: nb * n * 2].view(nb, n, 2).double().cpu()
    ref = _colsum_ref(y.float().cpu(), rows)
    # the epilogue sums the fp32 values BEFORE the storage rounding: 16-bit storage moves a sum of `rows` values by <= rows * 2^-9 |y|
    ulp = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}.get(prec_name, 2.0 ** -22)
    scale_s = float(y.float().abs().max()) * rows
    scale_q = float(y.float().abs().max()) ** 2 * rows
    es, eq = (got[..., 0] - ref[..., 0]).abs().max().item(), (got[..., 1] - ref[..., 1]).abs().max().item()
    print(f"gn_part[{prec_name}, tile {tile}]: rows per block {rows}, |sum err| {es:.3e} (scale {scale_s:.1f}), |sumsq err| {eq:.3e} (scale {scale_q:.1f})")
    assert es <= 0.25 * ulp * scale_s + 1e-3 and eq <= 0.5 * ulp * scale_q + 1e-3
    if grouped:      # per-group sums of the 32 groups behind the per-channel ones: the same numbers, summed over each group's 10 channels
        assert grouped == 32
        gg = part[nb * n * 2: nb * n * 2 + nb * 32 * 2].view(nb, 32, 2).double().cpu()
        gref = ref.view(nb, 32, n // 32, 2).sum(2)
        egs, egq = (gg[..., 0] - gref[..., 0]).abs().max().item(), (gg[..., 1] - gref[..., 1]).abs().max().item()
        assert egs <= 0.25 * ulp * scale_s * 10 + 1e-3 and egq <= 0.5 * ulp * scale_q * 10 + 1e-3, (egs, egq)
    y2 = ops.conv2d(x.to(DEV, prec.act), cw, **args)
    assert y2._gn_part[1:] == (rows, grouped) and torch.equal(y2._gn_part[0][: nb * (n + grouped) * 2], part[: nb * (n + grouped) * 2]) and torch.equal(y2, y)


@pytest.mark.parametrize("case", ["splitk", "odd_hw", "1x1", "up", "s2"])
def test_groupnorm_partial_sums_fallback_and_variants(case):
    """Launches whose epilogue cannot produce the sums (split-K reduce; an image size the tile's rows do not divide) leave them
    through the column-sum launch: same contract.  1x1 / upsampled / strided convs produce them in the epilogue."""
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(11)
    b, h, w, cin, n = (2, 24, 12, 64, 64) if case == "odd_hw" else (2, 32, 32, 64, 128)
    x = rb(torch.randn(b, h, w, cin, generator=g))
    kw = dict(gn_part=True)
    if case == "splitk":
        kw.update(splitk=2, tile=1)
    if case == "odd_hw":
        kw.update(tile=14)                        # hw = 288 = 2.25 x 128 rows: a row block would straddle two images
    if case == "up":
        kw.update(upsample=True)
    if case == "s2":
        kw.update(stride=2, padding=1)
    k = 1 if case == "1x1" else 3
    if case == "1x1":
        kw.update(padding=0)
    cw = ops.ConvWeight(rb(torch.randn(n, cin, k, k, generator=g) * 0.05), torch.randn(n, generator=g), prec, DEV)
    # hw must exceed 256 for ops.conv2d to ask (below that GroupNorm is one launch): call the descriptor level for the small case
    y = ops.conv2d(x.to(DEV, prec.act), cw, **kw)
    if case == "s2":
        assert not hasattr(y, "_gn_part")         # 16 x 16 output: GroupNorm's one-launch form, nothing asked
        return
    part, rows = y._gn_part[:2]
    hw = y.shape[1] * y.shape[2]
    assert hw % rows == 0 and (case not in ("splitk", "odd_hw") or rows in (32, 64, 128))
    nb = y.shape[0] * hw // rows
    got = part[: nb * n * 2].view(nb, n, 2).double().cpu()
    ref = _colsum_ref(y.float().cpu(), rows)
    tol_s, tol_q = 2.0 ** -9 * rows * float(y.float().abs().max()), 2.0 ** -8 * rows * float(y.float().abs().max()) ** 2
    assert (got[..., 0] - ref[..., 0]).abs().max() <= tol_s and (got[..., 1] - ref[..., 1]).abs().max() <= tol_q


@pytest.mark.parametrize("prec_name", ["bf16", "fp16", "f16x3", "fp32"])
@pytest.mark.parametrize("hw,splitk,tile", [(8, 4, 1), (8, 3, 3), (16, 2, 1), (16, 8, 6)])
def test_split_k_reduce_deferred_to_the_groupnorm_is_bit_identical(prec_name, hw, splitk, tile, monkeypatch):
    """mf_gemm_desc.defer_reduce + mf_groupnorm_desc.sk_ws: a split-K conv with bias + temb leaves the summing of its K slices to the
    GroupNorm that consumes it (a resnet's conv1 -> norm2, resnet.py:381-393) — same bits as reduce launch + GroupNorm, one launch less;
    a launch that does not split leaves nothing pending; the unwritten tensor is refused by every other op."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(21)
    b, cin, n = 4, 320, 320
    x = (torch.randn(b, hw, hw, cin, generator=g)).to(DEV, prec.act)
    cw = ops.ConvWeight(torch.randn(n, cin, 3, 3, generator=g) * 0.03, torch.randn(n, generator=g), prec, DEV)
    temb = torch.randn(b, n, generator=g).to(DEV)
    norm = (torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV))
    try:
        monkeypatch.setattr(hip, "DEFER_REDUCE", False)
        y_ref = ops.conv2d(x, cw, temb=temb, splitk=splitk, tile=tile, defer_reduce=True)
        assert getattr(y_ref, "_sk_pending", None) is None
        z_ref = ops.groupnorm(y_ref, norm, groups=32, eps=1e-5, silu=True, out_dtype=prec.act)
    except hip.MfhipError as e:
        assert "not instantiated" in str(e) or "does not apply" in str(e), e
        return
    monkeypatch.setattr(hip, "DEFER_REDUCE", True)
    y = ops.conv2d(x, cw, temb=temb, splitk=splitk, tile=tile, defer_reduce=True)
    assert y._sk_pending is not None and y._sk_pending[1] >= 2, "the forced split-K launch should have left its reduce"
    with pytest.raises(hip.MfhipError, match="deferred"):
        hip.add(y, y, prec.act)
    z = ops.groupnorm(y, norm, groups=32, eps=1e-5, silu=True, out_dtype=prec.act)
    assert y._sk_pending is None
    assert torch.equal(z, z_ref), f"deferred reduce differs from reduce + GroupNorm by {(z.float() - z_ref.float()).abs().max()}"
    y1 = ops.conv2d(x, cw, temb=temb, splitk=1, tile=tile, defer_reduce=True)       # no split: complete as always
    assert getattr(y1, "_sk_pending", None) is None
    tol = 2e-2 if prec_name in ("bf16", "fp16") else 1e-4
    check(f"defer_reduce split-K 1 [{prec_name}]", y1, y_ref.float().cpu(), tol, tol)
    y2 = ops.conv2d(x, cw, temb=temb, res0=y_ref, splitk=splitk, tile=tile, defer_reduce=True)   # a residual: not deferrable
    assert getattr(y2, "_sk_pending", None) is None


@pytest.mark.parametrize("prec_name", ["f16x3", "fp32", "fp16"])
def test_groupnorm_partial_sums_fallback_in_every_storage(prec_name):
    """The column-sum launch behind a split-K reduce reads the stored output in its own dtype: fp32 storage (the split-precision and
    fp32 modes) takes the 8-channel form too (two 16-byte loads), not the scalar walk."""
    prec = ops.Precision.get(prec_name)
    g = torch.Generator().manual_seed(12)
    b, h, w, cin, n = 2, 32, 32, 64, 128
    x = torch.randn(b, h, w, cin, generator=g)
    cw = ops.ConvWeight(torch.randn(n, cin, 3, 3, generator=g) * 0.05, torch.randn(n, generator=g), prec, DEV)
    y = ops.conv2d(x.to(DEV, prec.act), cw, gn_part=True, splitk=2, tile=1)
    part, rows = y._gn_part[:2]
    assert rows == 128
    nb = b * h * w // rows
    got = part[: nb * n * 2].view(nb, n, 2).double().cpu()
    ref = _colsum_ref(y.float().cpu(), rows)
    ymax = float(y.float().abs().max())
    assert (got[..., 0] - ref[..., 0]).abs().max() <= 2.0 ** -20 * rows * ymax and (got[..., 1] - ref[..., 1]).abs().max() <= 2.0 ** -19 * rows * ymax ** 2
    y2 = ops.conv2d(x.to(DEV, prec.act), cw, gn_part=True, splitk=2, tile=1)
    assert torch.equal(y2._gn_part[0][: nb * n * 2], part[: nb * n * 2])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("c,hw,rows,silu", [(320, 4096, 256, True), (640, 1024, 128, True), (1280, 1024, 128, False), (320, 4096, 64, False)])
def test_groupnorm_from_producer_group_sums_is_one_launch(dt, c, hw, rows, silu):
    """mf_groupnorm with grp0 (per-group sums of every block of `rows` rows, as a GEMM with gn_groups leaves them): no statistics
    pass and no finalize launch; the result equals the other two routes to the rounding of the statistic and is bit-reproducible."""
    g = torch.Generator().manual_seed(6)
    b, groups = 2, 32
    x = (torch.randn(b, hw, c, generator=g) * 1.5 + 0.7).to(DEV, dt)
    gamma, beta = torch.randn(c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
    v = x.float().view(-1, rows, c)
    chan = torch.stack([v.sum(1), (v * v).sum(1)], dim=-1)                               # [blocks, c, 2]
    grp = chan.view(-1, groups, c // groups, 2).sum(2)                                   # [blocks, groups, 2]
    x._gn_part = (torch.cat([chan.reshape(-1), grp.reshape(-1)]).contiguous(), rows, groups)
    y = hip.groupnorm(x, gamma, beta, groups=groups, eps=1e-5, silu=silu, out_dtype=dt)
    y2 = hip.groupnorm(x, gamma, beta, groups=groups, eps=1e-5, silu=silu, out_dtype=dt)
    assert torch.equal(y, y2)
    del x._gn_part
    ref = hip.groupnorm(x, gamma, beta, groups=groups, eps=1e-5, silu=silu, out_dtype=dt)
    tref = F.group_norm(x.float().cpu().permute(0, 2, 1), groups, gamma.cpu(), beta.cpu(), 1e-5).permute(0, 2, 1)
    if silu:
        tref = F.silu(tref)
    tol = {torch.bfloat16: 3e-2, torch.float32: 2e-5}[dt]
    check(f"groupnorm_from_groups[{dt}]", y, tref, tol, tol)
    assert (y.float() - ref.float()).abs().max().item() <= tol


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("c0,c1,hw,silu", [(320, 0, 4096, True), (640, 320, 1024, True), (1280, 640, 1024, False), (320, 0, 1024, False)])
def test_groupnorm_from_producer_partial_sums(dt, c0, c1, hw, silu):
    """mf_groupnorm with part0 / part1: the same result as computing the statistics from the tensor (to the rounding of the
    statistic), for one and two segments whose row blocks differ (256 and 128 rows), groups straddling the segment border
    (960 = 640 + 320 channels: 30 per group), and bit-identical from run to run."""
    g = torch.Generator().manual_seed(5)
    b, groups = 2, 32
    x0 = (torch.randn(b, hw, c0, generator=g) * 1.5 + 0.7).to(DEV, dt)
    x1 = (torch.randn(b, hw, c1, generator=g) * 0.5 - 0.3).to(DEV, dt) if c1 else None
    gamma, beta = torch.randn(c0 + c1, generator=g).to(DEV), torch.randn(c0 + c1, generator=g).to(DEV)

    def parts(x, rows):
        v = x.float().view(-1, rows, x.shape[-1])
        return torch.stack([v.sum(1), (v * v).sum(1)], dim=-1).contiguous().view(-1), rows

    x0._gn_part = parts(x0, 256)
    if x1 is not None:
        x1._gn_part = parts(x1, 128)
    y = hip.groupnorm(x0, gamma, beta, groups=groups, eps=1e-5, silu=silu, out_dtype=dt, x1=x1)
    y2 = hip.groupnorm(x0, gamma, beta, groups=groups, eps=1e-5, silu=silu, out_dtype=dt, x1=x1)
    assert torch.equal(y, y2)
    del x0._gn_part
    ref = hip.groupnorm(x0, gamma, beta, groups=groups, eps=1e-5, silu=silu, out_dtype=dt, x1=x1)
    xc = torch.cat([x0, x1], -1).float().cpu() if x1 is not None else x0.float().cpu()
    tref = F.group_norm(xc.permute(0, 2, 1), groups, gamma.cpu(), beta.cpu(), 1e-5).permute(0, 2, 1)
    if silu:
        tref = F.silu(tref)
    tol = {torch.bfloat16: 3e-2, torch.float16: 4e-3, torch.float32: 2e-5}[dt]
    check(f"groupnorm_from_parts[{dt}]", y, tref, tol, tol)
    d = (y.float() - ref.float()).abs().max().item()
    print(f"from partial sums vs own statistics pass: max diff {d:.3e}")
    assert d <= tol
