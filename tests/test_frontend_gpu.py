"""The image front-end on the device (csrc/frontend.hip): VaeImageProcessor against outputs of the REFERENCE's
own class (tests/golden/frontend.npz) and the host path it replaces; the dataset transforms (dataset.py:98-192) against the
oracle's numpy / torch restatement — PARITY UNPINNED for that one module: it imports h5py / torchvision / cv2, which this
image lacks, so the reference staticmethods cannot be run to pin the restatement (stated in each test)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mirrorfusion_ref as R  # noqa: E402
from reflecting_reality_amd import frontend, hip  # noqa: E402
from reflecting_reality_amd.pipeline import VaeImageProcessor  # noqa: E402

DEV = "cuda"


def test_preprocess_on_device_equals_host_path():
    """VaeImageProcessor.preprocess (image_processor.py:446-555): [0,1] tensors become 2x - 1, tensors that already hold
    negatives pass through (the decision is a device-side min), nearest resize when the size differs."""
    g = torch.Generator().manual_seed(1)
    p = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    for t in (torch.rand(2, 3, 64, 48, generator=g), torch.rand(2, 1, 64, 48, generator=g) * 2 - 1, torch.zeros(1, 3, 8, 8)):
        host = p.preprocess(t, height=t.shape[-2], width=t.shape[-1])
        dev = p.preprocess(t.to(DEV), height=t.shape[-2], width=t.shape[-1])
        assert dev.is_cuda and torch.equal(dev.cpu(), host)
    t = torch.rand(2, 3, 32, 32, generator=g)
    assert torch.equal(p.preprocess(t.to(DEV), height=64, width=48).cpu(), p.preprocess(t, height=64, width=48))
    mm = hip.minmax(t.to(DEV))
    assert float(mm[0]) == float(t.min()) and float(mm[1]) == float(t.max())


def test_mask_keep_concat_and_postprocess():
    g = torch.Generator().manual_seed(2)
    m3 = (torch.rand(3, 3, 40, 24, generator=g) > 0.5).float() * 2 - 1
    m3[0, :, :4] = torch.tensor([1.0, -0.5, -0.5]).view(3, 1, 1)            # sums to exactly 0: not < 0 -> hole
    assert torch.equal(hip.mask_keep(m3.to(DEV)).cpu(), (m3.sum(1)[:, None] < 0).float())
    a, b, c = torch.randn(4, 4, 8, 8, generator=g), torch.randn(2, 1, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    got = hip.concat_channels([a.to(DEV), b.to(DEV), c.to(DEV)], 4)
    assert torch.equal(got.cpu(), torch.cat([a, b.repeat(2, 1, 1, 1), c], 1))
    x = torch.randn(2, 3, 16, 20, generator=g) * 1.5
    ref = (x / 2 + 0.5).clamp(0, 1)
    assert torch.equal(hip.postprocess(x.to(DEV)).cpu(), ref)
    u8 = hip.postprocess(x.to(DEV), uint8=True).cpu().numpy()
    assert u8.dtype == np.uint8 and np.array_equal(u8, (ref.permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8"))
    p = VaeImageProcessor()
    pil_dev = p.postprocess(x.to(DEV), output_type="pil", do_denormalize=[True, True])
    pil_host = p.postprocess(x, output_type="pil", do_denormalize=[True, True])
    assert all(np.array_equal(np.array(i), np.array(j)) for i, j in zip(pil_dev, pil_host))
    assert np.array_equal(p.postprocess(x.to(DEV), output_type="np", do_denormalize=[True, True]),
                          p.postprocess(x, output_type="np", do_denormalize=[True, True]))


def test_device_front_end_matches_the_reference_vae_image_processor():
    """The device kernels (mf_minmax / mf_image_normalize / mf_nearest_resize / mf_postprocess through VaeImageProcessor on
    cuda tensors) against the outputs of the REFERENCE's VaeImageProcessor (image_processor.py:446-610) on the same seeded
    inputs (tests/golden/frontend.npz, tools/make_golden.py::frontend) — not against the package's own host path."""
    import os
    from test_host_logic import _frontend_inputs
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend.npz"))
    I = _frontend_inputs()
    ip = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    for name in ("t01", "tneg", "mask"):
        x = I[name].to(DEV)
        for tag, kw in (("same", dict(height=40, width=56)), ("resized", dict(height=32, width=48))):
            got = ip.preprocess(x, **kw)
            assert got.is_cuda and np.array_equal(got.cpu().numpy(), G[f"{name}_{tag}"]), (name, tag)
    post = I["post"].to(DEV)
    assert np.array_equal(ip.postprocess(post, output_type="pt", do_denormalize=[True, True]).cpu().numpy(), G["post_pt"])
    assert np.array_equal(ip.postprocess(post, output_type="np", do_denormalize=[True, True]), G["post_np"])
    assert np.array_equal(np.stack([np.array(im) for im in ip.postprocess(post, output_type="pil", do_denormalize=[True, True])]),
                          G["post_pil"])
    assert np.array_equal(ip.postprocess(post, output_type="pt", do_denormalize=[True, False]).cpu().numpy(), G["post_pt_mixed"])


@pytest.mark.parametrize("use_mask", [True, False])
@pytest.mark.parametrize("rng", [(-1, 1), (0, 1)])
def test_apply_transforms_depth_max_scene_depth(use_mask, rng):
    """examples/brushnet/dataset/dataset.py:98-145 (oracle restatement; the reference module itself cannot be imported:
    h5py / torchvision / cv2 are absent)."""
    g = np.random.default_rng(3)
    depth = (g.random((512, 512), dtype=np.float32) * 7.0).astype(np.float32)
    mask = np.zeros((512, 512, 3), dtype=np.uint8)
    mask[100:300, 200:420] = 255
    ref = R.apply_transforms_depth_ref(depth, mask if use_mask else None, max_scene_depth=5.0, norm_range=rng, delta=0.5)
    got = frontend.apply_transforms_depth(depth, mask if use_mask else None, max_scene_depth=5.0, norm_range=rng, delta=0.5)
    assert tuple(got.shape) == (1, 512, 512) and got.is_cuda
    assert float((got.cpu() - ref).abs().max()) < 2e-7
    with pytest.raises(ValueError):
        frontend.apply_transforms_depth(depth, norm_range=(0, 2))
    with pytest.raises(ValueError):
        frontend.apply_transforms_depth(depth, normalization_method="median")


@pytest.mark.parametrize("rng", [(-1, 1), (0, 1)])
def test_apply_transforms_depth_percentile(rng):
    """dataset.py:115-127: clip to np.percentile(d, 2) / np.percentile(d, 98) and map to the range.  The device path finds the four
    order statistics by a radix select (no sort) and interpolates like numpy's 'linear' method.  ORACLE UNPINNED: the
    reference's dataset module cannot be imported here (h5py / torchvision / cv2), the numpy restatement is the checker."""
    g = np.random.default_rng(4)
    depth = (g.standard_normal((512, 512)).astype(np.float32) * 2.0 + 3.0).astype(np.float32)
    depth[5, 7], depth[100, 3] = 1.0e4, -50.0                                             # outliers the percentiles exist to clip
    ref = R.apply_transforms_depth_ref(depth, normalization_method="percentile", norm_range=rng)
    got = frontend.apply_transforms_depth(depth, normalization_method="percentile", norm_range=rng)
    assert tuple(got.shape) == (1, 512, 512) and got.is_cuda
    assert float((got.cpu() - ref).abs().max()) < 2e-6
    # the selection itself, exactly: order statistics of a tensor with ties, negatives and both zeros
    x = torch.from_numpy(g.standard_normal(70001).astype(np.float32))
    x[:500] = 0.25
    x[500:600] = -0.0
    x[600:700] = 0.0
    srt = torch.sort(x).values
    ranks = [0, 1400, 35000, 70000]
    assert torch.equal(hip.select_ranks(x.to(DEV), ranks).cpu(), srt[ranks])


@pytest.mark.parametrize("antialias", [None, False])
@pytest.mark.parametrize("shape", [(384, 512), (512, 384), (300, 300), (512, 512), (600, 800), (1024, 1536), (2048, 1100), (515, 512)])
def test_bicubic_resize_center_crop_and_normals(shape, antialias):
    """torchvision's Resize(resolution, BICUBIC) + CenterCrop + Normalize of dataset.py:150-164,184-192 as one device kernel.
    antialias=None (the default) is the reference's pinned torchvision 0.18: Resize(antialias=True), whose tensor path calls
    torch.nn.functional.interpolate(mode='bicubic', align_corners=False, antialias=True) at every scale — up-sampling
    (384 -> 512), down-sampling by 1.2x ... 4x and the mixed case; antialias=False is the plain kernel.  The checker is that
    very interpolate on the CPU (ATen: the arithmetic torchvision calls; torchvision itself is absent from this image) inside
    the oracle's restatement of the two transforms, whose numpy lines stay PARITY UNPINNED (module header)."""
    g = np.random.default_rng(5)
    h, w = shape
    res = 512
    depth = (g.random((h, w), dtype=np.float32) * 4.0).astype(np.float32)
    normals = g.random((h, w, 3), dtype=np.float32)
    aa = antialias is not False
    ref_d = R.apply_transforms_depth_ref(depth, max_scene_depth=5.0, resolution=res, antialias=aa)
    got_d = frontend.apply_transforms_depth(depth, max_scene_depth=5.0, resolution=res, antialias=antialias)
    assert tuple(got_d.shape) == (1, res, res)
    err_d = float((got_d.cpu() - ref_d).abs().max())
    ref_n = R.apply_transforms_normals_ref(normals, res, antialias=aa)
    got_n = frontend.apply_transforms_normals(normals, res, antialias=antialias)
    assert tuple(got_n.shape) == (3, res, res)
    err_n = float((got_n.cpu() - ref_n).abs().max())
    print(f"{shape} antialias={antialias}: depth max err {err_d:.2e}, normals {err_n:.2e}")
    assert err_d < 5e-6 and err_n < 5e-6           # values in [-1, 1]: fp32 rounding of a <= 17 x 17-tap weighted sum
    if aa and min(h, w) > res:                     # the two kernels really differ when down-sampling (the low-pass filter)
        plain = frontend.apply_transforms_depth(depth, max_scene_depth=5.0, resolution=res, antialias=False)
        assert float((plain - got_d).abs().max()) > 1e-2
    with pytest.raises(NotImplementedError):
        frontend.apply_transforms_normals(normals, res, normals_conditioning_mode="ip_adapter")


def test_antialiased_bicubic_kernel_against_aten_on_ragged_sizes():
    """mf_bicubic_aa_resize_crop alone against F.interpolate(antialias=True) on the CPU: odd sizes, strong down-sampling (taps
    clipped at every border), up-sampling, and a crop window that is not centred."""
    g = torch.Generator().manual_seed(11)
    for (h, w, nh, nw) in ((37, 53, 16, 23), (96, 128, 48, 64), (100, 75, 33, 25), (40, 60, 40, 25), (17, 19, 64, 80), (511, 257, 96, 64)):
        x = torch.rand(2, h, w, generator=g) * 2.0 - 1.0
        ref = torch.nn.functional.interpolate(x[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)[0]
        got = hip.bicubic_resize_crop(x.to(DEV), (nh, nw), (0, 0), (nh, nw), antialias=True)
        assert float((got.cpu() - ref).abs().max()) < 3e-6, (h, w, nh, nw)
        top, left, ch, cw = nh // 5, nw // 7, nh // 2, nw // 2
        got = hip.bicubic_resize_crop(x.to(DEV), (nh, nw), (top, left), (ch, cw), 2.0, -1.0, antialias=True)
        assert float((got.cpu() - (2.0 * ref[:, top:top + ch, left:left + cw] - 1.0)).abs().max()) < 6e-6
