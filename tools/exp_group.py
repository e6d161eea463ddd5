"""Does ONE grouped launch of two same-shape convolutions (the BrushNet and the UNet instance of a low-resolution resnet conv)
beat the two launches the pipeline issues today on two HIP streams?  Times, from captured graphs on one box:
  (1) two launches on one stream, (2) one launch per stream (what the denoise graph does), (3) one z-batched launch (nz = 2).
(3) uses mf_gemm_conv's existing blockIdx.z batching with DISTINCT activations and weights per z — the arithmetic and the
memory traffic of a grouped launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops

prec = ops.Precision.get("bf16")
dev = "cuda"


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


side = torch.cuda.Stream()
for (b, hw, cin, cout) in [(8, 32, 640, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280), (8, 16, 2560, 1280), (8, 8, 2560, 1280)]:
    xall = torch.randn(2, b, hw, hw, cin, device=dev).bfloat16()
    x = [xall[0], xall[1]]
    w = [ops.ConvWeight(torch.randn(cout, cin, 3, 3) * 0.02, torch.randn(cout), prec, dev) for _ in range(2)]
    wall = torch.stack([w[0].w, w[1].w]).contiguous()          # DISTINCT weights per problem, like BrushNet's and the UNet's
    m = b * hw * hw
    # tuned single-problem launches
    for i in range(2):
        ops.conv2d(x[i], w[i])

    def one_stream():
        ops.conv2d(x[0], w[0]); ops.conv2d(x[1], w[1])

    def two_streams():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ops.conv2d(x[1], w[1])
        ops.conv2d(x[0], w[0])
        cur.wait_stream(side)

    out2 = torch.empty(2, b, hw, hw, cout, device=dev, dtype=torch.bfloat16)
    best_z = (1e9, None)
    key = None
    for tile in (0,):
        def grouped():
            hip.gemm_conv(x[0], wall[0], out2, dtype=prec.code, ldw=w[0].ldw, c0=cin, lda0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw,
                          w_out=hw, kh=3, kw=3, stride=1, pad_t=1, pad_l=1, n=cout, bias=w[0].bias, nz=2, zdiv=1, a_zs=(x[0].numel(), 0),
                          w_zs=(wall[0].numel(), 0), o_zs=(m * cout, 0))
        grouped()
        tz = bench(grouped)
    t1, t2 = bench(one_stream), bench(two_streams)
    fl = 2 * 2.0 * m * cout * 9 * cin
    print(f"conv3x3 {cin}->{cout} @{hw}x{hw} (M={m}): one stream {t1:.1f} us, two streams {t2:.1f} us, grouped (nz=2, autotuned) {tz:.1f} us "
          f"-> {fl / t1 / 1e6:.0f} / {fl / t2 / 1e6:.0f} / {fl / tz / 1e6:.0f} TF/s", flush=True)
