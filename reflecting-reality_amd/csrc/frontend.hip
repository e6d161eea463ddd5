// The image front-end of the pipeline on the device: what the reference does on the host with
// torch / numpy before and after the denoise loop.
//   VaeImageProcessor.preprocess  (image_processor.py:446-555): [0,1] -> [-1,1] unless the tensor already holds negatives
//                                 (:540-547: the decision depends on the data: min over the whole tensor)
//   mask 3 channels -> keep mask   (pipeline_brushnet.py:1139: (sum over channels < 0))
//   conditioning assembly          (pipeline_brushnet.py:1196-1215: torch.cat along channels, CFG duplication)
//   VaeImageProcessor.postprocess  (image_processor.py:557-610: (x / 2 + 0.5).clamp(0, 1), optional uint8 HWC)
//   HDF5Dataset.apply_transforms_depth, "max_scene_depth" method (examples/brushnet/dataset/dataset.py:98-145)
// Bandwidth-bound streaming kernels; reductions are two fixed-order stages; nothing synchronises with the host (a
// data-dependent decision is left in device memory and read by the next kernel).
#include <float.h>
#include "mf_common.h"

namespace {

inline unsigned fgrid(int64_t n, int cap = 4096) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

constexpr int MM_BLOCKS = 512;

// stage 1: per-block (min, max) of x (optionally only where mask > 0: then `mask` has n elements too)
__global__ __launch_bounds__(256) void minmax_stage1(const float* x, const float* mask, int64_t n, float* part) {
    __shared__ float smin[256], smax[256];
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (mask && !(mask[i] > 0.0f)) continue;
        const float v = x[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}
__global__ __launch_bounds__(256) void minmax_stage2(const float* part, int nparts, float* out) {
    __shared__ float smin[256], smax[256];
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = smin[0]; out[1] = smax[0]; }
}

// y = 2x - 1 when the tensor's minimum (minmax[0], device) is >= 0, else y = x   (image_processor.py:540-547)
__global__ __launch_bounds__(256) void image_normalize_kernel(const float* x, float* y, int64_t n, const float* minmax) {
    const bool norm = minmax[0] >= 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = norm ? 2.0f * x[i] - 1.0f : x[i];
}

// out[b][0][p] = (sum_c m[b][c][p] < 0) ? 1 : 0        (pipeline_brushnet.py:1139)
__global__ __launch_bounds__(256) void mask_keep_kernel(const float* m, float* out, int batch, int c, int64_t hw) {
    const int64_t total = (int64_t)batch * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw;
        float s = 0.0f;
        for (int cc = 0; cc < c; ++cc) s += m[(b * c + cc) * hw + p];
        out[i] = s < 0.0f ? 1.0f : 0.0f;
    }
}

struct ConcatArgs { const float* src[8]; int ch[8]; int bstride[8]; int nsrc, ctot; };
// out[b][c][p]: channels of up to 8 NCHW sources one after the other; source s contributes ch[s] channels and is read
// at batch index b % bstride[s]... (bstride = the source's own batch: a smaller batch is repeated: CFG duplication)
__global__ __launch_bounds__(256) void concat_channels_kernel(const ConcatArgs a, float* out, int batch, int64_t hw) {
    const int64_t total = (int64_t)batch * a.ctot * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % hw;
        int64_t t = i / hw;
        int c = (int)(t % a.ctot);
        const int b = (int)(t / a.ctot);
        int s = 0;
        while (c >= a.ch[s]) { c -= a.ch[s]; ++s; }
        out[i] = a.src[s][(((int64_t)(b % a.bstride[s])) * a.ch[s] + c) * hw + p];
    }
}

// postprocess: y = clamp(x / 2 + 0.5, 0, 1) as fp32 NCHW, or round(y * 255) as uint8 NHWC (image_processor.py:557-610)
__global__ __launch_bounds__(256) void postprocess_kernel(const float* x, float* y32, unsigned char* y8, int batch, int c, int64_t hw,
                                                          int denorm) {
    const int64_t total = (int64_t)batch * c * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = x[i];
        if (denorm) v = fminf(fmaxf(v * 0.5f + 0.5f, 0.0f), 1.0f);
        if (y32) y32[i] = v;
        if (y8) {
            const int64_t p = i % hw;
            const int64_t t = i / hw;
            const int cc = (int)(t % c);
            const int64_t b = t / c;
            y8[(b * hw + p) * c + cc] = (unsigned char)rintf(v * 255.0f);          // numpy .round(): half to even
        }
    }
}

// apply_transforms_depth, "max_scene_depth": scene = use_mask ? max over mask + delta : max_scene_depth;
// out = 2 * clip(d, 0, scene) / scene - 1  (norm_range [-1, 1]) or clip / scene ([0, 1])
__global__ __launch_bounds__(256) void depth_normalize_kernel(const float* d, float* out, int64_t n, const float* minmax, int use_mask,
                                                              float max_scene_depth, float delta, int signed_range) {
    const float scene = use_mask ? minmax[1] + delta : max_scene_depth;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = fminf(fmaxf(d[i], 0.0f), scene) / scene;
        out[i] = signed_range ? 2.0f * v - 1.0f : v;
    }
}


// ---- order statistics for apply_transforms_depth(normalization_method="percentile") (dataset.py:115-127) ---------------------
// np.percentile(d, q) = linear interpolation between the two order statistics around q/100 * (n - 1).  A two-level radix
// select on the order-preserving integer image of the floats finds up to four ranks without sorting: 65536-bin histogram of
// the high halves -> the bin that holds each rank -> 65536-bin histogram of the low halves inside that bin -> the exact key.
// Integer atomics only: the result does not depend on the order of the adds.
__device__ __forceinline__ unsigned ord_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_val(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
constexpr int SEL_MAX = 4;

__global__ __launch_bounds__(256) void sel_hist_hi_kernel(const float* x, int64_t n, unsigned* hist) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) atomicAdd(&hist[ord_key(x[i]) >> 16], 1u);
}

// one block of 1024 threads: for each rank, the bin whose cumulative count first exceeds it, and the rank inside that bin
__global__ __launch_bounds__(1024) void sel_pick_kernel(const unsigned* hist, int nsets, const int64_t* ranks, int nr, unsigned* bin_out,
                                                        int64_t* rem_out) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    for (int r = 0; r < nr; ++r) {
        const unsigned* h = hist + (nsets > 1 ? (int64_t)r * 65536 : 0);
        unsigned long long s = 0;
        for (int j = 0; j < 64; ++j) s += h[t * 64 + j];
        part[t] = s;
        __syncthreads();
        if (t == 0) {
            const unsigned long long want = (unsigned long long)(nsets > 1 ? rem_out[r] : ranks[r]);
            unsigned long long acc = 0;
            int blk = 0;
            while (blk < 1023 && acc + part[blk] <= want) { acc += part[blk]; ++blk; }
            int b = blk * 64;
            while (b < blk * 64 + 63 && acc + h[b] <= want) { acc += h[b]; ++b; }
            bin_out[r] = (unsigned)b;
            rem_out[r] = (int64_t)(want - acc);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sel_hist_lo_kernel(const float* x, int64_t n, const unsigned* bins, int nr, unsigned* hist2) {
    unsigned b[SEL_MAX];
    for (int r = 0; r < SEL_MAX; ++r) b[r] = r < nr ? bins[r] : 0xffffffffu;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned k = ord_key(x[i]);
        for (int r = 0; r < nr; ++r)
            if ((k >> 16) == b[r]) atomicAdd(&hist2[(int64_t)r * 65536 + (k & 0xffffu)], 1u);
    }
}

__global__ void sel_value_kernel(const unsigned* bins_hi, const unsigned* bins_lo, int nr, float* vals) {
    const int r = threadIdx.x;
    if (r < nr) vals[r] = ord_val((bins_hi[r] << 16) | bins_lo[r]);
}

// out = clip(d, d2, d98) mapped to [0, 1] or [-1, 1]; d2 / d98 interpolated from the four order statistics like numpy's
// 'linear' method (a + (b - a) * t, taken from the b side for t >= 0.5)
__global__ __launch_bounds__(256) void depth_percentile_kernel(const float* d, float* out, int64_t n, const float* v4, float t_lo, float t_hi,
                                                               int signed_range) {
    auto lerp = [](float a, float b, float t) { return t >= 0.5f ? b - (b - a) * (1.0f - t) : a + (b - a) * t; };
    const float d2 = lerp(v4[0], v4[1], t_lo), d98 = lerp(v4[2], v4[3], t_hi);
    const float inv = 1.0f / (d98 - d2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = (fminf(fmaxf(d[i], d2), d98) - d2) * inv;
        out[i] = signed_range ? 2.0f * v - 1.0f : v;
    }
}

// ---- torchvision Resize(bicubic) + CenterCrop (dataset.py:150-164, 184-192) ------------------------------------------------------
// PyTorch's upsample_bicubic2d (align_corners = False, A = -0.75, no antialiasing) evaluated only inside the crop window:
// out[p][y][x] = bicubic(src[p], (y + oy + 0.5) * sh - 0.5, (x + ox + 0.5) * sw - 0.5), then y = a * v + b (Normalize).
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }
__global__ __launch_bounds__(256) void bicubic_crop_kernel(const float* src, float* dst, int planes, int h_in, int w_in, int h_res, int w_res,
                                                           int oy, int ox, int h_out, int w_out, float a, float b) {
    const float A = -0.75f;
    const float sh = (float)h_in / (float)h_res, sw = (float)w_in / (float)w_res;
    const int64_t total = (int64_t)planes * h_out * w_out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w_out);
        const int64_t t = i / w_out;
        const int y = (int)(t % h_out);
        const int pl = (int)(t / h_out);
        const float fy = ((float)(y + oy) + 0.5f) * sh - 0.5f, fx = ((float)(x + ox) + 0.5f) * sw - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        const float ty = fy - (float)iy, tx = fx - (float)ix;
        const float wy[4] = {cubic2(ty + 1.0f, A), cubic1(ty, A), cubic1(1.0f - ty, A), cubic2(2.0f - ty, A)};
        const float wx[4] = {cubic2(tx + 1.0f, A), cubic1(tx, A), cubic1(1.0f - tx, A), cubic2(2.0f - tx, A)};
        const float* sp = src + (int64_t)pl * h_in * w_in;
        float acc = 0.0f;
        for (int j = 0; j < 4; ++j) {
            int yy = iy - 1 + j;
            yy = yy < 0 ? 0 : (yy > h_in - 1 ? h_in - 1 : yy);
            float row = 0.0f;
            for (int k = 0; k < 4; ++k) {
                int xx = ix - 1 + k;
                xx = xx < 0 ? 0 : (xx > w_in - 1 ? w_in - 1 : xx);
                row += sp[(int64_t)yy * w_in + xx] * wx[k];
            }
            acc += row * wy[j];
        }
        dst[i] = a * acc + b;
    }
}

// ---- the same with PyTorch's ANTIALIASED bicubic (F.interpolate(mode="bicubic", align_corners=False, antialias=True)) ------------
// What torchvision 0.18 (the reference's pinned version, MirrorFusion/README.md:34) computes for transforms.Resize(BICUBIC) on a
// tensor at EVERY scale: antialias defaults to True and _functional_tensor.resize hands it to interpolate unconditionally.
// ATen's _upsample_bicubic2d_aa: separable; per output index i along a dimension of size n_in -> n_out, scale = n_in / n_out,
// support = 2 * max(scale, 1), centre = scale * (i + 0.5); taps [xmin, xmin + xsize) with xmin = max(int(centre - support + 0.5), 0),
// xsize = min(int(centre + support + 0.5), n_in) - xmin; weight_j = cubic_aa((j + xmin - centre + 0.5) / max(scale, 1)) with the Keys
// kernel at a = -0.5 (NOT the -0.75 of the plain kernel), normalised by their sum.  Width first, then height, as ATen does.
__device__ __forceinline__ float cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
    return 0.0f;
}
struct AaTaps { int lo, n; float centre, inv, total; };
__device__ __forceinline__ AaTaps aa_taps(int i, int n_in, float scale) {
    AaTaps t;
    const float support = scale >= 1.0f ? 2.0f * scale : 2.0f;
    t.inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
    t.centre = scale * ((float)i + 0.5f);
    t.lo = (int)(t.centre - support + 0.5f);
    if (t.lo < 0) t.lo = 0;
    int hi = (int)(t.centre + support + 0.5f);
    if (hi > n_in) hi = n_in;
    t.n = hi - t.lo;
    t.total = 0.0f;
    for (int j = 0; j < t.n; ++j) t.total += cubic_aa(((float)(j + t.lo) - t.centre + 0.5f) * t.inv);
    return t;
}
__global__ __launch_bounds__(256) void bicubic_aa_crop_kernel(const float* src, float* dst, int planes, int h_in, int w_in, int h_res, int w_res,
                                                              int oy, int ox, int h_out, int w_out, float a, float b) {
    const float sh = (float)h_in / (float)h_res, sw = (float)w_in / (float)w_res;
    const int64_t total = (int64_t)planes * h_out * w_out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w_out);
        const int64_t t = i / w_out;
        const int y = (int)(t % h_out);
        const int pl = (int)(t / h_out);
        const AaTaps ty = aa_taps(y + oy, h_in, sh), tx = aa_taps(x + ox, w_in, sw);
        const float* sp = src + (int64_t)pl * h_in * w_in;
        float acc = 0.0f;
        for (int j = 0; j < ty.n; ++j) {
            const float* rowp = sp + (int64_t)(ty.lo + j) * w_in + tx.lo;
            float row = 0.0f;                       // the width pass' output pixel (ATen's intermediate image)
            for (int k = 0; k < tx.n; ++k) row += rowp[k] * (cubic_aa(((float)(k + tx.lo) - tx.centre + 0.5f) * tx.inv) / tx.total);
            acc += row * (cubic_aa(((float)(j + ty.lo) - ty.centre + 0.5f) * ty.inv) / ty.total);
        }
        dst[i] = a * acc + b;
    }
}

// y[c][p] = a * x[p][c] + b: HWC -> CHW with an affine map (apply_transforms_normals at the native resolution)
__global__ __launch_bounds__(256) void hwc_to_chw_affine_kernel(const float* x, float* y, int64_t hw, int c, float a, float b) {
    const int64_t total = hw * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % hw;
        const int cc = (int)(i / hw);
        y[i] = a * x[p * c + cc] + b;
    }
}

}  // namespace

extern "C" int64_t mf_minmax_ws_floats(void) { return 2 * MM_BLOCKS; }

extern "C" int mf_minmax(const float* x, const float* mask, int64_t n, float* out2, float* ws, void* stream) {
    MF_CHECK_ARG(x && out2 && ws && n >= 1, "mf_minmax: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = fgrid(n, MM_BLOCKS);
    hipLaunchKernelGGL(minmax_stage1, dim3(blocks), dim3(256), 0, s, x, mask, n, ws);
    MF_CHECK_LAUNCH("mf_minmax");
    hipLaunchKernelGGL(minmax_stage2, dim3(1), dim3(256), 0, s, ws, (int)blocks, out2);
    MF_CHECK_LAUNCH("mf_minmax(stage 2)");
    return MF_OK;
}

extern "C" int mf_image_normalize(const float* x, float* y, int64_t n, const float* minmax, void* stream) {
    MF_CHECK_ARG(x && y && minmax && n >= 1, "mf_image_normalize: bad arguments");
    hipLaunchKernelGGL(image_normalize_kernel, dim3(fgrid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, minmax);
    MF_CHECK_LAUNCH("mf_image_normalize");
    return MF_OK;
}

extern "C" int mf_mask_keep(const float* mask, float* out, int32_t batch, int32_t channels, int64_t hw, void* stream) {
    MF_CHECK_ARG(mask && out && batch >= 1 && channels >= 1 && hw >= 1, "mf_mask_keep: bad arguments");
    hipLaunchKernelGGL(mask_keep_kernel, dim3(fgrid((int64_t)batch * hw)), dim3(256), 0, (hipStream_t)stream, mask, out, batch, channels, hw);
    MF_CHECK_LAUNCH("mf_mask_keep");
    return MF_OK;
}

extern "C" int mf_concat_channels(const float* const* srcs, const int32_t* channels, const int32_t* batches, int32_t nsrc, float* out,
                                  int32_t batch, int64_t hw, void* stream) {
    MF_CHECK_ARG(srcs && channels && batches && out && nsrc >= 1 && nsrc <= 8 && batch >= 1 && hw >= 1, "mf_concat_channels: bad arguments");
    ConcatArgs a{};
    a.nsrc = nsrc;
    for (int i = 0; i < nsrc; ++i) {
        MF_CHECK_ARG(srcs[i] && channels[i] >= 1 && batches[i] >= 1 && batch % batches[i] == 0, "mf_concat_channels: bad source %d", i);
        a.src[i] = srcs[i]; a.ch[i] = channels[i]; a.bstride[i] = batches[i];
        a.ctot += channels[i];
    }
    hipLaunchKernelGGL(concat_channels_kernel, dim3(fgrid((int64_t)batch * a.ctot * hw)), dim3(256), 0, (hipStream_t)stream, a, out, batch, hw);
    MF_CHECK_LAUNCH("mf_concat_channels");
    return MF_OK;
}

extern "C" int mf_postprocess(const float* x, float* out_f32, void* out_u8, int32_t batch, int32_t channels, int64_t hw, int32_t denormalize,
                              void* stream) {
    MF_CHECK_ARG(x && (out_f32 || out_u8) && batch >= 1 && channels >= 1 && hw >= 1, "mf_postprocess: bad arguments");
    hipLaunchKernelGGL(postprocess_kernel, dim3(fgrid((int64_t)batch * channels * hw)), dim3(256), 0, (hipStream_t)stream, x, out_f32,
                       (unsigned char*)out_u8, batch, channels, hw, denormalize);
    MF_CHECK_LAUNCH("mf_postprocess");
    return MF_OK;
}

extern "C" int mf_depth_normalize(const float* depth, const float* mask, float* out, int64_t n, float max_scene_depth, float delta,
                                  int32_t signed_range, float* ws, void* stream) {
    MF_CHECK_ARG(depth && out && ws && n >= 1 && (mask || max_scene_depth > 0.0f), "mf_depth_normalize: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* mm = ws + 2 * MM_BLOCKS;                                    // ws: mf_minmax_ws_floats() + 2 floats
    if (mask) {
        const int rc = mf_minmax(depth, mask, n, mm, ws, stream);
        if (rc != MF_OK) return rc;
    }
    hipLaunchKernelGGL(depth_normalize_kernel, dim3(fgrid(n)), dim3(256), 0, s, depth, out, n, mm, mask != nullptr, max_scene_depth, delta,
                       signed_range);
    MF_CHECK_LAUNCH("mf_depth_normalize");
    return MF_OK;
}


extern "C" int64_t mf_select_ws_bytes(void) { return (int64_t)(65536 * (1 + SEL_MAX)) * 4 + SEL_MAX * (4 + 4 + 8) + 64; }

// the nr (<= 4) order statistics x_(ranks[i]) (0-based, ascending) of x[0..n) -> vals[i]; ranks / vals in device memory
static int select_ranks(const float* x, int64_t n, const int64_t* ranks_dev, int nr, float* vals, void* ws, hipStream_t s) {
    unsigned* hist = (unsigned*)ws;
    unsigned* hist2 = hist + 65536;
    unsigned* bin_hi = hist2 + (int64_t)SEL_MAX * 65536;
    unsigned* bin_lo = bin_hi + SEL_MAX;
    int64_t* rem = (int64_t*)(((uintptr_t)(bin_lo + SEL_MAX) + 7) & ~(uintptr_t)7);
    if (hipMemsetAsync(hist, 0, (size_t)65536 * (1 + SEL_MAX) * 4, s) != hipSuccess) return MF_ELAUNCH;
    hipLaunchKernelGGL(sel_hist_hi_kernel, dim3(fgrid(n)), dim3(256), 0, s, x, n, hist);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(1024), 0, s, hist, 1, ranks_dev, nr, bin_hi, rem);
    hipLaunchKernelGGL(sel_hist_lo_kernel, dim3(fgrid(n)), dim3(256), 0, s, x, n, bin_hi, nr, hist2);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(1024), 0, s, hist2, SEL_MAX, ranks_dev, nr, bin_lo, rem);
    hipLaunchKernelGGL(sel_value_kernel, dim3(1), dim3(64), 0, s, bin_hi, bin_lo, nr, vals);
    return MF_OK;
}

extern "C" int mf_select_ranks(const float* x, int64_t n, const int64_t* ranks, int32_t nr, float* vals, void* ws, void* stream) {
    MF_CHECK_ARG(x && ranks && vals && ws && n >= 1 && nr >= 1 && nr <= SEL_MAX, "mf_select_ranks: bad arguments");
    const int rc = select_ranks(x, n, ranks, nr, vals, ws, (hipStream_t)stream);
    if (rc != MF_OK) { mf_set_error("mf_select_ranks: memset failed"); return rc; }
    MF_CHECK_LAUNCH("mf_select_ranks");
    return MF_OK;
}

extern "C" int mf_depth_percentile_normalize(const float* depth, float* out, int64_t n, const int64_t* ranks4, float t_lo, float t_hi,
                                             int32_t signed_range, float* vals4, void* ws, void* stream) {
    MF_CHECK_ARG(depth && out && ranks4 && vals4 && ws && n >= 2, "mf_depth_percentile_normalize: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int rc = select_ranks(depth, n, ranks4, 4, vals4, ws, s);
    if (rc != MF_OK) { mf_set_error("mf_depth_percentile_normalize: memset failed"); return rc; }
    hipLaunchKernelGGL(depth_percentile_kernel, dim3(fgrid(n)), dim3(256), 0, s, depth, out, n, vals4, t_lo, t_hi, signed_range);
    MF_CHECK_LAUNCH("mf_depth_percentile_normalize");
    return MF_OK;
}

extern "C" int mf_bicubic_resize_crop(const float* src, float* dst, int32_t planes, int32_t h_in, int32_t w_in, int32_t h_res, int32_t w_res,
                                      int32_t crop_top, int32_t crop_left, int32_t h_out, int32_t w_out, float a, float b, void* stream) {
    MF_CHECK_ARG(src && dst && planes >= 1 && h_in >= 1 && w_in >= 1 && h_res >= 1 && w_res >= 1 && h_out >= 1 && w_out >= 1 && crop_top >= 0 &&
                     crop_left >= 0 && crop_top + h_out <= h_res && crop_left + w_out <= w_res,
                 "mf_bicubic_resize_crop: the crop window must lie inside the resized image");
    hipLaunchKernelGGL(bicubic_crop_kernel, dim3(fgrid((int64_t)planes * h_out * w_out)), dim3(256), 0, (hipStream_t)stream, src, dst, planes,
                       h_in, w_in, h_res, w_res, crop_top, crop_left, h_out, w_out, a, b);
    MF_CHECK_LAUNCH("mf_bicubic_resize_crop");
    return MF_OK;
}

extern "C" int mf_bicubic_aa_resize_crop(const float* src, float* dst, int32_t planes, int32_t h_in, int32_t w_in, int32_t h_res, int32_t w_res,
                                         int32_t crop_top, int32_t crop_left, int32_t h_out, int32_t w_out, float a, float b, void* stream) {
    MF_CHECK_ARG(src && dst && planes >= 1 && h_in >= 1 && w_in >= 1 && h_res >= 1 && w_res >= 1 && h_out >= 1 && w_out >= 1 && crop_top >= 0 &&
                     crop_left >= 0 && crop_top + h_out <= h_res && crop_left + w_out <= w_res,
                 "mf_bicubic_aa_resize_crop: the crop window must lie inside the resized image");
    hipLaunchKernelGGL(bicubic_aa_crop_kernel, dim3(fgrid((int64_t)planes * h_out * w_out)), dim3(256), 0, (hipStream_t)stream, src, dst, planes,
                       h_in, w_in, h_res, w_res, crop_top, crop_left, h_out, w_out, a, b);
    MF_CHECK_LAUNCH("mf_bicubic_aa_resize_crop");
    return MF_OK;
}

extern "C" int mf_hwc_to_chw_affine(const float* x, float* y, int64_t hw, int32_t channels, float a, float b, void* stream) {
    MF_CHECK_ARG(x && y && hw >= 1 && channels >= 1, "mf_hwc_to_chw_affine: bad arguments");
    hipLaunchKernelGGL(hwc_to_chw_affine_kernel, dim3(fgrid(hw * channels)), dim3(256), 0, (hipStream_t)stream, x, y, hw, channels, a, b);
    MF_CHECK_LAUNCH("mf_hwc_to_chw_affine");
    return MF_OK;
}
