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
