// Small HBM-bound kernels of the denoise loop: layout packing, injection add, GEGLU, timestep
// embedding, CFG + scheduler updates, VAE posterior sampling, nearest resize; plus the library's
// error-string plumbing.
#include <stdarg.h>
#include <string.h>
#include "mf_common.h"

static thread_local char g_err[512] = "";
void mf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mf_last_error(void) { return g_err; }
extern "C" int mf_abi_version(void) { return MF_ABI_VERSION; }
extern "C" int mf_sizeof_gemm_desc(void) { return (int)sizeof(mf_gemm_desc); }
extern "C" int mf_sizeof_groupnorm_desc(void) { return (int)sizeof(mf_groupnorm_desc); }

extern "C" int mf_split_overflow(int32_t reset, int32_t* raised, void* stream) {
    MF_CHECK_ARG(raised != nullptr, "mf_split_overflow: null pointer");
    unsigned* flags[3] = {mf_ovf_flag_gemm(), mf_ovf_flag_attention(), mf_ovf_flag_train()};
    hipStream_t s = (hipStream_t)stream;
    unsigned host[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        MF_CHECK_ARG(flags[i] != nullptr, "mf_split_overflow: flag symbol not found");
        if (hipMemcpyAsync(&host[i], flags[i], sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess) {
            mf_set_error("mf_split_overflow: copy failed");
            return MF_ELAUNCH;
        }
    }
    if (hipStreamSynchronize(s) != hipSuccess) {
        mf_set_error("mf_split_overflow: synchronise failed");
        return MF_ELAUNCH;
    }
    *raised = (int32_t)((host[0] ? 1 : 0) | (host[1] ? 2 : 0) | (host[2] ? 4 : 0));
    if (reset && *raised)
        for (int i = 0; i < 3; ++i) (void)hipMemsetAsync(flags[i], 0, sizeof(unsigned), s);
    return MF_OK;
}

namespace {

inline unsigned grid_for(int64_t n, int per_block = 256, int cap = 8192) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

__global__ void pack_nhwc_kernel(const float* s0, int c0, const float* s1, int c1, char* dst, int dst_dt, int c_pad,
                                 int batch, int hw) {
    const int64_t total = (int64_t)batch * hw * c_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c_pad);
        const int64_t bp = i / c_pad;
        const int p = (int)(bp % hw);
        const int b = (int)(bp / hw);
        float v = 0.0f;
        if (c < c0) v = s0[((int64_t)b * c0 + c) * hw + p];
        else if (c < c0 + c1) v = s1[((int64_t)b * c1 + (c - c0)) * hw + p];
        store_from_f32(dst, dst_dt, i, v);
    }
}

__global__ void unpack_nchw_kernel(const char* src, int src_dt, int64_t ld, float* dst, int c, int batch, int hw) {
    const int64_t total = (int64_t)batch * c * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const int64_t bc = i / hw;
        const int ch = (int)(bc % c);
        const int b = (int)(bc / c);
        dst[i] = load_as_f32(src, src_dt, ((int64_t)b * hw + p) * ld + ch);
    }
}

__global__ void add_kernel(const char* a, int adt, const char* b, int bdt, char* o, int odt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        store_from_f32(o, odt, i, load_as_f32(a, adt, i) + load_as_f32(b, bdt, i));
}

// all three tensors bf16, 16-byte aligned, n % 8 == 0: 8 elements per thread
__global__ void add_bf16x8_kernel(const uint4* a, const uint4* b, uint4* o, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        auto add2 = [](unsigned p, unsigned q) {
            return pack_bf16x2(__uint_as_float(p << 16) + __uint_as_float(q << 16),
                               __uint_as_float(p & 0xffff0000u) + __uint_as_float(q & 0xffff0000u));
        };
        uint4 r;
        r.x = add2(x.x, y.x); r.y = add2(x.y, y.y); r.z = add2(x.z, y.z); r.w = add2(x.w, y.w);
        o[i] = r;
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ void geglu_kernel(const char* h, int in_dt, char* out, int out_dt, int64_t rows, int c) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int j = (int)(i - r * c);
        const float v = load_as_f32(h, in_dt, r * 2 * c + j);
        const float g = load_as_f32(h, in_dt, r * 2 * c + c + j);
        store_from_f32(out, out_dt, i, v * gelu_erf(g));
    }
}

// bf16 [rows][2c] -> bf16 [rows][c], 8 channels per thread (16-byte loads and stores); c % 8 == 0
__global__ __launch_bounds__(256) void geglu16_kernel(const unsigned short* __restrict__ h, unsigned short* __restrict__ out, int64_t rows, int c8) {
    const int64_t total = rows * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c8;
        const int j = (int)(i - r * c8);
        const uint4 a = *reinterpret_cast<const uint4*>(h + (r * 2 * c8 + j) * 8);
        const uint4 g = *reinterpret_cast<const uint4*>(h + (r * 2 * c8 + c8 + j) * 8);
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = __uint_as_float(aw[e] << 16), a1 = __uint_as_float(aw[e] & 0xffff0000u);
            const float g0 = __uint_as_float(gw[e] << 16), g1 = __uint_as_float(gw[e] & 0xffff0000u);
            o[e] = pack_bf16x2(a0 * gelu_erf(g0), a1 * gelu_erf(g1));
        }
        *reinterpret_cast<uint4*>(out + i * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}

// embeddings.py:27-67 (scale = 1, max_period = 10000)
__global__ void timestep_embedding_kernel(const float* t, float* out, int n, int dim, int flip, float shift) {
    const int half = dim / 2;
    const int total = n * half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / half, k = i - r * half;
        float e = -9.210340371976184f * (float)k;   // -ln(10000) * k  (fp32, as torch)
        e = e / ((float)half - shift);
        const float arg = t[r] * expf(e);
        const float sn = sinf(arg), cs = cosf(arg);
        float* o = out + (int64_t)r * dim;
        if (flip) { o[k] = cs; o[half + k] = sn; }
        else { o[k] = sn; o[half + k] = cs; }
        if ((dim & 1) && k == 0) o[dim - 1] = 0.0f;
    }
}

__global__ void silu_f32_kernel(const float* x, float* o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = silu_precise(x[i]);
}

__global__ void cfg_ddim_kernel(const float* eu, const float* ec, float g, const float* x, float* xp, float sqrt_at,
                                float sqrt_1m_at, float sqrt_ap, float dir_coef, int pred_type, float clip,
                                float* eps_out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e = eu[i];
        if (g >= 0.0f) e = e + g * (ec[i] - e);
        if (eps_out) eps_out[i] = e;
        float x0, ep;
        if (pred_type == 0) {
            x0 = (x[i] - sqrt_1m_at * e) / sqrt_at;              // scheduling_ddim.py:412
            ep = e;
        } else {
            x0 = sqrt_at * x[i] - sqrt_1m_at * e;                // :418-420
            ep = sqrt_at * e + sqrt_1m_at * x[i];
        }
        if (clip > 0.0f) x0 = fminf(fmaxf(x0, -clip), clip);     // :427-430
        const float dir = dir_coef * ep;                          // :443
        xp[i] = sqrt_ap * x0 + dir;                               // :446
    }
}

__global__ void cfg_ddim_dev_kernel(const float* eu, const float* ec, float g, const float* x, float* xp,
                                    const float* coef, int pred_type, float clip, int64_t n) {
    const float sqrt_at = coef[0], sqrt_1m_at = coef[1], sqrt_ap = coef[2], dir_coef = coef[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e = eu[i];
        if (g >= 0.0f) e = e + g * (ec[i] - e);
        float x0, ep;
        const float xi = x[i];
        if (pred_type == 0) { x0 = (xi - sqrt_1m_at * e) / sqrt_at; ep = e; }
        else { x0 = sqrt_at * xi - sqrt_1m_at * e; ep = sqrt_at * e + sqrt_1m_at * xi; }
        if (clip > 0.0f) x0 = fminf(fmaxf(x0, -clip), clip);
        const float dir = dir_coef * ep;
        xp[i] = sqrt_ap * x0 + dir;
    }
}

__global__ void cfg_combine_kernel(const float* eu, const float* ec, float g, float* eps, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float u = eu[i];
        eps[i] = u + g * (ec[i] - u);
    }
}

struct AxpbyArgs { const float* x[6]; float c[6]; int nin; };
__global__ void axpby_kernel(AxpbyArgs a, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = a.c[0] * a.x[0][i];
        for (int k = 1; k < a.nin; ++k) v += a.c[k] * a.x[k][i];
        y[i] = v;
    }
}

// Training loss (train_brushnet_mirror.py:1433-1449): per-sample mean((pred - target)^2) * weight, then the
// mean over samples.  One block per sample, fp32 products accumulated in double, fixed reduction order.
__global__ void mse_rows_kernel(const float* pred, const float* target, const float* weights, float* per_row,
                                int64_t n) {
    __shared__ double part[256];
    const float* p = pred + (int64_t)blockIdx.x * n;
    const float* t = target + (int64_t)blockIdx.x * n;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = p[i] - t[i];
        acc += (double)(d * d);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float m = (float)(part[0] / (double)n);
        per_row[blockIdx.x] = weights ? m * weights[blockIdx.x] : m;
    }
}

__global__ void mean_rows_kernel(const float* per_row, float* loss, int rows) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < rows; ++i) acc += (double)per_row[i];
        loss[0] = (float)(acc / (double)rows);
    }
}

__global__ void vae_sample_kernel(const char* mom, int dt, int64_t ld, const float* noise, float* z, int c, int batch,
                                  int hw, float scaling) {
    const int64_t total = (int64_t)batch * c * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const int64_t bc = i / hw;
        const int ch = (int)(bc % c);
        const int b = (int)(bc / c);
        const int64_t row = ((int64_t)b * hw + p) * ld;
        const float mean = load_as_f32(mom, dt, row + ch);
        float logvar = load_as_f32(mom, dt, row + c + ch);
        logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);            // vae.py:776
        const float std = expf(0.5f * logvar);
        z[i] = (mean + std * noise[i]) * scaling;
    }
}

__global__ void nearest_resize_kernel(const float* src, float* dst, int planes, int hi, int wi, int ho, int wo) {
    const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
    const int64_t total = (int64_t)planes * ho * wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % wo);
        const int64_t py = i / wo;
        const int y = (int)(py % ho);
        const int pl = (int)(py / ho);
        int iy = (int)floorf((float)y * sy), ix = (int)floorf((float)x * sx);
        if (iy > hi - 1) iy = hi - 1;
        if (ix > wi - 1) ix = wi - 1;
        dst[i] = src[((int64_t)pl * hi + iy) * wi + ix];
    }
}

}  // namespace

extern "C" int mf_pack_nhwc(const float* src0, int32_t c0, const float* src1, int32_t c1, void* dst, int32_t dst_dtype,
                            int32_t c_pad, int32_t batch, int32_t hw, void* stream) {
    MF_CHECK_ARG(src0 && dst && c0 > 0 && c1 >= 0 && (src1 != nullptr) == (c1 > 0) && c_pad >= c0 + c1,
                 "mf_pack_nhwc: bad arguments");
    const int64_t total = (int64_t)batch * hw * c_pad;
    hipLaunchKernelGGL(pack_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src0, c0, src1, c1,
                       (char*)dst, dst_dtype, c_pad, batch, hw);
    MF_CHECK_LAUNCH("mf_pack_nhwc");
    return MF_OK;
}

extern "C" int mf_unpack_nchw(const void* src, int32_t src_dtype, int64_t ld, float* dst, int32_t c, int32_t batch,
                              int32_t hw, void* stream) {
    MF_CHECK_ARG(src && dst && c > 0 && ld >= c, "mf_unpack_nchw: bad arguments");
    const int64_t total = (int64_t)batch * hw * c;
    hipLaunchKernelGGL(unpack_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       src_dtype, ld, dst, c, batch, hw);
    MF_CHECK_LAUNCH("mf_unpack_nchw");
    return MF_OK;
}

extern "C" int mf_add(const void* a, int32_t a_dtype, const void* b, int32_t b_dtype, void* out, int32_t out_dtype,
                      int64_t n, void* stream) {
    MF_CHECK_ARG(a && b && out && n >= 0, "mf_add: bad arguments");
    if (n == 0) return MF_OK;
    if (a_dtype == MF_BF16 && b_dtype == MF_BF16 && out_dtype == MF_BF16 && n % 8 == 0 && mf_aligned16(a) && mf_aligned16(b) &&
        mf_aligned16(out))
        hipLaunchKernelGGL(add_bf16x8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a,
                           (const uint4*)b, (uint4*)out, n / 8);
    else
        hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const char*)a, a_dtype,
                           (const char*)b, b_dtype, (char*)out, out_dtype, n);
    MF_CHECK_LAUNCH("mf_add");
    return MF_OK;
}

// fp32 -> bf16 (nearest-even), 8 elements per thread; the scalar tail by the first threads
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, int64_t n8, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + i * 8);
        const float4 b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
        *reinterpret_cast<uint4*>(out + i * 8) = uint4{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
    }
    const int64_t t = n8 * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = f32_to_bf16(x[t]);
}

extern "C" int mf_cast_bf16(const float* x, void* out, int64_t n, void* stream) {
    MF_CHECK_ARG(x && out && n >= 0 && mf_aligned16(x) && mf_aligned16(out), "mf_cast_bf16: null or misaligned arguments");
    if (n == 0) return MF_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)out, n / 8, n);
    MF_CHECK_LAUNCH("mf_cast_bf16");
    return MF_OK;
}

extern "C" int mf_geglu(const void* h, int32_t in_dtype, void* out, int32_t out_dtype, int64_t rows, int32_t c,
                        void* stream) {
    MF_CHECK_ARG(h && out && rows >= 0 && c > 0, "mf_geglu: bad arguments");
    if (rows == 0) return MF_OK;
    if (in_dtype == MF_BF16 && out_dtype == MF_BF16 && c % 8 == 0 && mf_aligned16(h) && mf_aligned16(out)) {
        hipLaunchKernelGGL(geglu16_kernel, dim3(grid_for(rows * (c / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)h,
                           (unsigned short*)out, rows, c / 8);
        MF_CHECK_LAUNCH("mf_geglu");
        return MF_OK;
    }
    hipLaunchKernelGGL(geglu_kernel, dim3(grid_for(rows * c)), dim3(256), 0, (hipStream_t)stream, (const char*)h,
                       in_dtype, (char*)out, out_dtype, rows, c);
    MF_CHECK_LAUNCH("mf_geglu");
    return MF_OK;
}

extern "C" int mf_timestep_embedding(const float* t, float* out, int32_t n, int32_t dim, int32_t flip_sin_to_cos,
                                     float freq_shift, void* stream) {
    MF_CHECK_ARG(t && out && n > 0 && dim >= 2, "mf_timestep_embedding: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((int64_t)n * (dim / 2))), dim3(256), 0,
                       (hipStream_t)stream, t, out, n, dim, flip_sin_to_cos, freq_shift);
    MF_CHECK_LAUNCH("mf_timestep_embedding");
    return MF_OK;
}

extern "C" int mf_silu_f32(const float* x, float* out, int64_t n, void* stream) {
    MF_CHECK_ARG(x && out && n >= 0, "mf_silu_f32: bad arguments");
    if (n == 0) return MF_OK;
    hipLaunchKernelGGL(silu_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    MF_CHECK_LAUNCH("mf_silu_f32");
    return MF_OK;
}

extern "C" int mf_cfg_ddim_step(const float* eps_u, const float* eps_c, float g, const float* x, float* x_prev,
                                float sqrt_at, float sqrt_1m_at, float sqrt_ap, float dir_coef, int32_t pred_type,
                                float clip, float* eps_out, int64_t n, void* stream) {
    MF_CHECK_ARG(eps_u && x && x_prev && n > 0 && (g < 0.0f || eps_c), "mf_cfg_ddim_step: bad arguments");
    MF_CHECK_ARG(pred_type == 0 || pred_type == 1, "mf_cfg_ddim_step: pred_type must be 0 (epsilon) or 1 (v_prediction)");
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, eps_u, eps_c, g, x,
                       x_prev, sqrt_at, sqrt_1m_at, sqrt_ap, dir_coef, pred_type, clip, eps_out, n);
    MF_CHECK_LAUNCH("mf_cfg_ddim_step");
    return MF_OK;
}

extern "C" int mf_cfg_ddim_step_dev(const float* eps_u, const float* eps_c, float g, const float* x, float* x_prev,
                                    const float* coef4, int32_t pred_type, float clip, int64_t n, void* stream) {
    MF_CHECK_ARG(eps_u && x && x_prev && coef4 && n > 0 && (g < 0.0f || eps_c), "mf_cfg_ddim_step_dev: bad arguments");
    MF_CHECK_ARG(pred_type == 0 || pred_type == 1, "mf_cfg_ddim_step_dev: pred_type must be 0 or 1");
    hipLaunchKernelGGL(cfg_ddim_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, eps_u, eps_c, g, x,
                       x_prev, coef4, pred_type, clip, n);
    MF_CHECK_LAUNCH("mf_cfg_ddim_step_dev");
    return MF_OK;
}

extern "C" int mf_cfg_combine(const float* eps_u, const float* eps_c, float g, float* eps, int64_t n, void* stream) {
    MF_CHECK_ARG(eps_u && eps_c && eps && n > 0, "mf_cfg_combine: bad arguments");
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, eps_u, eps_c, g, eps,
                       n);
    MF_CHECK_LAUNCH("mf_cfg_combine");
    return MF_OK;
}

extern "C" int mf_axpby_n(const float* const* xs, const float* coefs, int32_t nin, float* y, int64_t n, void* stream) {
    MF_CHECK_ARG(xs && coefs && y && nin >= 1 && nin <= 6 && n > 0, "mf_axpby_n: bad arguments");
    AxpbyArgs a{};
    a.nin = nin;
    for (int i = 0; i < nin; ++i) { a.x[i] = xs[i]; a.c[i] = coefs[i]; }
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, y, n);
    MF_CHECK_LAUNCH("mf_axpby_n");
    return MF_OK;
}

extern "C" int mf_mse_loss(const float* pred, const float* target, const float* weights, float* per_sample,
                           float* loss, int32_t rows, int64_t n, void* stream) {
    MF_CHECK_ARG(pred && target && per_sample && loss && rows > 0 && n > 0, "mf_mse_loss: bad arguments");
    hipLaunchKernelGGL(mse_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, weights,
                       per_sample, n);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, per_sample, loss, rows);
    MF_CHECK_LAUNCH("mf_mse_loss");
    return MF_OK;
}

extern "C" int mf_vae_sample(const void* moments, int32_t m_dtype, int64_t ld, const float* noise, float* z,
                             int32_t c, int32_t batch, int32_t hw, float scaling, void* stream) {
    MF_CHECK_ARG(moments && noise && z && c > 0 && ld >= 2 * c, "mf_vae_sample: bad arguments");
    hipLaunchKernelGGL(vae_sample_kernel, dim3(grid_for((int64_t)batch * c * hw)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)moments, m_dtype, ld, noise, z, c, batch, hw, scaling);
    MF_CHECK_LAUNCH("mf_vae_sample");
    return MF_OK;
}

extern "C" int mf_nearest_resize(const float* src, float* dst, int32_t planes, int32_t h_in, int32_t w_in,
                                 int32_t h_out, int32_t w_out, void* stream) {
    MF_CHECK_ARG(src && dst && planes > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0,
                 "mf_nearest_resize: bad arguments");
    hipLaunchKernelGGL(nearest_resize_kernel, dim3(grid_for((int64_t)planes * h_out * w_out)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, planes, h_in, w_in, h_out, w_out);
    MF_CHECK_LAUNCH("mf_nearest_resize");
    return MF_OK;
}
