// Dynamic per-row fp8 (OCP e4m3) quantisation of activations, optionally fused behind a LayerNorm, for the fp8 GEMMs of
// the SDXL transformer blocks (BASELINE.json configs[4]).  The reference has no fp8 path: this is the
// MI355X-side precision choice for the layers whose arithmetic is attention.py:203-412 (LayerNorm -> Linear).
// One half-wave per row, 8-channel (16-byte) vectors, the row stays in registers between the passes: one read of x,
// one 1-byte-per-element write.  scale[r] = max|y| / 448 (448 = largest e4m3 value), q = fp8(y / scale).
#include "mf_common.h"

namespace {

__device__ __forceinline__ void load8f(const char* p, int dt, int64_t idx, float* o) {
    if (dt == MF_F32) {
        const float4 a = *reinterpret_cast<const float4*>(p + idx * 4);
        const float4 b = *reinterpret_cast<const float4*>(p + idx * 4 + 16);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(p + idx * 2);
        o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
        o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
        o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
        o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
    }
}

// LPR lanes per row (32: C <= 2048, two rows per wave; 64: C <= 8192, GEGLU outputs of the 1280-wide SDXL blocks)
template <bool LN, int LPR, int MAXV>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const char* x, int in_dt, unsigned char* out, float* scale,
                                                             const float* gamma, const float* beta, int64_t rows, int C, float eps) {
    const int l32 = threadIdx.x & (LPR - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + (threadIdx.x / LPR);
    const bool live = row < rows;                       // keep every lane alive for the shuffles
    float v[MAXV][8];
    const int c8n = C >> 3;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c8 = l32 + LPR * j;
        if (live && c8 < c8n) {
            load8f(x, in_dt, row * C + c8 * 8, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.0f;
        }
    }
    if constexpr (LN) {
#pragma unroll
        for (int off = LPR / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, LPR);
        const float mean = s / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (l32 + LPR * j < c8n) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; q += d * d; }
            }
#pragma unroll
        for (int off = LPR / 2; off >= 1; off >>= 1) q += __shfl_xor(q, off, LPR);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c8 = l32 + LPR * j;
            if (c8 < c8n) {
                float g[8], bb[8];
                load8f(reinterpret_cast<const char*>(gamma), MF_F32, c8 * 8, g);
                load8f(reinterpret_cast<const char*>(beta), MF_F32, c8 * 8, bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = (v[j][e] - mean) * rstd * g[e] + bb[e];
            }
        }
    }
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[j][e]));
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, LPR));
    if (!live) return;
    const float sc = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (l32 == 0) scale[row] = sc;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c8 = l32 + LPR * j;
        if (c8 < c8n) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = fminf(fmaxf(v[j][e] * inv, -448.0f), 448.0f);
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], hi, true);
            *reinterpret_cast<uint2*>(out + row * C + c8 * 8) = uint2{(unsigned)lo, (unsigned)hi};
        }
    }
}

}  // namespace

extern "C" int mf_quantize_rows_fp8(const void* x, int32_t in_dtype, void* out_q, float* scale, int64_t rows, int32_t c,
                                    const float* gamma, const float* beta, float eps, void* stream) {
    MF_CHECK_ARG(x && out_q && scale && rows >= 1 && c >= 8 && c % 8 == 0 && c <= 8192, "mf_quantize_rows_fp8: c must be a multiple of 8, <= 8192");
    MF_CHECK_ARG(in_dtype == MF_F32 || in_dtype == MF_BF16, "mf_quantize_rows_fp8: input must be fp32 or bf16");
    MF_CHECK_ARG((gamma != nullptr) == (beta != nullptr), "mf_quantize_rows_fp8: gamma and beta go together");
    if (!mf_aligned16(x) || (((uintptr_t)out_q) & 7)) {
        mf_set_error("mf_quantize_rows_fp8: x must be 16-byte and out_q 8-byte aligned");
        return MF_EALIGN;
    }
    hipStream_t s = (hipStream_t)stream;
    const char* xp = (const char*)x;
    unsigned char* qp = (unsigned char*)out_q;
    if (c <= 2048) {
        const dim3 grid((unsigned)((rows + 7) / 8));
        if (gamma) hipLaunchKernelGGL((quant_rows_fp8_kernel<true, 32, 8>), grid, dim3(256), 0, s, xp, in_dtype, qp, scale, gamma, beta, rows, c, eps);
        else hipLaunchKernelGGL((quant_rows_fp8_kernel<false, 32, 8>), grid, dim3(256), 0, s, xp, in_dtype, qp, scale, gamma, beta, rows, c, eps);
    } else {
        const dim3 grid((unsigned)((rows + 3) / 4));
        if (gamma) hipLaunchKernelGGL((quant_rows_fp8_kernel<true, 64, 16>), grid, dim3(256), 0, s, xp, in_dtype, qp, scale, gamma, beta, rows, c, eps);
        else hipLaunchKernelGGL((quant_rows_fp8_kernel<false, 64, 16>), grid, dim3(256), 0, s, xp, in_dtype, qp, scale, gamma, beta, rows, c, eps);
    }
    MF_CHECK_LAUNCH("mf_quantize_rows_fp8");
    return MF_OK;
}
