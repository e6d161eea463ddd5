// GroupNorm(+SiLU), LayerNorm and row softmax for NHWC / token-major activations on gfx950.
// All three are HBM-bound: one read + one write of the tensor (GroupNorm reads it twice: stats,
// then apply), fp32 statistics, vectorised 4-channel accesses, no atomics (bitwise reproducible).
#include "mf_common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 64;

struct GnArgs {
    const char* x0; const char* x1;
    int C0, C1, C, in_dt, HW, G, cpg, gslices, rows_per_chunk, nchunks;
    float eps;
    const float* gamma; const float* beta;
    int silu;
    char* out; int out_dt;
    float* ws;   // [batch][G][nchunks][2] = (mean, M2) of each chunk
    float* ws_ab;   // [batch][2][C] per-channel scale / shift
};

__device__ __forceinline__ float4 load4(const char* p, int dt, int64_t idx) {
    if (dt == MF_F32) return *reinterpret_cast<const float4*>(p + idx * 4);
    const uint2 u = *reinterpret_cast<const uint2*>(p + idx * 2);
    float4 r;
    r.x = __uint_as_float(u.x << 16);
    r.y = __uint_as_float(u.x & 0xffff0000u);
    r.z = __uint_as_float(u.y << 16);
    r.w = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void store4(char* p, int dt, int64_t idx, float4 v) {
    if (dt == MF_F32) {
        *reinterpret_cast<float4*>(p + idx * 4) = v;
    } else {
        uint2 u;
        u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2*>(p + idx * 2) = u;
    }
}

__device__ __forceinline__ void load8(const char* p, int dt, int64_t idx, float* o) {
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
__device__ __forceinline__ void store8(char* p, int dt, int64_t idx, const float* v) {
    if (dt == MF_F32) {
        *reinterpret_cast<float4*>(p + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + idx * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        uint4 u;
        u.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        u.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        u.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
        u.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
        *reinterpret_cast<uint4*>(p + idx * 2) = u;
    }
}

// grid (nchunks, gslices, batch), 256 threads = 4 waves; wave w takes rows r0+w, r0+w+4, ...;
// lanes take 4-channel column vectors of the slice.  Per-(wave, channel) sums go to LDS, then a
// fixed-order tree reduces them to one (mean, M2) per group of the slice.
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* chan = reinterpret_cast<float*>(smem_raw);   // [4 waves][slice_c][2]
    const int chunk = blockIdx.x, gs = blockIdx.y, b = blockIdx.z;
    const int gps = p.G / p.gslices;                 // groups per slice
    const int slice_c = gps * p.cpg;                 // channels per slice
    const int cs = gs * slice_c;
    const int c4n = slice_c >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = chunk * p.rows_per_chunk;
    int r1 = r0 + p.rows_per_chunk;
    if (r1 > p.HW) r1 = p.HW;

    if (p.C0 % 8 == 0 && p.C1 % 8 == 0 && slice_c % 8 == 0) {
        const int c8n = slice_c >> 3;
        for (int c8 = lane; c8 < c8n; c8 += 64) {
            const int c = cs + c8 * 8;
            const char* base; int64_t ld; int cc;
            if (c < p.C0) { base = p.x0; ld = p.C0; cc = c; }
            else { base = p.x1; ld = p.C1; cc = c - p.C0; }
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = r0 + w; r < r1; r += 4) {
                float v[8];
                load8(base, p.in_dt, ((int64_t)b * p.HW + r) * ld + cc, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                chan[((w * slice_c) + c8 * 8 + e) * 2 + 0] = s[e];
                chan[((w * slice_c) + c8 * 8 + e) * 2 + 1] = ss[e];
            }
        }
    } else
    for (int c4 = lane; c4 < c4n; c4 += 64) {
        const int c = cs + c4 * 4;
        const char* base; int64_t ld; int cc;
        if (c < p.C0) { base = p.x0; ld = p.C0; cc = c; }
        else { base = p.x1; ld = p.C1; cc = c - p.C0; }
        float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        for (int r = r0 + w; r < r1; r += 4) {
            const float4 v = load4(base, p.in_dt, ((int64_t)b * p.HW + r) * ld + cc);
            s[0] += v.x; ss[0] += v.x * v.x;
            s[1] += v.y; ss[1] += v.y * v.y;
            s[2] += v.z; ss[2] += v.z * v.z;
            s[3] += v.w; ss[3] += v.w * v.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            chan[((w * slice_c) + c4 * 4 + e) * 2 + 0] = s[e];
            chan[((w * slice_c) + c4 * 4 + e) * 2 + 1] = ss[e];
        }
    }
    __syncthreads();
    // 32 threads per group: thread (g, l) sums items l, l+32, ... of the 4*cpg (wave, channel) list
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    for (int gg = g; gg < gps; gg += 8) {
        double s = 0.0, ss = 0.0;
        const int items = 4 * p.cpg;
        for (int it = l; it < items; it += 32) {
            const int ww = it / p.cpg, ch = gg * p.cpg + (it - ww * p.cpg);
            s += (double)chan[(ww * slice_c + ch) * 2 + 0];
            ss += (double)chan[(ww * slice_c + ch) * 2 + 1];
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            s += __shfl_xor(s, off, 32);
            ss += __shfl_xor(ss, off, 32);
        }
        if (l == 0) {
            const double n = (double)(r1 - r0) * p.cpg;
            const double mean = s / n;
            double m2 = ss - s * mean;
            if (m2 < 0.0) m2 = 0.0;
            float* o = p.ws + (((int64_t)b * p.G + gs * gps + gg) * p.nchunks + chunk) * 2;
            o[0] = (float)mean;
            o[1] = (float)m2;
        }
    }
}

// grid (nblocks, batch): combine the chunk statistics (Chan et al., in double), build per-channel
// scale/shift in LDS, then stream rows: y = silu(x*a[c] + b[c]).  16-byte accesses (8 channels) when both
// segment widths are multiples of 8, else 4 channels; 32-bit index arithmetic only.
// grid (nblocks, batch).  Prologue (fully parallel, every load issued before the first use): 8 lanes per
// group Chan-combine that group's chunk statistics with a fixed-order butterfly, then every thread turns
// (mean, rstd, gamma, beta) into the per-channel affine y = x*a + b kept in LDS.  Body: stream rows with 16-byte
// accesses (8 channels) when both segment widths are multiples of 8, else 4 channels.
template <int VW>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs p, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sa = reinterpret_cast<float*>(smem_raw);   // [C] scale
    float* sb = sa + p.C;                              // [C] shift
    float* gm = sb + p.C;                              // [G] mean
    float* gr = gm + p.G;                              // [G] rstd
    const int b = blockIdx.y;
    for (int g0 = 0; g0 < p.G; g0 += 32) {
        const int g = g0 + (threadIdx.x >> 3), j = threadIdx.x & 7;
        float pm[8], pq[8];
        int nk = 0;
        if (g < p.G) {
            const float2* st = reinterpret_cast<const float2*>(p.ws + ((int64_t)b * p.G + g) * p.nchunks * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = j + 8 * u;
                if (k < p.nchunks) { const float2 t = st[k]; pm[u] = t.x; pq[u] = t.y; nk = u + 1; }
            }
        }
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int u = 0; u < nk; ++u) {
            const int k = j + 8 * u;
            int rows = p.rows_per_chunk;
            if ((k + 1) * p.rows_per_chunk > p.HW) rows = p.HW - k * p.rows_per_chunk;
            const double nb = (double)rows * p.cpg, nt = n + nb, delta = (double)pm[u] - mean;
            mean += delta * nb / nt;
            m2 += (double)pq[u] + delta * delta * n * nb / nt;
            n = nt;
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {          // fixed-order butterfly over the 8 lanes of a group
            const double n2 = __shfl_xor(n, off, 8), mean2 = __shfl_xor(mean, off, 8), m22 = __shfl_xor(m2, off, 8);
            const double nt = n + n2;
            if (nt > 0.0) {
                const double delta = mean2 - mean;
                const double mnew = (mean * n + mean2 * n2) / nt;      // symmetric: both partners agree bit for bit
                m2 = m2 + m22 + delta * delta * n * n2 / nt;
                mean = mnew;
                n = nt;
            }
        }
        if (g < p.G && j == 0) {
            gm[g] = (float)mean;
            gr[g] = (float)(1.0 / sqrt(m2 / n + (double)p.eps));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        const int g = c / p.cpg;
        const float a = gr[g] * p.gamma[c];
        sa[c] = a;
        sb[c] = p.beta[c] - gm[g] * a;
    }
    __syncthreads();
    const unsigned cvn = (unsigned)p.C / VW;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > p.HW) r1 = p.HW;
    const unsigned items = (unsigned)(r1 - r0) * cvn;
    const bool fast_silu = p.out_dt == MF_BF16;         // bf16 output: __expf is far inside the rounding
    for (unsigned it = threadIdx.x; it < items; it += 256) {
        const unsigned rr = it / cvn;
        const int c = (int)(it - rr * cvn) * VW;
        const char* base; int64_t ld; int cc;
        if (c < p.C0) { base = p.x0; ld = p.C0; cc = c; }
        else { base = p.x1; ld = p.C1; cc = c - p.C0; }
        const int64_t row = (int64_t)b * p.HW + r0 + (int)rr;
        float v[8];
        if (VW == 8) {
            load8(base, p.in_dt, row * ld + cc, v);
        } else {
            const float4 t = load4(base, p.in_dt, row * ld + cc);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            float y = v[j] * sa[c + j] + sb[c + j];
            if (p.silu) y = fast_silu ? silu_f(y) : silu_precise(y);
            v[j] = y;
        }
        if (VW == 8) store8(p.out, p.out_dt, row * p.C + c, v);
        else store4(p.out, p.out_dt, row * p.C + c, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// One wave per row, up to 8 x 256 channels.
__global__ __launch_bounds__(256) void layernorm_kernel(const char* x, int in_dt, char* out, int out_dt,
                                                        const float* gamma, const float* beta, int64_t rows, int C,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int MAXV = 8;
    float4 v[MAXV];
    const int c4n = C >> 2;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < c4n) {
            v[j] = load4(x, in_dt, row * C + c4 * 4);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        } else {
            v[j] = make_float4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < c4n) {
            const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < c4n) {
            const int c = c4 * 4;
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 bb = *reinterpret_cast<const float4*>(beta + c);
            float4 y;
            y.x = (v[j].x - mean) * rstd * g.x + bb.x;
            y.y = (v[j].y - mean) * rstd * g.y + bb.y;
            y.z = (v[j].z - mean) * rstd * g.z + bb.z;
            y.w = (v[j].w - mean) * rstd * g.w + bb.w;
            store4(out, out_dt, row * C + c, y);
        }
    }
}

// One wave per row: max, sum(exp), normalise; pad columns [cols, ld) are zeroed.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, char* out, int out_dt, int64_t rows,
                                                           int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = s + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, r[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.0f;
    for (int c = lane; c < cols; c += 64) sum += expf(r[c] - mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = 1.0f / sum;
    for (int c = lane; c < ld; c += 64) {
        const float pv = c < cols ? expf(r[c] - mx) * inv : 0.0f;
        store_from_f32(out, out_dt, row * ld + c, pv);
    }
}

}  // namespace

extern "C" int64_t mf_groupnorm_ws_floats(int32_t batch, int32_t groups, int32_t channels) {
    return (int64_t)batch * groups * GN_MAX_CHUNKS * 2 + (int64_t)batch * channels * 2;
}

extern "C" int mf_groupnorm(const mf_groupnorm_desc* d, void* stream) {
    MF_CHECK_ARG(d && d->x0 && d->out && d->gamma && d->beta && d->ws, "mf_groupnorm: null pointer");
    MF_CHECK_ARG((d->x1 != nullptr) == (d->c1 > 0) && d->c0 > 0, "mf_groupnorm: bad segments");
    const int C = d->c0 + d->c1;
    MF_CHECK_ARG(d->groups > 0 && C % d->groups == 0, "mf_groupnorm: C=%d not divisible by groups=%d", C, d->groups);
    MF_CHECK_ARG(d->c0 % 4 == 0 && d->c1 % 4 == 0, "mf_groupnorm: channel counts must be multiples of 4");
    MF_CHECK_ARG(d->batch >= 1 && d->hw >= 1, "mf_groupnorm: bad batch/hw");
    GnArgs a{};
    a.x0 = (const char*)d->x0; a.x1 = (const char*)d->x1;
    a.C0 = d->c0; a.C1 = d->c1; a.C = C; a.in_dt = d->in_dtype; a.HW = d->hw; a.G = d->groups;
    a.cpg = C / d->groups;
    // slices of whole groups whose channel count is a multiple of 4 (vector loads never straddle a slice)
    a.gslices = 1;
    for (int s = 4; s >= 2; s >>= 1)
        if (d->groups % s == 0 && ((d->groups / s) * a.cpg) % 4 == 0) { a.gslices = s; break; }
    a.rows_per_chunk = (d->hw + GN_MAX_CHUNKS - 1) / GN_MAX_CHUNKS;
    if (a.rows_per_chunk < 16) a.rows_per_chunk = 16;
    a.nchunks = (d->hw + a.rows_per_chunk - 1) / a.rows_per_chunk;
    a.eps = d->eps; a.gamma = d->gamma; a.beta = d->beta; a.silu = d->silu;
    a.out = (char*)d->out; a.out_dt = d->out_dtype; a.ws = d->ws;
    a.ws_ab = d->ws + (int64_t)d->batch * d->groups * GN_MAX_CHUNKS * 2;
    MF_CHECK_ARG(d->groups <= 64, "mf_groupnorm: at most 64 groups");
    const int slice_c = (a.G / a.gslices) * a.cpg;
    const size_t smem1 = (size_t)4 * slice_c * 2 * sizeof(float);
    MF_CHECK_ARG(smem1 <= 64 * 1024, "mf_groupnorm: slice of %d channels too large", slice_c);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nchunks, a.gslices, d->batch), dim3(256), smem1, s, a);
    MF_CHECK_LAUNCH("mf_groupnorm(stats)");
    const size_t smem2 = (size_t)(2 * C + 2 * a.G) * sizeof(float);
    MF_CHECK_ARG(smem2 <= 64 * 1024, "mf_groupnorm: C=%d too large", C);
    // ~8 blocks per CU worth of row blocks, at least 4 rows each
    int rows_per_block = (int)(((int64_t)d->hw * d->batch + 1023) / 1024);
    if (rows_per_block < 8) rows_per_block = 8;
    const int nblk = (d->hw + rows_per_block - 1) / rows_per_block;
    if (d->c0 % 8 == 0 && d->c1 % 8 == 0)
        hipLaunchKernelGGL(gn_apply_kernel<8>, dim3(nblk, d->batch), dim3(256), smem2, s, a, rows_per_block);
    else
        hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(nblk, d->batch), dim3(256), smem2, s, a, rows_per_block);
    MF_CHECK_LAUNCH("mf_groupnorm(apply)");
    return MF_OK;
}

extern "C" int mf_layernorm(const void* x, int32_t in_dtype, void* out, int32_t out_dtype, const float* gamma,
                            const float* beta, int64_t rows, int32_t c, float eps, void* stream) {
    MF_CHECK_ARG(x && out && gamma && beta, "mf_layernorm: null pointer");
    MF_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= 2048, "mf_layernorm: C=%d must be a multiple of 4 and <= 2048", c);
    if (rows <= 0) return MF_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)x, in_dtype, (char*)out, out_dtype, gamma, beta, rows, c, eps);
    MF_CHECK_LAUNCH("mf_layernorm");
    return MF_OK;
}

extern "C" int mf_softmax_rows(const float* scores, void* out, int32_t out_dtype, int64_t rows, int32_t cols,
                               int32_t ld, void* stream) {
    MF_CHECK_ARG(scores && out && cols >= 1 && ld >= cols, "mf_softmax_rows: bad arguments");
    if (rows <= 0) return MF_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       scores, (char*)out, out_dtype, rows, cols, ld);
    MF_CHECK_LAUNCH("mf_softmax_rows");
    return MF_OK;
}
