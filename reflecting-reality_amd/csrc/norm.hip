// GroupNorm(+SiLU), LayerNorm and row softmax for NHWC / token-major activations on gfx950.
// All three are HBM-bound: one read + one write of the tensor (GroupNorm above 16x16 reads it twice: stats,
// then apply), fp32 statistics, vectorised 4-channel accesses, no atomics (bitwise reproducible).
#include <stdlib.h>
#include "mf_common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 64;
constexpr int GN_BLK = 512;

struct GnArgs {
    const char* x0; const char* x1;
    int C0, C1, C, in_dt, HW, G, cpg, rows_per_chunk, nchunks;
    int cvn, tpr, rif;                 // vector columns per row, threads per row, rows in flight per block
    float eps;
    const float* gamma; const float* beta;
    int silu;
    char* out; int out_dt;
    float* ws;   // [batch][G][nchunks][2] = (mean, M2) of each chunk
    float* ws_ab;   // [batch][2][C] per-channel scale / shift (separate-finalize path)
    int fuse_finalize;
    float* stats_out;   // nullable [batch][G][2]: (mean, rstd) of every group, kept for mf_groupnorm_bwd (training)
    // statistics handed over by the producers (mf_gemm_desc.gn_part): per-channel (sum, sum of squares) of every block of rows
    const float2* part0; const float2* part1; int pr0, pr1;
    const float2* grp0; int nch0;      // per-GROUP sums of x0 from its producer: [batch][nch0][G]; fuse_finalize == 2
    // the input as a deferred split-K reduce (mf_groupnorm_desc.sk_ws): slabs [split][batch * HW][C] fp32, bias / temb / alpha
    const float* sk_ws; int sk_splits; const float* sk_bias; const float* sk_temb; int64_t sk_ld_temb; float sk_alpha; int64_t sk_mn;
};

template <bool F16>
__device__ __forceinline__ float4 load4(const char* p, int dt, int64_t idx) {
    if (dt == MF_F32) return *reinterpret_cast<const float4*>(p + idx * 4);
    const uint2 u = *reinterpret_cast<const uint2*>(p + idx * 2);
    float4 r;
    unpack_h2<F16>(u.x, r.x, r.y);
    unpack_h2<F16>(u.y, r.z, r.w);
    return r;
}
template <bool F16>
__device__ __forceinline__ void store4(char* p, int dt, int64_t idx, float4 v) {
    if (dt == MF_F32) {
        *reinterpret_cast<float4*>(p + idx * 4) = v;
    } else {
        uint2 u;
        u.x = pack_h2<F16>(v.x, v.y);
        u.y = pack_h2<F16>(v.z, v.w);
        *reinterpret_cast<uint2*>(p + idx * 2) = u;
    }
}

template <bool F16>
__device__ __forceinline__ void load8(const char* p, int dt, int64_t idx, float* o) {
    if (dt == MF_F32) {
        const float4 a = *reinterpret_cast<const float4*>(p + idx * 4);
        const float4 b = *reinterpret_cast<const float4*>(p + idx * 4 + 16);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        unpack_h8<F16>(*reinterpret_cast<const uint4*>(p + idx * 2), o);
    }
}
template <bool F16>
__device__ __forceinline__ void store8(char* p, int dt, int64_t idx, const float* v) {
    if (dt == MF_F32) {
        *reinterpret_cast<float4*>(p + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + idx * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        *reinterpret_cast<uint4*>(p + idx * 2) = pack_h8<F16>(v);
    }
}

template <int VW, bool F16>
__device__ __forceinline__ void loadv(const char* p, int dt, float* o) {
    if (VW == 8) {
        load8<F16>(p, dt, 0, o);
    } else {
        const float4 t = load4<F16>(p, dt, 0);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
}

// Thread geometry shared by both passes: a row of C channels is cvn = C/VW vector columns; tpr = min(cvn, 512)
// threads cover one row and rif = 512/tpr rows are in flight per block, so every lane issues a 16-byte access
// (VW = 8 bf16 / 2 x 16 bytes fp32) and a block's footprint is whole contiguous rows.  A thread keeps the same
// column(s) for all of its rows: per-channel state (sums, or scale/shift) lives in registers.
//
// Pass 1, grid (nchunks, batch): per-(thread-row, channel) fp32 sums -> LDS -> one (mean, M2) per group of the
// chunk, in double.  No atomics: bitwise reproducible.
template <int VW, int U = 4, bool F16 = false>
__global__ __launch_bounds__(GN_BLK) void gn_stats_kernel(const GnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* chan = reinterpret_cast<float2*>(smem_raw);   // [rif][C] (sum, sum of squares)
    const int chunk = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int lcol = t % p.tpr, trow = t / p.tpr;
    const int r0 = chunk * p.rows_per_chunk;
    int r1 = r0 + p.rows_per_chunk;
    if (r1 > p.HW) r1 = p.HW;
    const int esz = p.in_dt == MF_F32 ? 4 : 2;
    if (trow < p.rif) {
        for (int col = lcol; col < p.cvn; col += p.tpr) {
            const int c = col * VW;
            const char* base; int64_t ld; int cc;
            if (c < p.C0) { base = p.x0; ld = p.C0; cc = c; }
            else { base = p.x1; ld = p.C1; cc = c - p.C0; }
            float s[VW], ss[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) { s[e] = 0.0f; ss[e] = 0.0f; }
            const int64_t step = (int64_t)p.rif * ld * esz;
            const char* ptr = base + (((int64_t)b * p.HW + r0 + trow) * ld + cc) * esz;
            int r = r0 + trow;
            for (; r + (U - 1) * p.rif < r1; r += U * p.rif, ptr += U * step) {
                float v[U][VW];
#pragma unroll
                for (int u = 0; u < U; ++u) loadv<VW, F16>(ptr + u * step, p.in_dt, v[u]);
#pragma unroll
                for (int u = 0; u < U; u += 4)
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        s[e] += (v[u][e] + v[u + 1][e]) + (v[u + 2][e] + v[u + 3][e]);
                        ss[e] += (v[u][e] * v[u][e] + v[u + 1][e] * v[u + 1][e]) + (v[u + 2][e] * v[u + 2][e] + v[u + 3][e] * v[u + 3][e]);
                    }
            }
            for (; r < r1; r += p.rif, ptr += step) {
                float v0[VW];
                loadv<VW, F16>(ptr, p.in_dt, v0);
#pragma unroll
                for (int e = 0; e < VW; ++e) { s[e] += v0[e]; ss[e] += v0[e] * v0[e]; }
            }
#pragma unroll
            for (int e = 0; e < VW; ++e) chan[trow * p.C + c + e] = make_float2(s[e], ss[e]);
        }
    }
    __syncthreads();
    {   // 8 lanes per group: fixed-order partial sums over the (thread-row, channel) list, then a butterfly
        const int l = t & 7;
        const int items = p.rif * p.cpg;
        for (int g = t >> 3; g < p.G; g += (int)blockDim.x >> 3) {
            double s = 0.0, ss = 0.0;
            for (int it = l; it < items; it += 8) {
                const int tr = it / p.cpg;
                const float2 v = chan[tr * p.C + g * p.cpg + (it - tr * p.cpg)];
                s += (double)v.x;
                ss += (double)v.y;
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                s += __shfl_xor(s, off, 8);
                ss += __shfl_xor(ss, off, 8);
            }
            if (l == 0) {
                const double n = (double)(r1 - r0) * p.cpg;
                const double mean = s / n;
                double m2 = ss - s * mean;
                if (m2 < 0.0) m2 = 0.0;
                float* o = p.ws + (((int64_t)b * p.G + g) * p.nchunks + chunk) * 2;
                o[0] = (float)mean;
                o[1] = (float)m2;
            }
        }
    }
}

// Pass 1b, grid (batch): combine the chunk statistics of every group in a fixed order (double) and write the
// per-channel affine y = x*a[c] + b[c].  (An in-kernel "last block finalizes" variant needs agent-scope fences,
// whose L2 writeback/invalidate on this multi-XCD part cost more than this launch: measured 95 us vs 34 us.)
// (mean, rstd) of every group of sample b from the per-chunk (mean_k, M2_k) of pass 1, by all threads of the block:
// chunk k holds (mean_k, M2_k) over n_k elements: mean = sum n_k mean_k / N, M2 = sum M2_k + n_k (mean_k - mean)^2.
// 8 lanes per group, each over chunks j, j+8, ...; fixed-order butterflies (xor partners add the same two values), so
// every block that evaluates it gets the same bits.
__device__ __forceinline__ void gn_group_mean_rstd(const GnArgs& p, int b, float* gm, float* gr) {
    const int t = threadIdx.x;
    for (int g0 = 0; g0 < p.G; g0 += (int)blockDim.x >> 3) {
        const int g = g0 + (t >> 3), j = t & 7;
        const bool on = g < p.G;
        const float* st = p.ws + ((int64_t)b * p.G + (on ? g : 0)) * p.nchunks * 2;
        float pm[GN_MAX_CHUNKS / 8], pq[GN_MAX_CHUNKS / 8], pn[GN_MAX_CHUNKS / 8];
        double s1 = 0.0;
#pragma unroll
        for (int u = 0; u < GN_MAX_CHUNKS / 8; ++u) {
            const int k = j + 8 * u;
            pm[u] = 0.0f; pq[u] = 0.0f; pn[u] = 0.0f;
            if (on && k < p.nchunks) {
                const float2 v = *reinterpret_cast<const float2*>(st + 2 * k);
                int rows = p.rows_per_chunk;
                if ((k + 1) * p.rows_per_chunk > p.HW) rows = p.HW - k * p.rows_per_chunk;
                pm[u] = v.x; pq[u] = v.y; pn[u] = (float)(rows * p.cpg);
            }
            s1 += (double)pn[u] * (double)pm[u];
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) s1 += __shfl_xor(s1, off, 8);
        const double n = (double)p.HW * p.cpg;
        const double mean = s1 / n;
        double m2 = 0.0;
#pragma unroll
        for (int u = 0; u < GN_MAX_CHUNKS / 8; ++u) {
            const double d = (double)pm[u] - mean;
            m2 += (double)pq[u] + (double)pn[u] * d * d;
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) m2 += __shfl_xor(m2, off, 8);
        if (on && j == 0) {
            gm[g] = (float)mean;
            gr[g] = (float)(1.0 / sqrt(m2 / n + (double)p.eps));
        }
    }
}

// (mean, rstd) of every group of sample b from the per-(row block, group) sums the producing GEMM left (GnArgs::grp0): 8 lanes per
// group, each over blocks j, j + 8, ... in double; fixed-order butterflies, so every block that evaluates it gets the same bits.
__device__ __forceinline__ void gn_group_from_sums(const GnArgs& p, int b, float* gm, float* gr) {
    const int t = threadIdx.x;
    for (int g0 = 0; g0 < p.G; g0 += (int)blockDim.x >> 3) {
        const int g = g0 + (t >> 3), j = t & 7;
        const bool on = g < p.G;
        double s = 0.0, q = 0.0;
        if (on) {
            const float2* src = p.grp0 + (int64_t)b * p.nch0 * p.G + g;
            for (int k0 = j; k0 < p.nch0; k0 += 32) {                  // up to four independent loads in flight (nch0 <= 64), added in order
                float2 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = k0 + 8 * u < p.nch0 ? src[(int64_t)(k0 + 8 * u) * p.G] : make_float2(0.0f, 0.0f);
#pragma unroll
                for (int u = 0; u < 4; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            s += __shfl_xor(s, off, 8);
            q += __shfl_xor(q, off, 8);
        }
        if (on && j == 0) {
            const double n = (double)p.HW * p.cpg;
            const double mean = s / n;
            double m2 = q - s * mean;
            if (m2 < 0.0) m2 = 0.0;
            gm[g] = (float)mean;
            gr[g] = (float)(1.0 / sqrt(m2 / n + (double)p.eps));
        }
    }
}

__global__ __launch_bounds__(GN_BLK) void gn_finalize_kernel(const GnArgs p) {
    __shared__ float gm[64], gr[64];
    const int b = blockIdx.x, t = threadIdx.x;
    gn_group_mean_rstd(p, b, gm, gr);
    __syncthreads();
    if (p.stats_out)
        for (int g = t; g < p.G; g += blockDim.x) {
            p.stats_out[((int64_t)b * p.G + g) * 2] = gm[g];
            p.stats_out[((int64_t)b * p.G + g) * 2 + 1] = gr[g];
        }
    float* ab = p.ws_ab + (int64_t)b * 2 * p.C;
    for (int c = t; c < p.C; c += blockDim.x) {
        const int g = c / p.cpg;
        const float a = gr[g] * p.gamma[c];
        ab[c] = a;
        ab[p.C + c] = p.beta[c] - gm[g] * a;
    }
}

// Pass 1 replaced (round 6): the producing GEMMs left per-channel (sum, sum of squares) of every block of pr rows of their output
// (gemm_conv_kernel.h, GemmArgs::gn_part), so the statistics pass over the tensor is this launch over HW / pr * C float pairs per
// image.  grid (batch, slices): a block takes a contiguous run of groups; a thread sums one channel's blocks in block order
// (double), 8 lanes then combine a group's channels in a fixed order — every bit is reproducible.  Writes the per-channel
// affine y = x * a[c] + b[c] like gn_finalize_kernel.
__global__ __launch_bounds__(GN_BLK) void gn_finalize_part_kernel(const GnArgs p, int gps, int kparts) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2* chan = reinterpret_cast<double2*>(smem_raw);          // [kparts][gps * cpg] partial (sum, sum of squares), then row 0 = the totals
    __shared__ float gm[64], gr[64];
    const int b = blockIdx.x, t = threadIdx.x;
    const int g0 = blockIdx.y * gps;
    int g1 = g0 + gps;
    if (g1 > p.G) g1 = p.G;
    const int c0 = g0 * p.cpg, nc = (g1 - g0) * p.cpg, ncmax = gps * p.cpg;
    // (the affine parameters of the last stage are requested first: their latency hides under the partial sums')
    const float gam = t < nc ? p.gamma[c0 + t] : 0.0f, bet = t < nc ? p.beta[c0 + t] : 0.0f;
    // thread (channel j, part kp) sums the blocks kp, kp + kparts, ... of its channel: the launch is latency-bound (the partial sums
    // come from other XCDs' L2s through the fabric), so the loads of a channel are spread over kparts threads and issued together —
    // one thread per channel walking 32 blocks measured 11 us, as long as the statistics pass this launch replaces
    for (int idx = t; idx < nc * kparts; idx += blockDim.x) {
        const int kp = idx / nc, j = idx - kp * nc;
        const int c = c0 + j;
        const float2* src; int nb; int64_t ld;
        if (c < p.C0) { nb = p.HW / p.pr0; ld = p.C0; src = p.part0 + (int64_t)b * nb * ld + c; }
        else { nb = p.HW / p.pr1; ld = p.C1; src = p.part1 + (int64_t)b * nb * ld + (c - p.C0); }
        double s = 0.0, q = 0.0;
        int k = kp;
        for (; k + 3 * kparts < nb; k += 4 * kparts) {             // four independent loads in flight
            const float2 v0 = src[(int64_t)k * ld], v1 = src[(int64_t)(k + kparts) * ld], v2 = src[(int64_t)(k + 2 * kparts) * ld],
                         v3 = src[(int64_t)(k + 3 * kparts) * ld];
            s += (double)v0.x; q += (double)v0.y; s += (double)v1.x; q += (double)v1.y;
            s += (double)v2.x; q += (double)v2.y; s += (double)v3.x; q += (double)v3.y;
        }
        for (; k < nb; k += kparts) {
            const float2 v = src[(int64_t)k * ld];
            s += (double)v.x; q += (double)v.y;
        }
        chan[kp * ncmax + j] = make_double2(s, q);
    }
    __syncthreads();
    for (int j = t; j < nc; j += blockDim.x) {                     // parts in order: reproducible
        double s = 0.0, q = 0.0;
        for (int kp = 0; kp < kparts; ++kp) { const double2 v = chan[kp * ncmax + j]; s += v.x; q += v.y; }
        chan[j] = make_double2(s, q);                              // (row 0 is written by the thread that read column j of every row)
    }
    __syncthreads();
    {
        const int l = t & 7;
        for (int g = g0 + (t >> 3); g < g1; g += (int)blockDim.x >> 3) {
            double s = 0.0, q = 0.0;
            for (int it = l; it < p.cpg; it += 8) {
                const double2 v = chan[(g - g0) * p.cpg + it];
                s += v.x; q += v.y;
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                s += __shfl_xor(s, off, 8);
                q += __shfl_xor(q, off, 8);
            }
            if (l == 0) {
                const double n = (double)p.HW * p.cpg;
                const double mean = s / n;
                double m2 = q - s * mean;
                if (m2 < 0.0) m2 = 0.0;
                gm[g - g0] = (float)mean;
                gr[g - g0] = (float)(1.0 / sqrt(m2 / n + (double)p.eps));
                if (p.stats_out) {
                    p.stats_out[((int64_t)b * p.G + g) * 2] = gm[g - g0];
                    p.stats_out[((int64_t)b * p.G + g) * 2 + 1] = gr[g - g0];
                }
            }
        }
    }
    __syncthreads();
    float* ab = p.ws_ab + (int64_t)b * 2 * p.C;
    for (int j = t; j < nc; j += blockDim.x) {
        const int c = c0 + j, g = j / p.cpg;
        const float a = gr[g] * (j == t ? gam : p.gamma[c]);
        ab[c] = a;
        ab[p.C + c] = (j == t ? bet : p.beta[c]) - gm[g] * a;
    }
}

// Pass 2, grid (row blocks, batch): pure streaming y = silu(x*a[c] + b[c]) with the thread's a/b in registers.
template <int VW, int U = 4, bool F16 = false>
__global__ __launch_bounds__(GN_BLK) void gn_apply_kernel(const GnArgs p, int rows_per_block) {
    const int b = blockIdx.y, t = threadIdx.x;
    const int lcol = t % p.tpr, trow = t / p.tpr;
    __shared__ float gm[64], gr[64];
    if (p.fuse_finalize) {       // every block combines the chunk statistics itself: one launch (and its gap) less
        if (p.fuse_finalize == 2) gn_group_from_sums(p, b, gm, gr);
        else gn_group_mean_rstd(p, b, gm, gr);
        __syncthreads();
        if (p.stats_out && blockIdx.x == 0)
            for (int g = t; g < p.G; g += blockDim.x) {
                p.stats_out[((int64_t)b * p.G + g) * 2] = gm[g];
                p.stats_out[((int64_t)b * p.G + g) * 2 + 1] = gr[g];
            }
    }
    if (trow >= p.rif) return;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > p.HW) r1 = p.HW;
    const int esz = p.in_dt == MF_F32 ? 4 : 2, osz = p.out_dt == MF_F32 ? 4 : 2;
    const float* ab = p.ws_ab + (int64_t)b * 2 * p.C;
    const bool fast_silu = p.out_dt != MF_F32;          // 16-bit output: __expf is far inside the rounding
    for (int col = lcol; col < p.cvn; col += p.tpr) {
        const int c = col * VW;
        const char* base; int64_t ld; int cc;
        if (c < p.C0) { base = p.x0; ld = p.C0; cc = c; }
        else { base = p.x1; ld = p.C1; cc = c - p.C0; }
        float sa[VW], sb[VW];
        if (p.fuse_finalize) {
            loadv<VW, F16>(reinterpret_cast<const char*>(p.gamma + c), MF_F32, sa);
            loadv<VW, F16>(reinterpret_cast<const char*>(p.beta + c), MF_F32, sb);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const int g = (c + e) / p.cpg;
                sa[e] = gr[g] * sa[e];
                sb[e] = sb[e] - gm[g] * sa[e];
            }
        } else {
            loadv<VW, F16>(reinterpret_cast<const char*>(ab + c), MF_F32, sa);
            loadv<VW, F16>(reinterpret_cast<const char*>(ab + p.C + c), MF_F32, sb);
        }
        const int64_t step = (int64_t)p.rif * ld * esz, ostep = (int64_t)p.rif * p.C * osz;
        const char* ptr = base + (((int64_t)b * p.HW + r0 + trow) * ld + cc) * esz;
        char* optr = p.out + (((int64_t)b * p.HW + r0 + trow) * p.C + c) * osz;
        auto finish = [&](float* v, char* o) {
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                float y = v[e] * sa[e] + sb[e];
                if (p.silu) y = fast_silu ? silu_f(y) : silu_precise(y);
                v[e] = y;
            }
            if (VW == 8) store8<F16>(o, p.out_dt, 0, v);
            else store4<F16>(o, p.out_dt, 0, make_float4(v[0], v[1], v[2], v[3]));
        };
        int r = r0 + trow;
        for (; r + (U - 1) * p.rif < r1; r += U * p.rif, ptr += U * step, optr += U * ostep) {
            float v[U][VW];
#pragma unroll
            for (int u = 0; u < U; ++u) loadv<VW, F16>(ptr + u * step, p.in_dt, v[u]);      // U independent 16-byte loads in flight
#pragma unroll
            for (int u = 0; u < U; ++u) finish(v[u], optr + u * ostep);
        }
        for (; r < r1; r += p.rif, ptr += step, optr += ostep) {
            float v0[VW];
            loadv<VW, F16>(ptr, p.in_dt, v0);
            finish(v0, optr);
        }
    }
}

// One-launch GroupNorm for the lowest-resolution levels (HW <= 256: 16x16 / 8x8 latents), where the two-launch form
// above is a chain of dependent launches and memory round trips, not bytes (tools/bench_gn.py: 12 us at 8x8 for 2 MB).
// grid (C / SC, batch): a block owns a SLAB of SC = lcm(cpg, 8) channels = whole groups = nv 16-byte vector columns
// over ALL HW rows of one sample, so the statistics never leave the block and the rows never leave the registers.
// Thread t = lane * nv + vcol keeps vector column vcol of rows lane, lane + P, ... (at most ROWS): per-channel fp32
// (sum, sum of squares) -> LDS -> one wave per group combines them in double in a fixed order (bitwise
// reproducible) -> y = silu(x * a[c] + b[c]) from the registers.
template <int ROWS, bool F16 = false>
__global__ __launch_bounds__(1024) void gn_slab_kernel(const GnArgs p, int SC, int nv, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* chan = reinterpret_cast<float2*>(smem_raw);              // [P][SC or SC/8] (sum, sum of squares)
    __shared__ float gm[8], gr[8];
    const int slab = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int vcol = t % nv, lane = t / nv;
    const bool active = lane < P;                      // the block is padded to whole waves for the butterflies
    const int c = slab * SC + vcol * 8;
    const int esz = p.in_dt == MF_F32 ? 4 : 2, osz = p.out_dt == MF_F32 ? 4 : 2;
    const char* base; int64_t ld; int cc;
    if (c < p.C0) { base = p.x0; ld = p.C0; cc = c; }
    else { base = p.x1; ld = p.C1; cc = c - p.C0; }
    const int64_t step = (int64_t)P * ld * esz;
    const char* ptr = base + (((int64_t)b * p.HW + lane) * ld + cc) * esz;
    const bool whole = p.cpg % 8 == 0;                 // a thread's 8 channels lie in one group: pre-reduce them
    const int W = whole ? SC / 8 : SC;                 // LDS items per lane
    float v[ROWS][8];
    if (active) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if (lane + i * P < p.HW && p.sk_ws == nullptr) load8<F16>(ptr + i * step, p.in_dt, 0, v[i]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = 0.0f;
            }
        }
        if (p.sk_ws) {
            // the producer's split-K slabs: summed in slab order, + bias + temb, * alpha, rounded to the storage dtype — what
            // splitk_reduce_kernel + epilogue_store8 (csrc/gemm_conv.hip) would have stored and this kernel read back
            float bt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bt[e] = 0.0f;
            float tb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (p.sk_bias) load8<false>(reinterpret_cast<const char*>(p.sk_bias + c), MF_F32, 0, bt);
            if (p.sk_temb) load8<false>(reinterpret_cast<const char*>(p.sk_temb + (int64_t)b * p.sk_ld_temb + c), MF_F32, 0, tb);
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                if (lane + i * P >= p.HW) continue;
                const float* src = p.sk_ws + ((int64_t)b * p.HW + lane + i * P) * p.C + c;
                int z = 0;
                for (; z + 4 <= p.sk_splits; z += 4) {       // four slabs' loads in flight; the additions keep the slab order
                    float4 lo[4], hi[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        lo[u] = *reinterpret_cast<const float4*>(src + (z + u) * p.sk_mn);
                        hi[u] = *reinterpret_cast<const float4*>(src + (z + u) * p.sk_mn + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[i][0] += lo[u].x; v[i][1] += lo[u].y; v[i][2] += lo[u].z; v[i][3] += lo[u].w;
                        v[i][4] += hi[u].x; v[i][5] += hi[u].y; v[i][6] += hi[u].z; v[i][7] += hi[u].w;
                    }
                }
                for (; z < p.sk_splits; ++z) {
                    const float4 lo = *reinterpret_cast<const float4*>(src + z * p.sk_mn);
                    const float4 hi = *reinterpret_cast<const float4*>(src + z * p.sk_mn + 4);
                    v[i][0] += lo.x; v[i][1] += lo.y; v[i][2] += lo.z; v[i][3] += lo.w;
                    v[i][4] += hi.x; v[i][5] += hi.y; v[i][6] += hi.z; v[i][7] += hi.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[i][e];
                    if (p.sk_bias) x += bt[e];
                    if (p.sk_temb) x += tb[e];
                    x *= p.sk_alpha;
                    if (p.in_dt == MF_BF16) x = bf16_to_f32(f32_to_bf16(x));
                    else if (p.in_dt == MF_F16) x = (float)(_Float16)x;
                    v[i][e] = x;
                }
            }
        }
        float s[8], ss[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] = 0.0f; ss[e] = 0.0f;
#pragma unroll
            for (int i = 0; i < ROWS; i += 4) {
                s[e] += (v[i][e] + v[i + 1][e]) + (v[i + 2][e] + v[i + 3][e]);
                ss[e] += (v[i][e] * v[i][e] + v[i + 1][e] * v[i + 1][e]) + (v[i + 2][e] * v[i + 2][e] + v[i + 3][e] * v[i + 3][e]);
            }
        }
        if (whole) {
            const float s8 = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
            const float q8 = ((ss[0] + ss[1]) + (ss[2] + ss[3])) + ((ss[4] + ss[5]) + (ss[6] + ss[7]));
            chan[lane * W + vcol] = make_float2(s8, q8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) chan[lane * W + vcol * 8 + e] = make_float2(s[e], ss[e]);
        }
    }
    __syncthreads();
    {   // one wave per group of the slab: fixed-order partial sums in double, then a butterfly
        const int l = t & 63, ipg = whole ? p.cpg / 8 : p.cpg, items = P * ipg, gps = SC / p.cpg;
        for (int g = t >> 6; g < gps; g += (int)blockDim.x >> 6) {
            double s = 0.0, ss = 0.0;
            for (int it = l; it < items; it += 64) {
                const int tr = it / ipg;
                const float2 x = chan[tr * W + g * ipg + (it - tr * ipg)];
                s += (double)x.x;
                ss += (double)x.y;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                s += __shfl_xor(s, off, 64);
                ss += __shfl_xor(ss, off, 64);
            }
            if (l == 0) {
                const double n = (double)p.HW * p.cpg;
                const double mean = s / n;
                double m2 = ss - s * mean;
                if (m2 < 0.0) m2 = 0.0;
                gm[g] = (float)mean;
                gr[g] = (float)(1.0 / sqrt(m2 / n + (double)p.eps));
                if (p.stats_out) {
                    p.stats_out[((int64_t)b * p.G + slab * gps + g) * 2] = gm[g];
                    p.stats_out[((int64_t)b * p.G + slab * gps + g) * 2 + 1] = gr[g];
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
    float sa[8], sb[8];
    load8<false>(reinterpret_cast<const char*>(p.gamma + c), MF_F32, 0, sa);
    load8<false>(reinterpret_cast<const char*>(p.beta + c), MF_F32, 0, sb);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (vcol * 8 + e) / p.cpg;
        sa[e] = gr[g] * sa[e];
        sb[e] = sb[e] - gm[g] * sa[e];
    }
    const bool fast_silu = p.out_dt != MF_F32;
    const int64_t ostep = (int64_t)P * p.C * osz;
    char* optr = p.out + (((int64_t)b * p.HW + lane) * p.C + c) * osz;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        if (lane + i * P < p.HW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = v[i][e] * sa[e] + sb[e];
                if (p.silu) y = fast_silu ? silu_f(y) : silu_precise(y);
                v[i][e] = y;
            }
            store8<F16>(optr + i * ostep, p.out_dt, 0, v[i]);
        }
    }
}

// Half a wave per row (two rows per wave, 8 per block), 8-channel (16-byte) vectors: C % 8 == 0, C <= 2048.
// For the transformer widths of the path (320 / 640 / 1280) this moves twice the bytes per instruction of the
// 4-channel kernel below and halves the dependent shuffle chain (5 steps inside 32 lanes).
template <bool F16>
__global__ __launch_bounds__(256) void layernorm8_kernel(const char* x, int in_dt, char* out, int out_dt,
                                                         const float* gamma, const float* beta, int64_t rows, int C,
                                                         float eps) {
    const int l32 = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = row < rows;                       // keep every lane alive for the shuffles
    constexpr int MAXV = 8;
    float v[MAXV][8];
    const int c8n = C >> 3;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c8 = l32 + 32 * j;
        if (live && c8 < c8n) {
            load8<F16>(x, in_dt, row * C + c8 * 8, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.0f;
        }
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 32);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        if (l32 + 32 * j < c8n) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) q += __shfl_xor(q, off, 32);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c8 = l32 + 32 * j;
        if (c8 < c8n) {
            const int c = c8 * 8;
            float g[8], bb[8], y[8];
            load8<false>(reinterpret_cast<const char*>(gamma), MF_F32, c, g);
            load8<false>(reinterpret_cast<const char*>(beta), MF_F32, c, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (v[j][e] - mean) * rstd * g[e] + bb[e];
            store8<F16>(out, out_dt, row * C + c, y);
        }
    }
}

// One wave per row, up to 8 x 256 channels.
template <bool F16>
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
            v[j] = load4<F16>(x, in_dt, row * C + c4 * 4);
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
            store4<F16>(out, out_dt, row * C + c, y);
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
    MF_CHECK_ARG(!(mf_any_f16(d->in_dtype, d->out_dtype) && mf_any_bf16(d->in_dtype, d->out_dtype)), "mf_groupnorm: fp16 and bf16 operands in one launch");
    const bool f16 = mf_any_f16(d->in_dtype, d->out_dtype);
    GnArgs a{};
    a.x0 = (const char*)d->x0; a.x1 = (const char*)d->x1;
    a.C0 = d->c0; a.C1 = d->c1; a.C = C; a.in_dt = d->in_dtype; a.HW = d->hw; a.G = d->groups;
    a.cpg = C / d->groups;
    static const int gn_chunks = getenv("MFHIP_GN_CHUNKS") ? atoi(getenv("MFHIP_GN_CHUNKS")) : GN_MAX_CHUNKS;      // developer sweep
    const int want_chunks = gn_chunks >= 1 && gn_chunks <= GN_MAX_CHUNKS ? gn_chunks : GN_MAX_CHUNKS;
    a.rows_per_chunk = (d->hw + want_chunks - 1) / want_chunks;
    if (a.rows_per_chunk < 16) a.rows_per_chunk = 16;
    a.nchunks = (d->hw + a.rows_per_chunk - 1) / a.rows_per_chunk;
    a.eps = d->eps; a.gamma = d->gamma; a.beta = d->beta; a.silu = d->silu;
    a.out = (char*)d->out; a.out_dt = d->out_dtype; a.ws = d->ws; a.stats_out = d->stats_out;
    a.ws_ab = d->ws + (int64_t)d->batch * d->groups * GN_MAX_CHUNKS * 2;
    MF_CHECK_ARG(d->groups <= 64, "mf_groupnorm: at most 64 groups");
    static const bool gn3 = getenv("MFHIP_GN3") != nullptr;     // A/B switch: separate finalize launch
    // measured (tools/bench_gn.py): -1.5...2 us per GroupNorm up to 32x32, +1 us at 64x64 (64 chunks combined by 256 blocks)
    // (round 3: fetching a thread's first rows AHEAD of the combine, so that the fused form could also serve 64x64, was
    // measured neutral there — 26.3 vs 25.6 us, tools/bench_gn.py, and 16.70 vs 16.75 ms per denoise step — and removed)
    a.fuse_finalize = !gn3 && d->hw <= 1024 && mf_aligned16(d->gamma) && mf_aligned16(d->beta);
    const int vw = (d->c0 % 8 == 0 && d->c1 % 8 == 0) ? 8 : 4;
    hipStream_t s = (hipStream_t)stream;
    {   // one-launch slab kernel for the low-resolution levels
        // measured (tools/bench_gn.py, batch 8, us): 8x8 C 1280: 11.7 -> 5.0, 2560: 13.0 -> 6.0; 16x16 C 1280: 12.8 -> 8.1,
        // 2560: 17.4 -> 9.8
        static const bool two_pass = getenv("MFHIP_GN_2PASS") != nullptr;     // A/B switch: always the two-launch form
        int sc = a.cpg;                                  // lcm(cpg, 8)
        while (sc % 8) sc += a.cpg;
        const int nv = sc / 8;
        const int P = d->hw < 64 ? d->hw : 64, rows = 4;     // 32x32 (P 128, 8 rows, 128 blocks) measured no faster: 17.3 vs 17.6 us
        const int nthr = (nv * P + 63) / 64 * 64;
        const size_t smem = (size_t)P * sc * sizeof(float2);
        a.sk_ws = d->sk_ws; a.sk_splits = d->sk_splits; a.sk_bias = d->sk_bias; a.sk_temb = d->sk_temb; a.sk_ld_temb = d->sk_ld_temb;
        a.sk_alpha = d->sk_alpha; a.sk_mn = (int64_t)d->batch * d->hw * C;
        const bool slab_ok = vw == 8 && d->hw <= P * rows && C % sc == 0 && sc / a.cpg <= 8 && nthr <= 1024 && smem <= 64 * 1024 &&
                             mf_aligned16(d->gamma) && mf_aligned16(d->beta) && mf_aligned16(d->x0) && mf_aligned16(d->out) &&
                             (!d->x1 || mf_aligned16(d->x1));
        if (d->sk_ws) {
            MF_CHECK_ARG(slab_ok && d->c1 == 0 && d->sk_splits >= 2 && d->stats_out == nullptr && mf_aligned16(d->sk_ws) &&
                             (!d->sk_bias || mf_aligned16(d->sk_bias)) && (!d->sk_temb || (mf_aligned16(d->sk_temb) && d->sk_ld_temb % 4 == 0)),
                         "mf_groupnorm: a deferred split-K input (sk_ws) needs the one-launch form (hw <= 256, channels %% 8 == 0, one segment), "
                         ">= 2 slabs, 16-byte aligned slabs / bias / temb and no stats_out");
        }
        if ((!two_pass || d->sk_ws) && slab_ok) {
            if (f16) hipLaunchKernelGGL((gn_slab_kernel<4, true>), dim3(C / sc, d->batch), dim3(nthr), smem, s, a, sc, nv, P);
            else hipLaunchKernelGGL((gn_slab_kernel<4, false>), dim3(C / sc, d->batch), dim3(nthr), smem, s, a, sc, nv, P);
            MF_CHECK_LAUNCH("mf_groupnorm(slab)");
            return MF_OK;
        }
    }
    // statistics from the producers' partial sums: every present segment has them and their row blocks divide the image
    // (at most 128 partial blocks per image: the finalize launch walks them serially per channel)
    const bool from_parts = d->part0 != nullptr && d->part0_rows > 0 && d->hw % d->part0_rows == 0 && d->hw / d->part0_rows <= 128 &&
                            (d->c1 == 0 || (d->part1 != nullptr && d->part1_rows > 0 && d->hw % d->part1_rows == 0 && d->hw / d->part1_rows <= 128));
    a.part0 = (const float2*)d->part0; a.part1 = (const float2*)d->part1; a.pr0 = d->part0_rows; a.pr1 = d->part1_rows;
    if (from_parts) a.fuse_finalize = 0;
    // per-GROUP sums from the producer: no statistics pass and no finalize launch — every apply block combines its image's row blocks
    const bool from_groups = d->grp0 != nullptr && d->c1 == 0 && d->grp0_rows > 0 && d->hw % d->grp0_rows == 0 && d->hw / d->grp0_rows <= 64 &&
                             mf_aligned16(d->gamma) && mf_aligned16(d->beta);
    if (from_groups) { a.grp0 = (const float2*)d->grp0; a.nch0 = d->hw / d->grp0_rows; a.fuse_finalize = 2; }
    a.cvn = C / vw;
    a.tpr = a.cvn < GN_BLK ? a.cvn : GN_BLK;
    a.rif = GN_BLK / a.tpr;
    const int nthr = (a.tpr * a.rif + 63) / 64 * 64;
    const size_t smem1 = (size_t)a.rif * C * sizeof(float2) > (size_t)2 * d->groups * sizeof(float)
                             ? (size_t)a.rif * C * sizeof(float2) : (size_t)2 * d->groups * sizeof(float);
    MF_CHECK_ARG(smem1 <= 64 * 1024, "mf_groupnorm: C=%d too large", C);
    // ~4 row blocks per CU, at least 4 rows per thread
    static const int gn_blocks = getenv("MFHIP_GN_APPLY_BLOCKS") ? atoi(getenv("MFHIP_GN_APPLY_BLOCKS")) : 1024;   // developer sweep
    const int tgt = gn_blocks >= 64 ? gn_blocks : 1024;
    int rows_per_block = (int)(((int64_t)d->hw * d->batch + tgt - 1) / tgt);
    if (rows_per_block < 4 * a.rif) rows_per_block = 4 * a.rif;
    const int nblk = (d->hw + rows_per_block - 1) / rows_per_block;
    static const int gn_u = getenv("MFHIP_GN_UNROLL") ? atoi(getenv("MFHIP_GN_UNROLL")) : 4;      // developer sweep: 4 or 8 rows in flight
    // finalize from partial sums: 8 slices of the groups per image, and as many threads per channel as the block has room for
    const int gn_slices = d->groups >= 32 ? 8 : 1, gn_gps = (d->groups + gn_slices - 1) / gn_slices;
    int gn_kparts = GN_BLK / (gn_gps * a.cpg);
    {
        const int nb0 = from_parts ? d->hw / d->part0_rows : 1, nb1 = (from_parts && d->c1) ? d->hw / d->part1_rows : nb0;
        const int nbmin = nb0 < nb1 ? nb0 : nb1;
        if (gn_kparts > nbmin) gn_kparts = nbmin;
        if (gn_kparts < 1) gn_kparts = 1;
        while ((size_t)gn_kparts * gn_gps * a.cpg * sizeof(double2) > 48 * 1024 && gn_kparts > 1) --gn_kparts;
    }
#define MF_GN_LAUNCH(VW_, U_, F_)                                                                                              \
    do {                                                                                                                          \
        if (from_groups) {                                                                                                        \
        } else if (from_parts) {                                                                                                  \
            hipLaunchKernelGGL(gn_finalize_part_kernel, dim3(d->batch, gn_slices), dim3(GN_BLK), (size_t)gn_kparts * gn_gps * a.cpg * sizeof(double2), s, a, gn_gps, gn_kparts); \
            MF_CHECK_LAUNCH("mf_groupnorm(finalize from partial sums)");                                                          \
        } else {                                                                                                                  \
            hipLaunchKernelGGL((gn_stats_kernel<VW_, U_, F_>), dim3(a.nchunks, d->batch), dim3(nthr), smem1, s, a);               \
            MF_CHECK_LAUNCH("mf_groupnorm(stats)");                                                                               \
            if (!a.fuse_finalize) hipLaunchKernelGGL(gn_finalize_kernel, dim3(d->batch), dim3(GN_BLK), 0, s, a);                 \
        }                                                                                                                         \
        hipLaunchKernelGGL((gn_apply_kernel<VW_, U_, F_>), dim3(nblk, d->batch), dim3(nthr), 0, s, a, rows_per_block);            \
    } while (0)
    if (vw == 8 && gn_u == 8) { if (f16) MF_GN_LAUNCH(8, 8, true); else MF_GN_LAUNCH(8, 8, false); }
    else if (vw == 8) { if (f16) MF_GN_LAUNCH(8, 4, true); else MF_GN_LAUNCH(8, 4, false); }
    else { if (f16) MF_GN_LAUNCH(4, 4, true); else MF_GN_LAUNCH(4, 4, false); }
#undef MF_GN_LAUNCH
    MF_CHECK_LAUNCH("mf_groupnorm(apply)");
    return MF_OK;
}

extern "C" int mf_layernorm(const void* x, int32_t in_dtype, void* out, int32_t out_dtype, const float* gamma,
                            const float* beta, int64_t rows, int32_t c, float eps, void* stream) {
    MF_CHECK_ARG(x && out && gamma && beta, "mf_layernorm: null pointer");
    MF_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= 2048, "mf_layernorm: C=%d must be a multiple of 4 and <= 2048", c);
    if (rows <= 0) return MF_OK;
    static const bool ln4 = getenv("MFHIP_LN4") != nullptr;     // A/B switch: the 4-channel kernel
    MF_CHECK_ARG(!(mf_any_f16(in_dtype, out_dtype) && mf_any_bf16(in_dtype, out_dtype)), "mf_layernorm: fp16 and bf16 operands in one launch");
    const bool f16 = mf_any_f16(in_dtype, out_dtype);
    const bool v8 = !ln4 && c % 8 == 0 && mf_aligned16(x) && mf_aligned16(out) && mf_aligned16(gamma) && mf_aligned16(beta);
    const dim3 grid((unsigned)(v8 ? (rows + 7) / 8 : (rows + 3) / 4));
    auto* kern = v8 ? (f16 ? layernorm8_kernel<true> : layernorm8_kernel<false>) : (f16 ? layernorm_kernel<true> : layernorm_kernel<false>);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, in_dtype, (char*)out, out_dtype, gamma, beta, rows, c, eps);
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
