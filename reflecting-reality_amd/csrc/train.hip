// Backward / optimizer kernels of the MirrorFusion training step for gfx950 (examples/brushnet/train_brushnet_mirror.py
// :1459-1466: accelerator.backward, clip_grad_norm_(1.0), AdamW step).  The reference has no native code here: every
// gradient is ATen autograd.  These kernels are the arithmetic of that backward pass, hand-written:
//
//   mf_conv_wgrad      dW[n][tap][c] = sum_m dY[m][n] * A[pix(m, tap)][c]      (MFMA; the one GEMM form the forward
//                      kernel cannot express: both operands are pixel-major, the reduction runs over pixels)
//   mf_transpose       strided batched fp32 transpose (dgrad weight layout [C][taps flipped][N], P^T / dS^T / q^T / k^T)
//   mf_colsum          column sums per row segment (bias gradients, d temb, dgamma / dbeta partials)
//   mf_groupnorm_bwd   GroupNorm(+SiLU) backward over one or two NHWC segments
//   mf_layernorm_bwd, mf_softmax_bwd, mf_silu_bwd, mf_geglu_bwd, mf_zero_insert2x, mf_sumpool2x2, mf_mse_grad
//   mf_sumsq / mf_clip_coef / mf_adamw   gradient norm, clip coefficient (device-resident) and fused AdamW over flat
//                      fp32 arenas (master weights, gradients, exp_avg, exp_avg_sq)
//
// dgrad needs no kernel of its own: the data gradient of a stride-1 conv / linear is mf_gemm_conv on the transposed
// (and tap-flipped) weight; strided and upsampling convs go through mf_zero_insert2x / mf_sumpool2x2.
// Everything is deterministic (fixed-order reductions, no atomics): two runs give bit-identical gradients.
#include "mf_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* a0; const float* a1;
    int C0, Ctot; int64_t lda0, lda1;
    int Hin, Win, Ho, Wo, HoWo, KW, taps, stride, pad_t, pad_l, ups;
    const float* dy; int64_t lddy;
    int M, N;
    float* out; int64_t ldo;            // dW (splitm == 1) or the slab workspace
    int64_t slab;                       // floats per slab
    int splitm, m_per_split, tiles_k_per_tap, tiles_n;
    unsigned mul_howo, sh_howo, mul_wo, sh_wo;      // m / HoWo and rr / Wo by multiplication (wg_fastdiv)
    int slabs_xcd;                                  // the first slabs_xcd (a multiple of 8) slabs are pinned to XCDs
    int acc_out;                                    // out += (one slab accumulating into dW)
};

// floor(x / d) for 0 <= x < 2^31 as (x * mul) >> sh with l = ceil(log2 d), mul = floor(2^(31+l) / d) + 1, sh = 31 + l
// (Granlund-Montgomery: mul * d - 2^(31+l) <= d <= 2^l).  The 32-bit division it replaces expands to ~30 VALU instructions
// and the staging loop needs eight of them per 32 pixels: more issue cycles than the tile's MFMAs.
static inline void wg_fastdiv_make(unsigned d, unsigned* mul, unsigned* sh) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    *mul = (unsigned)(((1ull << (31 + l)) / d) + 1);
    *sh = 31 + l;
}
__device__ __forceinline__ int wg_fastdiv(int x, unsigned mul, unsigned sh) {
    return (int)(((unsigned long long)(unsigned)x * mul) >> sh);
}

constexpr int WG_BN = 128, WG_BC = 128, WG_BP = 32, WG_PITCH = 132;   // LDS rows of 128 floats + 4 pad (16-byte aligned rows)

// Block = 4 waves, output tile 128 (n) x 128 (c of one tap); wave (wn, wk) owns 64 x 64 = 2 x 2 accumulator blocks, so every
// gathered fragment feeds two MFMAs (the first version's 32 x 32 per wave fed one).  Per iteration 32 pixels are staged
// (register -> LDS, single buffered: correctness-first; the reduction over pixels is split over blockIdx.y).
__device__ unsigned g_split_ovf_train;    // raised when an MF_F16X3 wgrad operand exceeded the fp16 range (mf_common.h)

template <int DT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    float wg_amax = 0.0f;
    __shared__ __attribute__((aligned(16))) float sY[WG_BP * WG_PITCH];
    __shared__ __attribute__((aligned(16))) float sA[WG_BP * WG_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    int bid = blockIdx.x;
    const int tile_n = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tile_k = bid % p.tiles_k_per_tap;
    const int tap = bid / p.tiles_k_per_tap;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int n0 = tile_n * WG_BN, c0 = tile_k * WG_BC;
    const int m_begin = blockIdx.y * p.m_per_split;
    int m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // staging coordinates: thread -> (row = tid / 32 (+8 per pass), 4 consecutive columns of 128)
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;
    for (int m0 = m_begin; m0 < m_end; m0 += WG_BP) {
#pragma unroll
        for (int i = 0; i < WG_BP / 8; ++i) {
            const int row = srow + 8 * i;
            const int m = m0 + row;
            float4 vy = make_float4(0, 0, 0, 0), va = make_float4(0, 0, 0, 0);
            if (m < m_end) {
                const int n = n0 + scol;
                if (n + 4 <= p.N) vy = *reinterpret_cast<const float4*>(p.dy + (int64_t)m * p.lddy + n);
                else if (n < p.N) {
                    float t[4] = {0, 0, 0, 0};
                    for (int j = 0; j < 4 && n + j < p.N; ++j) t[j] = p.dy[(int64_t)m * p.lddy + n + j];
                    vy = make_float4(t[0], t[1], t[2], t[3]);
                }
                const int b = m / p.HoWo, rr = m - b * p.HoWo, oy = rr / p.Wo, ox = rr - oy * p.Wo;
                const int iy = oy * p.stride - p.pad_t + ky, ix = ox * p.stride - p.pad_l + kx;
                const int c = c0 + scol;
                if ((unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim && c < p.Ctot) {
                    const int64_t pix = (int64_t)b * p.Hin * p.Win + (int64_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
                    const float* src = c < p.C0 ? p.a0 + pix * p.lda0 + c : p.a1 + pix * p.lda1 + (c - p.C0);
                    va = *reinterpret_cast<const float4*>(src);          // channel counts are multiples of 4: no straddle
                }
            }
            *reinterpret_cast<float4*>(sY + row * WG_PITCH + scol) = vy;
            *reinterpret_cast<float4*>(sA + row * WG_PITCH + scol) = va;
        }
        __syncthreads();
        const float* py = sY + wn * 64 + r;
        const float* pa = sA + wk * 64 + r;
        if constexpr (DT == MF_F32) {
#pragma unroll
            for (int s = 0; s < WG_BP / 2; ++s) {
                const float y0 = py[(2 * s + h) * WG_PITCH], y1 = py[(2 * s + h) * WG_PITCH + 32];
                const float a0 = pa[(2 * s + h) * WG_PITCH], a1 = pa[(2 * s + h) * WG_PITCH + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, a0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, a1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1, a0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1, a1, acc[1][1], 0, 0, 0);
            }
        } else {
            // fp32 -> (hi, lo) fp16 halves, three MFMAs per product (the split precision of mf_gemm_conv)
#pragma unroll
            for (int s = 0; s < WG_BP / 16; ++s) {
                f16x8_t Yh[2], Yl[2], Ah[2], Al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    unsigned yh[4], yl[4], ah[4], al[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int pp = 16 * s + 8 * h + 2 * e;
                        const float y0 = py[pp * WG_PITCH + 32 * t], y1 = py[(pp + 1) * WG_PITCH + 32 * t];
                        const float a0 = pa[pp * WG_PITCH + 32 * t], a1 = pa[(pp + 1) * WG_PITCH + 32 * t];
                        if constexpr (DT == MF_BF16X1) {       // one bf16 MFMA per product: operands rounded to nearest-even
                            yh[e] = pack_bf16x2(y0, y1); ah[e] = pack_bf16x2(a0, a1);
                            yl[e] = 0; al[e] = 0;
                        } else {
                            wg_amax = mf_amax3(mf_amax3(wg_amax, y0, y1), a0, a1);
                            const auto hy = __builtin_amdgcn_cvt_pkrtz(y0, y1);
                            const auto ly = __builtin_amdgcn_cvt_pkrtz(y0 - (float)hy[0], y1 - (float)hy[1]);
                            const auto ha = __builtin_amdgcn_cvt_pkrtz(a0, a1);
                            const auto la = __builtin_amdgcn_cvt_pkrtz(a0 - (float)ha[0], a1 - (float)ha[1]);
                            yh[e] = __builtin_bit_cast(unsigned, hy); yl[e] = __builtin_bit_cast(unsigned, ly);
                            ah[e] = __builtin_bit_cast(unsigned, ha); al[e] = __builtin_bit_cast(unsigned, la);
                        }
                    }
                    Yh[t] = __builtin_bit_cast(f16x8_t, uint4{yh[0], yh[1], yh[2], yh[3]});
                    Yl[t] = __builtin_bit_cast(f16x8_t, uint4{yl[0], yl[1], yl[2], yl[3]});
                    Ah[t] = __builtin_bit_cast(f16x8_t, uint4{ah[0], ah[1], ah[2], ah[3]});
                    Al[t] = __builtin_bit_cast(f16x8_t, uint4{al[0], al[1], al[2], al[3]});
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (DT == MF_BF16X1) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Yh[i]),
                                                                                __builtin_bit_cast(bf16x8_t, Ah[j]), acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Yl[i], Ah[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Yh[i], Al[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Yh[i], Ah[j], acc[i][j], 0, 0, 0);
                        }
                    }
            }
        }
        __syncthreads();
    }
    if constexpr (DT == MF_F16X3) mf_raise_if_over(&g_split_ovf_train, wg_amax);
    // D[row = n (e & 3) + 8 (e >> 2) + 4 h][col = c r]
    float* out = p.out + (int64_t)blockIdx.y * p.slab;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wk * 64 + 32 * j + r;
            if (c >= p.Ctot) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < p.N) out[(int64_t)n * p.ldo + (int64_t)tap * p.Ctot + c] = acc[i][j][e];
            }
        }
}

// ---- the 16-bit forms (MF_F16X3, MF_BF16X1), second version -------------------------------------------------------------
// Same 128 (n) x 128 (c of one tap) tile and 64 x 64 per wave, but the operands are converted ONCE while they are staged
// (above: every element is read from LDS as fp32 by two waves, one ds_read_b32 per element, and split in both) into 16-bit
// planes [pixel][128 columns] (256-byte rows, 16-byte chunks XOR-swizzled), and the MFMA fragments — eight consecutive PIXELS of
// one column — come out of two ds_read_b64_tr_b16 (the hardware's transposing read: a quarter of the read instructions, twice
// the LDS bytes per clock of ds_read_b32).  Two LDS stages: the global loads of tile t+1 are in flight during the MFMAs of
// tile t and are converted and written behind them; one barrier per 32 pixels.
constexpr int WT_PLANE = WG_BP * 256;     // one 16-bit plane of a 32-pixel x 128-column tile: 8 KB

__device__ __forceinline__ int wt_off(int row, int ch) {      // byte offset of 16-byte chunk ch of pixel row `row`
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

typedef __attribute__((ext_vector_type(4))) short wt_s4;
__device__ __forceinline__ f16x8_t wt_frag(const char* base, int a0, int a1) {
    // two transposed reads = pixels 0..3 and 4..7 of this lane's column
    const wt_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s4*)(base + a0));
    const wt_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s4*)(base + a1));
    struct { wt_s4 a, b; } v{lo, hi};
    return __builtin_bit_cast(f16x8_t, v);
}

// IN16 (DT == MF_BF16X1 only): x and dy are ALREADY bf16 in memory (mf_conv_wgrad with dtype MF_BF16: the pre-rounded operand
// copies of the bf16x1 training mode) — a staged chunk of 8 columns is one 16-byte load written to LDS as it is: half the operand
// bytes and no conversion in the staging loop.
template <int DT, bool IN16 = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tr_kernel(const WgradArgs p) {
    static_assert(!IN16 || DT == MF_BF16X1, "16-bit inputs: the one-plane bf16 form");
    constexpr int NPL = DT == MF_F16X3 ? 2 : 1;              // planes per operand: (hi, lo) or the one bf16 plane
    constexpr int STAGE = 2 * NPL * WT_PLANE;                // Y planes, then A planes
    extern __shared__ __attribute__((aligned(16))) char wt_smem[];
    float wg_amax = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    // 1-D grid, workgroups go to the 8 XCDs round-robin: with the pixel slabs a multiple of 8, XCD x takes slabs x, x + 8, ... and
    // all the tiles of a slab run side by side on ONE XCD, marching through the same pixels: each dy / x tile crosses the fabric
    // once per XCD and the other (tiles - 1) reads hit that XCD's L2 (a slab's dy + x do not fit in 4 MB, the moving window does)
    const int tiles_all = p.tiles_n * p.tiles_k_per_tap * p.taps;
    int bid, slab_id;
    if ((int)blockIdx.x < p.slabs_xcd * tiles_all) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        slab_id = xcd + 8 * (idx / tiles_all);
        bid = idx % tiles_all;
    } else {                                                  // the remaining (< 8) slabs: spread over all XCDs
        slab_id = blockIdx.x / tiles_all;
        bid = blockIdx.x - slab_id * tiles_all;
    }
    const int tile_n = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tile_k = bid % p.tiles_k_per_tap;
    const int tap = bid / p.tiles_k_per_tap;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int n0 = tile_n * WG_BN, c0 = tile_k * WG_BC;
    const int m_begin = slab_id * p.m_per_split;
    int m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging: thread -> (pixel row tid / 8 of the 32, 16 consecutive columns of 128): ONE row's coordinates per thread and tile
    const int srow = tid >> 3, scol = (tid & 7) * 16;
    const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;
    float4 vy0[4], va0[4], vy1[4], va1[4];        // two register sets: the loads of tile t+2 are issued before the MFMAs of tile t
    auto load_tile = [&](int m0, float4 (&vy)[4], float4 (&va)[4]) {
        const int m = m0 + srow;
#pragma unroll
        for (int k = 0; k < 4; ++k) { vy[k] = make_float4(0, 0, 0, 0); va[k] = make_float4(0, 0, 0, 0); }
        if (m < m_end) {
            const int b = wg_fastdiv(m, p.mul_howo, p.sh_howo), rr = m - b * p.HoWo, oy = wg_fastdiv(rr, p.mul_wo, p.sh_wo), ox = rr - oy * p.Wo;
            const int iy = oy * p.stride - p.pad_t + ky, ix = ox * p.stride - p.pad_l + kx;
            const bool inside = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            const int64_t pix = (int64_t)b * p.Hin * p.Win + (int64_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
            if constexpr (IN16) {      // 8 bf16 = 16 bytes per chunk, carried in vy / va [0] and [2] (N, C0, Ctot multiples of 8: host check)
                const unsigned short* yrow = reinterpret_cast<const unsigned short*>(p.dy) + (int64_t)m * p.lddy;
                const unsigned short* r0 = reinterpret_cast<const unsigned short*>(p.a0) + pix * p.lda0;
                const unsigned short* r1 = reinterpret_cast<const unsigned short*>(p.a1) + pix * p.lda1 - p.C0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int n = n0 + scol + 8 * k, c = c0 + scol + 8 * k;
                    if (n < p.N) vy[2 * k] = *reinterpret_cast<const float4*>(yrow + n);
                    if (inside && c < p.Ctot) va[2 * k] = *reinterpret_cast<const float4*>((c < p.C0 ? r0 : r1) + c);
                }
            } else {
            const float* yrow = p.dy + (int64_t)m * p.lddy;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = n0 + scol + 4 * k;
                if (n + 4 <= p.N) vy[k] = *reinterpret_cast<const float4*>(yrow + n);
                else if (n < p.N) {
                    float t[4] = {0, 0, 0, 0};
                    for (int j = 0; j < 4 && n + j < p.N; ++j) t[j] = yrow[n + j];
                    vy[k] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            if (inside) {
                const float* r0 = p.a0 + pix * p.lda0;
                const float* r1 = p.a1 + pix * p.lda1 - p.C0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + scol + 4 * k;
                    if (c < p.Ctot) va[k] = *reinterpret_cast<const float4*>((c < p.C0 ? r0 : r1) + c);   // channel counts are multiples of 4
                }
            }
            }
        }
    };
    auto cvt = [&](const float4 v, uint2& hi, uint2& lo) {
        if constexpr (DT == MF_BF16X1) {
            hi = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        } else {
            wg_amax = mf_amax3(mf_amax3(wg_amax, v.x, v.y), v.z, v.w);
            mf_split_f16x2(v.x, v.y, hi.x, lo.x);
            mf_split_f16x2(v.z, v.w, hi.y, lo.y);
        }
    };
    auto put2 = [&](char* dst, const float4 u, const float4 v) {          // eight consecutive columns = one 16-byte chunk per plane
        if constexpr (IN16) {
            *reinterpret_cast<float4*>(dst) = u;                          // already 8 bf16
            return;
        }
        uint2 h0, l0, h1, l1;
        cvt(u, h0, l0); cvt(v, h1, l1);
        *reinterpret_cast<uint4*>(dst) = uint4{h0.x, h0.y, h1.x, h1.y};
        if constexpr (NPL == 2) *reinterpret_cast<uint4*>(dst + WT_PLANE) = uint4{l0.x, l0.y, l1.x, l1.y};
    };
    auto write_tile = [&](char* st, const float4 (&vy)[4], const float4 (&va)[4]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int off = wt_off(srow, (scol >> 3) + k);
            put2(st + off, vy[2 * k], vy[2 * k + 1]);
            put2(st + NPL * WT_PLANE + off, va[2 * k], va[2 * k + 1]);
        }
    };
    // transposed-read addresses: lane 4q + pp of the 16-lane group g supplies row q of the group's 4-pixel x 16-column block,
    // columns 4pp..4pp+3; group g = (pixel half h = g >> 1, column half g & 1) of the 32-column operand block
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int ay[2][2], aa[2][2];                                   // [32-column block][pixels 0..3 / 4..7]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * (g >> 1) + 4 * j + q;
            ay[i][j] = wt_off(row, ((wn * 64 + 32 * i + 16 * (g & 1)) >> 3) + (pp >> 1)) + 8 * (pp & 1);
            aa[i][j] = wt_off(row, ((wk * 64 + 32 * i + 16 * (g & 1)) >> 3) + (pp >> 1)) + 8 * (pp & 1) + NPL * WT_PLANE;
        }

    const int tiles = (m_end - m_begin + WG_BP - 1) / WG_BP;
    auto compute = [&](const char* st) {
#pragma unroll
        for (int s = 0; s < WG_BP / 16; ++s) {
            const char* sb = st + s * 16 * 256;               // 16 pixel rows further: the swizzle repeats every 16 rows
            f16x8_t Y[NPL][2], A[NPL][2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Y[pl][i] = wt_frag(sb + pl * WT_PLANE, ay[i][0], ay[i][1]);
                    A[pl][i] = wt_frag(sb + pl * WT_PLANE, aa[i][0], aa[i][1]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (DT == MF_BF16X1) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Y[0][i]),
                                                                            __builtin_bit_cast(bf16x8_t, A[0][j]), acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Y[NPL - 1][i], A[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Y[0][i], A[NPL - 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Y[0][i], A[0][j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    if (tiles > 0) {
        load_tile(m_begin, vy0, va0);
        write_tile(wt_smem, vy0, va0);
        if (tiles > 1) load_tile(m_begin + WG_BP, vy0, va0);
    }
    __syncthreads();
    for (int t = 0; t < tiles; t += 2) {
        // even tile t: stage 0; its successor t+1 sits in set 0, t+2 goes to set 1
        if (t + 2 < tiles) load_tile(m_begin + (t + 2) * WG_BP, vy1, va1);
        compute(wt_smem);
        if (t + 1 < tiles) write_tile(wt_smem + STAGE, vy0, va0);
        __syncthreads();
        if (t + 1 >= tiles) break;
        // odd tile t+1: stage 1; t+2 sits in set 1, t+3 goes to set 0
        if (t + 3 < tiles) load_tile(m_begin + (t + 3) * WG_BP, vy0, va0);
        compute(wt_smem + STAGE);
        if (t + 2 < tiles) write_tile(wt_smem, vy1, va1);
        __syncthreads();
    }
    if constexpr (DT == MF_F16X3) mf_raise_if_over(&g_split_ovf_train, wg_amax);
    // D[row = n (e & 3) + 8 (e >> 2) + 4 h][col = c r]
    float* out = p.out + (int64_t)slab_id * p.slab;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wk * 64 + 32 * j + r;
            if (c >= p.Ctot) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < p.N) {
                    float* o = out + (int64_t)n * p.ldo + (int64_t)tap * p.Ctot + c;
                    *o = p.acc_out ? *o + acc[i][j][e] : acc[i][j][e];
                }
            }
        }
}

// ---- bf16 inputs, staged by LDS-DMA ----------------------------------------------------------------------------------------
// conv_wgrad_tr_kernel<MF_BF16X1, IN16> with the register staging replaced by `buffer_load ... lds`: the operands are already bf16 in
// memory, so a tile's 16-byte chunks go from HBM / L2 straight to their (swizzled) LDS places — no staging registers, no ds_write, no
// conversion — and three stages are in flight (two tiles ahead of the MFMAs) at three blocks per CU (the register-staged form holds
// 206 VGPRs: two).  A DMA instruction writes 64 lanes x 16 B = four consecutive 256-byte pixel rows; lane L lands on row L >> 4, chunk
// position L & 15, so it FETCHES chunk (L & 15) ^ swizzle(row): the XOR swizzle of the transposing reads is applied on the global side.
// Out-of-range rows / columns / padding taps use an offset past the descriptor's range and arrive as zeros.  Same tiles, same MFMA
// order, same epilogue as conv_wgrad_tr_kernel: bit-identical results.  One input segment per tile (host check: no x1, or C0 % 128 == 0).
constexpr int WD_NST = 3;
__global__ __launch_bounds__(256, 3) void conv_wgrad_dma_kernel(const WgradArgs p) {
    constexpr int STAGE = 2 * WT_PLANE;                      // Y plane, then A plane
    extern __shared__ __attribute__((aligned(16))) char wt_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_all = p.tiles_n * p.tiles_k_per_tap * p.taps;
    int bid, slab_id;
    if ((int)blockIdx.x < p.slabs_xcd * tiles_all) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        slab_id = xcd + 8 * (idx / tiles_all);
        bid = idx % tiles_all;
    } else {
        slab_id = blockIdx.x / tiles_all;
        bid = blockIdx.x - slab_id * tiles_all;
    }
    const int tile_n = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tile_k = bid % p.tiles_k_per_tap;
    const int tap = bid / p.tiles_k_per_tap;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int n0 = tile_n * WG_BN, c0 = tile_k * WG_BC;
    const int m_begin = slab_id * p.m_per_split;
    int m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- staging by DMA: wave w owns pixel rows [8w, 8w + 8) of every tile, two 4-row instructions per plane
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)wt_smem);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;
    const bool seg1 = c0 >= p.C0;                              // the tile's columns lie in x1
    const int64_t ldx = seg1 ? p.lda1 : p.lda0;
    const int cseg = seg1 ? c0 - p.C0 : c0, climit = seg1 ? p.Ctot - p.C0 : p.C0;
    const int batch = p.M / p.HoWo;
    const srd_t srdX = make_srd(reinterpret_cast<const char*>(seg1 ? p.a1 : p.a0), (unsigned)((int64_t)batch * p.Hin * p.Win * ldx * 2));
    const srd_t srdY = make_srd(reinterpret_cast<const char*>(p.dy) + ((int64_t)m_begin * p.lddy + n0) * 2,
                                (unsigned)(((int64_t)(m_end - m_begin - 1) * p.lddy + (p.N - n0)) * 2));
    int rowi[2];
    unsigned yoff[2], xcol[2];
    bool yok[2], xok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = 8 * wv + 4 * i + (lane >> 4);
        const int ch = (lane & 15) ^ (((R & 3) << 2) | ((R >> 2) & 3));
        rowi[i] = R;
        yok[i] = n0 + 8 * ch < p.N;
        yoff[i] = (unsigned)(((int64_t)R * p.lddy + 8 * ch) * 2);
        xok[i] = cseg + 8 * ch < climit;
        xcol[i] = (unsigned)((cseg + 8 * ch) * 2);
    }
    const unsigned ystep = (unsigned)(WG_BP * p.lddy * 2);
    const unsigned ldx2 = (unsigned)(ldx * 2);
    auto issue_tile = [&](int stage, int t) {
        const int m0 = m_begin + t * WG_BP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned ly = lds0 + stage * STAGE + (8 * wv + 4 * i) * 256;
            const int m = m0 + rowi[i];
            const bool row_ok = m < m_end;
            dma16_buf((row_ok && yok[i]) ? yoff[i] + (unsigned)t * ystep : 0x80000000u, srdY, ly);
            // 32-bit arithmetic throughout (the host guarantees every byte offset < 2^31): this address computation runs once per
            // row and tile, and as 64-bit code it cost more VALU issue than the tile's eight MFMAs (10 VALU per MFMA under the counters)
            unsigned xo = 0x80000000u;
            {
                const int b = wg_fastdiv(m, p.mul_howo, p.sh_howo), rr = m - b * p.HoWo, oy = wg_fastdiv(rr, p.mul_wo, p.sh_wo), ox = rr - oy * p.Wo;
                const int iy = oy * p.stride - p.pad_t + ky, ix = ox * p.stride - p.pad_l + kx;
                const unsigned pix = (unsigned)((b * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups));
                if (row_ok && xok[i] && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim) xo = pix * ldx2 + xcol[i];
            }
            dma16_buf(xo, srdX, ly + WT_PLANE);
        }
    };

    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int ay[2][2], aa[2][2];                                   // [32-column block][pixels 0..3 / 4..7] (conv_wgrad_tr_kernel's addresses)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * (g >> 1) + 4 * j + q;
            ay[i][j] = wt_off(row, ((wn * 64 + 32 * i + 16 * (g & 1)) >> 3) + (pp >> 1)) + 8 * (pp & 1);
            aa[i][j] = wt_off(row, ((wk * 64 + 32 * i + 16 * (g & 1)) >> 3) + (pp >> 1)) + 8 * (pp & 1) + WT_PLANE;
        }
    auto compute = [&](const char* st) {
#pragma unroll
        for (int s = 0; s < WG_BP / 16; ++s) {
            const char* sb = st + s * 16 * 256;
            f16x8_t Y[2], A[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Y[i] = wt_frag(sb, ay[i][0], ay[i][1]);
                A[i] = wt_frag(sb, aa[i][0], aa[i][1]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Y[i]), __builtin_bit_cast(bf16x8_t, A[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
    };

    const int tiles = (m_end - m_begin + WG_BP - 1) / WG_BP;
    for (int s = 0; s < WD_NST - 1 && s < tiles; ++s) issue_tile(s, s);
    int st_c = 0, st_i = WD_NST - 1;                         // stage computed this iteration / stage the next issue goes to
    for (int t = 0; t < tiles; ++t) {
        // tiles newer than t that may stay in flight: WD_NST - 2 (4 DMA instructions each), fewer at the tail
        if (tiles - t - 1 >= WD_NST - 2) wait_vmcnt<4 * (WD_NST - 2)>();
        else wait_vmcnt<0>();
        __syncthreads();                                       // tile t is in LDS for everybody; everybody is done with tile t - 1's stage
        if (t + WD_NST - 1 < tiles) issue_tile(st_i, t + WD_NST - 1);
        compute(wt_smem + st_c * STAGE);
        st_c = st_c + 1 == WD_NST ? 0 : st_c + 1;
        st_i = st_i + 1 == WD_NST ? 0 : st_i + 1;
    }
    float* out = p.out + (int64_t)slab_id * p.slab;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wk * 64 + 32 * j + r;
            if (c >= p.Ctot) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < p.N) {
                    float* o = out + (int64_t)n * p.ldo + (int64_t)tap * p.Ctot + c;
                    *o = p.acc_out ? *o + acc[i][j][e] : acc[i][j][e];
                }
            }
        }
}

// fp32 [rows][k] (row stride ldw) -> mf_gemm_desc's w_split layout: per 32 k, [32 high halves | 32 low halves], rows zero-padded
// to kp = round_up(k, 32); hi = RNE(w), lo = RNE(w - hi) (bit-identical to the host's torch split).  The trainable weights
// change every optimizer step: one pass over them buys the pre-split GEMM forms for the step's forward and data gradients.
template <int DT>      // MF_F16X3 or MF_BF16X3
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, int64_t ldw, unsigned short* __restrict__ out, int64_t rows,
                                                         int k, int kp) {
    const int64_t quads = rows * (kp >> 2);
    float amax = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (kp >> 2);
        const int c = (int)(i - row * (kp >> 2)) * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = c + j < k ? w[row * ldw + c + j] : 0.0f;
        unsigned short hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (DT == MF_F16X3) {
                amax = fmaxf(amax, fabsf(v[j]));
                const _Float16 h = (_Float16)v[j];
                const _Float16 l = (_Float16)(v[j] - (float)h);
                hi[j] = __builtin_bit_cast(unsigned short, h); lo[j] = __builtin_bit_cast(unsigned short, l);
            } else {
                hi[j] = f32_to_bf16(v[j]);
                lo[j] = f32_to_bf16(v[j] - bf16_to_f32(hi[j]));
            }
        }
        unsigned short* dst = out + row * 2 * (int64_t)kp + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<uint2*>(dst) = uint2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
        *reinterpret_cast<uint2*>(dst + 32) = uint2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
    }
    if constexpr (DT == MF_F16X3) mf_raise_if_over(&g_split_ovf_train, amax);
}

// ---- 160 x 160 tiles, five waves ------------------------------------------------------------------------------------------
// The same scheme on a 160 (n) x 160 (c) tile: wave w owns the 32 columns w of the c side and all five 32-row blocks of the n
// side (5 accumulator blocks; one A fragment feeds five products).  The step's channel counts are multiples of 160 (320, 640,
// 960, 1280, 1920, 2560): no padded third tile as with 128 (320 = 2.5 x 128), and 320 staged columns feed 160 x 160 products
// per pixel where 256 fed 128 x 128 (1.56 x the products per byte: the kernel is bound by its operand traffic).  Planes are
// [32 pixels][160 columns] with a 320-byte pitch: row q of a transposed read's 4-row block starts 16 banks after row q - 1, so
// the reads are conflict-free without a swizzle.  80 KB of LDS (two stages) = two blocks per CU.
constexpr int W160 = 160, W160_PITCH = 2 * W160, W160_PLANE = WG_BP * W160_PITCH;     // 10 KB per plane

template <int DT, int NST, bool IN16 = false>
__global__ __launch_bounds__(320, 2) void conv_wgrad_tr160_kernel(const WgradArgs p) {
    static_assert(!IN16 || DT == MF_BF16X1, "16-bit inputs: the one-plane bf16 form");
    constexpr int NPL = DT == MF_F16X3 ? 2 : 1;
    constexpr int STAGE = 2 * NPL * W160_PLANE;
    extern __shared__ __attribute__((aligned(16))) char wt_smem[];
    float wg_amax = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_all = p.tiles_n * p.tiles_k_per_tap * p.taps;
    int bid, slab_id;
    if ((int)blockIdx.x < p.slabs_xcd * tiles_all) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        slab_id = xcd + 8 * (idx / tiles_all);
        bid = idx % tiles_all;
    } else {
        slab_id = blockIdx.x / tiles_all;
        bid = blockIdx.x - slab_id * tiles_all;
    }
    const int tile_n = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tile_k = bid % p.tiles_k_per_tap;
    const int tap = bid / p.tiles_k_per_tap;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int n0 = tile_n * W160, c0 = tile_k * W160;
    const int m_begin = slab_id * p.m_per_split;
    int m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;

    f32x16_t acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    // staging: 32 pixel rows x 20 chunks of 8 columns per operand = 640 slots, two per thread (rows 16 apart)
    const int srow = tid / 20, sch = tid - srow * 20;
    const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;
    float4 vy[2][2], va[2][2];
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = m0 + srow + 16 * k;
            vy[k][0] = vy[k][1] = va[k][0] = va[k][1] = make_float4(0, 0, 0, 0);
            if (m < m_end) {
                const int b = wg_fastdiv(m, p.mul_howo, p.sh_howo), rr = m - b * p.HoWo, oy = wg_fastdiv(rr, p.mul_wo, p.sh_wo), ox = rr - oy * p.Wo;
                const int iy = oy * p.stride - p.pad_t + ky, ix = ox * p.stride - p.pad_l + kx;
                const bool inside = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
                const int64_t pix = (int64_t)b * p.Hin * p.Win + (int64_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
                if constexpr (IN16) {      // one 16-byte chunk of 8 bf16 per operand and row, carried in vy / va [k][0]
                    const int n = n0 + 8 * sch, c = c0 + 8 * sch;
                    if (n < p.N) vy[k][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(p.dy) + (int64_t)m * p.lddy + n);
                    if (inside && c < p.Ctot) {
                        const unsigned short* r0 = reinterpret_cast<const unsigned short*>(p.a0) + pix * p.lda0;
                        const unsigned short* r1 = reinterpret_cast<const unsigned short*>(p.a1) + pix * p.lda1 - p.C0;
                        va[k][0] = *reinterpret_cast<const float4*>((c < p.C0 ? r0 : r1) + c);
                    }
                } else {
                const float* yrow = p.dy + (int64_t)m * p.lddy;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + 8 * sch + 4 * j;
                    if (n + 4 <= p.N) vy[k][j] = *reinterpret_cast<const float4*>(yrow + n);
                    else if (n < p.N) {
                        float t[4] = {0, 0, 0, 0};
                        for (int u = 0; u < 4 && n + u < p.N; ++u) t[u] = yrow[n + u];
                        vy[k][j] = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                if (inside) {
                    const float* r0 = p.a0 + pix * p.lda0;
                    const float* r1 = p.a1 + pix * p.lda1 - p.C0;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = c0 + 8 * sch + 4 * j;
                        if (c < p.Ctot) va[k][j] = *reinterpret_cast<const float4*>((c < p.C0 ? r0 : r1) + c);
                    }
                }
                }
            }
        }
    };
    auto cvt = [&](const float4 v, uint2& hi, uint2& lo) {
        if constexpr (DT == MF_BF16X1) {
            hi = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        } else {
            wg_amax = mf_amax3(mf_amax3(wg_amax, v.x, v.y), v.z, v.w);
            mf_split_f16x2(v.x, v.y, hi.x, lo.x);
            mf_split_f16x2(v.z, v.w, hi.y, lo.y);
        }
    };
    auto put2 = [&](char* dst, const float4 u, const float4 v) {
        if constexpr (IN16) {
            *reinterpret_cast<float4*>(dst) = u;
            return;
        }
        uint2 h0, l0, h1, l1;
        cvt(u, h0, l0); cvt(v, h1, l1);
        *reinterpret_cast<uint4*>(dst) = uint4{h0.x, h0.y, h1.x, h1.y};
        if constexpr (NPL == 2) *reinterpret_cast<uint4*>(dst + W160_PLANE) = uint4{l0.x, l0.y, l1.x, l1.y};
    };
    auto write_tile = [&](char* st) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int off = (srow + 16 * k) * W160_PITCH + 16 * sch;
            put2(st + off, vy[k][0], vy[k][1]);
            put2(st + NPL * W160_PLANE + off, va[k][0], va[k][1]);
        }
    };
    // transposed-read addresses (see conv_wgrad_tr_kernel): lane 4q + pp of group g -> row 8 (g >> 1) + 4 j + q, columns 4 pp ..
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int ay[2], aa[2];                                         // pixels 0..3 / 4..7; n-block i adds 64 bytes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * (g >> 1) + 4 * j + q;
        ay[j] = row * W160_PITCH + 2 * (16 * (g & 1) + 4 * pp);
        aa[j] = ay[j] + 64 * wave + NPL * W160_PLANE;
    }

    const int tiles = (m_end - m_begin + WG_BP - 1) / WG_BP;
    if (tiles > 0) {
        load_tile(m_begin);
        write_tile(wt_smem);
    }
    __syncthreads();
    for (int t = 0; t < tiles; ++t) {
        const char* st = wt_smem + (NST == 2 ? (t & 1) * STAGE : 0);
        if (t + 1 < tiles) load_tile(m_begin + (t + 1) * WG_BP);
#pragma unroll
        for (int s = 0; s < WG_BP / 16; ++s) {
            const char* sb = st + s * 16 * W160_PITCH;
            f16x8_t A[NPL], Y[2][NPL];        // the n-block i + 1 fragments are read while block i's MFMAs run (left to itself the
                                              // compiler reuses one register set: read, wait, three MFMAs, read, wait ...)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                A[pl] = wt_frag(sb + pl * W160_PLANE, aa[0], aa[1]);
                Y[0][pl] = wt_frag(sb + pl * W160_PLANE, ay[0], ay[1]);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (i + 1 < 5) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) Y[(i + 1) & 1][pl] = wt_frag(sb + pl * W160_PLANE + 64 * (i + 1), ay[0], ay[1]);
                }
                const f16x8_t* Yc = Y[i & 1];
                if constexpr (DT == MF_BF16X1) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Yc[0]), __builtin_bit_cast(bf16x8_t, A[0]), acc[i], 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Yc[NPL - 1], A[0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Yc[0], A[NPL - 1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Yc[0], A[0], acc[i], 0, 0, 0);
                }
            }
        }
        if (NST == 1) __syncthreads();                        // one stage: every wave is done reading before the tile is replaced
        if (t + 1 < tiles) write_tile(wt_smem + (NST == 2 ? ((t + 1) & 1) * STAGE : 0));
        __syncthreads();
    }
    if constexpr (DT == MF_F16X3) mf_raise_if_over(&g_split_ovf_train, wg_amax);
    float* out = p.out + (int64_t)slab_id * p.slab;
    const int c = c0 + 32 * wave + r;
    if (c < p.Ctot) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < p.N) {
                    float* o = out + (int64_t)n * p.ldo + (int64_t)tap * p.Ctot + c;
                    *o = p.acc_out ? *o + acc[i][e] : acc[i][e];
                }
            }
    }
}

// out[i] (+)= sum_z slabs[z][i]   (rows of `cols` floats, output row stride ldo)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* ws, int nslab, int64_t slab, float* out, int64_t ldo,
                                                        int rows, int cols, int accumulate) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / cols), col = (int)(i - (int64_t)row * cols);
        float s = 0.0f;
        for (int z = 0; z < nslab; ++z) s += ws[(int64_t)z * slab + i];
        float* o = out + (int64_t)row * ldo + col;
        *o = accumulate ? *o + s : s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// strided batched transpose: y[z][c][r] = x[z][r][c]
// ------------------------------------------------------------------------------------------------------------
// OUT16: y is bf16 (nearest-even): the transposed data-gradient weight of the bf16x1 mode is rounded while it is laid out
// bf16 -> bf16, 64 x 64 tiles, 16-byte loads along the input rows and 16-byte stores along the output rows (8 consecutive input
// rows of one channel).  cols, ldx, ldy multiples of 8; ldy >= rows rounded up to 8 (the pad columns are written as zeros).
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int rows, int cols,
                                                          int64_t ldx, int64_t ldy, int64_t zsx, int64_t zsy) {
    __shared__ unsigned short t[64][72];
    const unsigned short* xs = x + (int64_t)blockIdx.z * zsx;
    unsigned short* ys = y + (int64_t)blockIdx.z * zsy;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = (threadIdx.x >> 3) + 32 * i, cg = (threadIdx.x & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + rr < rows && c0 + cg < cols) v = *reinterpret_cast<const uint4*>(xs + (int64_t)(r0 + rr) * ldx + c0 + cg);
        *reinterpret_cast<uint4*>(&t[rr][cg]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cc = (threadIdx.x >> 3) + 32 * i, rg = (threadIdx.x & 7) * 8;
        if (c0 + cc < cols && r0 + rg < rows) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = t[rg + j][cc];
            *reinterpret_cast<uint4*>(ys + (int64_t)(c0 + cc) * ldy + r0 + rg) =
                uint4{e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16)};
        }
    }
}

// fp32 -> bf16 (nearest-even), 64 x 64 tiles: 16-byte loads along the input rows, 16-byte stores of 8 consecutive input rows of one
// column.  cols / ldx / zsx multiples of 4, ldy / zsy multiples of 8, 16-byte aligned bases (mf_transpose_bf16 checks).
__global__ __launch_bounds__(256) void transpose_f2b_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int rows, int cols,
                                                            int64_t ldx, int64_t ldy, int64_t zsx, int64_t zsy) {
    __shared__ float t[64][65];
    const float* xs = x + (int64_t)blockIdx.z * zsx;
    unsigned short* ys = y + (int64_t)blockIdx.z * zsy;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (threadIdx.x >> 4) + 16 * i, cg = (threadIdx.x & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + rr < rows && c0 + cg < cols) v = *reinterpret_cast<const float4*>(xs + (int64_t)(r0 + rr) * ldx + c0 + cg);
        t[rr][cg] = v.x; t[rr][cg + 1] = v.y; t[rr][cg + 2] = v.z; t[rr][cg + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cc = (threadIdx.x >> 3) + 32 * i, rg = (threadIdx.x & 7) * 8;
        if (c0 + cc >= cols || r0 + rg >= rows) continue;
        unsigned short* o = ys + (int64_t)(c0 + cc) * ldy + r0 + rg;
        if (r0 + rg + 8 <= rows)
            *reinterpret_cast<uint4*>(o) = uint4{pack_bf16x2(t[rg][cc], t[rg + 1][cc]), pack_bf16x2(t[rg + 2][cc], t[rg + 3][cc]),
                                                 pack_bf16x2(t[rg + 4][cc], t[rg + 5][cc]), pack_bf16x2(t[rg + 6][cc], t[rg + 7][cc])};
        else
            for (int j = 0; r0 + rg + j < rows; ++j) o[j] = f32_to_bf16(t[rg + j][cc]);
    }
}

template <bool OUT16>
__global__ __launch_bounds__(256) void transpose_kernel(const float* x, void* yv, int rows, int cols, int64_t ldx, int64_t ldy,
                                                        int64_t zsx, int64_t zsy) {
    __shared__ float t[32][33];
    const float* xs = x + (int64_t)blockIdx.z * zsx;
    if constexpr (OUT16) {
        unsigned short* ys = reinterpret_cast<unsigned short*>(yv) + (int64_t)blockIdx.z * zsy;
        const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = r0 + ty + 8 * i, cc = c0 + tx;
            t[ty + 8 * i][tx] = (rr < rows && cc < cols) ? xs[(int64_t)rr * ldx + cc] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = c0 + ty + 8 * i, rr = r0 + tx;
            if (cc < cols && rr < rows) ys[(int64_t)cc * ldy + rr] = f32_to_bf16(t[tx][ty + 8 * i]);
        }
        return;
    }
    float* ys = reinterpret_cast<float*>(yv) + (int64_t)blockIdx.z * zsy;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + ty + 8 * i, cc = c0 + tx;
        t[ty + 8 * i][tx] = (rr < rows && cc < cols) ? xs[(int64_t)rr * ldx + cc] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + 8 * i, rr = r0 + tx;
        if (cc < cols && rr < rows) ys[(int64_t)cc * ldy + rr] = t[tx][ty + 8 * i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// column sums per row segment: out[s][n] = sum_{m in segment s} x[m][n]; two deterministic stages
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_stage1(const float* x, int64_t ldx, float* part, int n, int64_t rows_per_seg,
                                                     int chunks, int64_t rows_per_chunk) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int seg = blockIdx.y / chunks, ch = blockIdx.y - seg * chunks;
    const int64_t r0 = (int64_t)seg * rows_per_seg + (int64_t)ch * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk;
    if (r1 > (int64_t)(seg + 1) * rows_per_seg) r1 = (int64_t)(seg + 1) * rows_per_seg;
    float s = 0.0f;
    if (col < n)
        for (int64_t rr = r0 + rl; rr < r1; rr += 4) s += x[rr * ldx + col];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < n)
        part[((int64_t)seg * chunks + ch) * n + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void colsum_stage2(const float* part, float* out, int64_t ldo, int n, int segs, int chunks,
                                                     int accumulate) {
    const int64_t total = (int64_t)segs * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int seg = (int)(i / n), col = (int)(i - (int64_t)seg * n);
        float s = 0.0f;
        for (int c = 0; c < chunks; ++c) s += part[((int64_t)seg * chunks + c) * n + col];
        float* o = out + (int64_t)seg * ldo + col;
        *o = accumulate ? *o + s : s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 -> bf16 copy of a gradient WITH its column sums (per row segment) from the same read: the bf16x1 training mode rounds
// every conv / linear output gradient once (the operand of its weight and data gradients), and the bias / time-embedding
// gradients are column sums of the very same tensor.  Stage 1: a block owns `cpb` 8-column groups x one row chunk, a thread one
// 8-column group of every rpp-th row; stage 2: 4 lanes per column over the chunk partials.  Fixed order, no atomics.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_colsum_stage1(const float* __restrict__ x, unsigned short* __restrict__ out, float* __restrict__ part,
                                                          int n, int cpb, int rpp, int64_t rows_per_seg, int chunks, int64_t rows_per_chunk) {
    __shared__ float red[256 * 8];
    const int rl = threadIdx.x / cpb, cl = threadIdx.x - rl * cpb;
    const int col = (blockIdx.x * cpb + cl) * 8;
    const int seg = blockIdx.y / chunks, ch = blockIdx.y - seg * chunks;
    const int64_t r0 = (int64_t)seg * rows_per_seg + (int64_t)ch * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk;
    if (r1 > (int64_t)(seg + 1) * rows_per_seg) r1 = (int64_t)(seg + 1) * rows_per_seg;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < rpp && col < n) {
        const float* xp = x + col;
        unsigned short* op = out + col;
        int64_t rr = r0 + rl;
        for (; rr + 3 * (int64_t)rpp < r1; rr += 4 * (int64_t)rpp) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const float4*>(xp + (rr + (int64_t)u * rpp) * n);
                b[u] = *reinterpret_cast<const float4*>(xp + (rr + (int64_t)u * rpp) * n + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                *reinterpret_cast<uint4*>(op + (rr + (int64_t)u * rpp) * n) =
                    uint4{pack_bf16x2(a[u].x, a[u].y), pack_bf16x2(a[u].z, a[u].w), pack_bf16x2(b[u].x, b[u].y), pack_bf16x2(b[u].z, b[u].w)};
                s[0] += a[u].x; s[1] += a[u].y; s[2] += a[u].z; s[3] += a[u].w;
                s[4] += b[u].x; s[5] += b[u].y; s[6] += b[u].z; s[7] += b[u].w;
            }
        }
        for (; rr < r1; rr += rpp) {
            const float4 a = *reinterpret_cast<const float4*>(xp + rr * n), b = *reinterpret_cast<const float4*>(xp + rr * n + 4);
            *reinterpret_cast<uint4*>(op + rr * n) = uint4{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
            s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
            s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[j * 256 + threadIdx.x] = s[j];
    __syncthreads();
    // thread t < cpb * 8: column (t >> 3 group, t & 7) summed over the row lanes in order
    for (int t = threadIdx.x; t < cpb * 8; t += 256) {
        const int g = t >> 3, j = t & 7;
        const int c = (blockIdx.x * cpb + g) * 8 + j;
        if (c >= n) continue;
        float acc = 0.0f;
        for (int q = 0; q < rpp; ++q) acc += red[j * 256 + q * cpb + g];
        part[((int64_t)seg * chunks + ch) * n + c] = acc;
    }
}

// Stage 2: a block owns 16 columns, 16 lanes per column walk the chunk partials (four independent loads in flight per lane), then
// the 16 lane sums of every (segment, column) are added in order.  segs <= CC2_MAX_SEGS.
constexpr int CC2_MAX_SEGS = 32;
__global__ __launch_bounds__(256) void cast_colsum_stage2(const float* __restrict__ part, float* seg_out, int64_t ldo, int seg_acc, float* tot_out,
                                                          int tot_acc, int n, int segs, int chunks) {
    __shared__ float red[CC2_MAX_SEGS][16][17];
    __shared__ float segsum[CC2_MAX_SEGS][16];
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    for (int seg = 0; seg < segs; ++seg) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        if (col < n) {
            const float* pp = part + (int64_t)seg * chunks * n + col;
            int c = j;
            for (; c + 48 < chunks; c += 64) {
                const float v0 = pp[(int64_t)c * n], v1 = pp[(int64_t)(c + 16) * n], v2 = pp[(int64_t)(c + 32) * n], v3 = pp[(int64_t)(c + 48) * n];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; c < chunks; c += 16) a0 += pp[(int64_t)c * n];
        }
        red[seg][j][cl] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < segs * 16; t += 256) {
        const int seg = t >> 4, c = t & 15;
        float sv = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sv += red[seg][q][c];
        segsum[seg][c] = sv;
        const int cc = blockIdx.x * 16 + c;
        if (seg_out && cc < n) { float* o = seg_out + (int64_t)seg * ldo + cc; *o = seg_acc ? *o + sv : sv; }
    }
    __syncthreads();
    if (threadIdx.x < 16 && col < n && tot_out) {
        float tot = 0.0f;
        for (int seg = 0; seg < segs; ++seg) tot += segsum[seg][cl];
        tot_out[col] = tot_acc ? tot_out[col] + tot : tot;
    }
}

// ------------------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward.  One block per (batch, group).  y = act(xhat * gamma + beta), xhat = (x - mean) * rstd.
// ------------------------------------------------------------------------------------------------------------
struct GnBwdArgs {
    const float* x0; const float* x1; int c0, c1;
    const float* dy;                       // [b][hw][c0 + c1]
    const float* gamma; const float* beta;
    float* dx0; float* dx1;                // [b][hw][c0], [b][hw][c1]
    const float* add0; const float* add1;  // nullable, laid out like dx0 / dx1: dx = (the gradient) + add (a residual's gradient)
    float* dgamma_part; float* dbeta_part; // [b][c0 + c1] or null
    int batch, hw, groups, silu; float eps;
};

__device__ __forceinline__ double block_sum(double v, double* red) {       // 256 threads, fixed order
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const GnBwdArgs p) {
    __shared__ double red[4];
    __shared__ float cg[256], cb[256];
    const int C = p.c0 + p.c1, cpg = C / p.groups;
    const int b = blockIdx.x / p.groups, g = blockIdx.x - b * p.groups;
    const int rif = 256 / cpg;                         // pixel rows in flight
    const int ch = threadIdx.x % cpg, rl = threadIdx.x / cpg;
    const bool active = rl < rif;
    const int c = g * cpg + ch;
    const bool seg1 = c >= p.c0;
    const float* xb = seg1 ? p.x1 + (int64_t)b * p.hw * p.c1 + (c - p.c0) : p.x0 + (int64_t)b * p.hw * p.c0 + c;
    const int ldx = seg1 ? p.c1 : p.c0;
    const float* dyb = p.dy + (int64_t)b * p.hw * C + c;
    // ---- statistics ----
    double s = 0.0, ss = 0.0;
    if (active)
        for (int px = rl; px < p.hw; px += rif) {
            const double v = xb[(int64_t)px * ldx];
            s += v; ss += v * v;
        }
    const double n = (double)p.hw * cpg;
    const double mean = block_sum(s, red) / n;
    double var = block_sum(ss, red) / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float fmean = (float)mean;
    const float gam = active ? p.gamma[c] : 0.0f, bet = active ? p.beta[c] : 0.0f;
    auto dz_of = [&](float xv, float dyv, float& xhat) {
        xhat = (xv - fmean) * rstd;
        if (!p.silu) return dyv;
        const float z = xhat * gam + bet;
        const float sg = 1.0f / (1.0f + expf(-z));
        return dyv * (sg * (1.0f + z * (1.0f - sg)));
    };
    // ---- sums ----
    double s1 = 0.0, s2 = 0.0;
    float dg = 0.0f, db = 0.0f;
    if (active)
        for (int px = rl; px < p.hw; px += rif) {
            float xhat;
            const float dz = dz_of(xb[(int64_t)px * ldx], dyb[(int64_t)px * C], xhat);
            s1 += (double)(dz * gam); s2 += (double)(dz * gam * xhat);
            dg += dz * xhat; db += dz;
        }
    const float m1 = (float)(block_sum(s1, red) / n), m2 = (float)(block_sum(s2, red) / n);
    if (p.dgamma_part) {
        cg[threadIdx.x] = dg; cb[threadIdx.x] = db;
        __syncthreads();
        if (threadIdx.x < cpg) {
            float a = 0.0f, bb = 0.0f;
            for (int j = 0; j < rif; ++j) { a += cg[j * cpg + threadIdx.x]; bb += cb[j * cpg + threadIdx.x]; }
            p.dgamma_part[(int64_t)b * C + c] = a;
            p.dbeta_part[(int64_t)b * C + c] = bb;
        }
    }
    // ---- dx ----
    if (active) {
        float* dxb = seg1 ? p.dx1 + (int64_t)b * p.hw * p.c1 + (c - p.c0) : p.dx0 + (int64_t)b * p.hw * p.c0 + c;
        const float* adb = seg1 ? (p.add1 ? p.add1 + (int64_t)b * p.hw * p.c1 + (c - p.c0) : nullptr)
                                : (p.add0 ? p.add0 + (int64_t)b * p.hw * p.c0 + c : nullptr);
        for (int px = rl; px < p.hw; px += rif) {
            float xhat;
            const float dz = dz_of(xb[(int64_t)px * ldx], dyb[(int64_t)px * C], xhat);
            const float v = rstd * (dz * gam - m1 - xhat * m2);
            dxb[(int64_t)px * ldx] = adb ? v + adb[(int64_t)px * ldx] : v;
        }
    }
}

// ---- the streaming form for hw >= 256: five launches, every load a coalesced float4 over channels --------------------
// A: per (image, pixel chunk) per-channel  sum x, sum x^2            -> part      R1: per (image, group) mean, rstd
// B: per (image, pixel chunk) per-channel  sum dz, sum dz * xhat     -> part      R2: dgamma / dbeta partials, m1, m2
// C: dx = rstd * (dz * gamma - m1 - xhat * m2)
// One block per (image, group) as above leaves one block per CU and 40-byte runs per pixel at 320 channels (10 per group):
// ~125 us on average over the step's GroupNorms; the sums are fp32 over the <= 64 pixels of a chunk, double across chunks.
constexpr int GN_BWD_MAX_BATCH = 64;          // images per call of the fused parameter-gradient form (gn_bwd2_reduce_acc_kernel)
struct GnBwd2Args {
    GnBwdArgs a;
    float* part;        // [batch][chunks][2][C]
    float* stats;       // [batch][groups][4]: mean, rstd, m1, m2
    int chunks, rpc;    // pixel rows per chunk
    const float* mr;    // nullable [batch][groups][2]: (mean, rstd) kept by the forward pass (mf_groupnorm stats_out): phase 0 is skipped
};

__device__ __forceinline__ float4 gn_ld4(const GnBwdArgs& p, int b, int px, int q) {
    const int c = 4 * q;
    return c < p.c0 ? *reinterpret_cast<const float4*>(p.x0 + ((int64_t)b * p.hw + px) * p.c0 + c)
                    : *reinterpret_cast<const float4*>(p.x1 + ((int64_t)b * p.hw + px) * p.c1 + (c - p.c0));
}

template <int PHASE>      // 0: sums of x, x^2; 1: sums of dz, dz * xhat; 2: dx
__global__ __launch_bounds__(256) void gn_bwd2_kernel(const GnBwd2Args q2) {
    const GnBwdArgs& p = q2.a;
    __shared__ float red[2][4][256];
    const int C = p.c0 + p.c1, Q = C >> 2, cpg = C / p.groups;
    const int b = blockIdx.x / q2.chunks, chunk = blockIdx.x - b * q2.chunks;
    const int px0 = chunk * q2.rpc, px1 = min(p.hw, px0 + q2.rpc);
    const int t = threadIdx.x;
    const int rif = Q >= 256 ? 1 : 256 / Q;
    const int rl = Q >= 256 ? 0 : t / Q;
    for (int q = Q >= 256 ? t : t - rl * Q; q < Q; q += 256) {         // Q < 256: one pass, rif pixel rows in flight
        const bool active = rl < rif;
        float mean[4], rstd[4], gam[4], bet[4], m1[4], m2[4];
        if (PHASE >= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * q + j, g = c / cpg;
                const float* st = q2.stats + ((int64_t)b * p.groups + g) * 4;
                if (q2.mr) { mean[j] = q2.mr[((int64_t)b * p.groups + g) * 2]; rstd[j] = q2.mr[((int64_t)b * p.groups + g) * 2 + 1]; }
                else { mean[j] = st[0]; rstd[j] = st[1]; }
                if (PHASE == 2) { m1[j] = st[2]; m2[j] = st[3]; }
                gam[j] = p.gamma[c]; bet[j] = p.beta[c];
            }
        }
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
        if (active)
            for (int px = px0 + rl; px < px1; px += rif) {
                const float4 xv4 = gn_ld4(p, b, px, q);
                const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
                if (PHASE == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s0[j] += xv[j]; s1[j] += xv[j] * xv[j]; }
                } else {
                    const float4 dy4 = *reinterpret_cast<const float4*>(p.dy + ((int64_t)b * p.hw + px) * C + 4 * q);
                    const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xhat = (xv[j] - mean[j]) * rstd[j];
                        float dz = dyv[j];
                        if (p.silu) {
                            const float z = xhat * gam[j] + bet[j];
                            const float sg = 1.0f / (1.0f + expf(-z));
                            dz = dyv[j] * (sg * (1.0f + z * (1.0f - sg)));
                        }
                        if (PHASE == 1) { s0[j] += dz; s1[j] += dz * xhat; }
                        else o[j] = rstd[j] * (dz * gam[j] - m1[j] - xhat * m2[j]);
                    }
                    if (PHASE == 2) {
                        const int c = 4 * q;
                        const int64_t off = c < p.c0 ? ((int64_t)b * p.hw + px) * p.c0 + c : ((int64_t)b * p.hw + px) * p.c1 + (c - p.c0);
                        const float* ad = c < p.c0 ? p.add0 : p.add1;
                        if (ad) {
                            const float4 a4 = *reinterpret_cast<const float4*>(ad + off);
                            o[0] += a4.x; o[1] += a4.y; o[2] += a4.z; o[3] += a4.w;
                        }
                        *reinterpret_cast<float4*>((c < p.c0 ? p.dx0 : p.dx1) + off) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        if (PHASE < 2) {
            float* dst = q2.part + (((int64_t)b * q2.chunks + chunk) * 2) * C + 4 * q;
            if (rif == 1) {
                if (active) {
                    *reinterpret_cast<float4*>(dst) = make_float4(s0[0], s0[1], s0[2], s0[3]);
                    *reinterpret_cast<float4*>(dst + C) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                }
            } else {                                                    // fixed-order sum over the rif row lanes of a quad
#pragma unroll
                for (int j = 0; j < 4; ++j) { red[0][j][t] = s0[j]; red[1][j][t] = s1[j]; }
                __syncthreads();
                if (rl == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a0 = 0.0f, a1 = 0.0f;
                        for (int r = 0; r < rif; ++r) { a0 += red[0][j][r * Q + q]; a1 += red[1][j][r * Q + q]; }
                        s0[j] = a0; s1[j] = a1;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(s0[0], s0[1], s0[2], s0[3]);
                    *reinterpret_cast<float4*>(dst + C) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                }
            }
        }
    }
}

// R1 / R2: one block per (image, group).  Each wave takes channels of the group in turn, its lanes the pixel chunks (double,
// fixed order): PH 0 -> mean, rstd; PH 1 -> the image's dgamma / dbeta partials and m1, m2.
template <int PH>
__global__ __launch_bounds__(256) void gn_bwd2_reduce_kernel(const GnBwd2Args q2) {
    const GnBwdArgs& p = q2.a;
    __shared__ double red[2][4];
    const int C = p.c0 + p.c1, cpg = C / p.groups;
    const int b = blockIdx.x / p.groups, g = blockIdx.x - b * p.groups;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double ga = 0.0, gb = 0.0;                 // this wave's share of the group sums (identical in every lane)
    for (int cc = wave; cc < cpg; cc += 4) {
        const int c = g * cpg + cc;
        double a = 0.0, bb = 0.0;
        for (int ch = lane; ch < q2.chunks; ch += 64) {
            const float* src = q2.part + (((int64_t)b * q2.chunks + ch) * 2) * C + c;
            a += (double)src[0]; bb += (double)src[C];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); bb += __shfl_xor(bb, o, 64); }
        if (PH == 1) {
            if (lane == 0 && p.dgamma_part) {
                p.dbeta_part[(int64_t)b * C + c] = (float)a;
                p.dgamma_part[(int64_t)b * C + c] = (float)bb;
            }
            const double gm = (double)p.gamma[c];
            a *= gm; bb *= gm;
        }
        ga += a; gb += bb;
    }
    if (lane == 0) { red[0][wave] = ga; red[1][wave] = gb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double n = (double)p.hw * cpg;
        float* st = q2.stats + ((int64_t)b * p.groups + g) * 4;
        if (PH == 0) {
            const double mean = s / n;
            double var = ss / n - mean * mean;
            if (var < 0.0) var = 0.0;
            st[0] = (float)mean;
            st[1] = (float)(1.0 / sqrt(var + (double)p.eps));
        } else {
            st[2] = (float)(s / n);
            st[3] = (float)(ss / n);
        }
    }
}

// R2 with the parameter gradients finished in the same launch: one block per GROUP, images in turn; dgamma / dbeta of a channel are
// summed over the images in a register (fixed order) and ADDED to the gradient arena — the per-image partials and the two
// mf_colsum calls (four launches) that used to follow every GroupNorm backward are gone.
__global__ __launch_bounds__(256) void gn_bwd2_reduce_acc_kernel(const GnBwd2Args q2, float* dgamma_acc, float* dbeta_acc) {
    const GnBwdArgs& p = q2.a;
    __shared__ double red[2][4][GN_BWD_MAX_BATCH];
    const int C = p.c0 + p.c1, cpg = C / p.groups;
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 4 * GN_BWD_MAX_BATCH; i += 256) (&red[0][0][0])[i] = 0.0;
    __syncthreads();
    for (int cc = wave; cc < cpg; cc += 4) {
        const int c = g * cpg + cc;
        const double gm = (double)p.gamma[c];
        double da = 0.0, db = 0.0;
        // eight images per round: their partial loads are all in flight before the first reduction (one memory round trip per
        // round instead of one per image — this kernel runs on `groups` blocks only and was latency-bound at ~50 us)
        for (int b0 = 0; b0 < p.batch; b0 += 8) {
            double a[8], bb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = 0.0; bb[u] = 0.0;
                if (b0 + u < p.batch)
                    for (int ch = lane; ch < q2.chunks; ch += 64) {
                        const float* src = q2.part + (((int64_t)(b0 + u) * q2.chunks + ch) * 2) * C + c;
                        a[u] += (double)src[0]; bb[u] += (double)src[C];
                    }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { a[u] += __shfl_xor(a[u], o, 64); bb[u] += __shfl_xor(bb[u], o, 64); }
                if (b0 + u < p.batch) {
                    da += a[u]; db += bb[u];
                    if (lane == 0) { red[0][wave][b0 + u] += a[u] * gm; red[1][wave][b0 + u] += bb[u] * gm; }   // this wave's share of the image's group sums
                }
            }
        }
        if (lane == 0) {
            dbeta_acc[c] += (float)da;
            dgamma_acc[c] += (float)db;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < p.batch; b += 256) {
        const double s = (red[0][0][b] + red[0][1][b]) + (red[0][2][b] + red[0][3][b]);
        const double ss = (red[1][0][b] + red[1][1][b]) + (red[1][2][b] + red[1][3][b]);
        const double n = (double)p.hw * cpg;
        float* st = q2.stats + ((int64_t)b * p.groups + g) * 4;
        st[2] = (float)(s / n);
        st[3] = (float)(ss / n);
    }
}

// The fused parameter-gradient form, second version: the per-image partials of gn_bwd2_reduce_kernel<1> (batch x groups blocks)
// summed over the batch in image order and ADDED to the arena — two short launches on many blocks instead of one block per group
// walking every image and channel (51 us on 2560-channel inputs; now ~8 + 3).
__global__ __launch_bounds__(256) void gn_bwd2_batchsum_acc_kernel(const float* __restrict__ dg_part, const float* __restrict__ db_part,
                                                                   float* dgamma_acc, float* dbeta_acc, int batch, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double g = 0.0, b = 0.0;
    for (int i = 0; i < batch; ++i) { g += (double)dg_part[(int64_t)i * C + c]; b += (double)db_part[(int64_t)i * C + c]; }
    dgamma_acc[c] += (float)g;
    dbeta_acc[c] += (float)b;
}

static inline int gn_bwd2_rpc(int batch, int hw) {
    int64_t r = ((int64_t)hw * batch + 1023) / 1024;
    return (int)(r < 8 ? 8 : (r > 64 ? 64 : r));
}
static inline bool gn_bwd2_applies(int hw, int c0, int c1) { return hw >= 256 && c0 % 4 == 0 && c1 % 4 == 0 && c0 + c1 <= 4096; }

// ------------------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, everything of a row in registers (NK channels per lane, one read of x and dy, the
// next row's loads issued before this row's reductions); `rpw` consecutive rows per wave, per-block dgamma / dbeta partials
// ------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXK = 32;      // channels per lane: c <= 2048
template <int NK>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* dg_part, float* db_part, int64_t rows, int c, float eps, int rpw,
                                                            const float* __restrict__ add) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gm[NK], ag[NK], ab[NK], xv[NK], dv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int ch = lane + 64 * k;
        gm[k] = ch < c ? gamma[ch] : 0.0f;
        ag[k] = 0.0f; ab[k] = 0.0f;
    }
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rpw;
    auto load_row = [&](int64_t row, float (&xr)[NK], float (&dr)[NK]) {
        const bool ok = row < rows;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = lane + 64 * k;
            const bool in = ok && ch < c;
            xr[k] = in ? x[row * c + ch] : 0.0f;
            dr[k] = in ? dy[row * c + ch] : 0.0f;
        }
    };
    load_row(r0, xv, dv);
    const float inv_c = 1.0f / (float)c;
    for (int i = 0; i < rpw; ++i) {
        const int64_t row = r0 + i;
        if (row >= rows) break;
        float xn[NK], dn[NK];
        load_row(i + 1 < rpw ? row + 1 : rows, xn, dn);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; ++k) s += xv[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_c;
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float t = (lane + 64 * k < c) ? xv[k] - mean : 0.0f;
            ss += t * t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = 1.0f / sqrtf(ss * inv_c + eps);
        float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float xh = (lane + 64 * k < c) ? (xv[k] - mean) * rstd : 0.0f, dgm = dv[k] * gm[k];
            xv[k] = xh;
            a1 += dgm; a2 += dgm * xh;
            ag[k] += dv[k] * xh; ab[k] += dv[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
        a1 *= inv_c; a2 *= inv_c;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = lane + 64 * k;
            if (ch < c) {
                const float v = rstd * (dv[k] * gm[k] - a1 - xv[k] * a2);
                dx[row * c + ch] = add ? v + add[row * c + ch] : v;
            }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) { xv[k] = xn[k]; dv[k] = dn[k]; }
    }
    if (dg_part) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = lane + 64 * k;
            red[wave][lane] = ag[k];
            __syncthreads();
            if (wave == 0 && ch < c) dg_part[(int64_t)blockIdx.x * c + ch] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            __syncthreads();
            red[wave][lane] = ab[k];
            __syncthreads();
            if (wave == 0 && ch < c) db_part[(int64_t)blockIdx.x * c + ch] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            __syncthreads();
        }
    }
}

// rows per wave: enough blocks to fill the chip (>= 1024 where the row count allows), at most 16 rows per wave
static inline int ln_bwd_rpw(int64_t rows) {
    int64_t r = (rows + 4095) / 4096;
    return (int)(r < 2 ? 2 : (r > 16 ? 16 : r));
}

// dS = scale * P * (dP - sum_j dP P) per row (block per row)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* pr, const float* dp, float* ds, int cols, int ld, float scale) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    const float* P = pr + row * ld;
    const float* D = dp + row * ld;
    double s = 0.0;
    for (int j = threadIdx.x; j < cols; j += 256) s += (double)(P[j] * D[j]);
    const float dot = (float)block_sum(s, red);
    float* o = ds + row * ld;
    for (int j = threadIdx.x; j < ld; j += 256) o[j] = j < cols ? scale * P[j] * (D[j] - dot) : 0.0f;
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* x, const float* dy, float* dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float z = x[i], sg = 1.0f / (1.0f + expf(-z));
        dx[i] = dy[i] * (sg * (1.0f + z * (1.0f - sg)));
    }
}

// h = [a | g] (2c per row), out = a * gelu(g): dh = [dout * gelu(g) | dout * a * (Phi(g) + g phi(g))]
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* hin, const float* dout, float* dh, int64_t rows, int c) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / c;
        const int j = (int)(i - row * c);
        const float a = hin[row * 2 * c + j], g = hin[row * 2 * c + c + j], d = dout[i];
        const float Phi = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
        const float phi = 0.39894228040143267794f * expf(-0.5f * g * g);
        dh[row * 2 * c + j] = d * (g * Phi);
        dh[row * 2 * c + c + j] = d * a * (Phi + g * phi);
    }
}

// The same on a bf16 hidden tensor (the bf16x1 mode keeps FeedForward's [rows][2c] pre-activation — its largest activation — in
// bf16, as the reference's autocast does): dh comes out in bf16, the operand of ff.net.0.proj's weight / data gradients, and its
// column sums (that layer's bias gradient) leave the same pass as chunk partials (cast_colsum_stage2 finishes them).  A thread
// owns 8 value + 8 gate columns of every rpp-th row of its chunk.  The sums are of the ROUNDED values: the gradient tensor the
// bias gradient is defined on is the bf16 one.
__global__ __launch_bounds__(256) void geglu_bwd16_kernel(const unsigned short* __restrict__ hin, const float* __restrict__ dout,
                                                          unsigned short* __restrict__ dh, float* __restrict__ part, int c, int cpb, int rpp,
                                                          int64_t rows, int chunks, int64_t rows_per_chunk) {
    __shared__ float red[256 * 16];
    const int rl = threadIdx.x / cpb, cl = threadIdx.x - rl * cpb;
    const int col = (blockIdx.x * cpb + cl) * 8;
    const int ch = blockIdx.y;
    const int64_t r0 = (int64_t)ch * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    float s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = 0.0f;
    if (rl < rpp && col < c) {
        for (int64_t rr = r0 + rl; rr < r1; rr += rpp) {
            const uint4 a = *reinterpret_cast<const uint4*>(hin + rr * 2 * c + col);
            const uint4 g = *reinterpret_cast<const uint4*>(hin + rr * 2 * c + c + col);
            const float4 d0 = *reinterpret_cast<const float4*>(dout + rr * c + col), d1 = *reinterpret_cast<const float4*>(dout + rr * c + col + 4);
            const unsigned aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
            const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            unsigned oa[4], og[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ra[2], rg[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float av = q ? __uint_as_float(aw[e] & 0xffff0000u) : __uint_as_float(aw[e] << 16);
                    const float gv = q ? __uint_as_float(gw[e] & 0xffff0000u) : __uint_as_float(gw[e] << 16);
                    const float Phi = 0.5f * (1.0f + erff(gv * 0.70710678118654752440f));
                    const float phi = 0.39894228040143267794f * expf(-0.5f * gv * gv);
                    ra[q] = dv[2 * e + q] * (gv * Phi);
                    rg[q] = dv[2 * e + q] * av * (Phi + gv * phi);
                }
                oa[e] = pack_bf16x2(ra[0], ra[1]);
                og[e] = pack_bf16x2(rg[0], rg[1]);
                s[2 * e] += __uint_as_float(oa[e] << 16); s[2 * e + 1] += __uint_as_float(oa[e] & 0xffff0000u);
                s[8 + 2 * e] += __uint_as_float(og[e] << 16); s[8 + 2 * e + 1] += __uint_as_float(og[e] & 0xffff0000u);
            }
            *reinterpret_cast<uint4*>(dh + rr * 2 * c + col) = uint4{oa[0], oa[1], oa[2], oa[3]};
            *reinterpret_cast<uint4*>(dh + rr * 2 * c + c + col) = uint4{og[0], og[1], og[2], og[3]};
        }
    }
    if (!part) return;
#pragma unroll
    for (int j = 0; j < 16; ++j) red[j * 256 + threadIdx.x] = s[j];
    __syncthreads();
    for (int t = threadIdx.x; t < cpb * 16; t += 256) {
        const int g = t >> 4, j = t & 15;
        const int cc = (blockIdx.x * cpb + g) * 8 + (j & 7);
        if (cc >= c) continue;
        float acc = 0.0f;
        for (int q = 0; q < rpp; ++q) acc += red[j * 256 + q * cpb + g];
        part[(int64_t)ch * 2 * c + (j >> 3) * c + cc] = acc;
    }
}

// y[b][2h][2w][c]: y[2i][2j] = x[i][j], zeros elsewhere (data gradient of a stride-2 conv as a stride-1 conv)
__global__ __launch_bounds__(256) void zero_insert2x_kernel(const float* x, float* y, int b, int h, int w, int c4) {
    const int64_t total = (int64_t)b * 2 * h * 2 * w * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cc = (int)(i % c4);
        int64_t t = i / c4;
        const int xx = (int)(t % (2 * w)); t /= 2 * w;
        const int yy = (int)(t % (2 * h));
        const int bb = (int)(t / (2 * h));
        float4 v = make_float4(0, 0, 0, 0);
        if (!(xx & 1) && !(yy & 1)) v = reinterpret_cast<const float4*>(x)[(((int64_t)bb * h + (yy >> 1)) * w + (xx >> 1)) * c4 + cc];
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

// y[b][h][w][c] = sum of the 2x2 block of x[b][2h][2w][c] (backward of the nearest-2x upsample)
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const float* x, float* y, int b, int h, int w, int c4) {
    const int64_t total = (int64_t)b * h * w * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cc = (int)(i % c4);
        int64_t t = i / c4;
        const int xx = (int)(t % w); t /= w;
        const int yy = (int)(t % h);
        const int bb = (int)(t / h);
        const float4* src = reinterpret_cast<const float4*>(x);
        const int64_t base = (((int64_t)bb * 2 * h + 2 * yy) * 2 * w + 2 * xx) * c4 + cc;
        const float4 a = src[base], b2 = src[base + c4], c2 = src[base + (int64_t)2 * w * c4], d = src[base + (int64_t)2 * w * c4 + c4];
        reinterpret_cast<float4*>(y)[i] = make_float4((a.x + b2.x) + (c2.x + d.x), (a.y + b2.y) + (c2.y + d.y),
                                                      (a.z + b2.z) + (c2.z + d.z), (a.w + b2.w) + (c2.w + d.w));
    }
}

// d pred of loss = mean_r( w_r * mean_i (pred - target)^2 )
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* pred, const float* target, const float* w, float* d, int rows,
                                                       int64_t n) {
    const int64_t total = (int64_t)rows * n;
    const float k = 2.0f / ((float)n * (float)rows);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        d[i] = k * (w ? w[i / n] : 1.0f) * (pred[i] - target[i]);
}

// ------------------------------------------------------------------------------------------------------------
// gradient norm, clip coefficient, AdamW over flat arenas
// ------------------------------------------------------------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_stage1(const float* x, int64_t n, double* part) {
    __shared__ double red[4];
    // 16-byte loads, two of them in flight per thread, four double accumulators (fixed order: deterministic)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(x)[i + stride];
        s0 += (double)a.x * (double)a.x + (double)b.x * (double)b.x; s1 += (double)a.y * (double)a.y + (double)b.y * (double)b.y;
        s2 += (double)a.z * (double)a.z + (double)b.z * (double)b.z; s3 += (double)a.w * (double)a.w + (double)b.w * (double)b.w;
    }
    for (; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        s0 += (double)a.x * (double)a.x; s1 += (double)a.y * (double)a.y; s2 += (double)a.z * (double)a.z; s3 += (double)a.w * (double)a.w;
    }
    for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) s0 += (double)x[j] * (double)x[j];
    const double t = block_sum((s0 + s1) + (s2 + s3), red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void sumsq_stage2(const double* part, int nparts, double* out, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + t : t;
}
// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6)); norm_out receives the fp32 norm.  `unscale`
// undoes a loss scale: the arena holds gradients x S, norm = sqrt(sumsq) / S and the coefficient AdamW multiplies the stored
// gradients with becomes coef / S.
__global__ void clip_coef_kernel(const double* sumsq, float max_norm, float unscale, float* coef, float* norm_out) {
    const float norm = (float)(sqrt(sumsq[0]) * (double)unscale);
    if (norm_out) norm_out[0] = norm;
    const float c = max_norm / (norm + 1e-6f);
    coef[0] = (c < 1.0f ? c : 1.0f) * unscale;
}
// torch.optim.AdamW (single tensor, no amsgrad): decoupled decay, bias-corrected moments; g is scaled by *gscale
__global__ __launch_bounds__(256) void adamw_kernel(float* w, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* gscale) {
    const float gs = gscale ? gscale[0] : 1.0f;
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gs;
        float wi = w[i] * (1.0f - lr * wd);
        const float mi = m[i] * b1 + gi * (1.0f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        wi -= step * (mi / denom);
        w[i] = wi; m[i] = mi; v[i] = vi;
    }
}

inline int grid_for(int64_t n, int cap = 8192) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

unsigned* mf_ovf_flag_train() {
    unsigned* p = nullptr;
    (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_split_ovf_train));
    return p;
}

extern "C" int mf_sizeof_wgrad_desc(void) { return (int)sizeof(mf_wgrad_desc); }
extern "C" int mf_sizeof_groupnorm_bwd_desc(void) { return (int)sizeof(mf_groupnorm_bwd_desc); }

extern "C" int64_t mf_conv_wgrad_ws_floats(const mf_wgrad_desc* d) {
    if (!d) return 0;
    const int64_t M = (int64_t)d->batch * d->h_out * d->w_out;
    const int64_t K = (int64_t)d->kh * d->kw * (d->c0 + d->c1);
    int64_t sm = d->splitm;
    if (sm <= 0) sm = M / 256 > 64 ? 64 : (M / 256 < 1 ? 1 : M / 256);
    return sm * d->n * K;
}

// Zero a list of float ranges of one arena in ONE launch (a block per range, float4 where aligned): the gradient arena's small
// parameters (biases, norm scales / shifts), whose gradients are accumulated, while the conv / linear weights — 99.9 % of the arena —
// are WRITTEN by their first weight gradient of the step (mf_conv_wgrad with accumulate = 0) and need no clearing.
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* base, const int64_t* __restrict__ offs, const int64_t* __restrict__ lens) {
    float* q = base + offs[blockIdx.x];
    const int64_t n = lens[blockIdx.x];
    if ((reinterpret_cast<uintptr_t>(q) & 15) == 0) {
        const int64_t n4 = n / 4;
        for (int64_t i = threadIdx.x; i < n4; i += 256) reinterpret_cast<float4*>(q)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) q[i] = 0.0f;
    } else {
        for (int64_t i = threadIdx.x; i < n; i += 256) q[i] = 0.0f;
    }
}

extern "C" int mf_zero_ranges(float* base, const int64_t* offs, const int64_t* lens, int32_t count, void* stream) {
    MF_CHECK_ARG(base && offs && lens && count >= 0 && count < (1 << 30), "mf_zero_ranges: bad arguments");
    if (count == 0) return MF_OK;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, base, offs, lens);
    MF_CHECK_LAUNCH("mf_zero_ranges");
    return MF_OK;
}

static int g_wgrad_dma = getenv("MFHIP_WGRAD_NO_DMA") ? 0 : 1;      // the LDS-DMA form of the bf16-input weight gradient
extern "C" void mf_debug_set_wgrad_dma(int on) { g_wgrad_dma = on; }  // developer / test switch (the two forms are bit-identical)

extern "C" int mf_conv_wgrad(const mf_wgrad_desc* d, void* stream) {
    MF_CHECK_ARG(d != nullptr, "mf_conv_wgrad: null descriptor");
    MF_CHECK_ARG(d->dtype == MF_F32 || d->dtype == MF_F16X3 || d->dtype == MF_BF16X1 || d->dtype == MF_BF16,
                 "mf_conv_wgrad: dtype must be MF_F32, MF_F16X3, MF_BF16X1 or MF_BF16");
    const bool in16 = d->dtype == MF_BF16;        // x and dy are bf16 tensors (the pre-rounded operand copies of the bf16x1 mode)
    MF_CHECK_ARG(!in16 || (d->c0 % 8 == 0 && d->c1 % 8 == 0 && d->n % 8 == 0 && d->lda0 % 8 == 0 && d->lda1 % 8 == 0 && d->lddy % 8 == 0),
                 "mf_conv_wgrad: MF_BF16 operands need channel counts and strides that are multiples of 8");
    MF_CHECK_ARG(d->a0 && d->dy && d->dw, "mf_conv_wgrad: null a0/dy/dw");
    MF_CHECK_ARG(d->c0 > 0 && d->c1 >= 0 && (d->a1 != nullptr) == (d->c1 > 0) && d->c0 % 4 == 0 && d->c1 % 4 == 0 &&
                     d->lda0 % 4 == 0 && d->lda1 % 4 == 0,
                 "mf_conv_wgrad: channel counts / pixel strides must be multiples of 4");
    MF_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->batch >= 1 && d->h_in >= 1 && d->w_in >= 1 && d->h_out >= 1 &&
                     d->w_out >= 1 && d->n >= 1 && (d->upsample == 0 || d->upsample == 1),
                 "mf_conv_wgrad: bad geometry");
    if (!mf_aligned16(d->a0) || (d->a1 && !mf_aligned16(d->a1)) || !mf_aligned16(d->dy) || d->lddy % 4 != 0) {
        mf_set_error("mf_conv_wgrad: a0/a1/dy must be 16-byte aligned, lddy a multiple of 4");
        return MF_EALIGN;
    }
    static const bool wgrad_xcd = getenv("MFHIP_WGRAD_NO_XCD") == nullptr;            // developer A/B
    WgradArgs a{};
    a.a0 = d->a0; a.a1 = d->a1; a.C0 = d->c0; a.Ctot = d->c0 + d->c1; a.lda0 = d->lda0; a.lda1 = d->lda1;
    a.Hin = d->h_in; a.Win = d->w_in; a.Ho = d->h_out; a.Wo = d->w_out; a.HoWo = d->h_out * d->w_out;
    a.KW = d->kw; a.taps = d->kh * d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.ups = d->upsample;
    a.dy = d->dy; a.lddy = d->lddy;
    wg_fastdiv_make((unsigned)a.HoWo, &a.mul_howo, &a.sh_howo);
    wg_fastdiv_make((unsigned)a.Wo, &a.mul_wo, &a.sh_wo);
    const int64_t M64 = (int64_t)d->batch * d->h_out * d->w_out;
    MF_CHECK_ARG(M64 < (1ll << 31), "mf_conv_wgrad: M too large");
    a.M = (int)M64; a.N = d->n;
    const int64_t K = (int64_t)a.taps * a.Ctot;
    MF_CHECK_ARG(d->lddw >= K, "mf_conv_wgrad: lddw < K");
    const bool tr_form = d->dtype != MF_F32 && getenv("MFHIP_WGRAD_V1") == nullptr;
    static const bool no160 = getenv("MFHIP_WGRAD_128") != nullptr;                   // developer A/B
    // Tile form (128 x 128 / four waves, or 160 x 160 / five waves) and pixel split are chosen together by a cost model: rounds of
    // 512 resident blocks (2 per CU) x 32-pixel steps per block x the measured step time of the form (2.3 / 3.7 us: the 160 form
    // does 1.56 x the products per step and pads 320-multiples less, but has fewer, longer blocks to spread), plus the slabs'
    // write + read (N K floats each way per slab at ~4 TB/s) unless one slab adds into dW itself.
    int cap = a.M / 256 < 1 ? 1 : a.M / 256;
    if (cap > 64) cap = 64;
    if (d->ws == nullptr) cap = 1;
    else if ((int64_t)cap * a.N * K > d->ws_floats) cap = (int)(d->ws_floats / ((int64_t)a.N * K)) < 1 ? 1 : (int)(d->ws_floats / ((int64_t)a.N * K));
    auto tiles_of = [&](int w) { return (int64_t)((a.N + w - 1) / w) * ((a.Ctot + w - 1) / w) * a.taps; };
    auto model = [&](int w, double step_us, int* best_c) {
        double best = 1e300;
        *best_c = 1;
        for (int c = 1; c <= cap; ++c) {
            const int64_t blocks = tiles_of(w) * c, rounds = (blocks + 511) / 512;
            const int64_t steps = (((int64_t)a.M + c - 1) / c + WG_BP - 1) / WG_BP;
            const double slab_us = c == 1 ? 0.0 : (double)c * a.N * K * 8.0 / 4.0e6 + 4.0;
            const double cost = (double)rounds * steps * step_us + slab_us;
            if (cost < best * 0.98) { best = cost; *best_c = c; }                        // a larger split has to pay for itself
        }
        return best;
    };
    int c128 = 1, c160 = 1;
    const double cost128 = model(WG_BN, 2.3, &c128);
    const double cost160 = tr_form && !no160 ? model(W160, 3.7, &c160) : 1e300;
    const bool t160 = cost160 < cost128;
    const int tile_w = t160 ? W160 : WG_BN;
    a.tiles_n = (a.N + tile_w - 1) / tile_w;
    a.tiles_k_per_tap = (a.Ctot + tile_w - 1) / tile_w;
    const int64_t tiles = tiles_of(tile_w);
    int sm = d->splitm;
    if (sm <= 0) {
        if (!tr_form) {
            sm = (int)((1024 + tiles - 1) / tiles);
            if (sm > cap) sm = cap;
        } else {
            sm = t160 ? c160 : c128;
        }
    }
    MF_CHECK_ARG(sm == 1 || (d->ws && (int64_t)sm * a.N * K <= d->ws_floats), "mf_conv_wgrad: split-M=%d needs %lld workspace floats", sm,
                 (long long)sm * a.N * K);
    a.m_per_split = ((a.M + sm - 1) / sm + WG_BP - 1) / WG_BP * WG_BP;
    a.splitm = (a.M + a.m_per_split - 1) / a.m_per_split;
    // one slab: the 16-bit kernel adds into dW itself (single writer per element: still deterministic)
    const bool direct = a.splitm == 1 && (!d->accumulate || tr_form);
    a.acc_out = direct && d->accumulate;
    a.out = direct ? d->dw : d->ws;
    a.ldo = direct ? d->lddw : K;
    a.slab = (int64_t)a.N * K;
    a.slabs_xcd = wgrad_xcd ? a.splitm / 8 * 8 : 0;
    MF_CHECK_ARG(direct || d->ws, "mf_conv_wgrad: accumulate needs a workspace");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)a.splitm);
    const dim3 grid1((unsigned)(tiles * a.splitm));
    static const bool old_form = getenv("MFHIP_WGRAD_V1") != nullptr;        // developer A/B: the first version of the 16-bit forms
    if (d->dtype == MF_F32) hipLaunchKernelGGL(conv_wgrad_kernel<MF_F32>, grid, dim3(256), 0, s, a);
    else if (old_form && !in16) {
        if (d->dtype == MF_BF16X1) hipLaunchKernelGGL(conv_wgrad_kernel<MF_BF16X1>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_wgrad_kernel<MF_F16X3>, grid, dim3(256), 0, s, a);
    } else if (t160) {
        static const bool attr160 = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr160_kernel<MF_F16X3, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * W160_PLANE);
            return true;
        }();
        (void)attr160;
        // (one LDS stage + a second barrier per step measured the same as two stages: 383 vs 371 us on 8 x 64 x 64 320 -> 320 3x3)
        if (in16) hipLaunchKernelGGL((conv_wgrad_tr160_kernel<MF_BF16X1, 2, true>), grid1, dim3(320), 2 * 2 * W160_PLANE, s, a);
        else if (d->dtype == MF_BF16X1) hipLaunchKernelGGL((conv_wgrad_tr160_kernel<MF_BF16X1, 2>), grid1, dim3(320), 2 * 2 * W160_PLANE, s, a);
        else hipLaunchKernelGGL((conv_wgrad_tr160_kernel<MF_F16X3, 2>), grid1, dim3(320), 2 * 4 * W160_PLANE, s, a);
    } else if (in16 && g_wgrad_dma && (!a.a1 || a.C0 % WG_BC == 0) && (int64_t)a.M * a.lddy * 2 < 0x7fffffffll &&
               (int64_t)(a.M / a.HoWo) * a.Hin * a.Win * (a.lda0 > a.lda1 ? a.lda0 : a.lda1) * 2 < 0x7fffffffll) {
        hipLaunchKernelGGL(conv_wgrad_dma_kernel, grid1, dim3(256), WD_NST * 2 * WT_PLANE, s, a);
    } else if (in16) {
        hipLaunchKernelGGL((conv_wgrad_tr_kernel<MF_BF16X1, true>), grid1, dim3(256), 2 * 2 * WT_PLANE, s, a);
    } else if (d->dtype == MF_BF16X1) {
        hipLaunchKernelGGL(conv_wgrad_tr_kernel<MF_BF16X1>, grid1, dim3(256), 2 * 2 * WT_PLANE, s, a);
    } else {
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr_kernel<MF_F16X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      2 * 4 * WT_PLANE);
            return true;
        }();
        (void)attr;
        hipLaunchKernelGGL(conv_wgrad_tr_kernel<MF_F16X3>, grid1, dim3(256), 2 * 4 * WT_PLANE, s, a);
    }
    MF_CHECK_LAUNCH("mf_conv_wgrad");
    if (!direct) {
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid_for((int64_t)a.N * K)), dim3(256), 0, s, d->ws, a.splitm, a.slab, d->dw, d->lddw, a.N,
                           (int)K, d->accumulate);
        MF_CHECK_LAUNCH("mf_conv_wgrad(sum slabs)");
    }
    return MF_OK;
}

extern "C" int mf_split_pack(const float* w, int64_t ldw, void* out, int64_t rows, int32_t k, int32_t dtype, void* stream) {
    MF_CHECK_ARG(w && out && rows >= 1 && k >= 1 && ldw >= k, "mf_split_pack: bad arguments");
    MF_CHECK_ARG(dtype == MF_F16X3 || dtype == MF_BF16X3, "mf_split_pack: dtype must be MF_F16X3 or MF_BF16X3");
    MF_CHECK_ARG((((uintptr_t)out) & 7) == 0, "mf_split_pack: out must be 8-byte aligned");
    const int kp = (k + 31) / 32 * 32;
    const unsigned blocks = grid_for(rows * (kp / 4));
    if (dtype == MF_F16X3)
        hipLaunchKernelGGL(split_pack_kernel<MF_F16X3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, ldw, (unsigned short*)out, rows, k, kp);
    else
        hipLaunchKernelGGL(split_pack_kernel<MF_BF16X3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, ldw, (unsigned short*)out, rows, k, kp);
    MF_CHECK_LAUNCH("mf_split_pack");
    return MF_OK;
}

extern "C" int mf_transpose(const float* x, float* y, int32_t nz, int32_t rows, int32_t cols, int64_t ldx, int64_t ldy, int64_t zsx,
                            int64_t zsy, void* stream) {
    MF_CHECK_ARG(x && y && nz >= 1 && rows >= 1 && cols >= 1 && nz < 65536, "mf_transpose: bad arguments");
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, nz);
    MF_CHECK_ARG(grid.y < 65536, "mf_transpose: too many rows");
    hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (void*)y, rows, cols, ldx, ldy, zsx, zsy);
    MF_CHECK_LAUNCH("mf_transpose");
    return MF_OK;
}

extern "C" int mf_transpose_bf16(const float* x, void* y, int32_t nz, int32_t rows, int32_t cols, int64_t ldx, int64_t ldy, int64_t zsx,
                                 int64_t zsy, void* stream) {
    MF_CHECK_ARG(x && y && nz >= 1 && rows >= 1 && cols >= 1 && nz < 65536, "mf_transpose_bf16: bad arguments");
    if (cols % 4 == 0 && ldx % 4 == 0 && zsx % 4 == 0 && ldy % 8 == 0 && zsy % 8 == 0 && mf_aligned16(x) && mf_aligned16(y) && rows >= 32 &&
        (rows + 63) / 64 < 65536) {
        hipLaunchKernelGGL(transpose_f2b_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, nz), dim3(256), 0, (hipStream_t)stream, x,
                           (unsigned short*)y, rows, cols, ldx, ldy, zsx, zsy);
        MF_CHECK_LAUNCH("mf_transpose_bf16");
        return MF_OK;
    }
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, nz);
    MF_CHECK_ARG(grid.y < 65536, "mf_transpose_bf16: too many rows");
    hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, ldx, ldy, zsx, zsy);
    MF_CHECK_LAUNCH("mf_transpose_bf16");
    return MF_OK;
}

extern "C" int mf_transpose_bf16_bf16(const void* x, void* y, int32_t nz, int32_t rows, int32_t cols, int64_t ldx, int64_t ldy, int64_t zsx,
                                      int64_t zsy, void* stream) {
    MF_CHECK_ARG(x && y && nz >= 1 && rows >= 1 && cols >= 8 && nz < 65536 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && zsx % 8 == 0 &&
                     zsy % 8 == 0 && ldy >= (rows + 7) / 8 * 8,
                 "mf_transpose_bf16_bf16: cols, the leading dimensions and the batch strides are multiples of 8; ldy covers rows rounded up to 8");
    if (!mf_aligned16(x) || !mf_aligned16(y)) {
        mf_set_error("mf_transpose_bf16_bf16: pointers must be 16-byte aligned");
        return MF_EALIGN;
    }
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, nz);
    MF_CHECK_ARG(grid.y < 65536, "mf_transpose_bf16_bf16: too many rows");
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, rows, cols, ldx,
                       ldy, zsx, zsy);
    MF_CHECK_LAUNCH("mf_transpose_bf16_bf16");
    return MF_OK;
}

extern "C" int64_t mf_colsum_ws_floats(int32_t segs, int64_t rows_per_seg, int32_t n) {
    int64_t chunks = (rows_per_seg + 63) / 64;
    if (chunks > 64) chunks = 64;
    if (chunks < 1) chunks = 1;
    return (int64_t)segs * chunks * n;
}

extern "C" int mf_colsum(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t segs, int64_t rows_per_seg, int32_t n,
                         int32_t accumulate, float* ws, void* stream) {
    MF_CHECK_ARG(x && out && ws && segs >= 1 && rows_per_seg >= 1 && n >= 1, "mf_colsum: bad arguments");
    int64_t chunks = (rows_per_seg + 63) / 64;
    if (chunks > 64) chunks = 64;
    const int64_t rpc = (rows_per_seg + chunks - 1) / chunks;
    MF_CHECK_ARG((int64_t)segs * chunks < 65536, "mf_colsum: too many segments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_stage1, dim3((n + 63) / 64, (unsigned)(segs * chunks)), dim3(256), 0, s, x, ldx, ws, n, rows_per_seg,
                       (int)chunks, rpc);
    MF_CHECK_LAUNCH("mf_colsum");
    hipLaunchKernelGGL(colsum_stage2, dim3(grid_for((int64_t)segs * n, 1024)), dim3(256), 0, s, ws, out, ldo, n, segs, (int)chunks, accumulate);
    MF_CHECK_LAUNCH("mf_colsum(stage 2)");
    return MF_OK;
}

static void cast_colsum_plan(int32_t segs, int64_t rows_per_seg, int32_t n, int& ncb, int& cpb, int& rpp, int64_t& chunks) {
    const int n8 = n / 8;
    ncb = (n8 + 255) / 256;
    cpb = (n8 + ncb - 1) / ncb;                    // 8-column groups per block (<= 256)
    rpp = 256 / cpb;
    if (rpp > 32) rpp = 32;
    chunks = (512 + segs - 1) / segs;              // ~512 blocks over the row dimension
    const int64_t most = (rows_per_seg + rpp - 1) / rpp;
    if (chunks > most) chunks = most;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
}

extern "C" int64_t mf_cast_bf16_colsum_ws_floats(int32_t segs, int64_t rows_per_seg, int32_t n) {
    int ncb, cpb, rpp; int64_t chunks;
    cast_colsum_plan(segs, rows_per_seg, n, ncb, cpb, rpp, chunks);
    return (int64_t)segs * chunks * n;
}

extern "C" int mf_cast_bf16_colsum(const float* x, void* out16, int32_t segs, int64_t rows_per_seg, int32_t n, float* seg_out, int64_t ldo,
                                   int32_t seg_accumulate, float* tot_out, int32_t tot_accumulate, float* ws, void* stream) {
    MF_CHECK_ARG(x && out16 && ws && segs >= 1 && segs <= CC2_MAX_SEGS && rows_per_seg >= 1 && n >= 8 && n % 8 == 0 && (seg_out || tot_out),
                 "mf_cast_bf16_colsum: bad arguments (n a multiple of 8, at most 32 segments, at least one of seg_out / tot_out)");
    if (!mf_aligned16(x) || !mf_aligned16(out16)) {
        mf_set_error("mf_cast_bf16_colsum: x / out16 must be 16-byte aligned");
        return MF_EALIGN;
    }
    int ncb, cpb, rpp; int64_t chunks;
    cast_colsum_plan(segs, rows_per_seg, n, ncb, cpb, rpp, chunks);
    MF_CHECK_ARG((int64_t)segs * chunks < 65536, "mf_cast_bf16_colsum: too many segments");
    const int64_t rpc = (rows_per_seg + chunks - 1) / chunks;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cast_colsum_stage1, dim3(ncb, (unsigned)(segs * chunks)), dim3(256), 0, s, x, (unsigned short*)out16, ws, n, cpb, rpp,
                       rows_per_seg, (int)chunks, rpc);
    MF_CHECK_LAUNCH("mf_cast_bf16_colsum");
    hipLaunchKernelGGL(cast_colsum_stage2, dim3((n + 15) / 16), dim3(256), 0, s, ws, seg_out, ldo, seg_accumulate, tot_out, tot_accumulate, n, segs,
                       (int)chunks);
    MF_CHECK_LAUNCH("mf_cast_bf16_colsum(stage 2)");
    return MF_OK;
}

extern "C" int mf_groupnorm_bwd_streams(int32_t batch, int32_t hw, int32_t c0, int32_t c1) {
    return gn_bwd2_applies(hw, c0, c1) && batch <= GN_BWD_MAX_BATCH;
}

extern "C" int mf_groupnorm_bwd(const mf_groupnorm_bwd_desc* d, void* stream) {
    MF_CHECK_ARG(d && d->x0 && d->dy && d->gamma && d->beta && d->dx0, "mf_groupnorm_bwd: null pointer");
    const int C = d->c0 + d->c1;
    MF_CHECK_ARG(d->c0 > 0 && d->c1 >= 0 && (d->x1 != nullptr) == (d->c1 > 0) && (d->dx1 != nullptr) == (d->c1 > 0),
                 "mf_groupnorm_bwd: bad segments");
    MF_CHECK_ARG(d->groups >= 1 && C % d->groups == 0 && C / d->groups <= 256, "mf_groupnorm_bwd: groups must divide channels, <= 256 channels per group");
    MF_CHECK_ARG((d->dgamma_part != nullptr) == (d->dbeta_part != nullptr), "mf_groupnorm_bwd: dgamma / dbeta partials go together");
    MF_CHECK_ARG((d->dgamma_acc != nullptr) == (d->dbeta_acc != nullptr) && !(d->dgamma_acc && d->dgamma_part),
                 "mf_groupnorm_bwd: dgamma_acc / dbeta_acc go together and replace the partials");
    MF_CHECK_ARG(!d->dgamma_acc || (d->ws && mf_groupnorm_bwd_streams(d->batch, d->hw, d->c0, d->c1)),
                 "mf_groupnorm_bwd: dgamma_acc / dbeta_acc need the streaming form (mf_groupnorm_bwd_streams) and its workspace");
    GnBwdArgs a{};
    a.x0 = d->x0; a.x1 = d->x1; a.c0 = d->c0; a.c1 = d->c1; a.dy = d->dy; a.gamma = d->gamma; a.beta = d->beta;
    a.dx0 = d->dx0; a.dx1 = d->dx1; a.add0 = d->add0; a.add1 = d->add1; a.dgamma_part = d->dgamma_part; a.dbeta_part = d->dbeta_part;
    a.batch = d->batch; a.hw = d->hw; a.groups = d->groups; a.silu = d->silu; a.eps = d->eps;
    hipStream_t s = (hipStream_t)stream;
    if (d->ws && gn_bwd2_applies(d->hw, d->c0, d->c1)) {
        MF_CHECK_ARG(mf_aligned16(d->x0) && mf_aligned16(d->x1) && mf_aligned16(d->dy) && mf_aligned16(d->dx0) && mf_aligned16(d->dx1) &&
                     mf_aligned16(d->ws) && mf_aligned16(d->add0) && mf_aligned16(d->add1), "mf_groupnorm_bwd: tensors must be 16-byte aligned");
        GnBwd2Args q{};
        q.a = a;
        q.rpc = gn_bwd2_rpc(d->batch, d->hw);
        q.chunks = (d->hw + q.rpc - 1) / q.rpc;
        q.stats = d->ws;
        q.part = d->ws + (((int64_t)d->batch * d->groups * 4 + 3) & ~3ll);
        q.mr = d->stats_in;
        const dim3 grid((unsigned)(d->batch * q.chunks));
        if (!q.mr) {       // no statistics from the forward pass: recompute them (one more read of x)
            hipLaunchKernelGGL(gn_bwd2_kernel<0>, grid, dim3(256), 0, s, q);
            hipLaunchKernelGGL(gn_bwd2_reduce_kernel<0>, dim3((unsigned)(d->batch * d->groups)), dim3(256), 0, s, q);
        }
        hipLaunchKernelGGL(gn_bwd2_kernel<1>, grid, dim3(256), 0, s, q);
        static const bool acc_v1 = getenv("MFHIP_GN_ACC_V1") != nullptr;       // developer A/B: one block per group over all images
        if (d->dgamma_acc && acc_v1)
            hipLaunchKernelGGL(gn_bwd2_reduce_acc_kernel, dim3((unsigned)d->groups), dim3(256), 0, s, q, d->dgamma_acc, d->dbeta_acc);
        else if (d->dgamma_acc) {
            float* pg = q.part + (int64_t)d->batch * q.chunks * 2 * C;                 // [2][batch][C] per-image partials (workspace tail)
            q.a.dgamma_part = pg; q.a.dbeta_part = pg + (int64_t)d->batch * C;
            hipLaunchKernelGGL(gn_bwd2_reduce_kernel<1>, dim3((unsigned)(d->batch * d->groups)), dim3(256), 0, s, q);
            hipLaunchKernelGGL(gn_bwd2_batchsum_acc_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, q.a.dgamma_part, q.a.dbeta_part,
                               d->dgamma_acc, d->dbeta_acc, d->batch, C);
            q.a.dgamma_part = nullptr; q.a.dbeta_part = nullptr;
        } else hipLaunchKernelGGL(gn_bwd2_reduce_kernel<1>, dim3((unsigned)(d->batch * d->groups)), dim3(256), 0, s, q);
        hipLaunchKernelGGL(gn_bwd2_kernel<2>, grid, dim3(256), 0, s, q);
        MF_CHECK_LAUNCH("mf_groupnorm_bwd(streaming)");
        return MF_OK;
    }
    hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3((unsigned)(d->batch * d->groups)), dim3(256), 0, s, a);
    MF_CHECK_LAUNCH("mf_groupnorm_bwd");
    return MF_OK;
}

extern "C" int64_t mf_groupnorm_bwd_ws_floats(int32_t batch, int32_t hw, int32_t channels, int32_t groups) {
    if (batch < 1 || hw < 1 || channels < 1 || groups < 1) return 0;
    const int rpc = gn_bwd2_rpc(batch, hw);
    const int64_t chunks = (hw + rpc - 1) / rpc;
    return (((int64_t)batch * groups * 4 + 3) & ~3ll) + (int64_t)batch * chunks * 2 * channels + (int64_t)2 * batch * channels;
}

extern "C" int64_t mf_layernorm_bwd_parts(int64_t rows) {
    const int rpw = ln_bwd_rpw(rows);
    return (rows + 4 * rpw - 1) / (4 * rpw);
}

extern "C" int mf_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma_part, float* dbeta_part,
                                int64_t rows, int32_t c, float eps, const float* add, void* stream) {
    MF_CHECK_ARG(x && dy && gamma && dx && rows >= 1 && c >= 1 && c <= 64 * LN_MAXK, "mf_layernorm_bwd: bad arguments (c <= %d)", 64 * LN_MAXK);
    MF_CHECK_ARG((dgamma_part != nullptr) == (dbeta_part != nullptr), "mf_layernorm_bwd: dgamma / dbeta partials go together");
    const int rpw = ln_bwd_rpw(rows);
    const dim3 grid((unsigned)mf_layernorm_bwd_parts(rows));
    const int nk = (c + 63) / 64;
#define MF_LN_BWD(NK)                                                                                                              \
    hipLaunchKernelGGL(layernorm_bwd_kernel<NK>, grid, dim3(256), 0, (hipStream_t)stream, x, dy, gamma, dx, dgamma_part, dbeta_part, \
                       rows, c, eps, rpw, add)
    if (nk <= 1) MF_LN_BWD(1);
    else if (nk <= 2) MF_LN_BWD(2);
    else if (nk <= 5) MF_LN_BWD(5);
    else if (nk <= 10) MF_LN_BWD(10);
    else if (nk <= 20) MF_LN_BWD(20);
    else MF_LN_BWD(32);
#undef MF_LN_BWD
    MF_CHECK_LAUNCH("mf_layernorm_bwd");
    return MF_OK;
}

extern "C" int mf_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int32_t cols, int32_t ld, float scale, void* stream) {
    MF_CHECK_ARG(p && dp && ds && rows >= 1 && cols >= 1 && ld >= cols && rows < (1ll << 31), "mf_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, ds, cols, ld, scale);
    MF_CHECK_LAUNCH("mf_softmax_bwd");
    return MF_OK;
}

extern "C" int mf_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    MF_CHECK_ARG(x && dy && dx && n >= 0, "mf_silu_bwd: bad arguments");
    if (n == 0) return MF_OK;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
    MF_CHECK_LAUNCH("mf_silu_bwd");
    return MF_OK;
}

extern "C" int mf_geglu_bwd(const float* h, const float* dout, float* dh, int64_t rows, int32_t c, void* stream) {
    MF_CHECK_ARG(h && dout && dh && rows >= 1 && c >= 1, "mf_geglu_bwd: bad arguments");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * c)), dim3(256), 0, (hipStream_t)stream, h, dout, dh, rows, c);
    MF_CHECK_LAUNCH("mf_geglu_bwd");
    return MF_OK;
}

static void geglu_bwd16_plan(int64_t rows, int32_t c, int& ncb, int& cpb, int& rpp, int64_t& chunks) {
    const int c8 = c / 8;
    ncb = (c8 + 127) / 128;                        // a thread keeps 16 sums: at most 128 column groups per block
    cpb = (c8 + ncb - 1) / ncb;
    rpp = 256 / cpb;
    if (rpp > 32) rpp = 32;
    chunks = (1024 + ncb - 1) / ncb;
    const int64_t most = (rows + rpp - 1) / rpp;
    if (chunks > most) chunks = most;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
}

extern "C" int64_t mf_geglu_bwd_bf16_ws_floats(int64_t rows, int32_t c) {
    int ncb, cpb, rpp; int64_t chunks;
    geglu_bwd16_plan(rows, c, ncb, cpb, rpp, chunks);
    return chunks * 2 * c;
}

extern "C" int mf_geglu_bwd_bf16(const void* h16, const float* dout, void* dh16, int64_t rows, int32_t c, float* bias_grad, float* ws,
                                 void* stream) {
    MF_CHECK_ARG(h16 && dout && dh16 && rows >= 1 && c >= 8 && c % 8 == 0 && (!bias_grad || ws),
                 "mf_geglu_bwd_bf16: bad arguments (c a multiple of 8; the bias gradient needs the workspace)");
    if (!mf_aligned16(h16) || !mf_aligned16(dout) || !mf_aligned16(dh16)) {
        mf_set_error("mf_geglu_bwd_bf16: pointers must be 16-byte aligned");
        return MF_EALIGN;
    }
    int ncb, cpb, rpp; int64_t chunks;
    geglu_bwd16_plan(rows, c, ncb, cpb, rpp, chunks);
    const int64_t rpc = (rows + chunks - 1) / chunks;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(geglu_bwd16_kernel, dim3(ncb, (unsigned)chunks), dim3(256), 0, s, (const unsigned short*)h16, dout, (unsigned short*)dh16,
                       bias_grad ? ws : nullptr, c, cpb, rpp, rows, (int)chunks, rpc);
    MF_CHECK_LAUNCH("mf_geglu_bwd_bf16");
    if (bias_grad) {
        hipLaunchKernelGGL(cast_colsum_stage2, dim3((2 * c + 15) / 16), dim3(256), 0, s, ws, (float*)nullptr, (int64_t)0, 0, bias_grad, 1, 2 * c, 1,
                           (int)chunks);
        MF_CHECK_LAUNCH("mf_geglu_bwd_bf16(bias gradient)");
    }
    return MF_OK;
}

extern "C" int mf_zero_insert2x(const float* x, float* y, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
    MF_CHECK_ARG(x && y && batch >= 1 && h >= 1 && w >= 1 && c >= 4 && c % 4 == 0, "mf_zero_insert2x: bad arguments (c %% 4 == 0)");
    hipLaunchKernelGGL(zero_insert2x_kernel, dim3(grid_for((int64_t)batch * 4 * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, y,
                       batch, h, w, c / 4);
    MF_CHECK_LAUNCH("mf_zero_insert2x");
    return MF_OK;
}

extern "C" int mf_sumpool2x2(const float* x, float* y, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
    MF_CHECK_ARG(x && y && batch >= 1 && h >= 1 && w >= 1 && c >= 4 && c % 4 == 0, "mf_sumpool2x2: bad arguments (c %% 4 == 0)");
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid_for((int64_t)batch * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, y, batch, h,
                       w, c / 4);
    MF_CHECK_LAUNCH("mf_sumpool2x2");
    return MF_OK;
}

extern "C" int mf_mse_grad(const float* pred, const float* target, const float* weights, float* dpred, int32_t rows, int64_t n,
                           void* stream) {
    MF_CHECK_ARG(pred && target && dpred && rows >= 1 && n >= 1, "mf_mse_grad: bad arguments");
    hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_for((int64_t)rows * n)), dim3(256), 0, (hipStream_t)stream, pred, target, weights, dpred,
                       rows, n);
    MF_CHECK_LAUNCH("mf_mse_grad");
    return MF_OK;
}

extern "C" int64_t mf_sumsq_ws_doubles(void) { return SUMSQ_BLOCKS; }

extern "C" int mf_sumsq(const float* x, int64_t n, double* out, int32_t accumulate, double* ws, void* stream) {
    MF_CHECK_ARG(x && out && ws && n >= 0, "mf_sumsq: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = grid_for(n, SUMSQ_BLOCKS);
    hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, s, x, n, ws);
    MF_CHECK_LAUNCH("mf_sumsq");
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, s, ws, blocks, out, accumulate);
    MF_CHECK_LAUNCH("mf_sumsq(stage 2)");
    return MF_OK;
}

extern "C" int mf_clip_coef(const double* sumsq, float max_norm, float unscale, float* coef, float* norm_out, void* stream) {
    MF_CHECK_ARG(sumsq && coef && unscale > 0.0f, "mf_clip_coef: null pointer or unscale <= 0");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, unscale, coef, norm_out);
    MF_CHECK_LAUNCH("mf_clip_coef");
    return MF_OK;
}

extern "C" int mf_adamw(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int32_t step, const float* grad_scale, void* stream) {
    MF_CHECK_ARG(w && g && m && v && n >= 0 && step >= 1, "mf_adamw: bad arguments");
    if (n == 0) return MF_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    MF_CHECK_LAUNCH("mf_adamw");
    return MF_OK;
}
