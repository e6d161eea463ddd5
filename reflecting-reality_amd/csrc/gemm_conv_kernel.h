// Implicit-GEMM convolution / linear / strided-batched NT GEMM for gfx950 (MI355X) on MFMA.
//
// One kernel family serves every contraction of the MirrorFusion hot path (see include/mfhip.h,
// mf_gemm_conv): conv3x3 (stride 1/2, symmetric or asymmetric zero padding, optional fused
// nearest-2x upsample, optional two-tensor channel concat), conv1x1 / nn.Linear, and the batched
// QK^T / PV products of the unfused attention path.
//
// Design (CDNA4):
//   * M = batch*Ho*Wo pixels, N = Cout, K = kh*kw*Cin.  Activations NHWC, weights [N][K].
//   * A K-tile is 128 bytes of K per row in BOTH precisions (64 bf16 / 32 f32), so staging, swizzle and
//     fragment addressing are byte-identical for bf16 and f32; only the MFMA differs:
//       bf16: one v_mfma_f32_32x32x16_bf16 per 16-byte fragment (k = 8h + j),
//       f32 : four v_mfma_f32_32x32x2_f32 per 16-byte fragment (element e covers k = 4h + e of the
//             8-wide step; A and W use the same k permutation so the dot product is unchanged).
//   * Staging is LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write.  The DMA destination
//     is lane-linear (wave base + lane*16 B), i.e. 8 lanes fill one 128-B row, so the XOR swizzle
//     chunk ^ ((row >> 1) & 7) is applied to the per-lane SOURCE address (lane l fetches logical chunk
//     (l & 7) ^ key) and again on the fragment read: conflict-free ds_read_b128 for the 32x32 fragment
//     pattern.  Zero padding / tails: a lane whose element is out of range reads a 16-byte zero page, so the
//     gather has no divergent branches.  Two LDS stages: the DMA of tile t+1 flies while tile t is multiplied.
//   * fp32 accumulate.  Epilogue: each wave transposes its accumulators through LDS (32-row slabs) so that
//     every lane owns 8 consecutive output channels of one pixel: bias / temb / residual reads and the store
//     are 16-byte vectors; alpha*(acc + bias + temb) + res0 + res1, SiLU, dtype cast are fused.
//   * split-K (deterministic fp32 slabs + a reduce kernel running the same epilogue) for small-spatial layers
//     (8x8 / 16x16 latents) that cannot fill 256 CUs otherwise; XCD-aware block order (consecutive tiles of
//     one XCD share A rows and stay in that XCD's L2).
//   * activations stored as fp32 with bf16 compute (A_F32) use a register-staged path that converts on load.
//
// Translation units: this header holds the kernel template; gemm_*.hip each instantiate one group of tiles (so the library builds
// in parallel), conv_halo.hip holds the resident-patch 3x3 kernels, gemm_conv.hip the host entry point (mf_gemm_conv).
#pragma once
#include "mf_common.h"

#include <type_traits>

// developer switch (A/B builds): 0 restores the one-barrier-per-tile consumer loop without the cross-tile fragment pipeline
#ifndef MF_XTAP
#define MF_XTAP 1
#endif
// developer switch: 0 issues the dx-reuse A window together with the tap that opens its group (round 4) instead of one tap earlier
#ifndef MF_AEARLY
#define MF_AEARLY 1
#endif

namespace mfgemm {

static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[16];   // zero-initialised module global (one per translation unit)

// Developer build only (-DMF_STAMPS: tools/stamps.py): wave 0 of every block (and the first staging wave of a warp-specialised
// block, slots 8+) writes the 100 MHz real-time counter at its phase boundaries, so that a launch's time can be split into
// ramp / prologue / first DMA round trip / main loop / epilogue per block.  Compiled out of the product library.
#ifdef MF_STAMPS
#define MF_STAMP(slot) do { if (p.stamps) { const int t_ = (int)threadIdx.x; \
    if (t_ == 0) p.stamps[(size_t)blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    else if (t_ == (int)blockDim.x - 256) p.stamps[(size_t)blockIdx.x * 32 + 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define MF_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")     // slot 5 = "the epilogue's stores have left"
// who waits for whom in the warp-specialised main loops: shader-clock sums per wave (slots 13-15 of wave 0 / the first staging wave)
#define MF_CLK(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define MF_SUM(acc, a, b) acc += (b) - (a)
#define MF_PUT(slot, v) do { if (p.stamps) { const int t_ = (int)threadIdx.x; \
    if (t_ == 0) p.stamps[(size_t)blockIdx.x * 32 + (slot)] = (v); \
    else if (t_ == (int)blockDim.x - 256) p.stamps[(size_t)blockIdx.x * 32 + 16 + (slot)] = (v); } } while (0)
#else
#define MF_STAMP(slot) do { } while (0)
#define MF_STAMP_DRAIN() do { } while (0)
#define MF_CLK(v) do { } while (0)
#define MF_SUM(acc, a, b) do { } while (0)
#define MF_PUT(slot, v) do { } while (0)
#endif

struct GemmArgs {
    const char* a0; const char* a1;
    int C0, Ctot;
    int ld0b, ld1b;          // pixel strides in BYTES of the A storage dtype
    int Hin, Win, Ho, Wo, HoWo, KW, stride, pad_t, pad_l, ups;
    const char* w; int64_t ldw;
    int M, N, K;
    int zdiv; int64_t a_zs_o, a_zs_i, w_zs_o, w_zs_i, o_zs_o, o_zs_i;
    int splitk, kt_per_split, nkt, nz;
    float* ws;
    unsigned* sk_tickets;    // in-launch split-K combine: one arrival counter per output tile (zero between launches), or nullptr
    const float* bias; int bias_mode;
    const float* rs; const float* cs; int64_t rs_zs, cs_zs;   // dequantisation: acc * rs[m] * cs[n]
    const float* temb; int64_t ld_temb;
    const char* res0; int res0_dt; int64_t ld_res0;
    const char* res1; int res1_dt; int64_t ld_res1;
    int res1_rows;           // > 0: res1 has this many rows, output row m adds row m % res1_rows (shared by batch replicas)
    float alpha; int act;
    char* out; int out_dt; int64_t ldc;
    int tiles_n, tiles_m, ord_mfast, ord_pw, nblk, vec_ok, fast, dbg_no_res_pre;
    int dbg_epi;             // developer switch (MFHIP_DBG_EPI, stamped builds): bit 0 skip the epilogue's global stores, bit 1 skip its slab reads
    int pointwise;           // kh = kw = 1, stride 1, no padding, no upsample, same extent: input pixel index == output row
    // Warp-specialised tiles: the per-column epilogue operands of the block's BN columns (bias, the folded LayerNorm's column sums,
    // the time-embedding row of each image the tile touches) are fetched into LDS by the staging waves' FIRST DMAs, so the
    // epilogue's item loops contain no global load: loads and stores share vmcnt and return in order, and a bias load issued after
    // the previous item's store used to wait for that store's round trip (~1.5 us per 64-item chunk, stamped).
    // prologue without integer divisions (each costs ~30 VALU instructions and a kernel's start-up held a dozen of them): log2 of the
    // extents that are powers of two (-1: not one — the division stays), and a magic multiplier for the dx-reuse window's row pitch
    int howo_sh, wo_sh, ho_sh;      // Ho * Wo, Wo, Ho
    unsigned wfr_magic;             // floor(2^20 / (min(Wo, 256) + 2)) + 1: x / pitch == (x * magic) >> 20 for x < 4096 (dx-reuse window rows)
    int epb;                 // 1: the rows are staged (the host checked the geometry: epb_layout below)
    int epb_sh;              // log2(Ho * Wo) (a power of two whenever temb is staged): image of output row m = m >> epb_sh
    unsigned* ovf;           // device flag raised when an MF_F16X3 operand exceeded the fp16 range (mf_common.h)
    unsigned long long* stamps;   // developer build (-DMF_STAMPS) only: [blocks][32] phase time stamps, or nullptr
    // LayerNorm folded into this GEMM (warp-specialised ring tiles): ln_cs[n] = sum_k W'[n][k] of the gamma-scaled weight;
    // the staging waves accumulate every A row's (sum, sum of squares) while they wait, the epilogue applies
    // rstd[m] * (acc - mean[m] * ln_cs[n]).  vt_out: columns n >= vt_n0 are written TRANSPOSED ([image][n - vt_n0][token]).
    const float* ln_cs; float ln_eps;
    char* vt_out; int vt_n0, vt_tokens; int64_t vt_ld;
    int nloop;               // tile 69 (gemm_nloop.hip): output tiles of 160 columns one block walks
    // GroupNorm statistics from the producer (round 6): when non-null, every block also writes the per-channel (sum, sum of squares)
    // of the FINAL values of its BM rows (after bias / temb / residuals / activation, fp32, before the storage rounding) to
    // gn_part[tile_m][n] — the consumer GroupNorm's statistics pass over the stored tensor disappears (norm.hip, gn_finalize_part).
    // Fixed summation order (slab rows, then the block's wave rows): bitwise reproducible.  The host only sets it for launches the
    // epilogue can serve (no split-K, no GEGLU, no transposed columns, N % 8 == 0, nz == 1).
    float2* gn_part;
    // ... and per-GROUP sums gn_grp[tile_m * (N / gn_cpg) + group] when the consumer's groups are whole inside a tile's columns
    // (BN % gn_cpg == 0; the host checks): the consumer GroupNorm then needs no finalize launch
    float2* gn_grp; int gn_cpg;
};

__device__ __forceinline__ int div_sh(int x, int d, int sh) { return sh >= 0 ? x >> sh : x / d; }

// res1 shared by batch replicas (the BrushNet residual of both classifier-free-guidance halves): a handful of replicas,
// so a subtract loop, not a division
__device__ __forceinline__ int res1_row(const GemmArgs& p, int m) {
    if (p.res1_rows > 0)
        while (m >= p.res1_rows) m -= p.res1_rows;
    return m;
}

// Scalar epilogue for one output element (tails, misaligned outputs, split-K reduce).
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int64_t zo, int m, int n, float v, int zq = 0) {
    if (p.rs) v *= p.rs[zq * p.rs_zs + m];
    if (p.cs) v *= p.cs[zq * p.cs_zs + n];
    if (p.bias) v += p.bias_mode ? p.bias[m] : p.bias[n];
    if (p.temb) v += p.temb[(int64_t)div_sh(m, p.HoWo, p.howo_sh) * p.ld_temb + n];
    v *= p.alpha;
    if (p.res0) v += load_as_f32(p.res0, p.res0_dt, (int64_t)m * p.ld_res0 + n);
    if (p.res1) v += load_as_f32(p.res1, p.res1_dt, (int64_t)res1_row(p, m) * p.ld_res1 + n);
    if (p.act == MF_ACT_SILU) v = silu_precise(v);
    store_from_f32(p.out, p.out_dt, zo + (int64_t)m * p.ldc + n, v);
}

template <bool F16 = false>
__device__ __forceinline__ void load8_as_f32(const char* p, int dt, int64_t idx, float* o) {
    if (dt == MF_F32) {
        const float4 a = *reinterpret_cast<const float4*>(p + idx * 4);
        const float4 b = *reinterpret_cast<const float4*>(p + idx * 4 + 16);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        unpack_h8<F16>(*reinterpret_cast<const uint4*>(p + idx * 2), o);
    }
}

// Vector epilogue: 8 consecutive output channels of one pixel.
__device__ __forceinline__ void unpack8_bf16(const uint4& u, float* o) {
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}

// `pre`: the bf16 residual vectors of this item were fetched ahead of the LDS transposition (q0 / q1).
// `eb` (warp-specialised tiles with staged epilogue rows): LDS address of this item's 8 columns in the bias row; `et`: in the
// time-embedding row of the item's image (rows are `pitch` floats apart: see epb_off).
// F16: the 16-bit flavour of the launch (fp16 storage mode), a compile-time property of the instantiation (DT == MF_F16).
template <bool F16>
__device__ __forceinline__ void epilogue_store8(const GemmArgs& p, int64_t zo, int m, int n, float* v, bool pre = false,
                                                const uint4& q0 = uint4{0, 0, 0, 0}, const uint4& q1 = uint4{0, 0, 0, 0}, int zq = 0,
                                                const char* eb = nullptr, const char* et = nullptr) {
    if (p.rs || p.cs) {
        const float r = p.rs ? p.rs[zq * p.rs_zs + m] : 1.0f;
        float c8[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        if (p.cs) load8_as_f32((const char*)(p.cs + zq * p.cs_zs), MF_F32, n, c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= r * c8[j];
    }
    if (p.bias) {
        if (p.bias_mode) {
            const float b = p.bias[m];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b;
        } else {
            float b[8];
            if (eb) load8_as_f32(eb, MF_F32, 0, b);
            else load8_as_f32((const char*)p.bias, MF_F32, n, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j];
        }
    }
    if (p.temb) {
        float t[8];
        if (et) load8_as_f32(et, MF_F32, 0, t);
        else load8_as_f32((const char*)p.temb, MF_F32, (int64_t)div_sh(m, p.HoWo, p.howo_sh) * p.ld_temb + n, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += t[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
    if (p.res0) {
        float r[8];
        if (pre) unpack_h8<F16>(q0, r);
        else load8_as_f32<F16>(p.res0, p.res0_dt, (int64_t)m * p.ld_res0 + n, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    if (p.res1) {
        float r[8];
        if (pre) unpack_h8<F16>(q1, r);
        else load8_as_f32<F16>(p.res1, p.res1_dt, (int64_t)res1_row(p, m) * p.ld_res1 + n, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    if (p.act == MF_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_precise(v[j]);
    }
    if (p.act == MF_ACT_GEGLU4) {
        // weight rows are interleaved [4 values | 4 gates]: out[n/2 + j] = v[j] * gelu_erf(v[4 + j])  (activations.py:100-103)
        float g[4];
        if (p.out_dt != MF_F32) {
            // 16-bit output: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far inside the bf16 rounding) on
            // v_rcp_f32 / v_exp_f32 -- about half the VALU work of erff, and this epilogue runs once per 5 K-tiles
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = v[4 + j];
                const float z = fabsf(x) * 0.70710678118654752440f;
                const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
                const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
                const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
                g[j] = v[j] * (0.5f * x + 0.5f * fabsf(x) * e);              // 0.5 x (1 + sign(x) erf(|z|))
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = v[j] * (0.5f * v[4 + j] * (1.0f + erff(v[4 + j] * 0.70710678118654752440f)));
        }
        const int64_t o = zo + (int64_t)m * p.ldc + (n >> 1);
        if (p.out_dt == MF_F32) {
            *reinterpret_cast<float4*>(p.out + o * 4) = make_float4(g[0], g[1], g[2], g[3]);
        } else {
            uint2 u;
            u.x = pack_h2<F16>(g[0], g[1]);
            u.y = pack_h2<F16>(g[2], g[3]);
            *reinterpret_cast<uint2*>(p.out + o * 2) = u;
        }
        return;
    }
    const int64_t o = zo + (int64_t)m * p.ldc + n;
    if (p.out_dt == MF_F32) {
        *reinterpret_cast<float4*>(p.out + o * 4) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p.out + o * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        *reinterpret_cast<uint4*>(p.out + o * 2) = pack_h8<F16>(v);
    }
}

// 8 consecutive fp32 -> 8 bf16 (RNE) packed in 16 bytes
__device__ __forceinline__ uint4 ld8f_to_bf16(const char* p) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 16);
    uint4 r;
    r.x = pack_bf16x2(lo.x, lo.y);
    r.y = pack_bf16x2(lo.z, lo.w);
    r.z = pack_bf16x2(hi.x, hi.y);
    r.w = pack_bf16x2(hi.z, hi.w);
    return r;
}

typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
    // 64 lanes x 16 B -> LDS [lds_wave_base, +1 KiB), lane-linear; lds_wave_base must be wave-uniform.
    // Issued through inline asm ON PURPOSE: hipcc's waitcnt pass cannot tell which LDS bytes a
    // __builtin_amdgcn_global_load_lds writes, so it drains vmcnt(0) before the next ds_read and serialises the
    // DMA of tile t+1 behind the MFMAs of tile t.  An asm DMA is invisible to that pass; the ring below orders it
    // with its own counted s_waitcnt vmcnt(N) + s_barrier (cdna_hip_programming.md 5.7 item 1).
    const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_off) : "memory");
}

// one 16-bit MFMA by compute code (MF_BF16 / MF_F16): fragments are passed as bf16x8_t bit patterns
template <int DT>
__device__ __forceinline__ f32x4_t mfma16x32(bf16x8_t a, bf16x8_t b, f32x4_t c, int, int, int) {
    if constexpr (DT == MF_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ f32x16_t mfma32x16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    if constexpr (DT == MF_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Feature bits of an instantiation (template argument FX).  The three round-5 loop forms change the register allocation of the whole
// kernel (the cross-tile pipeline keeps two fragment sets: +20 ... +80 VGPRs, a wave per SIMD less on most tiles), and in the denoise
// step a block shares its CU with the other stream's blocks — so they are separate tile numbers the tuner may pick, not a rewrite of
// the round-4 tiles: FX = 0 compiles to the round-4 loops.
#define MF_FX_XT 1     // cross-tile fragment software pipeline of the warp-specialised 16-bit loops
#define MF_FX_EPB 2    // bias / LayerNorm column sums / time-embedding rows staged in LDS by the staging waves' first DMAs
#define MF_FX_AE 4     // dx-reuse convs: the A window of group g + 1 issued one tap early
#define MF_FX_ALL 7
#define MF_FX_XQ 8     // the cross-tile pipeline at k16 granularity (32x32x16 forms with 64-row wave tiles: two sets of MT + NT fragments)

// LDS layout of the staged epilogue rows of a warp-specialised tile: [bias][ln_colsum][temb of image 0 .. nimg-1], fp32, each row
// padded to whole 64-float DMA pieces; placed behind the ring (and the LayerNorm statistics).  nimg = 0: the tile has no room.
constexpr int epb_pitch(int bn) { return (bn + 63) / 64 * 64; }
constexpr int epb_off(int bm, int bn, int stages, bool dx, bool ws) {
    return dx ? 2 * (bm + 32) * 128 + (ws ? stages : 2) * bn * 128 : stages * (bm + bn) * 128 + (ws ? bm * 8 : 0);
}
constexpr int epb_nimg(int bm, int bn, int stages, bool dx, bool ws) {
    if (!ws) return 0;
    const int room = 160 * 1024 - epb_off(bm, bn, stages, dx, ws), row = epb_pitch(bn) * 4;
    return room >= 6 * row ? 4 : (room >= 3 * row ? 1 : 0);
}

// waves per SIMD the LDS footprint allows (register budget follows from it: 512 / waves per lane)
constexpr int min_waves(int bm, int bn, int stages, int nthr) {
    const int smem = stages * (bm + bn) * 128;
    int blocks = 160 * 1024 / smem;
    if (blocks > 8) blocks = 8;
    int w = blocks * (nthr / 64) / 4;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}

// In-launch split-K combine (mf_gemm_desc.sk_tickets).  Every K-slice block has written its fp32 slab; the block that arrives
// LAST at the tile's ticket sums the slabs in slice order (s = 0 .. splitk-1: the result does not depend on which block that is)
// and runs the epilogue — no second launch, and the kernel boundary behind ~10-20 MB of dirty partials goes with it.
// Cross-workgroup visibility (cdna_hip_programming.md Guideline 16, the counter form): every storing wave drains vmcnt, the
// block's barrier, ONE lane's agent-scope release + drain, the relaxed agent-scope ticket; the last arriver's ONE agent-scope
// acquire + drain, a barrier, then plain loads by every wave.  Correct for any placement of a tile's slices over XCDs / CUs.
// The last arriver re-arms the ticket, so the counters are zero again when the launch ends.
template <int BM, int BN, int NT_ALL, bool F16>
__device__ __forceinline__ void splitk_combine_tail(const GemmArgs& p, char* smem, int tile_m, int tile_n, int z, int zq, int64_t zo, int t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's slab stores have left
    __syncthreads();                                                     // ... and every other wave's (the LDS is free too)
    unsigned* last_flag = reinterpret_cast<unsigned*>(smem);
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the compiler may drop the fence's own wait (Pitfall 12)
        unsigned* tk = p.sk_tickets + ((int64_t)z * p.tiles_m + tile_m) * p.tiles_n + tile_n;
        const unsigned prev = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = prev == (unsigned)p.splitk - 1u;
        if (last) {
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *last_flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*last_flag == 0u) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    constexpr int CPR = BN / 8;
    const int64_t mn = (int64_t)p.M * p.N;
    const bool vec = p.vec_ok && (p.N & 3) == 0;
    for (int it = t; it < BM * CPR; it += NT_ALL) {
        const int row = it / CPR, ec = (it - row * CPR) * 8;
        const int m = m0 + row, n = n0 + ec;
        if (m >= p.M || n >= p.N) continue;
        const float* src = p.ws + (int64_t)z * mn + (int64_t)m * p.N + n;
        if (vec && n + 8 <= p.N) {
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = 0; s < p.splitk; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)s * p.nz * mn);
                const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)s * p.nz * mn + 4);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
            epilogue_store8<F16>(p, zo, m, n, v, false, uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}, zq);
        } else {
            for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) {
                float v = 0.0f;
                for (int s = 0; s < p.splitk; ++s) v += src[(int64_t)s * p.nz * mn + jj];
                epilogue_store(p, zo, m, n + jj, v, zq);
            }
        }
    }
}

// DT: MF_BF16, MF_F32, or a split code (MF_F16X3 / MF_BF16X3: fp32 operands staged exactly like MF_F32, every 8-wide
// fragment split in registers into 16-bit (hi, lo) halves, three 32x32x16 MFMAs per product).  WPK (split codes only):
// the W operand was split ahead of time ([32 hi | 32 lo] 16-bit values per block of 32 k — the same 128 bytes as 32
// floats, so its staging is byte-identical too) and needs no conversion.
// M16 (bf16 only): the same 32x32 accumulator blocks computed as four 16x16 tiles with v_mfma_f32_16x16x32_bf16 — the same
// LDS reads and matrix-pipe cycles as 32x32x16, but the chip sustains a higher clock on this shape under load
// (MI355X_MICROARCH.md, DVFS give-back (7): 1.12-1.14x in LDS-fed loops).
// WS (dx-reuse convs only): warp specialisation.  The block has four more waves: waves [0, W) compute (MFMA + epilogue),
// the last four (one per SIMD) only stage operands (LDS-DMA issue + counted waits), two taps ahead through a 3-deep W ring.  A wave's K
// tile costs ~640 MFMA cycles AND ~1000 cycles of DMA issue when one wave does both; split over two waves of
// the same SIMD the two streams issue from different ports and overlap.
// SKF: the in-launch split-K combine (splitk_combine_tail) is compiled in.  A separate instantiation on purpose: these kernels
// sit at their register / SGPR budget, and the tail's extra scalar state costs the 256 x 160 forms 368 bytes of scratch and
// several others a wave per SIMD even when it never runs; only the tiles that small-M, deep-K calls use carry an SKF twin.
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, bool A_F32, int STAGES, bool DXR = false, bool WPK = false, bool M16 = false,
          bool WS = false, bool P16 = false, bool SKF = false, int FX = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64 + (WS ? 256 : 0),
                             WS ? (WAVES_M * WAVES_N + 4) / 4 : min_waves(BM + (DXR ? 32 : 0), BN, STAGES, WAVES_M* WAVES_N * 64))
void gemm_conv_kernel(const GemmArgs p) {
    constexpr int NTHR = WAVES_M * WAVES_N * 64;      // compute threads (also the staging threads unless WS)
    constexpr int NSTG = WS ? 256 : NTHR;             // staging threads: WS adds four producer waves, one per SIMD
    constexpr bool H16 = DT == MF_BF16 || DT == MF_F16;      // 16-bit operands in memory: one MFMA per product (MF_F16: the f16 forms)
    static_assert(!WS || ((H16 || (DT == MF_F16X3 && WPK && !M16 && !P16)) && !A_F32 && STAGES == 3),
                  "warp specialisation: bf16 / fp16 (or the parity mode with a pre-split W), LDS-DMA staging, 3-deep ring");
    static_assert(!M16 || (H16 && !A_F32), "the 16x16x32 form is instantiated for bf16 / fp16 only");
    constexpr bool X1 = DT == MF_BF16X1;      // fp32 operands rounded to bf16 (RNE) in registers, ONE MFMA per product
    constexpr bool SPLIT = (DT == MF_F16X3 || DT == MF_BF16X3 || X1);
    static_assert(!WPK || (SPLIT && !X1), "a pre-split W operand only exists for the three-MFMA split codes");
    constexpr bool FP8 = (DT == MF_FP8);
    constexpr int ES = H16 ? 2 : (FP8 ? 1 : 4);   // element size of the operands in memory / LDS
    constexpr int AES = A_F32 ? 4 : ES;          // element size of the A storage dtype
    constexpr int VEC = 16 / ES;                 // elements per 16-byte LDS chunk
    constexpr int BK = 128 / ES;                 // K elements per tile (128 bytes per row)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    // P16 (warp-specialised dx-reuse convs, bf16): the wave tile is made of 16x16 MFMA tiles only (v_mfma_f32_16x16x32_bf16), so
    // WN needs to be a multiple of 16, not 32: 4x2 compute waves of 32x80 put TWO compute waves on every SIMD for a
    // 128x160 block (a CU's whole share of the 32x32-level convs)
    // (round 5: 64-row wave tiles too — 4x2 waves of 64x80 on a 256x160 block read 9 fragments per 20 MFMAs where 8x1 waves of
    // 32x160 read 12, and their two register sets of the cross-tile pipeline fit the 168 registers of three waves per SIMD)
    static_assert(!P16 || (H16 && !M16 && !A_F32 && (WM == 32 || WM == 64) && WN % 16 == 0), "P16: 16-bit operands, 32- or 64-row wave tiles");
    constexpr int MT = WM / 32, NT = P16 ? 1 : WN / 32;
    constexpr int MT16 = WM / 16, NT16 = WN / 16;
    constexpr int RPP = NSTG / 8;                // rows staged per pass (8 lanes per 128-B row)
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int EP_RS = (WN + 4) * 4;          // epilogue slab row stride (bytes)
    static_assert(WM % 32 == 0 && (P16 || WN % 32 == 0), "wave tile must be a multiple of 32x32");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    constexpr int SR = (WAVES_M * WAVES_N * 32 * EP_RS <= STAGES * STAGE_BYTES) ? 32 : 16;   // rows per epilogue slab
    static_assert(WAVES_M * WAVES_N * SR * EP_RS <= STAGES * STAGE_BYTES, "epilogue slabs must fit in the staging LDS");
    static_assert(STAGES >= 2 && STAGES <= 6, "2 to 6 LDS stages");
    static_assert(!A_F32 || STAGES == 2, "the register-staged path is double buffered");
    static_assert(!A_F32 || DT == MF_BF16, "A_F32 only converts fp32 activations for bf16 compute");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    MF_STAMP(0);

    // wave-uniform, and provably so (a scalar compare): the staging code keeps descriptors and loop state in SGPRs
    const bool producer = WS && __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= NTHR;
    const int tid = producer ? (int)threadIdx.x - NTHR : (int)threadIdx.x;   // index within the role
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    if constexpr (WS) {
        // developer switches (MFHIP_DBG_EPI bits 16 / 32): static priority 1 for the compute / the staging waves of a warp-specialised block
        if ((p.dbg_epi & 16) && !producer) __builtin_amdgcn_s_setprio(1);
        if ((p.dbg_epi & 32) && producer) __builtin_amdgcn_s_setprio(1);
    }

    // XCD-aware order: the 8 XCDs take blocks round-robin, so give each XCD a contiguous run of the LOGICAL order
    // (K split outermost, then the tiles of one split; blockIdx.x counts all of them).  The order inside a split is
    // chosen on the host (choose_order): n fastest (an XCD owns rows of A, W is shared) or m fastest (an XCD owns
    // columns of W), optionally in panels of ord_pw tiles of the fast dimension so that the ~64 tiles an XCD runs at
    // once are a compact rectangle and the panel's operand stays in that XCD's L2 from round to round.
    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int per_split = p.tiles_m * p.tiles_n;
    const int ksplit = bid >= per_split ? bid / per_split : 0;
    int tile_m, tile_n;
    {
        const int r = bid - ksplit * per_split;
        const int Li = p.ord_mfast ? p.tiles_m : p.tiles_n, Lo = p.ord_mfast ? p.tiles_n : p.tiles_m;
        const int panel = r >= Lo * p.ord_pw ? r / (Lo * p.ord_pw) : 0, rp = r - panel * Lo * p.ord_pw;
        int w = Li - panel * p.ord_pw;
        if (w > p.ord_pw) w = p.ord_pw;
        const int o = w == 1 ? rp : rp / w, i = panel * p.ord_pw + (rp - o * w);
        tile_m = p.ord_mfast ? i : o;
        tile_n = p.ord_mfast ? o : i;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;
    const int zq = z ? z / p.zdiv : 0, zr = z - zq * p.zdiv;

    // ---- staged epilogue rows (see GemmArgs::epb): the staging waves' first DMAs ----------------------------------------
    constexpr int EPB_NIMG = (FX & MF_FX_EPB) ? epb_nimg(BM, BN, STAGES, DXR, WS) : 0, EPB_PITCH = epb_pitch(BN), EPB_OFF = epb_off(BM, BN, STAGES, DXR, WS);
    const bool epb = EPB_NIMG > 0 && p.epb != 0;
    if constexpr (EPB_NIMG > 0) {
        if (producer && epb) {
            constexpr int PIECES = EPB_PITCH / 64;
            const int img0 = m0 >> p.epb_sh;
            const unsigned lds_e = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem) + EPB_OFF;
            const int nb = p.M >> p.epb_sh;                           // images (only read when temb is staged)
            for (int row = 0; row < 2 + EPB_NIMG; ++row) {
                const float* base = row == 0 ? p.bias : row == 1 ? p.ln_cs
                                    : (p.temb && img0 + row - 2 < nb) ? p.temb + (int64_t)(img0 + row - 2) * p.ld_temb : nullptr;
                if (base == nullptr) continue;
                const srd_t srd = make_srd(reinterpret_cast<const char*>(base), (unsigned)p.N * 4u);
#pragma unroll
                for (int q = 0; q < PIECES; ++q) {
                    if (((row * PIECES + q) & 3) != wave) continue;   // dealt over the four staging waves
                    const int col = q * 64 + lane;
                    dma4_buf(col < BN ? (unsigned)(n0 + col) * 4u : 0x80000000u, srd, lds_e + (row * EPB_PITCH + q * 64) * 4);
                }
            }
        }
    }

    const char* a0 = p.a0 + (zq * p.a_zs_o + zr * p.a_zs_i) * AES;
    const char* a1 = p.a1 ? p.a1 + (zq * p.a_zs_o + zr * p.a_zs_i) * AES : a0;
    const char* wbase = p.w + (zq * p.w_zs_o + zr * p.w_zs_i) * ES;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // ---- per-thread staging coordinates -------------------------------------------------
    const int lrow = tid >> 3;                                   // row within a staging pass
    const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);             // LOGICAL 16-B chunk this lane fetches (swizzled)
    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + lrow + i * RPP;
        if (m < p.M) {
            if (p.pointwise) {          // 1x1 / stride 1 / no padding / no upsample: output pixel m reads input pixel m
                a_pix[i] = m;           // (two integer divisions per row less: ~0.4 us of every Linear's prologue)
                a_iy0[i] = 0;
                a_ix0[i] = 0;
            } else {
                const int b = div_sh(m, p.HoWo, p.howo_sh);
                const int r = m - b * p.HoWo;
                const int oy = div_sh(r, p.Wo, p.wo_sh);
                const int ox = r - oy * p.Wo;
                a_pix[i] = b * p.Hin * p.Win;
                a_iy0[i] = oy * p.stride - p.pad_t;
                a_ix0[i] = ox * p.stride - p.pad_l;
            }
        } else {
            a_pix[i] = 0;
            a_iy0[i] = -(1 << 28);   // fails the bounds test -> zero page
            a_ix0[i] = 0;
        }
    }
    const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;
    const char* w_row[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + lrow + i * RPP;
        w_row[i] = n < p.N ? wbase + (int64_t)n * p.ldw * ES : nullptr;
    }

    const int kt_begin = ksplit * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > p.nkt) kt_end = p.nkt;
    const int nt = kt_end - kt_begin;

    int kk = kt_begin * BK + chunk * VEC;        // this thread's K element index in the current tile
    int c, ky, kx;
    {
        const int tap = kk >= p.Ctot ? kk / p.Ctot : 0;
        c = kk - tap * p.Ctot;
        ky = tap ? tap / p.KW : 0;
        kx = tap - ky * p.KW;
    }

    // source address of this lane's 16-B (or 32-B when A_F32) vector of A row i in the current K tile
    auto a_src = [&](int i, bool kvalid, const char* base, int ldb, int ccb) -> const char* {
        const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        const bool ok = kvalid && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
        const int pix = a_pix[i] + (iy >> p.ups) * p.Win + (ix >> p.ups);
        const char* src = base + ((int64_t)pix * ldb + ccb);
        return ok ? src : nullptr;
    };
    auto advance_k = [&]() {
        kk += BK;
        c += BK;
        while (c >= p.Ctot) {
            c -= p.Ctot;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    // ---- staging: LDS-DMA (default) --------------------------------------------------------
    auto issue_tile = [&](int stage) {
        char* As = smem + stage * STAGE_BYTES + wave * (8 * 128);      // this wave's first row of each pass
        char* Bs = As + BM * 128;
        const bool kvalid = kk < p.K;
        const bool seg0 = c < p.C0;
        const char* base = seg0 ? a0 : a1;
        const int ldb = seg0 ? p.ld0b : p.ld1b;
        const int ccb = (seg0 ? c : c - p.C0) * AES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const char* src = a_src(i, kvalid, base, ldb, ccb);
            dma16(src ? src : zero, As + i * RPP * 128);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            // a pre-split W row is zero-padded to whole blocks of 32 k: chunk -> k is not monotonic there, every chunk of
            // an existing K tile is readable
            const char* src = ((kvalid || WPK) && w_row[i]) ? w_row[i] + (int64_t)kk * ES : zero;
            dma16(src, Bs + i * RPP * 128);
        }
        advance_k();
    };

    // ---- fast staging path: every K tile lies inside one (tap, segment) -----------------------------
    // (all channel counts multiples of BK, tensors < 2 GiB).  Sources are buffer descriptors + a per-lane 32-bit
    // byte offset that advances by a constant per tile (ONE v_add per row per tile); offsets are recomputed only when
    // the tap or the segment changes, which is uniform for the whole block.  Out-of-image taps and rows past M / N
    // carry an offset >= 2^31 > num_records, so the hardware range check writes zeros for them.
    unsigned aoff[A_IT], woff[B_IT];
    srd_t srdA0, srdA1, srdW, srdCur;
    int f_ky = 0, f_kx = 0, f_seg = 0, f_left = 0, f_cin = 0;    // scalar (block-uniform) state
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem) + wave * 1024;
    auto fast_retarget = [&]() {
        srdCur = f_seg ? srdA1 : srdA0;
        const int ldb = f_seg ? p.ld1b : p.ld0b;
        const int ccb = (f_cin + chunk * VEC) * AES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + f_ky, ix = a_ix0[i] + f_kx;
            const bool ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            const int pix = a_pix[i] + (iy >> p.ups) * p.Win + (ix >> p.ups);
            aoff[i] = ok ? (unsigned)(pix * ldb + ccb) : 0x80000000u;
        }
    };
    auto fast_init = [&]() {
        const int nb = div_sh(p.M, p.HoWo, p.howo_sh);                  // images in the batch
        srdA0 = make_srd(a0, (unsigned)((nb * p.Hin * p.Win - 1) * p.ld0b + p.C0 * AES));
        srdA1 = make_srd(a1, (unsigned)((nb * p.Hin * p.Win - 1) * p.ld1b + (p.Ctot - p.C0) * AES));
        srdW = make_srd(wbase, (unsigned)(((int64_t)(p.N - 1) * p.ldw + p.K) * ES));
        const int k0 = kt_begin * BK;
        const int tap = k0 ? k0 / p.Ctot : 0;
        const int c0 = k0 - tap * p.Ctot;
        f_ky = tap ? tap / p.KW : 0; f_kx = tap - f_ky * p.KW;
        f_seg = c0 >= p.C0;
        f_cin = f_seg ? c0 - p.C0 : c0;
        f_left = ((f_seg ? p.Ctot - p.C0 : p.C0) - f_cin) / BK;
        fast_retarget();
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + lrow + i * RPP;
            woff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + k0 + chunk * VEC) * ES) : 0x80000000u;
        }
    };
    auto issue_tile_fast = [&](int stage) {
        const unsigned la = lds_base + stage * STAGE_BYTES;
        const unsigned lb = la + BM * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            dma16_buf(aoff[i], srdCur, la + i * RPP * 128);
            aoff[i] += BK * AES;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            dma16_buf(woff[i], srdW, lb + i * RPP * 128);
            woff[i] += 128;
        }
        if (--f_left == 0) {          // block-uniform: next tile starts a new segment or tap
            if (f_seg == 0 && p.Ctot > p.C0) {
                f_seg = 1; f_left = (p.Ctot - p.C0) / BK;
            } else {
                f_seg = 0; f_left = p.C0 / BK;
                if (++f_kx == p.KW) { f_kx = 0; ++f_ky; }
            }
            f_cin = 0;
            fast_retarget();
        }
    };

    // ---- staging: registers with fp32 -> bf16 conversion (A_F32) -------------------------------
    uint4 ra[A_F32 ? A_IT : 1], rb[A_F32 ? B_IT : 1];
    auto load_tile_regs = [&]() {
        const bool kvalid = kk < p.K;
        const bool seg0 = c < p.C0;
        const char* base = seg0 ? a0 : a1;
        const int ldb = seg0 ? p.ld0b : p.ld1b;
        const int ccb = (seg0 ? c : c - p.C0) * AES;
#pragma unroll
        for (int i = 0; i < (A_F32 ? A_IT : 1); ++i) {
            const char* src = a_src(i, kvalid, base, ldb, ccb);
            ra[i] = src ? ld8f_to_bf16(src) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < (A_F32 ? B_IT : 1); ++i) {
            const char* src = (kvalid && w_row[i]) ? w_row[i] + (int64_t)kk * ES : zero;
            rb[i] = *reinterpret_cast<const uint4*>(src);
        }
        advance_k();
    };
    auto store_tile_regs = [&](int stage) {
        char* As = smem + stage * STAGE_BYTES + tid * 16;               // lane-linear, like the DMA
        char* Bs = As + BM * 128;
#pragma unroll
        for (int i = 0; i < (A_F32 ? A_IT : 1); ++i) *reinterpret_cast<uint4*>(As + i * RPP * 128) = ra[i];
#pragma unroll
        for (int i = 0; i < (A_F32 ? B_IT : 1); ++i) *reinterpret_cast<uint4*>(Bs + i * RPP * 128) = rb[i];
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    f32x4_t acc16[P16 ? MT16 : 1][P16 ? NT16 : 1];          // P16: the accumulators (acc above stays unused)
#pragma unroll
    for (int a = 0; a < (P16 ? MT16 : 1); ++a)
#pragma unroll
        for (int b = 0; b < (P16 ? NT16 : 1); ++b) acc16[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};

    const int frow = lane & 31;              // fragment row (A: m, W: n) within a 32-row tile
    const int fh = lane >> 5;                // which half of the k-step this lane holds
    const int fkey = (frow >> 1) & 7;        // swizzle key (tile bases are multiples of 32 rows)

    // ---- split codes: fp32 -> (hi, lo) 16-bit halves in registers ---------------------------------------------
    // c0, c1: 8 consecutive fp32 (k = 16 ks + 8 fh + 0..7 of this lane's row).  hi = x toward zero, lo = x - hi (exact in
    // fp32) toward zero; the error of hi + lo against x is below 2^-22 |x| (fp16 halves) / 2^-16 |x| (bf16 halves).
    float split_amax = 0.0f;          // MF_F16X3: running max |operand| of this lane (range guard, mf_common.h)
    auto split8 = [&](const uint4& c0, const uint4& c1, uint4& hi, uint4& lo) {
        const float x[8] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w),
                            __uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w)};
        if constexpr (DT == MF_F16X3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) split_amax = mf_amax3(split_amax, x[2 * e], x[2 * e + 1]);
        }
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (X1) {
                h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);                                   // nearest-even; no low half
                l[e] = 0;
            } else if constexpr (DT == MF_F16X3) {
                mf_split_f16x2(x[2 * e], x[2 * e + 1], h[e], l[e]);                           // v_cvt_pkrtz + 2 v_fma_mix + v_cvt_pkrtz
            } else {
                const unsigned ua = __float_as_uint(x[2 * e]), ub = __float_as_uint(x[2 * e + 1]);
                h[e] = (ua >> 16) | (ub & 0xffff0000u);                                       // truncated bf16 pair
                l[e] = pack_bf16x2(x[2 * e] - __uint_as_float(ua & 0xffff0000u), x[2 * e + 1] - __uint_as_float(ub & 0xffff0000u));
            }
        }
        hi = uint4{h[0], h[1], h[2], h[3]};
        lo = uint4{l[0], l[1], l[2], l[3]};
    };
    auto mma16 = [&](const uint4& a, const uint4& b, f32x16_t& c) {
        if constexpr (DT == MF_F16X3)
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    };
    // One K tile (32 k = two 16-wide MFMA steps) of a split code.  Ap[i] / Ak[i]: LDS row base and swizzle key of this
    // lane's row of A tile i; Bp: row base of W tile 0 (tile j at + j * 32 rows), key fkey.
    auto compute_split = [&](const char* const (&Ap)[MT], const int (&Ak)[MT], const char* Bp) {
        if constexpr (FP8) {
            // fp8 e4m3: a K tile is 128 elements = two 64-wide steps of v_mfma_scale_f32_32x32x64_f8f6f4 (unit block
            // scales: E8M0 127); a lane's fragment is 32 consecutive bytes (chunks 4 ks + 2 fh, + 1) of its row — A and W
            // use the same byte -> k map, so the dot product does not depend on the instruction's internal k order
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                i32x8_t fa[MT], fb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint4 c0 = *reinterpret_cast<const uint4*>(Ap[i] + (((4 * ks + 2 * fh) ^ Ak[i]) << 4));
                    const uint4 c1 = *reinterpret_cast<const uint4*>(Ap[i] + (((4 * ks + 2 * fh + 1) ^ Ak[i]) << 4));
                    fa[i] = i32x8_t{(int)c0.x, (int)c0.y, (int)c0.z, (int)c0.w, (int)c1.x, (int)c1.y, (int)c1.z, (int)c1.w};
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const uint4 c0 = *reinterpret_cast<const uint4*>(Bp + j * 32 * 128 + (((4 * ks + 2 * fh) ^ fkey) << 4));
                    const uint4 c1 = *reinterpret_cast<const uint4*>(Bp + j * 32 * 128 + (((4 * ks + 2 * fh + 1) ^ fkey) << 4));
                    fb[j] = i32x8_t{(int)c0.x, (int)c0.y, (int)c0.z, (int)c0.w, (int)c1.x, (int)c1.y, (int)c1.z, (int)c1.w};
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[i], fb[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint4 c0 = *reinterpret_cast<const uint4*>(Ap[i] + (((4 * ks + 2 * fh) ^ Ak[i]) << 4));
                const uint4 c1 = *reinterpret_cast<const uint4*>(Ap[i] + (((4 * ks + 2 * fh + 1) ^ Ak[i]) << 4));
                split8(c0, c1, ah[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (WPK) {
                    bh[j] = *reinterpret_cast<const uint4*>(Bp + j * 32 * 128 + (((2 * ks + fh) ^ fkey) << 4));
                    bl[j] = *reinterpret_cast<const uint4*>(Bp + j * 32 * 128 + (((4 + 2 * ks + fh) ^ fkey) << 4));
                } else {
                    const uint4 c0 = *reinterpret_cast<const uint4*>(Bp + j * 32 * 128 + (((4 * ks + 2 * fh) ^ fkey) << 4));
                    const uint4 c1 = *reinterpret_cast<const uint4*>(Bp + j * 32 * 128 + (((4 * ks + 2 * fh + 1) ^ fkey) << 4));
                    split8(c0, c1, bh[j], bl[j]);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (!X1) {
                        mma16(al[i], bh[j], acc[i][j]);      // the small terms first
                        mma16(ah[i], bl[j], acc[i][j]);
                    }
                    mma16(ah[i], bh[j], acc[i][j]);
                }
        }
    };

    // ---- M16: one K tile (64 k = two 32-wide steps) on 16x16x32 MFMAs ---------------------------------------------
    // acc[i][j] element e = (a*2 + b)*4 + r is row 16a + 4 (lane >> 4) + r, column 16b + (lane & 15) of the 32x32 block.
    // Lane (r16 = lane & 15, kg = lane >> 4) of a 16-row fragment holds k = 32 ks + 8 kg + 0..7: chunk 4 ks + kg of its row;
    // tile bases are multiples of 16 rows, so the swizzle key is ((r16 >> 1) & 7) for every fragment (conflict-free for
    // ds_read_b128's lane groups: the four chunks kg = 0..3 of 16 rows cover the 16 slots of a 256-byte bank row once).
    const int r16 = lane & 15, kg = lane >> 4;
    auto compute_m16 = [&](const char* const (&Ap)[2 * MT], const int (&Ak)[2 * MT], const char* Bp) {
        // Ap[2i + a]: LDS row base of this lane's row of A half-tile (i, a); Bp: row base of W half-tile 0 (this lane's row)
        const int key16 = (r16 >> 1) & 7;
        uint4 fa[2][2 * MT], fb[2][2 * NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 2 * MT; ++t) fa[ks][t] = *reinterpret_cast<const uint4*>(Ap[t] + (((4 * ks + kg) ^ Ak[t]) << 4));
#pragma unroll
            for (int t = 0; t < 2 * NT; ++t) fb[ks][t] = *reinterpret_cast<const uint4*>(Bp + t * 16 * 128 + (((4 * ks + kg) ^ key16) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4_t c = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        c = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, fa[ks][2 * i + (q >> 1)]),
                                                                    __builtin_bit_cast(bf16x8_t, fb[ks][2 * j + (q & 1)]), c, 0, 0, 0);
                        acc[i][j][4 * q] = c[0]; acc[i][j][4 * q + 1] = c[1]; acc[i][j][4 * q + 2] = c[2]; acc[i][j][4 * q + 3] = c[3];
                    }
    };

    auto mma = [&](const uint4 (&fa)[MT], const uint4 (&fb)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (DT == MF_BF16) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
                } else if constexpr (DT == MF_F16) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8_t, fa[i]), __builtin_bit_cast(f16x8_t, fb[j]), acc[i][j], 0, 0, 0);
                } else {
                    const f32x4_t av = __builtin_bit_cast(f32x4_t, fa[i]);
                    const f32x4_t bv = __builtin_bit_cast(f32x4_t, fb[j]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[i][j], 0, 0, 0);
                }
            }
    };
    // Fragment reads are software-pipelined: the ds_read_b128s of k-step ks+1 are in flight while the MFMAs of
    // k-step ks run, so one wave alone keeps its matrix pipe fed across the LDS latency.
    auto compute = [&](int stage) {
        if constexpr (P16) {      // 16x16x32 MFMA tiles only (see compute3): tile bases are multiples of 16 rows
            const char* A16 = smem + stage * STAGE_BYTES + (wm * WM + r16) * 128;
            const char* B16 = smem + stage * STAGE_BYTES + BM * 128 + (wn * WN + r16) * 128;
            const int key16 = (r16 >> 1) & 7;
            uint4 fa[2][MT16], fb[2][NT16];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int t = 0; t < MT16; ++t) fa[ks][t] = *reinterpret_cast<const uint4*>(A16 + t * 16 * 128 + (((4 * ks + kg) ^ key16) << 4));
#pragma unroll
                for (int b = 0; b < NT16; ++b) fb[ks][b] = *reinterpret_cast<const uint4*>(B16 + b * 16 * 128 + (((4 * ks + kg) ^ key16) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < MT16; ++a)
#pragma unroll
                    for (int b = 0; b < NT16; ++b)
                        acc16[a][b] = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, fa[ks][a]),
                                                                              __builtin_bit_cast(bf16x8_t, fb[ks][b]), acc16[a][b], 0, 0, 0);
            return;
        }
        const char* As = smem + stage * STAGE_BYTES + (wm * WM + frow) * 128;
        const char* Bs = smem + stage * STAGE_BYTES + BM * 128 + (wn * WN + frow) * 128;
        if constexpr (SPLIT || FP8) {
            const char* Ap[MT]; int Ak[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) { Ap[i] = As + i * 32 * 128; Ak[i] = fkey; }
            compute_split(Ap, Ak, Bs);
            return;
        }
        if constexpr (M16) {
            const char* A16 = smem + stage * STAGE_BYTES + (wm * WM + r16) * 128;
            const char* B16 = smem + stage * STAGE_BYTES + BM * 128 + (wn * WN + r16) * 128;
            const char* Ap[2 * MT]; int Ak[2 * MT];
#pragma unroll
            for (int t = 0; t < 2 * MT; ++t) { Ap[t] = A16 + t * 16 * 128; Ak[t] = (r16 >> 1) & 7; }
            compute_m16(Ap, Ak, B16);
            return;
        }
        uint4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        auto ldfrag = [&](int ks, uint4 (&fa)[MT], uint4 (&fb)[NT]) {
            const int coff = (((2 * ks + fh) ^ fkey) << 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + j * 32 * 128 + coff);
        };
        // sched_barrier(0) pins the phase order: without it hipcc folds the two fragment sets into one register set
        // and re-serialises every k-step behind its own ds_read latency
        ldfrag(0, fa0, fb0);
        ldfrag(1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(3, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        mma(fa1, fb1);
    };

    // ---- cross-tap software pipeline of the warp-specialised bf16 forms (XT) -------------------------------------------
    // The compute waves of a block meet at one barrier per K tile, so without it they all read their fragments at the same
    // time and then all multiply: the LDS pipe and the matrix pipe take turns (measured: 128x160 / 8 + 4 waves, 1200 clocks per
    // K tile = 580 of LDS traffic + 640 of MFMA).  Here a K tile is two halves with a register set each (xa / xb [0], [1]):
    //     load half 1 of tile t | MFMAs of half 0 | lgkmcnt(0), barrier #(t + 1) | load half 0 of tile t + 1 | MFMAs of half 1
    // i.e. the barrier sits in the MIDDLE of a tile's arithmetic and every fragment read flies under the other half's MFMAs.
    // The producer protocol is unchanged: barrier #k still means "tile k has landed" and "tile k - 1 is no longer read" — the
    // consumers now arrive when tile k - 1's READS are complete instead of its MFMAs.  Same accumulation order: bit-identical.
    // Not for 8 x 1 compute waves of 32x160 (tiles 37 / 39 / 42): two sets of 12 fragments beside 80 accumulators exceed the 168
    // registers of three waves per SIMD and spill INSIDE the loop; tiles 49 / 50 are their 4 x 2 (64x80) successors.
    // XQ: the same pipeline at k16 granularity for the 32x32x16 forms — a K tile is FOUR parts, part q in register set q & 1, the barrier
    // between the MFMAs of parts 2 and 3.  Two sets of MT + NT fragments instead of 2 MT + 2 NT: 64x160 wave tiles (160 accumulators +
    // 56 fragment registers) fit the 256 registers of two waves per SIMD, i.e. FOUR compute waves of 64x160 + four staging waves per
    // 256x160 block.  Why that shape: a wave tile of WM x WN reads (WM + WN) x 32 bytes of LDS per k16 and multiplies for WM x WN / 32
    // clocks; over four SIMDs 32x80 needs 179 B/clk, 32x160 154, 64x80 115, 64x160 90 of the CU's 128 B/clk — the smaller wave tiles
    // are LDS-bound before the DMAs even take their share (stamps, round 5: tile 48 936 clocks per K tile against 640 of MFMA).
    constexpr bool XQ = (FX & MF_FX_XQ) != 0 && WS && H16 && !A_F32 && !M16 && !P16;
    constexpr bool XT = XQ || ((FX & MF_FX_XT) != 0 && WS && H16 && !A_F32 && (MF_XTAP != 0) && !(WAVES_M == 8 && WAVES_N == 1));
    constexpr int XNA = !XT ? 1 : XQ ? MT : (P16 ? MT16 : 2 * MT), XNB = !XT ? 1 : XQ ? NT : (P16 ? NT16 : 2 * NT);
    uint4 xa[2][XNA], xb[2][XNB];
    // fragments of part P (compile-time; XT: half P of two, XQ: k16 step P of four) of one K tile into register set S.  arow(t): LDS row
    // of this lane's row of A fragment t (P16 / M16: 16-row fragments, r16; else 32-row fragments, frow); Ab: the A buffer; Bb: the W
    // tile's first row of this wave (lane row included).
    auto xt_load = [&](auto S, auto P, const char* Ab, auto arow, const char* Bb) {
        constexpr int ss = decltype(S)::value, hh = decltype(P)::value;
        if constexpr (XQ) {
            const int ch = 2 * hh + fh;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = arow(i);
                xa[ss][i] = *reinterpret_cast<const uint4*>(Ab + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) xb[ss][j] = *reinterpret_cast<const uint4*>(Bb + j * 32 * 128 + ((ch ^ fkey) << 4));
        } else if constexpr (P16 || M16) {
            const int key16 = (r16 >> 1) & 7;
#pragma unroll
            for (int t = 0; t < XNA; ++t) {
                const int r = arow(t);
                xa[ss][t] = *reinterpret_cast<const uint4*>(Ab + r * 128 + (((4 * hh + kg) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int t = 0; t < XNB; ++t) xb[ss][t] = *reinterpret_cast<const uint4*>(Bb + t * 16 * 128 + (((4 * hh + kg) ^ key16) << 4));
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = 2 * (2 * hh + kk) + fh;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int r = arow(i);
                    xa[ss][kk * MT + i] = *reinterpret_cast<const uint4*>(Ab + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) xb[ss][kk * NT + j] = *reinterpret_cast<const uint4*>(Bb + j * 32 * 128 + ((ch ^ fkey) << 4));
            }
        }
    };
    auto xt_mma = [&](auto S) {
        constexpr int hh = decltype(S)::value;
        if constexpr (XQ) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma32x16<DT>(__builtin_bit_cast(bf16x8_t, xa[hh][i]), __builtin_bit_cast(bf16x8_t, xb[hh][j]), acc[i][j]);
        } else if constexpr (P16) {
#pragma unroll
            for (int a = 0; a < MT16; ++a)
#pragma unroll
                for (int b = 0; b < NT16; ++b)
                    acc16[a][b] = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, xa[hh][a]),
                                                                          __builtin_bit_cast(bf16x8_t, xb[hh][b]), acc16[a][b], 0, 0, 0);
        } else if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4_t c = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        c = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, xa[hh][2 * i + (q >> 1)]),
                                                                    __builtin_bit_cast(bf16x8_t, xb[hh][2 * j + (q & 1)]), c, 0, 0, 0);
                        acc[i][j][4 * q] = c[0]; acc[i][j][4 * q + 1] = c[1]; acc[i][j][4 * q + 2] = c[2]; acc[i][j][4 * q + 3] = c[3];
                    }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = mfma32x16<DT>(__builtin_bit_cast(bf16x8_t, xa[hh][kk * MT + i]), __builtin_bit_cast(bf16x8_t, xb[hh][kk * NT + j]), acc[i][j]);
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using H2 = std::integral_constant<int, 2>;
    using H3 = std::integral_constant<int, 3>;
    // the consumer loop of the cross-tile pipeline.  ld(S, P): part P of the current tile into set S; adv(): step to the next tile.
    // (the last tile is peeled: a straight-line loop body lets hipcc's waitcnt pass count the reads in flight across the back edge
    // instead of draining lgkmcnt(0) in front of the first MFMA)
    auto xt_loop = [&](auto&& ld, auto&& adv) {
        auto head = [&]() {              // every part of a tile but the last: the next part's reads fly under this part's MFMAs
            ld(H1{}, H1{});
            __builtin_amdgcn_sched_barrier(0);
            xt_mma(H0{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XQ) {
                ld(H0{}, H2{});
                __builtin_amdgcn_sched_barrier(0);
                xt_mma(H1{});
                __builtin_amdgcn_sched_barrier(0);
                ld(H1{}, H3{});
                __builtin_amdgcn_sched_barrier(0);
                xt_mma(H0{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        ld(H0{}, H0{});
        [[maybe_unused]] unsigned long long w_bar = 0;
        MF_CLK(tl0);
        for (int t = 0; t + 1 < nt; ++t) {
            head();
            adv();
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): every read of tile t has returned
            MF_CLK(tb0);
            __builtin_amdgcn_s_barrier();               // #(t + 1): tile t + 1 has landed
            MF_CLK(tb1);
            MF_SUM(w_bar, tb0, tb1);
            ld(H0{}, H0{});
            __builtin_amdgcn_sched_barrier(0);
            xt_mma(H1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        head();
        xt_mma(H1{});
        MF_CLK(tl1);
        MF_PUT(13, w_bar); MF_PUT(14, tl1 - tl0); MF_PUT(15, (unsigned long long)nt);
    };

    // ---- main loop ------------------------------------------------------------------------
    MF_STAMP(1);
    if (nt > 0) {
        if constexpr (DXR) {
            // 3x3 / stride 1 convolution with dx-tap reuse of the A tile.  K runs (ky, 128-byte-row chunk, kx): the
            // three kx taps of a (ky, chunk) group read the SAME input pixels shifted by one, so a group stages ONE A
            // window and the MFMAs of tap kx read it at row offset kx.  The window is the tile's image rows (BM / W
            // of them when W <= BM, a BM-pixel piece of one row otherwise) each framed by its left and right
            // neighbour pixel: W + 2 (or BM + 2) LDS rows per image row, the frame pixels outside the image being
            // hardware zeros of the DMA range check — the horizontal padding needs no per-fragment masking.
            // A-side DMAs drop to ~40 % of the per-tap scheme; the vector-memory path, not the bytes in L2, is what
            // bounds these kernels.  LDS: [A window x 2][W tile x 2].
            static_assert(!A_F32 && (WS ? STAGES == 3 : STAGES == 2), "dx reuse: DMA staging; W ring 2 deep, 3 when WS (a 4-deep ring measured no faster)");
            constexpr int WST = WS ? STAGES : 2;                   // W ring depth
            constexpr int AROWS = BM + RPP, A3_IT = AROWS / RPP;
            constexpr int AB = AROWS * 128, WB = BN * 128;
            static_assert(WAVES_M * WAVES_N * SR * EP_RS <= 2 * AB + WST * WB, "epilogue slabs must fit in this loop's LDS");
            const int nck = p.Ctot / BK;                           // K chunks per tap
            const int weff = p.Wo < BM ? p.Wo : BM;                // pixels of one image row inside the tile
            const int wfr = weff + 2;                              // ... plus the frame
            const int weff_sh = p.Wo < BM ? p.wo_sh : 31 - __builtin_clz(BM);       // log2(weff) when Wo is a power of two
            const int nrows_img = div_sh(p.M, p.Wo, p.wo_sh);      // image rows in the whole batch
            const int gy0 = div_sh(m0, p.Wo, p.wo_sh), gx0 = p.Wo < BM ? 0 : m0 - gy0 * p.Wo;
            int ay[A3_IT], apix[A3_IT];
#pragma unroll
            for (int i = 0; i < A3_IT; ++i) {
                const int r = lrow + i * RPP;                      // LDS row of the window
                const int ir = (int)(((unsigned)r * p.wfr_magic) >> 20), c = r - ir * wfr;      // r / wfr (r < BM + 32)
                const int gy = gy0 + ir, x = gx0 + c - 1;
                if (ir * weff < BM && gy < nrows_img && (unsigned)x < (unsigned)p.Wo) {
                    const int b = div_sh(gy, p.Ho, p.ho_sh);
                    ay[i] = gy - b * p.Ho;
                    apix[i] = gy * p.Wo + x;                       // pixel index at ky = 1
                } else {
                    ay[i] = -(1 << 28);
                    apix[i] = 0;
                }
            }
            unsigned w3[B_IT];
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int n = n0 + lrow + i * RPP;
                w3[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + chunk * VEC) * ES) : 0x80000000u;
            }
            const int nb = div_sh(p.M, p.HoWo, p.howo_sh);
            const srd_t sA0 = make_srd(a0, (unsigned)((nb * p.Hin * p.Win - 1) * p.ld0b + p.C0 * AES));
            const srd_t sA1 = make_srd(a1, (unsigned)((nb * p.Hin * p.Win - 1) * p.ld1b + (p.Ctot - p.C0) * AES));
            const srd_t sW = make_srd(wbase, (unsigned)(((int64_t)(p.N - 1) * p.ldw + p.K) * ES));
            const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem) + wave * 1024;
            // issue state: group (ky, global chunk) and kx of the next tile to stage; kt_begin is a multiple of 3
            int i_ky = kt_begin ? (kt_begin / 3) / nck : 0, i_cg = (kt_begin / 3) - i_ky * nck, i_kx = 0, i_grp = 0;
            // AE (warp-specialised forms): the A window of group g + 1 is issued one tap EARLIER, behind the W tile of group g's last tap
            // — its buffer has been free since barrier #3g, and a tap that carries a window (A3_IT more DMAs) then has three taps of
            // flight time instead of two: the staging waves used to wait for exactly those taps (stamps, round 5: 26 % of the loop)
            constexpr bool AE = (FX & MF_FX_AE) != 0 && WS && (MF_AEARLY != 0);
            int i_tap = 0;
            auto issue_a = [&](int ky, int cg, int grp) {
                const int c = cg * BK;
                const bool seg = c >= p.C0;
                const int ldb = seg ? p.ld1b : p.ld0b;
                const int ccb = ((seg ? c - p.C0 : c) + chunk * VEC) * AES;
                const srd_t srd = seg ? sA1 : sA0;
                const unsigned la = ldsb + (grp & 1) * AB;
#pragma unroll
                for (int i = 0; i < A3_IT; ++i) {
                    const int iy = ay[i] + ky - 1;
                    const bool ok = (unsigned)iy < (unsigned)p.Hin;
                    const unsigned off = ok ? (unsigned)((apix[i] + (ky - 1) * p.Win) * ldb + ccb) : 0x80000000u;
                    dma16_buf(off, srd, la + i * RPP * 128);
                }
            };
            auto issue3 = [&](int wstage) {
                if (AE ? i_tap == 0 : i_kx == 0) issue_a(i_ky, i_cg, i_grp);
                const unsigned wk = (unsigned)(((i_ky * 3 + i_kx) * p.Ctot + i_cg * BK) * ES);
                const unsigned lb = ldsb + 2 * AB + wstage * WB;
#pragma unroll
                for (int i = 0; i < B_IT; ++i) dma16_buf(w3[i] + wk, sW, lb + i * RPP * 128);
                if (AE && i_kx == 2 && i_tap + 1 < nt) {           // behind this group's last W tile: the next group's window
                    int n_cg = i_cg + 1, n_ky = i_ky;
                    if (n_cg == nck) { n_cg = 0; ++n_ky; }
                    issue_a(n_ky, n_cg, i_grp + 1);
                }
                ++i_tap;
                if (++i_kx == 3) {
                    i_kx = 0; ++i_grp;
                    if (++i_cg == nck) { i_cg = 0; ++i_ky; }
                }
            };
            int arow0[MT];                                         // window row of this lane's output pixel at kx = 0
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = wm * WM + i * 32 + frow;
                const int ir = p.wo_sh >= 0 ? m >> weff_sh : m / weff;
                arow0[i] = ir * wfr + (m - ir * weff);
            }
            int arow16[2 * MT];                                    // the same for the 16-row fragments of the M16 form
#pragma unroll
            for (int t = 0; t < 2 * MT; ++t) {
                const int m = wm * WM + t * 16 + r16;
                const int ir = p.wo_sh >= 0 ? m >> weff_sh : m / weff;
                arow16[t] = ir * wfr + (m - ir * weff);
            }
            auto compute3 = [&](int abuf, int wstage, int kx) {
                const char* Ab = smem + abuf * AB;
                if constexpr (P16) {
                    // 16-row A fragments of this wave's 32 rows, 16-column W fragments of its WN columns; W tile bases are
                    // multiples of 16 rows, so every fragment's swizzle key is ((r16 >> 1) & 7)
                    const char* Bp = smem + 2 * AB + wstage * WB + (wn * WN + r16) * 128;
                    const int key16 = (r16 >> 1) & 7;
                    const char* Ap[MT16]; int Ak[MT16];
#pragma unroll
                    for (int t = 0; t < MT16; ++t) {
                        const int r = arow16[t] + kx;
                        Ap[t] = Ab + r * 128;
                        Ak[t] = (r >> 1) & 7;
                    }
                    uint4 fa[2][MT16], fb[2][NT16];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                        for (int t = 0; t < MT16; ++t) fa[ks][t] = *reinterpret_cast<const uint4*>(Ap[t] + (((4 * ks + kg) ^ Ak[t]) << 4));
#pragma unroll
                        for (int b = 0; b < NT16; ++b) fb[ks][b] = *reinterpret_cast<const uint4*>(Bp + b * 16 * 128 + (((4 * ks + kg) ^ key16) << 4));
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int a = 0; a < MT16; ++a)
#pragma unroll
                            for (int b = 0; b < NT16; ++b)
                                acc16[a][b] = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, fa[ks][a]),
                                                                                      __builtin_bit_cast(bf16x8_t, fb[ks][b]), acc16[a][b], 0, 0, 0);
                    return;
                }
                const char* Bs = smem + 2 * AB + wstage * WB + (wn * WN + frow) * 128;
                if constexpr (M16) {
                    const char* Ap[2 * MT]; int Ak[2 * MT];
#pragma unroll
                    for (int t = 0; t < 2 * MT; ++t) {
                        const int r = arow16[t] + kx;
                        Ap[t] = Ab + r * 128;
                        Ak[t] = (r >> 1) & 7;
                    }
                    compute_m16(Ap, Ak, smem + 2 * AB + wstage * WB + (wn * WN + r16) * 128);
                    return;
                }
                int aoffs[MT], akey[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int r = arow0[i] + kx;
                    aoffs[i] = r * 128;
                    akey[i] = (r >> 1) & 7;
                }
                if constexpr (SPLIT || FP8) {
                    const char* Ap[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) Ap[i] = Ab + aoffs[i];
                    compute_split(Ap, akey, Bs);
                    return;
                }
                uint4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
                auto ldfrag = [&](int ks, uint4 (&fa)[MT], uint4 (&fb)[NT]) {
                    const int cb = (((2 * ks + fh) ^ fkey) << 4);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        fa[i] = *reinterpret_cast<const uint4*>(Ab + aoffs[i] + (((2 * ks + fh) ^ akey[i]) << 4));
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + j * 32 * 128 + cb);
                };
                ldfrag(0, fa0, fb0);
                ldfrag(1, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(2, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(3, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa0, fb0);
                mma(fa1, fb1);
            };
            if constexpr (WS) {
                // Producers run PFD = WST - 1 taps ahead: tap t + PFD goes to W stage (t + PFD) % WST (last read in tap t - 1,
                // which every consumer finished before barrier #t) and, when it opens a group, to the A buffer that group
                // g - 2 used (its last tap is at most t - 1 for PFD <= 3).
                // Loads return in order: leaving exactly the newest batch in flight means tap t + 1 has landed.
                // Barrier #k (k = 0 .. nt - 1) separates "tap k landed / tap k - 1 consumed" for both roles.
                constexpr int PFD = WST - 1;                       // taps in flight ahead of the one being multiplied
                if (producer) {
                    int kx2 = 0, st2 = 0, issued = 0;              // kx, W stage and index of the next tap to issue
                    auto issue_next = [&]() {
                        issue3(st2);
                        st2 = st2 == WST - 1 ? 0 : st2 + 1;
                        kx2 = kx2 == 2 ? 0 : kx2 + 1;
                        ++issued;
                    };
                    // wait until tap `need` has landed: everything issued after it may stay in flight.  Those are at most
                    // PFD - 1 <= 2 taps, at most one of which opens a group (and carries the A window's DMAs too).
                    // AE: a window rides BEHIND the W tile of tap q when q % 3 == 2 (and tap q + 1 exists), so it may also stay in flight
                    // when it was issued with `need` itself
                    auto wait_for = [&](int need) {
                        const int newer = issued - 1 - need;       // taps issued after `need`
                        bool with_a = false;
                        if constexpr (AE) {
                            for (int q = need; q < issued; ++q) with_a |= (q % 3 == 2 && q + 1 < nt);
                        } else {
                            for (int q = need + 1; q < issued; ++q) with_a |= (q % 3 == 0);
                        }
                        if (newer <= 0) { if (with_a) wait_vmcnt<A3_IT>(); else wait_vmcnt<0>(); }
                        else if (newer == 1) { if (with_a) wait_vmcnt<B_IT + A3_IT>(); else wait_vmcnt<B_IT>(); }
                        else { if (with_a) wait_vmcnt<2 * B_IT + A3_IT>(); else wait_vmcnt<2 * B_IT>(); }
                    };
                    static_assert(2 * B_IT + A3_IT < 64, "vmcnt is a 6-bit counter");
                    for (int k = 0; k < PFD && k < nt; ++k) issue_next();
                    wait_for(0);
                    __builtin_amdgcn_s_barrier();                  // #0
                    MF_STAMP(2);
                    [[maybe_unused]] unsigned long long w_iss = 0, w_vm = 0, w_bar = 0;
                    for (int t = 0; t + 1 < nt; ++t) {
                        MF_CLK(c0);
                        if (t + PFD < nt) issue_next();            // tap t + PFD
                        MF_CLK(c1);
                        wait_for(t + 1);
                        MF_CLK(c2);
                        __builtin_amdgcn_s_barrier();              // #(t + 1)
                        MF_CLK(c3);
                        MF_SUM(w_iss, c0, c1); MF_SUM(w_vm, c1, c2); MF_SUM(w_bar, c2, c3);
                    }
                    MF_PUT(13, w_vm); MF_PUT(14, w_bar); MF_PUT(15, w_iss);
                } else {
                    int c_kx = 0, c_grp = 0, c_st = 0;
                    __builtin_amdgcn_s_barrier();                  // #0
                    MF_STAMP(2);
                    if constexpr (XT) {
                        // tap (c_grp, c_st, c_kx): A window buffer, W stage and the window's row shift
                        auto ld = [&](auto S, auto P) {
                            const int kx = c_kx;
                            const char* Ab = smem + (c_grp & 1) * AB;
                            const char* Bb = smem + 2 * AB + c_st * WB + (wn * WN + ((P16 || M16) ? r16 : frow)) * 128;
                            if constexpr (P16 || M16) xt_load(S, P, Ab, [&](int t) { return arow16[t] + kx; }, Bb);
                            else xt_load(S, P, Ab, [&](int i) { return arow0[i] + kx; }, Bb);
                        };
                        xt_loop(ld, [&]() {
                            c_st = c_st == WST - 1 ? 0 : c_st + 1;
                            if (++c_kx == 3) { c_kx = 0; ++c_grp; }
                        });
                    } else
                    for (int t = 0; t < nt; ++t) {
                        compute3(c_grp & 1, c_st, c_kx);
                        c_st = c_st == WST - 1 ? 0 : c_st + 1;
                        if (++c_kx == 3) { c_kx = 0; ++c_grp; }
                        if (t + 1 < nt) __builtin_amdgcn_s_barrier();   // #(t + 1)
                    }
                }
            } else {
            issue3(0);
            int c_kx = 0, c_grp = 0;
            for (int t = 0; t < nt; ++t) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (t + 1 < nt) issue3((t + 1) & 1);
                compute3(c_grp & 1, t & 1, c_kx);
                if (++c_kx == 3) { c_kx = 0; ++c_grp; }
            }
            }
        } else if constexpr (!A_F32) {
            // LDS ring, STAGES-1 tiles in flight while tile t is multiplied.  Only a COUNTED vmcnt (all but the newer
            // tiles' G = A_IT + B_IT DMAs each) and a raw s_barrier order the ring.  RAW: a tile is read only after
            // every wave's vmcnt + the barrier; WAR: stage (t + PF) % STAGES was last read in compute(t - 1), which
            // every wave finished (its MFMAs consumed the ds_reads) before arriving at this barrier.
            constexpr int G = A_IT + B_IT;
            constexpr int PF = STAGES - 1;
            auto ring = [&](auto issue) {
                for (int s0 = 0; s0 < PF; ++s0)
                    if (s0 < nt) issue(s0);
                int st_c = 0, st_i = PF;                 // stage being computed / stage being filled
                for (int t = 0; t < nt; ++t) {
                    {   // allow the DMAs of the (up to PF-1) newer tiles to stay in flight
                        const int newer = nt - 1 - t < PF - 1 ? nt - 1 - t : PF - 1;
                        static_assert((PF - 1) * G < 64, "vmcnt is a 6-bit counter");
                        if (PF >= 5 && newer == 4) wait_vmcnt<4 * G>();
                        else if (PF >= 4 && newer == 3) wait_vmcnt<3 * G>();
                        else if (PF >= 3 && newer == 2) wait_vmcnt<2 * G>();
                        else if (PF >= 2 && newer == 1) wait_vmcnt<G>();
                        else wait_vmcnt<0>();
                    }
                    __builtin_amdgcn_s_barrier();
                    if (t + PF < nt) issue(st_i);
                    compute(st_c);
                    st_c = st_c == STAGES - 1 ? 0 : st_c + 1;
                    st_i = st_i == STAGES - 1 ? 0 : st_i + 1;
                }
            };
            // Warp-specialised form of the same ring (see the dx-reuse loop): the producer waves keep PF tiles in flight
            // and meet the compute waves at one barrier per K tile; barrier #k separates "tile k landed / tile k - 1 consumed".
            auto ring_ws = [&](auto issue) {
                if (producer) {
                    int st_i = 0, issued = 0;
                    auto issue_next = [&]() { issue(st_i); st_i = st_i == STAGES - 1 ? 0 : st_i + 1; ++issued; };
                    auto wait_for = [&](int need) {                // tile `need` landed; the (<= PF - 1) newer ones may fly
                        const int newer = issued - 1 - need;
                        static_assert((PF - 1) * G < 64, "vmcnt is a 6-bit counter");
                        if (PF >= 3 && newer >= 2) wait_vmcnt<2 * G>();      // PF <= 3: at most two newer tiles
                        else if (PF >= 2 && newer == 1) wait_vmcnt<G>();
                        else wait_vmcnt<0>();
                    };
                    // LayerNorm fold: this thread reads back the 16-byte chunks of the A tile its own DMAs wrote (tile t has
                    // landed for this wave since wait_for(t); its stage is next written by this wave's own issue of tile
                    // t + STAGES) and accumulates them into the row's sum / sum of squares: 8 v_dot2c_f32_bf16 per chunk on
                    // waves that otherwise only wait.  A row's 8 lanes are combined once after the loop.
                    const bool lnf = p.ln_cs != nullptr;
                    float ln_s[A_IT], ln_q[A_IT];
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) { ln_s[i] = 0.0f; ln_q[i] = 0.0f; }
                    int st_s = 0;
                    auto ln_tile = [&]() {
                        if constexpr (DT == MF_BF16) {
                            typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
                            const bf2_t ones = __builtin_bit_cast(bf2_t, 0x3F803F80u);
                            const char* As = smem + st_s * STAGE_BYTES + tid * 16;
#pragma unroll
                            for (int i = 0; i < A_IT; ++i) {
                                const uint4 u = *reinterpret_cast<const uint4*>(As + i * RPP * 128);
                                const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const bf2_t v = __builtin_bit_cast(bf2_t, w4[e]);
                                    ln_s[i] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, ln_s[i], false);
                                    ln_q[i] = __builtin_amdgcn_fdot2_f32_bf16(v, v, ln_q[i], false);
                                }
                            }
                        } else if constexpr (DT == MF_F16) {
                            const char* As = smem + st_s * STAGE_BYTES + tid * 16;
#pragma unroll
                            for (int i = 0; i < A_IT; ++i) {
                                const uint4 u = *reinterpret_cast<const uint4*>(As + i * RPP * 128);
                                const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const mf_f16x2_t v = __builtin_bit_cast(mf_f16x2_t, w4[e]);
                                    const float a0 = (float)v[0], a1 = (float)v[1];
                                    ln_s[i] += a0 + a1;
                                    ln_q[i] = fmaf(a0, a0, fmaf(a1, a1, ln_q[i]));
                                }
                            }
                        }
                        st_s = st_s == STAGES - 1 ? 0 : st_s + 1;
                    };
                    for (int k = 0; k < PF && k < nt; ++k) issue_next();
                    wait_for(0);
                    __builtin_amdgcn_s_barrier();                  // #0
                    MF_STAMP(2);
                    [[maybe_unused]] unsigned long long w_iss = 0, w_vm = 0, w_bar = 0;
                    for (int t = 0; t + 1 < nt; ++t) {
                        MF_CLK(c0);
                        if (t + PF < nt) issue_next();             // into the stage of tile t - 1: the DMA goes out FIRST ...
                        if (lnf) ln_tile();                        // ... and tile t is summed while it flies
                        MF_CLK(c1);
                        wait_for(t + 1);
                        MF_CLK(c2);
                        __builtin_amdgcn_s_barrier();              // #(t + 1)
                        MF_CLK(c3);
                        MF_SUM(w_iss, c0, c1); MF_SUM(w_vm, c1, c2); MF_SUM(w_bar, c2, c3);
                    }
                    MF_PUT(13, w_vm); MF_PUT(14, w_bar); MF_PUT(15, w_iss);
                    if (lnf) {
                        ln_tile();                                 // tile nt - 1
                        float2* lnst = reinterpret_cast<float2*>(smem + STAGES * STAGE_BYTES);   // [BM] (mean, rstd): past the ring
                        const float invk = 1.0f / (float)p.K;
#pragma unroll
                        for (int i = 0; i < A_IT; ++i) {
                            float s1 = ln_s[i], q1 = ln_q[i];
#pragma unroll
                            for (int off = 1; off < 8; off <<= 1) {
                                s1 += __shfl_xor(s1, off, 64);
                                q1 += __shfl_xor(q1, off, 64);
                            }
                            const float mean = s1 * invk;
                            float var = q1 * invk - mean * mean;
                            if (var < 0.0f) var = 0.0f;
                            if ((tid & 7) == 0) lnst[lrow + i * RPP] = make_float2(mean, 1.0f / sqrtf(var + p.ln_eps));
                        }
                    }
                } else {
                    int st_c = 0;
                    __builtin_amdgcn_s_barrier();                  // #0
                    MF_STAMP(2);
                    if constexpr (XT) {
                        const int arb = wm * WM + ((P16 || M16) ? r16 : frow);      // this lane's row of A fragment 0
                        auto ld = [&](auto S, auto P) {
                            const char* Ab = smem + st_c * STAGE_BYTES;
                            const char* Bb = Ab + BM * 128 + (wn * WN + ((P16 || M16) ? r16 : frow)) * 128;
                            if constexpr (P16 || M16) xt_load(S, P, Ab, [&](int t) { return arb + t * 16; }, Bb);
                            else xt_load(S, P, Ab, [&](int i) { return arb + i * 32; }, Bb);
                        };
                        xt_loop(ld, [&]() { st_c = st_c == STAGES - 1 ? 0 : st_c + 1; });
                    } else
                    for (int t = 0; t < nt; ++t) {
                        compute(st_c);
                        st_c = st_c == STAGES - 1 ? 0 : st_c + 1;
                        if (t + 1 < nt) __builtin_amdgcn_s_barrier();   // #(t + 1)
                    }
                }
            };
            if constexpr (WS) {
                if (p.fast) {
                    if (producer) fast_init();
                    ring_ws(issue_tile_fast);
                } else {
                    ring_ws(issue_tile);
                }
            } else if (p.fast) {
                fast_init();
                ring(issue_tile_fast);
            } else {
                ring(issue_tile);
            }
        } else {
            load_tile_regs();
            store_tile_regs(0);
            if (nt > 1) load_tile_regs();
            __syncthreads();
            for (int t = 0; t < nt; ++t) {
                compute(t & 1);
                if (t + 1 < nt) {
                    store_tile_regs((t + 1) & 1);
                    if (t + 2 < nt) load_tile_regs();
                }
                __syncthreads();
            }
        }
    }
    MF_STAMP(3);
    if constexpr (EPB_NIMG > 0) {
        if (producer && epb) wait_vmcnt<0>();     // (the loop's own waits cover these oldest DMAs; this is for nt == 0)
    }
    __syncthreads();   // every wave is done reading the staging LDS: reuse it for the epilogue slabs
    MF_STAMP(4);
    if constexpr (DT == MF_F16X3) mf_raise_if_over(p.ovf, split_amax);

    // ---- epilogue ---------------------------------------------------------------------------
    char* slab = smem + wave * (SR * EP_RS);        // private to this wave: [SR rows][WN + 4] fp32
    constexpr int CPR = WN / 8;                      // 8-channel groups per output row
    constexpr int ITEMS = SR * CPR;                  // (row, group) items per slab
    float* ws = p.splitk > 1 ? p.ws + ((int64_t)ksplit * p.nz + z) * (int64_t)p.M * p.N : nullptr;
    const int64_t zo = zq * p.o_zs_o + zr * p.o_zs_i;
    // bf16 residuals of a slab's items are fetched BEFORE its LDS transposition: loads and stores share vmcnt and
    // return in order, so a load issued after the previous item's store waits for that store's round trip
    const bool res_pre = !ws && p.vec_ok && (p.res0 || p.res1) && (!p.res0 || p.res0_dt != MF_F32) &&
                         (!p.res1 || p.res1_dt != MF_F32) && !p.dbg_no_res_pre;
    constexpr int NIT = (ITEMS + 63) / 64;
    if constexpr (WS) {
        // Warp-specialised blocks: the four staging waves have nothing left to stage, so they take their share of the
        // epilogue's memory phase.  The compute waves transpose their accumulators into their slabs; after a block
        // barrier the 64-item chunks of ALL slabs are dealt round-robin to all waves (compute and staging alike), each
        // of which fetches its chunks' residuals ahead of the barrier and then reads, finishes and stores them.
        constexpr int NCW = WAVES_M * WAVES_N, TW = NCW + 4, NCHUNK = NCW * NIT, MAXC = (NCHUNK + TW - 1) / TW;
        const int g = producer ? NCW + wave : wave;
        const bool gn = p.gn_part != nullptr;
        constexpr int GNP = (NCW * WN + TW * 64 - 1) / (TW * 64);      // (slab, column) pairs per thread of the statistics pass
        float gns[GNP], gnq[GNP];
#pragma unroll
        for (int k = 0; k < GNP; ++k) { gns[k] = 0.0f; gnq[k] = 0.0f; }
#pragma unroll
        for (int ih = 0; ih < MT * (32 / SR); ++ih) {
            const int i = ih / (32 / SR), half = ih % (32 / SR);
            uint4 q0[MAXC], q1[MAXC];
            auto coords = [&](int u, int& sw, int& row, int& ec, int& m, int& n) -> bool {
                const int c = g + u * TW;
                sw = c / NIT;
                const int it = (c - sw * NIT) * 64 + lane;
                row = it / CPR; ec = (it - row * CPR) * 8;
                const int swm = sw / WAVES_N, swn = sw - swm * WAVES_N;
                m = m0 + swm * WM + i * 32 + half * SR + row;
                n = n0 + swn * WN + ec;
                return c < NCHUNK && (ITEMS % 64 == 0 || it < ITEMS);
            };
            if (res_pre) {
#pragma unroll
                for (int u = 0; u < MAXC; ++u) {
                    int sw, row, ec, m, n;
                    const bool on = coords(u, sw, row, ec, m, n);
                    q0[u] = uint4{0, 0, 0, 0}; q1[u] = uint4{0, 0, 0, 0};
                    if (on && m < p.M && n + 8 <= p.N) {
                        if (p.res0) q0[u] = *reinterpret_cast<const uint4*>(p.res0 + ((int64_t)m * p.ld_res0 + n) * 2);
                        if (p.res1) q1[u] = *reinterpret_cast<const uint4*>(p.res1 + ((int64_t)res1_row(p, m) * p.ld_res1 + n) * 2);
                    }
                }
            }
            if (!producer) {
                if constexpr (P16) {      // acc16[2i + a][b] element r: row 32i + 16a + 4 kg + r, column 16b + r16 of the wave tile
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < NT16; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * a + 4 * kg + r - half * SR;
                                if (SR == 32 || a == half)
                                    *reinterpret_cast<float*>(slab + row * EP_RS + (16 * b + r16) * 4) = acc16[2 * i + a][b][r];
                            }
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int e = half * (SR / 2); e < half * (SR / 2) + SR / 2; ++e) {
                            if constexpr (M16) {
                                const int row = 16 * (e >> 3) + 4 * kg + (e & 3) - half * SR;
                                *reinterpret_cast<float*>(slab + row * EP_RS + (j * 32 + 16 * ((e >> 2) & 1) + r16) * 4) = acc[i][j][e];
                            } else {
                                const int row = (e & 3) + 8 * (e >> 2) + 4 * fh - half * SR;
                                *reinterpret_cast<float*>(slab + row * EP_RS + (j * 32 + frow) * 4) = acc[i][j][e];
                            }
                        }
                }
            }
            MF_STAMP(7 + 3 * ih);                         // (stamps: round ih — residuals requested, accumulators in the slab)
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's slab writes are done ...
            __builtin_amdgcn_s_barrier();                // ... and everybody else's
            MF_STAMP(8 + 3 * ih);
#pragma unroll
            for (int u = 0; u < MAXC; ++u) {
                int sw, row, ec, m, n;
                if (!coords(u, sw, row, ec, m, n)) continue;
                const char* sl = smem + sw * (SR * EP_RS);
                const float2* lnst = reinterpret_cast<const float2*>(smem + STAGES * STAGE_BYTES);
                if (p.vt_out && n0 + (sw % WAVES_N) * WN >= p.vt_n0) {
                    // transposed columns (the V third of a fused q | k | v projection): this slab is dealt as (column, group of 8
                    // rows) items; a lane gathers 8 consecutive tokens of ONE channel from the slab's column and stores them as
                    // 16 bytes of V^T[image][channel][token] (attention reads V^T with keys contiguous)
                    static_assert(SR % 8 == 0 && (SR / 8) * WN == ITEMS, "transposed items cover the slab");
                    const int c = g + u * TW, it = (c - sw * NIT) * 64 + lane;
                    // consecutive lanes take consecutive 8-token groups of one channel: SR / 8 lanes write 2 * SR contiguous bytes
                    const int rg = it % (SR / 8), col = it / (SR / 8);
                    const int swm = sw / WAVES_N, swn = sw - swm * WAVES_N;
                    const int mt = m0 + swm * WM + i * 32 + half * SR + rg * 8, nt_ = n0 + swn * WN + col;
                    if (mt < p.M && nt_ < p.N) {
                        float tv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) tv[j] = *reinterpret_cast<const float*>(sl + (rg * 8 + j) * EP_RS + col * 4);
                        const char* erow = smem + EPB_OFF + (nt_ - n0) * 4;           // this column in the staged bias row
                        if (p.ln_cs) {
                            const float cs = epb ? *reinterpret_cast<const float*>(erow + EPB_PITCH * 4) : p.ln_cs[nt_];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float2 st = lnst[mt - m0 + j];
                                tv[j] = st.y * (tv[j] - st.x * cs);
                            }
                        }
                        const float b = p.bias ? (epb ? *reinterpret_cast<const float*>(erow) : p.bias[nt_]) : 0.0f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) tv[j] = (tv[j] + b) * p.alpha;
                        const int img = mt / p.vt_tokens, tok = mt - img * p.vt_tokens;
                        const uint4 o = pack_h8<DT == MF_F16>(tv);
                        *reinterpret_cast<uint4*>(p.vt_out + (((int64_t)img * (p.N - p.vt_n0) + (nt_ - p.vt_n0)) * p.vt_ld + tok) * 2) = o;
                    }
                    continue;
                }
                float v[8];
                float4 lo, hi;
#ifdef MF_STAMPS
                if (p.dbg_epi & 2) { lo = make_float4(1.0f, 2.0f, 3.0f, (float)u); hi = lo; } else
#endif
                {
                    lo = *reinterpret_cast<const float4*>(sl + row * EP_RS + ec * 4);
                    hi = *reinterpret_cast<const float4*>(sl + row * EP_RS + ec * 4 + 16);
                }
#ifdef MF_STAMPS
                if (p.dbg_epi & 1) { asm volatile("" ::"v"(lo.x), "v"(hi.w)); continue; }
#endif
                if (p.ln_cs && m < p.M && n + 8 <= p.N) {            // LayerNorm fold: rstd * (acc - mean * colsum)
                    const float2 st = lnst[m - m0];
                    const float* lc = epb ? reinterpret_cast<const float*>(smem + EPB_OFF + (EPB_PITCH + n - n0) * 4) : p.ln_cs + n;
                    const float4 c0 = *reinterpret_cast<const float4*>(lc);
                    const float4 c1 = *reinterpret_cast<const float4*>(lc + 4);
                    lo.x = st.y * (lo.x - st.x * c0.x); lo.y = st.y * (lo.y - st.x * c0.y);
                    lo.z = st.y * (lo.z - st.x * c0.z); lo.w = st.y * (lo.w - st.x * c0.w);
                    hi.x = st.y * (hi.x - st.x * c1.x); hi.y = st.y * (hi.y - st.x * c1.y);
                    hi.z = st.y * (hi.z - st.x * c1.z); hi.w = st.y * (hi.w - st.x * c1.w);
                }
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                if (m < p.M && n < p.N) {
                    if (ws) {
                        if (n + 8 <= p.N && (p.N & 3) == 0) {
                            *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) = lo;
                            *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n + 4) = hi;
                        } else {
                            for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) ws[(int64_t)m * p.N + n + jj] = v[jj];
                        }
                    } else if (p.vec_ok && n + 8 <= p.N) {
                        if (epb) {
                            const char* eb = smem + EPB_OFF + (n - n0) * 4;
                            epilogue_store8<DT == MF_F16>(p, zo, m, n, v, res_pre, q0[u], q1[u], zq, eb, eb + (2 + (m >> p.epb_sh) - (m0 >> p.epb_sh)) * (EPB_PITCH * 4));
                        } else {
                            epilogue_store8<DT == MF_F16>(p, zo, m, n, v, res_pre, q0[u], q1[u], zq);
                        }
                        if (gn) {      // the final values go back into the slab for the column pass below
                            char* slw = smem + sw * (SR * EP_RS) + row * EP_RS + ec * 4;
                            *reinterpret_cast<float4*>(slw) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(slw + 16) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                    } else {
                        for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) epilogue_store(p, zo, m, n + jj, v[jj], zq);
                    }
                }
            }
            MF_STAMP(9 + 3 * ih);                         // (stamps: this wave's stores of round ih are issued)
            if (gn) {
                // column pass: the slabs now hold the round's final values; thread (g, lane) sums the SR rows of its (slab, column)
                // pairs — the same pairs every round, so a column's sum over the wave tile's rows stays in one register pair
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int k = 0; k < GNP; ++k) {
                    const int pid = (g + k * TW) * 64 + lane;
                    if (pid < NCW * WN) {
                        const int sw = pid / WN, col = pid - sw * WN;
                        const char* sl = smem + sw * (SR * EP_RS) + col * 4;
                        float a = 0.0f, b = 0.0f;
#pragma unroll
                        for (int r = 0; r < SR; ++r) {
                            const float x = *reinterpret_cast<const float*>(sl + r * EP_RS);
                            a += x; b = fmaf(x, x, b);
                        }
                        gns[k] += a; gnq[k] += b;
                    }
                }
            }
            if (ih + 1 < MT * (32 / SR) || gn) {
                __builtin_amdgcn_s_waitcnt(0xc07f);      // slab reads done before the next round overwrites the slabs
                __builtin_amdgcn_s_barrier();
            }
        }
        if (gn) {
            // the block's WAVES_M wave rows of every column, summed in wave-row order by one thread per column
            float2* red = reinterpret_cast<float2*>(smem);
#pragma unroll
            for (int k = 0; k < GNP; ++k) {
                const int pid = (g + k * TW) * 64 + lane;
                if (pid < NCW * WN) {
                    const int sw = pid / WN, col = pid - sw * WN;
                    const int swm = sw / WAVES_N, swn = sw - swm * WAVES_N;
                    red[swm * BN + swn * WN + col] = make_float2(gns[k], gnq[k]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            const int tcol = (int)threadIdx.x;
            float a = 0.0f, b = 0.0f;
            if (tcol < BN && n0 + tcol < p.N) {
#pragma unroll
                for (int w2 = 0; w2 < WAVES_M; ++w2) { const float2 x = red[w2 * BN + tcol]; a += x.x; b += x.y; }
                p.gn_part[(int64_t)tile_m * p.N + n0 + tcol] = make_float2(a, b);
            }
            if (p.gn_grp) {       // whole groups of gn_cpg columns inside this tile: their sums, columns in order
                float2* tot = red + WAVES_M * BN;
                if (tcol < BN) tot[tcol] = make_float2(a, b);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                const int cpg = p.gn_cpg, c0g = tcol * cpg;
                if (c0g < BN && n0 + c0g < p.N) {
                    float ga = 0.0f, gb = 0.0f;
                    for (int j = 0; j < cpg; ++j) { const float2 x = tot[c0g + j]; ga += x.x; gb += x.y; }
                    p.gn_grp[(int64_t)tile_m * (p.N / cpg) + (n0 + c0g) / cpg] = make_float2(ga, gb);
                }
            }
        }
        MF_STAMP_DRAIN();
        MF_STAMP(5);
        if constexpr (SKF) {
            if (ws) splitk_combine_tail<BM, BN, NTHR + 256, DT == MF_F16>(p, smem, tile_m, tile_n, z, zq, zo, (int)threadIdx.x);
        }
        MF_STAMP(6);
        return;
    }
    const bool gn = p.gn_part != nullptr;
    constexpr int GNC = (WN + 63) / 64;          // columns per lane of the statistics pass over this wave's slab
    float gns[GNC], gnq[GNC];
#pragma unroll
    for (int k = 0; k < GNC; ++k) { gns[k] = 0.0f; gnq[k] = 0.0f; }
#pragma unroll
    for (int ih = 0; ih < MT * (32 / SR); ++ih) {
        const int i = ih / (32 / SR), half = ih % (32 / SR);     // accumulator rows [half*SR, half*SR + SR) of tile i
        uint4 q0[NIT], q1[NIT];
        if (res_pre) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int it = k * 64 + lane;
                const int row = it / CPR, ec = (it - row * CPR) * 8;
                const int m = m0 + wm * WM + i * 32 + half * SR + row;
                const int n = n0 + wn * WN + ec;
                q0[k] = uint4{0, 0, 0, 0}; q1[k] = uint4{0, 0, 0, 0};
                if ((ITEMS % 64 == 0 || it < ITEMS) && m < p.M && n + 8 <= p.N) {
                    if (p.res0) q0[k] = *reinterpret_cast<const uint4*>(p.res0 + ((int64_t)m * p.ld_res0 + n) * 2);
                    if (p.res1) q1[k] = *reinterpret_cast<const uint4*>(p.res1 + ((int64_t)res1_row(p, m) * p.ld_res1 + n) * 2);
                }
            }
        }
        if constexpr (P16) {      // acc16[2i + a][b] element r: row 32i + 16a + 4 kg + r, column 16b + r16 of the wave tile
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NT16; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + 4 * kg + r - half * SR;
                        if (SR == 32 || a == half)
                            *reinterpret_cast<float*>(slab + row * EP_RS + (16 * b + r16) * 4) = acc16[2 * i + a][b][r];
                    }
        } else
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = half * (SR / 2); e < half * (SR / 2) + SR / 2; ++e) {
                if constexpr (M16) {      // e = (a*2 + b)*4 + r: row 16a + 4 kg + r, column 16b + r16 (rows [16a, 16a + 16) = half a at SR 16)
                    const int row = 16 * (e >> 3) + 4 * kg + (e & 3) - half * SR;
                    *reinterpret_cast<float*>(slab + row * EP_RS + (j * 32 + 16 * ((e >> 2) & 1) + r16) * 4) = acc[i][j][e];
                } else {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * fh - half * SR;
                    *reinterpret_cast<float*>(slab + row * EP_RS + (j * 32 + frow) * 4) = acc[i][j][e];
                }
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the slab is written (LDS ops are in order per wave)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it0 = 0; it0 < ITEMS; it0 += 64) {
            const int it = it0 + lane;
            if (ITEMS % 64 != 0 && it >= ITEMS) continue;
            const int row = it / CPR, ec = (it - row * CPR) * 8;
            const int m = m0 + wm * WM + i * 32 + half * SR + row;
            const int n = n0 + wn * WN + ec;
            float v[8];
            const float4 lo = *reinterpret_cast<const float4*>(slab + row * EP_RS + ec * 4);
            const float4 hi = *reinterpret_cast<const float4*>(slab + row * EP_RS + ec * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            if (m < p.M && n < p.N) {
                if (ws) {
                    if (n + 8 <= p.N && (p.N & 3) == 0) {
                        *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) = lo;
                        *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n + 4) = hi;
                    } else {
                        for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) ws[(int64_t)m * p.N + n + jj] = v[jj];
                    }
                } else if (p.vec_ok && n + 8 <= p.N) {
                    epilogue_store8<DT == MF_F16>(p, zo, m, n, v, res_pre, q0[it0 / 64], q1[it0 / 64], zq);
                    if (gn) {      // the final values go back into the slab for the column pass below
                        *reinterpret_cast<float4*>(slab + row * EP_RS + ec * 4) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(slab + row * EP_RS + ec * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                } else {
                    for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) epilogue_store(p, zo, m, n + jj, v[jj], zq);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // slab reads done before the next slab overwrites it
        __builtin_amdgcn_wave_barrier();
        if (gn) {      // column pass over this wave's slab of final values (rows past M hold zeros: their A rows were zero pages)
#pragma unroll
            for (int k = 0; k < GNC; ++k) {
                const int col = lane + 64 * k;
                if (WN % 64 == 0 || col < WN) {
                    float a = 0.0f, b = 0.0f;
#pragma unroll
                    for (int r = 0; r < SR; ++r) {
                        const float x = *reinterpret_cast<const float*>(slab + r * EP_RS + col * 4);
                        a += x; b = fmaf(x, x, b);
                    }
                    gns[k] += a; gnq[k] += b;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (gn) {
        // the block's WAVES_M wave rows of every column, summed in wave-row order by one thread per column
        __syncthreads();                              // every wave is done with its slab
        float2* red = reinterpret_cast<float2*>(smem);
#pragma unroll
        for (int k = 0; k < GNC; ++k) {
            const int col = lane + 64 * k;
            if (WN % 64 == 0 || col < WN) red[wm * BN + wn * WN + col] = make_float2(gns[k], gnq[k]);
        }
        __syncthreads();
        float2* tot = red + WAVES_M * BN;
        for (int tcol = (int)threadIdx.x; tcol < BN; tcol += NTHR) {
            float a = 0.0f, b = 0.0f;
            if (n0 + tcol < p.N) {
#pragma unroll
                for (int w2 = 0; w2 < WAVES_M; ++w2) { const float2 x = red[w2 * BN + tcol]; a += x.x; b += x.y; }
                p.gn_part[(int64_t)tile_m * p.N + n0 + tcol] = make_float2(a, b);
            }
            if (p.gn_grp) tot[tcol] = make_float2(a, b);
        }
        if (p.gn_grp) {           // whole groups of gn_cpg columns inside this tile: their sums, columns in order
            __syncthreads();
            const int cpg = p.gn_cpg;
            for (int c0g = (int)threadIdx.x * cpg; c0g < BN; c0g += NTHR * cpg) {
                if (n0 + c0g < p.N) {
                    float ga = 0.0f, gb = 0.0f;
                    for (int j = 0; j < cpg; ++j) { const float2 x = tot[c0g + j]; ga += x.x; gb += x.y; }
                    p.gn_grp[(int64_t)tile_m * (p.N / cpg) + (n0 + c0g) / cpg] = make_float2(ga, gb);
                }
            }
        }
    }
    MF_STAMP_DRAIN();
    MF_STAMP(5);
    if constexpr (SKF) {
        if (ws) splitk_combine_tail<BM, BN, NTHR, DT == MF_F16>(p, smem, tile_m, tile_n, z, zq, zo, (int)threadIdx.x);
    }
    MF_STAMP(6);
}

template <int DT, int BM, int BN, int WMv, int WNv, bool AF, int ST, bool DX = false, bool WPK = false, bool M16 = false, bool WS = false,
          bool P16 = false, bool SKF = false, int FX = 0>
void launch_one(const GemmArgs& a, dim3 grid, hipStream_t s) {
    // warp-specialised ring tiles keep BM (mean, rstd) pairs of a folded LayerNorm past the ring
    constexpr int smem_k = DX ? 2 * (BM + (WS ? 32 : WMv * WNv * 8)) * 128 + (WS ? ST : 2) * BN * 128 : ST * (BM + BN) * 128 + (WS ? BM * 8 : 0);
    constexpr int smem_e = (FX & MF_FX_EPB) && epb_nimg(BM, BN, ST, DX, WS) > 0 ? (2 + epb_nimg(BM, BN, ST, DX, WS)) * epb_pitch(BN) * 4 : 0;   // staged epilogue rows
    static_assert(smem_k == epb_off(BM, BN, ST, DX, WS) || !WS, "epb_off must equal the ring's footprint");
    static_assert(smem_k + smem_e <= 160 * 1024, "LDS");
    // experiment switch: MFHIP_SMEM_MIN=<bytes> raises the LDS request (occupancy control for ring-depth A/B runs)
    static const int smem_min = getenv("MFHIP_SMEM_MIN") ? atoi(getenv("MFHIP_SMEM_MIN")) : 0;
    const int smem = smem_k + smem_e > smem_min ? smem_k + smem_e : smem_min;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_kernel<DT, BM, BN, WMv, WNv, AF, ST, DX, WPK, M16, WS, P16, SKF, FX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_conv_kernel<DT, BM, BN, WMv, WNv, AF, ST, DX, WPK, M16, WS, P16, SKF, FX>), grid, dim3(WMv * WNv * 64 + (WS ? 256 : 0)), smem, s, a);
}

// a tile with an SKF twin: the twin when the call carries tickets, the plain kernel otherwise
template <int DT, int BM, int BN, int WMv, int WNv, bool AF, int ST, bool DX = false, bool WPK = false, bool M16 = false, bool WS = false,
          bool P16 = false, int FX = 0>
void launch_skf(const GemmArgs& a, dim3 grid, hipStream_t s) {
    if (a.sk_tickets) launch_one<DT, BM, BN, WMv, WNv, AF, ST, DX, WPK, M16, WS, P16, true, FX>(a, grid, s);
    else launch_one<DT, BM, BN, WMv, WNv, AF, ST, DX, WPK, M16, WS, P16, false, FX>(a, grid, s);
}

// ---- tile groups, one translation unit each (false: the tile is not instantiated in that group) ----------------------
bool launch_nloop(int dtype, const GemmArgs& a, hipStream_t s);                                 // 69: the persistent short-K GEMM (gemm_nloop.hip)
bool launch_pers(int dtype, const GemmArgs& a, hipStream_t s);                                  // 70: its 128-row successor, 16 waves (gemm_pers.hip)
bool launch_bf16_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool a_f32);   // tiles 1-6 (+ fp32 activations, + in-launch split-K twins)
bool launch_bf16_b(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);               // 7-15, 20-24
bool launch_bf16_c(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);               // 25-36
bool launch_bf16_ws_dx(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);           // 37-40, 47, 49, 51: warp-specialised dx-reuse convs
bool launch_bf16_ws_ring(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);         // 41-46, 48, 50, 52: warp-specialised plain ring
bool launch_f16_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool a_f32);    // fp16: the bf16 groups' tiles on the f16 MFMA forms
bool launch_f16_b(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);
bool launch_f16_c(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);
bool launch_f16_ws_dx(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);
bool launch_f16_ws_ring(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);
bool launch_f32_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);                // fp32 MFMA, tiles 1-12
bool launch_f32_b(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);                // fp32 MFMA, tiles 13-15, 20-24, 31-36
bool launch_f16x3(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool wpk);
bool launch_f16x3_ws(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);             // pre-split W only
bool launch_bf16x3(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool wpk);
bool launch_bf16x1(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);
bool launch_fp8(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);
bool launch_halo_family(int tile, const GemmArgs& a, dim3 grid, hipStream_t s);          // 16-19 (conv_halo.hip)

}  // namespace mfgemm
