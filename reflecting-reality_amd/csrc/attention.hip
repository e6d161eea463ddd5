// Flash-style fused attention for gfx950, bf16 operands / fp32 softmax + accumulation, plus the same kernel in
// split precision (SP: every operand given as two fp16 planes hi + lo = the fp32 value to 22 bits, three MFMAs per
// product, P split in registers, fp32 output) — the parity mode runs THIS kernel, not a separate code path.
//
// Replaces F.scaled_dot_product_attention (reference models/attention_processor.py:1266-1268) for
// the SD1.5 head dims (40 / 80 / 160; also 8 and 64 for tiny test configs).
//
// Formulation (CDNA4): a workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32
// queries.  Per 64-key tile the wave computes S^T = K.Q^T with v_mfma_f32_32x32x16_bf16 (keys on the
// accumulator rows = registers, queries on the lanes).  With the query on the lane the online
// softmax needs no cross-lane traffic except one lane^32 exchange for the row max / sum, and the
// S^T accumulator, converted pairwise to bf16, is *already* the B operand of the second product
// O^T = V^T.P^T (accumulator rows -> k index, no LDS round trip).  V arrives pre-transposed
// ([head*d][keys], produced that way by the to_v GEMM with swapped operands) so both K and V^T
// fragments are single 16-byte LDS reads.  K and V^T tiles are staged by LDS-DMA (buffer_load ... lds, double
// buffered, no staging registers or ds_writes); the k-permutation implied by the accumulator layout
// (k = 16s + 8(j>>2) + 4h + (j&3)) is applied to the ROWS of the K tile (row R holds key R with bits 2 and 3
// swapped — a per-lane source offset), so P^T comes out in natural key order and V^T is staged as it lies in memory.
// The softmax rescale factor is per query = per lane, so rescaling O^T is a plain register multiply.
// O^T is transposed once through LDS at the end so the global stores are row-contiguous.
#include <stdlib.h>
#include <type_traits>
#include "mf_common.h"

namespace {

struct AttnArgs {
    const char* q; int64_t ldq;
    const char* k; int64_t ldk;
    const char* vt; int64_t ldvt;
    const char* q2; const char* k2; const char* vt2;   // SP: the low-half planes (same layout as q / k / vt)
    char* out; int64_t ldo;
    float* lse;   // optional [batch][heads][sq]: log2 of the row's softmax denominator in the exp2 domain (m + log2 l), for the backward pass
    int heads, sq, skv, batch;
    float c;   // softmax scale * log2(e)
    int no_xcd_order;   // A/B switch (MFHIP_ATTN_NOXCD=1): keep the hardware's round-robin block order
};

__device__ unsigned g_split_ovf_attn;     // raised when an fp16-split operand of this file exceeded the fp16 range (mf_common.h)

__device__ __forceinline__ uint4 ldg16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

__device__ __forceinline__ f32x16_t mfma16(const uint4& a, const uint4& b, const f32x16_t& c, bool f16) {
    return f16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0)
               : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// F16 (!SP): fp16 operands and output (MF_F16 storage) on the f16 MFMA forms — the byte layout of every tile equals the bf16 one; the
// packs / unpacks (Q~ scaling, P, the three pieces of the exponent offset, the ones) use fp16 instead of bf16.
template <int HD, bool DB, bool SP = false, bool F16 = false>
__global__ __launch_bounds__(256, (HD <= 80 && !SP) ? 3 : (SP && HD > 64 ? 1 : 2)) void attn_fwd_kernel(const AttnArgs p) {
    static_assert(!(SP && F16), "F16 is the single-plane form");
    constexpr unsigned ONE2 = F16 ? 0x3C003C00u : 0x3F803F80u;   // (1.0, 1.0)
    constexpr int NP = SP ? 2 : 1;            // operand planes (hi, lo)
    constexpr int OES = SP ? 4 : 2;           // output element size
    constexpr int DK = (HD + 15) / 16 * 16;   // QK^T reduction length, padded to the MFMA k-step
    constexpr int KS = DK / 16;
    constexpr int DV = (HD + 31) / 32 * 32;   // O^T rows, padded to the MFMA tile
    constexpr int DT = DV / 32;
    constexpr int RBK = DK * 2 + 16;          // K tile row stride (odd multiple of 16 B: conflict-free b128 reads)
    constexpr int RBV = 144;                  // V^T tile row stride: 64 keys * 2 B + 16
    constexpr int RBO = DV * OES + 16;        // epilogue transpose row stride
    constexpr int KCPR = RBK / 16, VCPR = RBV / 16;          // 16-byte chunks per LDS row (data + pad)
    constexpr int KI = KCPR;                                  // wave-instructions per K tile: 64 rows x KCPR chunks / 64 lanes
    constexpr int VI = (HD * VCPR + 63) / 64;                 // ... per V^T tile (rows < HD; the tail spills zeros into pad rows)
    constexpr int KPW = (KI + 3) / 4, VPW = (VI + 3) / 4;     // per wave
    constexpr int K_BYTES = 64 * RBK;
    constexpr int V_BYTES = (DV * RBV > VI * 1024 ? DV * RBV : VI * 1024);
    constexpr int O_BYTES = 4 * 32 * RBO;
    constexpr int BUF_BYTES = NP * (K_BYTES + V_BYTES);     // one K + V^T tile: [K planes][V^T planes]
    constexpr int V0 = NP * K_BYTES;                        // offset of the first V^T plane inside a buffer
    constexpr int NBUF = 2;
    constexpr int LDS_BYTES = (NBUF * BUF_BYTES) > O_BYTES ? (NBUF * BUF_BYTES) : O_BYTES;
    static_assert(DB, "K / V^T tiles are double buffered");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // MJ ("max injected"): head dims whose QK^T reduction has >= 3 spare (zero) k slots (40 -> 48, 8 -> 16) run the softmax
    // exponent ON THE MATRIX PIPE: Q is pre-multiplied by scale*log2(e) once, and the spare slots of the Q~ fragment carry
    // -m (the running exponent offset as three bf16 pieces, exact to 24 bits) against ones in K~, so the MFMA result already
    // is  log2(e)*scale*q.k - m  and the 32 v_fma per lane and tile disappear (the loop is VALU-issue bound: PMC round 2).  m only
    // moves when some score exceeds the offset by more than MJ_T (defer-max: P <= 2^MJ_T, exact in the fp32 sums), so the
    // rescale of O^T and the update of the Q~ slots are rare wave-uniform branches.
    // (Round 3 also measured a 64-queries-per-wave form of this kernel — every K / V^T fragment feeding two MFMAs, 256-query blocks,
    // a 2 / 3 / 4-deep K/V ring under counted vmcnt waits: 252 / 264 / 268 us against 247 us for this kernel on the same box
    // (gpurun_out/r03e): neither the L2 -> LDS fill nor the LDS reads bound the d = 40 loop, so the form was removed.)
    constexpr bool MJ = !SP && (DK - HD >= 3);
    constexpr int SP_SHIFT = 8;
    constexpr float MJ_T = 5.0f;
    static_assert(!MJ || HD % 16 == 8, "MJ: the pad k slots must be the whole last fragment of the h = 1 half-wave");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware order: hardware deals consecutive workgroups round-robin over the 8 XCDs; give each XCD a contiguous run of
    // logical blocks so that the query blocks of one (batch, head) — which stream the same K / V^T — share one XCD's L2
    // (the plain (x, head, batch) grid spread every head over all 8 L2s: 340 MB fetched for 84 MB of operands, PMC round 2)
    int bid = blockIdx.x;
    if (!p.no_xcd_order) {
        const int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + j;
    }
    const int qblocks = (p.sq + 127) >> 7;
    const int bx = bid % qblocks, bh = bid / qblocks;
    const int b = bh / p.heads, head = bh - b * p.heads;
    const int q0 = bx * 128 + wave * 32;
    const int qi = q0 + r;

    // zero the whole staging area once: pad columns / rows must never hold NaN bit patterns
    for (int i = tid * 16; i < NBUF * BUF_BYTES; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
    // Head dims that leave pad rows in the 32-row V^T tiles (40, 80, 8) get a row of ones there: O^T row HD then
    // accumulates sum_k P[k][q] on the matrix pipe, rescaled with O like every other row, and the VALU row sum goes away.
    constexpr bool ONES = DV > HD;
    constexpr int ONES_ROW = DV - 1;                          // the last pad row: the V^T DMA's zero spill never reaches it
    constexpr int L_D = ONES_ROW / 32, L_RR = ONES_ROW % 32, L_E = (L_RR & 3) + 4 * (L_RR >> 3), L_H = (L_RR >> 2) & 1;
    if (ONES) {
        __syncthreads();
        for (int i = tid; i < NBUF * 32; i += 256)           // 64 keys = 128 B = 32 dwords per buffer
            *reinterpret_cast<uint32_t*>(smem + (i >> 5) * BUF_BYTES + V0 + ONES_ROW * RBV + (i & 31) * 4) = SP ? 0x3C003C00u : ONE2;   // 1.0 (hi plane only)
    }

    // Q fragments (B operand of S^T = K.Q^T): lane (query r, half h) holds Q[q][16ks + 8h + j]
    uint4 qf[NP][KS];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const char* qrow = (pl ? p.q2 : p.q) + (((int64_t)b * p.sq + (qi < p.sq ? qi : 0)) * p.ldq + head * HD) * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 16 * ks + 8 * h;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kk < HD && qi < p.sq) v = ldg16(qrow + kk * 2);
            if constexpr (MJ) {          // Q~ = bf16(q * scale * log2 e): scores leave the MFMA in exp2 units
                unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0, x1;
                    unpack_h2<F16>(w[e], x0, x1);
                    w[e] = pack_h2<F16>(x0 * p.c, x1 * p.c);
                }
            }
            qf[pl][ks] = v;
        }
    }

    const int64_t kb_off = ((int64_t)b * p.skv * p.ldk + head * HD) * 2;
    const int64_t vb_off = ((int64_t)(b * p.heads + head) * HD) * p.ldvt * 2;

    // LDS-DMA staging.  A wave-instruction fills 64 consecutive 16-byte chunks of the (padded) LDS tile; lane chunk g
    // of the K tile is (row R = g / KCPR, chunk c = g % KCPR) and reads key row perm(R) of this head (pad chunks and
    // rows past the tensor: offset >= num_records -> zeros); the V^T tile is (row = g / VCPR, c) -> 8 keys of channel
    // `row`.  Offsets advance by a constant per 64-key tile.  Keys past skv inside the last tile read whatever follows
    // in memory (the next batch's rows, finite, or zeros past the tensor): their scores are masked to -inf below and
    // their probabilities are exact zeros.
    const int64_t k_left = ((int64_t)(p.batch - b) * p.skv * p.ldk - head * HD) * 2;
    const int64_t v_left = ((int64_t)((p.batch - b) * p.heads - head) * HD * p.ldvt) * 2;
    srd_t srdK[NP], srdV[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        srdK[pl] = make_srd((pl ? p.k2 : p.k) + kb_off, (unsigned)(k_left < 0x7fffffff ? k_left : 0x7fffffff));
        srdV[pl] = make_srd((pl ? p.vt2 : p.vt) + vb_off, (unsigned)(v_left < 0x7fffffff ? v_left : 0x7fffffff));
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned koff[KPW], voff[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int g = (wv + 4 * i) * 64 + lane;
        const int R = g / KCPR, c = g - R * KCPR;
        const int kr = (R & ~12) | ((R & 4) << 1) | ((R & 8) >> 1);
        const int64_t off = ((int64_t)kr * p.ldk + c * 8) * 2;
        koff[i] = (c < HD / 8 && off < 0x7fffffff) ? (unsigned)off : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const int g = (wv + 4 * i) * 64 + lane;
        const int row = g / VCPR, c = g - row * VCPR;
        const int64_t off = ((int64_t)row * p.ldvt + c * 8) * 2;
        voff[i] = (row < HD && c < 8 && off < 0x7fffffff) ? (unsigned)off : 0x80000000u;
    }
    const unsigned kstep = (unsigned)(64 * p.ldk * 2);
    auto issue_tile = [&](int buf) {
        const unsigned lk = lds0 + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                if (wv + 4 * i < KI) dma16_buf(koff[i], srdK[pl], lk + pl * K_BYTES + (wv + 4 * i) * 1024);
            koff[i] += kstep;          // an out-of-range lane stays out of range: 0x80000000 + n * kstep < 2^32 for every tile
        }
#pragma unroll
        for (int i = 0; i < VPW; ++i) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                if (wv + 4 * i < VI) dma16_buf(voff[i], srdV[pl], lk + V0 + pl * V_BYTES + (wv + 4 * i) * 1024);
            voff[i] += 128;
        }
    };

    f32x16_t o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.0f;
    float m = MJ ? 0.0f : -INFINITY, l = 0.0f;
    // MJ: bf16 ones in k slots HD, HD+1, HD+2 of the last K fragment, for the half-wave that reads the pad chunk
    const unsigned ones_x = (MJ && h) ? ONE2 : 0u, ones_y = (MJ && h) ? (ONE2 & 0xffffu) : 0u;
    (void)ones_x; (void)ones_y;

    const int ntiles = (p.skv + 63) / 64;
    __syncthreads();   // zero-fill (and the ones rows) done
    issue_tile(0);
    wait_vmcnt<0>();
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        const char* Ks = smem + (t & 1) * BUF_BYTES;
        const char* Vs = Ks + V0;
        // the other buffer was last read in iteration t-1, which every wave finished before the barrier that ended it
        if (t + 1 < ntiles) issue_tile((t + 1) & 1);       // the DMA flies under this tile's MFMAs and softmax

        // ---- S^T = K . Q^T : two 32-key tiles ----
        f32x16_t st[2];
        const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // inline-constant src2, no register zeroing
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const uint4 a = *reinterpret_cast<const uint4*>(Ks + (32 * tt + r) * RBK + (16 * ks + 8 * h) * 2);
                if constexpr (SP) {
                    const uint4 a2 = *reinterpret_cast<const uint4*>(Ks + K_BYTES + (32 * tt + r) * RBK + (16 * ks + 8 * h) * 2);
                    st[tt] = mfma16(a2, qf[0][ks], ks == 0 ? zero16 : st[tt], true);      // the small terms first
                    st[tt] = mfma16(a, qf[1][ks], st[tt], true);
                    st[tt] = mfma16(a, qf[0][ks], st[tt], true);
                } else if constexpr (MJ) {
                    uint4 am = a;
                    if (ks == KS - 1) { am.x |= ones_x; am.y |= ones_y; }     // K~[key][HD .. HD+2] = 1 (the DMA left zeros there)
                    st[tt] = mfma16(am, qf[0][ks], ks == 0 ? zero16 : st[tt], F16);
                } else {
                    st[tt] = mfma16(a, qf[0][ks], ks == 0 ? zero16 : st[tt], F16);
                }
            }
        }
        if (kv0 + 64 > p.skv) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + 32 * tt + (e & 3) + 4 * (e >> 2 & 1) + 16 * (e >> 3) + 8 * h;   // row with bits 2, 3 swapped
                    if (key >= p.skv) st[tt][e] = -INFINITY;
                }
        }
        // ---- online softmax (per query = per lane; the two lane halves hold disjoint keys) ----
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(fmaxf(mx, st[0][e]), st[1][e]);     // v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float rs = 0.0f;
        if constexpr (MJ) {
            // st already is score - m.  Move the offset only when some lane's scores run more than MJ_T above it (and on
            // the first tile, where the offset is still 0): wave-uniform, rare after the first few tiles
            if (t == 0 || __builtin_amdgcn_ballot_w64(mx > MJ_T) != 0) {
                const float want = m + (t == 0 ? mx : fmaxf(mx, 0.0f));
                // the offset the matrix pipe can subtract exactly: three 16-bit pieces (bf16: 24 bits; fp16: 33, the last piece
                // may be an fp16 subnormal — the matrix pipe keeps those)
                float f0, f1, f2, dummy;
                const unsigned b0 = pack_h2<F16>(want, 0.0f) & 0xffffu;
                unpack_h2<F16>(b0, f0, dummy);
                const float r1 = want - f0;
                const unsigned b1 = pack_h2<F16>(r1, 0.0f) & 0xffffu;
                unpack_h2<F16>(b1, f1, dummy);
                const float r2 = r1 - f1;
                const unsigned b2 = pack_h2<F16>(r2, 0.0f) & 0xffffu;
                unpack_h2<F16>(b2, f2, dummy);
                const float m_rep = f0 + f1 + f2;
                const float dlt = m_rep - m;
                m = m_rep;
                if (h) {               // Q~[q][HD .. HD+2] = -(b0, b1, b2): the half-wave whose last fragment is the pad slots
                    qf[0][KS - 1].x = (b0 | (b1 << 16)) ^ 0x80008000u;
                    qf[0][KS - 1].y = b2 ^ 0x8000u;
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) st[tt][e] -= dlt;
                if (t != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-dlt);
                    l *= alpha;
#pragma unroll
                    for (int d = 0; d < DT; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(st[tt][e]);
                    st[tt][e] = pv;
                    if (!ONES) rs += pv;
                }
        } else {
            const float m_new = fmaxf(m, mx * p.c);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            // SP: P * 2^SP_SHIFT (<= 256), so that the low fp16 half of a small probability is not an fp16 subnormal with a
            // handful of bits; l and O^T carry the same factor and it cancels in O = O^T / l (lse takes it back out)
            const float m_sub = SP ? m_new - (float)SP_SHIFT : m_new;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[tt][e], p.c, -m_sub));
                    st[tt][e] = pv;
                    if (!ONES) rs += pv;
                }
            // the running max settles after the first few tiles: skip the O^T rescale when no lane's max moved
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                l *= alpha;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
            }
        }
        if (!ONES) l += rs;
        // ---- P^T fragments: accumulator registers 8s..8s+7 of tile tt are k-step 2*tt+s ----
        uint4 pf[NP][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (SP) {
                    // P = hi + lo in fp16 halves (hi toward zero, lo = P - hi exact in fp32); P <= 2^SP_SHIFT, and fp16
                    // subnormals survive both v_cvt_pkrtz_f16_f32 and the matrix pipe (tools/micro/f16_denorm.hip)
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = st[tt][8 * s + 2 * e], a1 = st[tt][8 * s + 2 * e + 1];
                        mf_split_f16x2(a0, a1, hw[e], lw[e]);
                    }
                    pf[0][2 * tt + s] = uint4{hw[0], hw[1], hw[2], hw[3]};
                    pf[1][2 * tt + s] = uint4{lw[0], lw[1], lw[2], lw[3]};
                } else {
                    uint4 u;
                    u.x = pack_h2<F16>(st[tt][8 * s + 0], st[tt][8 * s + 1]);
                    u.y = pack_h2<F16>(st[tt][8 * s + 2], st[tt][8 * s + 3]);
                    u.z = pack_h2<F16>(st[tt][8 * s + 4], st[tt][8 * s + 5]);
                    u.w = pack_h2<F16>(st[tt][8 * s + 6], st[tt][8 * s + 7]);
                    pf[0][2 * tt + s] = u;
                }
            }
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int kst = 0; kst < 4; ++kst)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const uint4 a = *reinterpret_cast<const uint4*>(Vs + (32 * d + r) * RBV + (16 * kst + 8 * h) * 2);
                if constexpr (SP) {
                    const uint4 a2 = *reinterpret_cast<const uint4*>(Vs + V_BYTES + (32 * d + r) * RBV + (16 * kst + 8 * h) * 2);
                    o[d] = mfma16(a2, pf[0][kst], o[d], true);
                    o[d] = mfma16(a, pf[1][kst], o[d], true);
                    o[d] = mfma16(a, pf[0][kst], o[d], true);
                } else {
                    o[d] = mfma16(a, pf[0][kst], o[d], F16);
                }
            }
        wait_vmcnt<0>();      // this wave's share of tile t+1 has landed ...
        __syncthreads();      // ... and so has everybody else's; every wave is done reading tile t
    }

    // ---- epilogue: normalise, transpose through LDS, row-contiguous stores ----
    if (ONES) l = __shfl(o[L_D][L_E], (lane & 31) + 32 * L_H, 64);      // the half-wave that holds accumulator row ONES_ROW
    else l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (p.lse && h == 0 && qi < p.sq) p.lse[((int64_t)b * p.heads + head) * p.sq + qi] = m + __builtin_amdgcn_logf(l) - (SP ? (float)SP_SHIFT : 0.0f);   // v_log_f32 = log2
    char* Os = smem + wave * 32 * RBO;   // safe: the loop ended on a barrier
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (SP) {
                *reinterpret_cast<float4*>(Os + r * RBO + (32 * d + 8 * g + 4 * h) * 4) =
                    make_float4(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
            } else {
                uint2 u;
                u.x = pack_h2<F16>(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
                u.y = pack_h2<F16>(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(Os + r * RBO + (32 * d + 8 * g + 4 * h) * 2) = u;
            }
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    constexpr int EPV = 16 / OES;         // elements per 16-byte vector of the output
    constexpr int OV = 32 * (HD / EPV);
    for (int v = lane; v < OV; v += 64) {
        const int row = v / (HD / EPV), cv = v - row * (HD / EPV);
        const int qq = q0 + row;
        if (qq < p.sq) {
            const uint4 val = *reinterpret_cast<const uint4*>(Os + row * RBO + cv * 16);
            *reinterpret_cast<uint4*>(p.out + (((int64_t)b * p.sq + qq) * p.ldo + head * HD + cv * EPV) * OES) = val;
        }
    }
}



// =====================================================================================================================
// Flash-style attention BACKWARD in split precision (round 3).  Replaces the unfused backward of the training step
// (autograd.record_attention: scores / softmax / softmax-backward over [B*heads][Sq][Skv] fp32 tensors, two of them transposed:
// ~60 GB of HBM traffic per 4096-token layer) by two launches of ONE kernel that recompute P tile by tile from Q, K and the
// forward's row statistics (lse, in the exp2 domain) — nothing of size Sq x Skv is ever written:
//   KV = false: a wave owns 32 QUERIES (lanes), streams key tiles:   dQ^T += K^T . dS^T
//   KV = true : a wave owns 32 KEYS (lanes), streams query tiles:    dK^T += Q^T . dS,  dV^T += dO^T . P
// with  T = (row tile) . (column fragments) = q.k,   U = (second row tile) . (second fragments) = dO.v = dP,
//       P = exp2(T c - lse),   dS = scale * P * (U - D),   D[q] = sum_c dO[q][c] O[q][c]   (mf_rowdot_heads).
// The operand roles are the forward kernel's: a row-major tile with permuted rows is the MFMA A operand against register-resident
// column fragments (accumulator rows = tile rows in natural order, lane = column item), and the accumulator, split into
// (hi, lo) fp16 halves, is directly the B operand of the product with a TRANSPOSED tile (channels x items).  Every operand is
// given as two fp16 planes (mf_split_halves); three MFMAs per product.  Deterministic: no atomics (dQ has its own pass).
struct AttnBwdArgs {
    const char* c1[2]; int64_t ldc1;      // column-side fragments, operand 1: dQ pass Q, dK/dV pass K      [B][Sc][ld] planes (hi, lo)
    const char* c2[2]; int64_t ldc2;      // column-side fragments, operand 2: dQ pass dO, dK/dV pass V
    const char* r1[2]; int64_t ldr1;      // streamed row-major tiles, operand 1: dQ pass K, dK/dV pass Q       [B][Sr][ld]
    const char* r2[2]; int64_t ldr2;      // streamed row-major tiles, operand 2: dQ pass V, dK/dV pass dO
    const char* t1[2]; int64_t ldt1;      // streamed transposed tiles for out1: dQ pass K^T, dK/dV pass Q^T       [B][heads*d][ldt]
    const char* t2[2]; int64_t ldt2;      // dK/dV pass only: dO^T (for out2 = dV)
    const float* lse; const float* dd;    // [B][heads][Sq]
    float* out1; float* out2; int64_t ldo;   // fp32 (out16: bf16) [B][Sc][ldo]: dQ, or dK and dV
    int out16;
    int heads, sc, sr, batch, sq;         // sc / sr: items on the column / row side; sq: queries (lse / dd row length)
    float c, scale;                       // scale * log2(e), scale
    // P and dS are split into fp16 halves whose low half is an fp16 SUBNORMAL for the small probabilities of a long row
    // (P ~ 1 / 4096 keeps ~12 bits): both are computed times 2^pshift (folded into the exp2 argument, exact) and the
    // accumulators are scaled back by 2^-pshift in the epilogue
    float pshift, inv_pscale;
};

// Eight waves per block (256 column items against one streamed tile): two waves per SIMD, so one wave's exp2 / split VALU work
// runs under the other's MFMAs, and each staged tile feeds twice the products of the first version's four waves.
constexpr int BWD_NW = 8;
// B16: the same kernel on ONE bf16 plane per operand and one bf16 MFMA per product (P and dS rounded to bf16): the attention
// backward of the bf16x1 training mode on pre-rounded operands — a third of the matrix work of the split form.
template <int HD, bool KV, bool B16 = false>
__global__ __launch_bounds__(BWD_NW * 64, 1) void attn_bwd_kernel(const AttnBwdArgs p) {
    constexpr int NP = B16 ? 1 : 2, NW = BWD_NW;
    constexpr int DK = (HD + 15) / 16 * 16, KS = DK / 16;
    constexpr int DV = (HD + 31) / 32 * 32, DT = DV / 32;
    constexpr int RBK = DK * 2 + 16, RBV = 144, RBO = DV * 4 + 16;
    constexpr int KCPR = RBK / 16, VCPR = RBV / 16;
    constexpr int KI = KCPR, VI = (HD * VCPR + 63) / 64;
    constexpr int KPW = (KI + NW - 1) / NW, VPW = (VI + NW - 1) / NW;
    constexpr int K_BYTES = 64 * RBK;
    constexpr int V_BYTES = (DV * RBV > VI * 1024 ? DV * RBV : VI * 1024);
    constexpr int NT = KV ? 2 : 1;                            // transposed operands per tile
    constexpr int ST_BYTES = KV ? 2 * 1024 : 0;                // lse | dd of the 64 tile rows (dK/dV pass): one 1-KB DMA span each
    constexpr int R2_OFF = NP * K_BYTES, T1_OFF = 2 * NP * K_BYTES, T2_OFF = T1_OFF + NP * V_BYTES, ST_OFF = T1_OFF + NT * NP * V_BYTES;
    constexpr int BUF_BYTES = ST_OFF + ST_BYTES;
    constexpr int O_BYTES = NW * 32 * RBO;
    constexpr int LDS_BYTES = 2 * BUF_BYTES > O_BYTES ? 2 * BUF_BYTES : O_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cblocks = (p.sc + NW * 32 - 1) / (NW * 32);
    const int bx = blockIdx.x % cblocks, bh = blockIdx.x / cblocks;
    const int b = bh / p.heads, head = bh - b * p.heads;
    const int c0 = bx * (NW * 32) + wave * 32;
    const int ci = c0 + r;                                    // this lane's column item (query or key)

    for (int i = tid * 16; i < 2 * BUF_BYTES; i += NW * 64 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

    // column-side fragments: lane (item r, half h) holds X[item][16 ks + 8 h + j] of both planes
    uint4 f1[NP][KS], f2[NP][KS];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const char* row1 = p.c1[pl] + (((int64_t)b * p.sc + (ci < p.sc ? ci : 0)) * p.ldc1 + head * HD) * 2;
        const char* row2 = p.c2[pl] + (((int64_t)b * p.sc + (ci < p.sc ? ci : 0)) * p.ldc2 + head * HD) * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 16 * ks + 8 * h;
            const bool on = kk < HD && ci < p.sc;
            f1[pl][ks] = on ? ldg16(row1 + kk * 2) : make_uint4(0, 0, 0, 0);
            f2[pl][ks] = on ? ldg16(row2 + kk * 2) : make_uint4(0, 0, 0, 0);
        }
    }
    float lse_l = 0.0f, dd_l = 0.0f;                           // dQ pass: the lane's own query
    if (!KV && ci < p.sc) {
        lse_l = p.lse[((int64_t)b * p.heads + head) * p.sq + ci] - p.pshift;
        dd_l = p.dd[((int64_t)b * p.heads + head) * p.sq + ci];
    }

    // LDS-DMA staging of the streamed tiles (the forward kernel's scheme: permuted rows for the row-major tiles, natural
    // order for the transposed ones; out-of-range lanes write zeros)
    srd_t srdR1[NP], srdR2[NP], srdT1[NP], srdT2[NP];
    {
        const int64_t rb1 = ((int64_t)b * p.sr * p.ldr1 + head * HD) * 2, rl1 = ((int64_t)(p.batch - b) * p.sr * p.ldr1 - head * HD) * 2;
        const int64_t rb2 = ((int64_t)b * p.sr * p.ldr2 + head * HD) * 2, rl2 = ((int64_t)(p.batch - b) * p.sr * p.ldr2 - head * HD) * 2;
        const int64_t tb1 = ((int64_t)(b * p.heads + head) * HD) * p.ldt1 * 2, tl1 = ((int64_t)((p.batch - b) * p.heads - head) * HD * p.ldt1) * 2;
        const int64_t tb2 = KV ? ((int64_t)(b * p.heads + head) * HD) * p.ldt2 * 2 : 0;
        const int64_t tl2 = KV ? ((int64_t)((p.batch - b) * p.heads - head) * HD * p.ldt2) * 2 : 0;
        auto lim = [](int64_t v) { return (unsigned)(v < 0x7fffffff ? v : 0x7fffffff); };
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            srdR1[pl] = make_srd(p.r1[pl] + rb1, lim(rl1));
            srdR2[pl] = make_srd(p.r2[pl] + rb2, lim(rl2));
            srdT1[pl] = make_srd(p.t1[pl] + tb1, lim(tl1));
            srdT2[pl] = make_srd((KV ? p.t2[pl] : p.t1[pl]) + tb2, lim(KV ? tl2 : tl1));
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned roff1[KPW], roff2[KPW], toff1[VPW], toff2[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int g = (wv + NW * i) * 64 + lane;
        const int R = g / KCPR, c = g - R * KCPR;
        const int kr = (R & ~12) | ((R & 4) << 1) | ((R & 8) >> 1);
        const int64_t o1 = ((int64_t)kr * p.ldr1 + c * 8) * 2, o2 = ((int64_t)kr * p.ldr2 + c * 8) * 2;
        roff1[i] = (c < HD / 8 && o1 < 0x7fffffff) ? (unsigned)o1 : 0x80000000u;
        roff2[i] = (c < HD / 8 && o2 < 0x7fffffff) ? (unsigned)o2 : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const int g = (wv + NW * i) * 64 + lane;
        const int row = g / VCPR, c = g - row * VCPR;
        const int64_t o1 = ((int64_t)row * p.ldt1 + c * 8) * 2, o2 = ((int64_t)row * p.ldt2 + c * 8) * 2;
        toff1[i] = (row < HD && c < 8 && o1 < 0x7fffffff) ? (unsigned)o1 : 0x80000000u;
        toff2[i] = (KV && row < HD && c < 8 && o2 < 0x7fffffff) ? (unsigned)o2 : 0x80000000u;
    }
    const unsigned rstep1 = (unsigned)(64 * p.ldr1 * 2), rstep2 = (unsigned)(64 * p.ldr2 * 2);
    // the tile's 64 lse / dd values arrive by DMA too (lanes 0..15 of wave 0 fetch 16 bytes each; rows past the end read zeros)
    const int64_t st_base = ((int64_t)b * p.heads + head) * p.sq;
    const srd_t srdL = make_srd(reinterpret_cast<const char*>(p.lse + st_base), (unsigned)(p.sq * 4));
    const srd_t srdD = make_srd(reinterpret_cast<const char*>(p.dd + st_base), (unsigned)(p.sq * 4));
    unsigned soff = lane < 16 ? (unsigned)(lane * 16) : 0x80000000u;
    auto issue_tile = [&](int buf, int row0) {
        const unsigned lb = lds0 + buf * BUF_BYTES;
        if (KV && wv == 0) {
            dma16_buf(soff, srdL, lb + ST_OFF);
            dma16_buf(soff, srdD, lb + ST_OFF + 1024);
            soff += 256;
        }
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                if (wv + NW * i < KI) {
                    dma16_buf(roff1[i], srdR1[pl], lb + pl * K_BYTES + (wv + NW * i) * 1024);
                    dma16_buf(roff2[i], srdR2[pl], lb + R2_OFF + pl * K_BYTES + (wv + NW * i) * 1024);
                }
            roff1[i] += rstep1;
            roff2[i] += rstep2;
        }
#pragma unroll
        for (int i = 0; i < VPW; ++i) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                if (wv + NW * i < VI) {
                    dma16_buf(toff1[i], srdT1[pl], lb + T1_OFF + pl * V_BYTES + (wv + NW * i) * 1024);
                    if (KV) dma16_buf(toff2[i], srdT2[pl], lb + T2_OFF + pl * V_BYTES + (wv + NW * i) * 1024);
                }
            toff1[i] += 128;
            toff2[i] += 128;
        }
        (void)row0;
    };

    f32x16_t acc1[DT], acc2[KV ? DT : 1];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[d][e] = 0.0f; if (KV) acc2[d][e] = 0.0f; }
    float amax = 0.0f;                                         // range guard of the fp16 split (dS can be large)

    const int ntiles = (p.sr + 63) / 64;
    __syncthreads();
    issue_tile(0, 0);
    wait_vmcnt<0>();
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int row0 = t * 64;
        const char* B0 = smem + (t & 1) * BUF_BYTES;
        if (t + 1 < ntiles) issue_tile((t + 1) & 1, row0 + 64);

        // ---- T = R1 . f1^T and U = R2 . f2^T : [64 tile rows (registers)] x [32 column items (lanes)] ----
        f32x16_t T[2], U[2];
        const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int fo = (32 * tt + r) * RBK + (16 * ks + 8 * h) * 2;
                const uint4 ah = *reinterpret_cast<const uint4*>(B0 + fo);
                const uint4 bh_ = *reinterpret_cast<const uint4*>(B0 + R2_OFF + fo);
                if constexpr (B16) {
                    T[tt] = mfma16(ah, f1[0][ks], ks == 0 ? zero16 : T[tt], false);
                    U[tt] = mfma16(bh_, f2[0][ks], ks == 0 ? zero16 : U[tt], false);
                } else {
                    const uint4 al = *reinterpret_cast<const uint4*>(B0 + K_BYTES + fo);
                    T[tt] = mfma16(al, f1[0][ks], ks == 0 ? zero16 : T[tt], true);
                    T[tt] = mfma16(ah, f1[NP - 1][ks], T[tt], true);
                    T[tt] = mfma16(ah, f1[0][ks], T[tt], true);
                    const uint4 bl_ = *reinterpret_cast<const uint4*>(B0 + R2_OFF + K_BYTES + fo);
                    U[tt] = mfma16(bl_, f2[0][ks], ks == 0 ? zero16 : U[tt], true);
                    U[tt] = mfma16(bh_, f2[NP - 1][ks], U[tt], true);
                    U[tt] = mfma16(bh_, f2[0][ks], U[tt], true);
                }
            }
        // ---- P = exp2(T c - lse), dS = scale P (U - D); accumulator register e of sub-tile tt is tile row
        //      32 tt + (e & 3) + 4 ((e >> 2) & 1) + 16 (e >> 3) + 8 h (natural order: the tile's rows are stored permuted) ----
        uint4 ph[4], plo[4], sh[4], slo[4];
        if constexpr (!B16) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float lv[16], dv[16];
            if (KV) {
                const float* stl = reinterpret_cast<const float*>(B0 + ST_OFF);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = 32 * tt + (e & 7) + 16 * (e >> 3) + 8 * h;
                    lv[e] = stl[n] - p.pshift;
                    dv[e] = stl[256 + n];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) { lv[e] = lse_l; dv[e] = dd_l; }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = row0 + 32 * tt + (e & 7) + 16 * (e >> 3) + 8 * h;
                float pv = __builtin_amdgcn_exp2f(fmaf(T[tt][e], p.c, -lv[e]));
                if (n >= p.sr) pv = 0.0f;
                const float ds = p.scale * pv * (U[tt][e] - dv[e]);
                if constexpr (!B16) amax = mf_amax3(amax, ds, 0.0f);
                T[tt][e] = pv;
                U[tt][e] = ds;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                unsigned a_h[4], a_l[4], b_h[4], b_l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p0 = T[tt][8 * s2 + 2 * e], p1 = T[tt][8 * s2 + 2 * e + 1];
                    const float d0 = U[tt][8 * s2 + 2 * e], d1 = U[tt][8 * s2 + 2 * e + 1];
                    if constexpr (B16) {
                        a_h[e] = pack_bf16x2(p0, p1); b_h[e] = pack_bf16x2(d0, d1); a_l[e] = 0; b_l[e] = 0;
                    } else {
                        mf_split_f16x2(p0, p1, a_h[e], a_l[e]);
                        mf_split_f16x2(d0, d1, b_h[e], b_l[e]);
                    }
                }
                ph[2 * tt + s2] = uint4{a_h[0], a_h[1], a_h[2], a_h[3]};
                plo[2 * tt + s2] = uint4{a_l[0], a_l[1], a_l[2], a_l[3]};
                sh[2 * tt + s2] = uint4{b_h[0], b_h[1], b_h[2], b_h[3]};
                slo[2 * tt + s2] = uint4{b_l[0], b_l[1], b_l[2], b_l[3]};
            }
        }
        } else {
        // MASK: only the LAST row tile can reach past the sequence (its DMA rows are zeros, exp2(0 - lse) is not): every other tile
        // runs the form without the per-element row test — a third of this section's VALU instructions, and the section, not the
        // MFMAs, bounds the single-plane (B16) kernel.  Pairs of elements are written as 2-vectors: v_pk_fma / v_pk_add / v_pk_mul.
        auto elementwise = [&](auto mask_tag) {
            constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float lv[16], dv[16];
                if (KV) {
                    const float* stl = reinterpret_cast<const float*>(B0 + ST_OFF);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = 32 * tt + (e & 7) + 16 * (e >> 3) + 8 * h;
                        lv[e] = stl[n] - p.pshift;
                        dv[e] = stl[256 + n];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { lv[e] = lse_l; dv[e] = dd_l; }
                }
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const mf_f32x2_t tv = {T[tt][e], T[tt][e + 1]}, l2 = {lv[e], lv[e + 1]}, uv = {U[tt][e], U[tt][e + 1]}, d2 = {dv[e], dv[e + 1]};
                    const mf_f32x2_t c2 = {p.c, p.c}, s2v = {p.scale, p.scale};
                    const mf_f32x2_t arg = __builtin_elementwise_fma(tv, c2, -l2);
                    mf_f32x2_t pv = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                    if constexpr (MASK) {
                        const int n = row0 + 32 * tt + (e & 7) + 16 * (e >> 3) + 8 * h;
                        if (n >= p.sr) pv.x = 0.0f;
                        if (n + 1 >= p.sr) pv.y = 0.0f;
                    }
                    const mf_f32x2_t ds = (s2v * pv) * (uv - d2);
                    T[tt][e] = pv.x; T[tt][e + 1] = pv.y;
                    U[tt][e] = ds.x; U[tt][e + 1] = ds.y;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    unsigned a_h[4], a_l[4], b_h[4], b_l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p0 = T[tt][8 * s2 + 2 * e], p1 = T[tt][8 * s2 + 2 * e + 1];
                        const float d0 = U[tt][8 * s2 + 2 * e], d1 = U[tt][8 * s2 + 2 * e + 1];
                        if constexpr (B16) {
                            a_h[e] = pack_bf16x2(p0, p1); b_h[e] = pack_bf16x2(d0, d1); a_l[e] = 0; b_l[e] = 0;
                        } else {
                            mf_split_f16x2(p0, p1, a_h[e], a_l[e]);
                            mf_split_f16x2(d0, d1, b_h[e], b_l[e]);
                        }
                    }
                    ph[2 * tt + s2] = uint4{a_h[0], a_h[1], a_h[2], a_h[3]};
                    plo[2 * tt + s2] = uint4{a_l[0], a_l[1], a_l[2], a_l[3]};
                    sh[2 * tt + s2] = uint4{b_h[0], b_h[1], b_h[2], b_h[3]};
                    slo[2 * tt + s2] = uint4{b_l[0], b_l[1], b_l[2], b_l[3]};
                }
            }
        };
        // (two copies of the section cost registers: the split-precision forms and head dim 80 sit at their budget and keep the one
        // masked copy — they are matrix-bound anyway)
        if constexpr (HD <= 40) {
            if (row0 + 64 > p.sr) elementwise(std::true_type{});
            else elementwise(std::false_type{});
        } else {
            elementwise(std::true_type{});
        }
        }
        // ---- out1^T += T1 . dS   (and out2^T += T2 . P): [channels (registers)] x [32 column items (lanes)] ----
#pragma unroll
        for (int kst = 0; kst < 4; ++kst)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int fo = (32 * d + r) * RBV + (16 * kst + 8 * h) * 2;
                const uint4 ah = *reinterpret_cast<const uint4*>(B0 + T1_OFF + fo);
                if constexpr (B16) {
                    acc1[d] = mfma16(ah, sh[kst], acc1[d], false);
                    if (KV) {
                        const uint4 bh_ = *reinterpret_cast<const uint4*>(B0 + T2_OFF + fo);
                        acc2[d] = mfma16(bh_, ph[kst], acc2[d], false);
                    }
                } else {
                const uint4 al = *reinterpret_cast<const uint4*>(B0 + T1_OFF + V_BYTES + fo);
                acc1[d] = mfma16(al, sh[kst], acc1[d], true);
                acc1[d] = mfma16(ah, slo[kst], acc1[d], true);
                acc1[d] = mfma16(ah, sh[kst], acc1[d], true);
                if (KV) {
                    const uint4 bh_ = *reinterpret_cast<const uint4*>(B0 + T2_OFF + fo);
                    const uint4 bl_ = *reinterpret_cast<const uint4*>(B0 + T2_OFF + V_BYTES + fo);
                    acc2[d] = mfma16(bl_, ph[kst], acc2[d], true);
                    acc2[d] = mfma16(bh_, plo[kst], acc2[d], true);
                    acc2[d] = mfma16(bh_, ph[kst], acc2[d], true);
                }
                }
            }
        wait_vmcnt<0>();
        __syncthreads();
    }
    if constexpr (!B16) mf_raise_if_over(&g_split_ovf_attn, amax);

    // ---- epilogue: transpose through LDS, row-contiguous fp32 stores ----
    char* Os = smem + wave * 32 * RBO;
#pragma unroll
    for (int which = 0; which < (KV ? 2 : 1); ++which) {
        float* outp = which == 0 ? p.out1 : p.out2;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16_t& a = (which == 0 || !KV) ? acc1[d] : acc2[d];
                const float k2 = p.inv_pscale;
                *reinterpret_cast<float4*>(Os + r * RBO + (32 * d + 8 * g + 4 * h) * 4) =
                    make_float4(a[4 * g + 0] * k2, a[4 * g + 1] * k2, a[4 * g + 2] * k2, a[4 * g + 3] * k2);
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        constexpr int OV = 32 * (HD / 4);
        for (int v = lane; v < OV; v += 64) {
            const int row = v / (HD / 4), cv = v - row * (HD / 4);
            const int cc = c0 + row;
            if (cc < p.sc) {
                const uint4 val = *reinterpret_cast<const uint4*>(Os + row * RBO + cv * 16);
                const int64_t eo = ((int64_t)b * p.sc + cc) * p.ldo + head * HD + cv * 4;
                if (B16 && p.out16)
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(outp) + eo * 2) =
                        uint2{pack_bf16x2(__uint_as_float(val.x), __uint_as_float(val.y)), pack_bf16x2(__uint_as_float(val.z), __uint_as_float(val.w))};
                else
                    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(outp) + eo * 4) = val;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// D[b][head][q] = sum_c dO[b][q][head*d + c] * O[b][q][head*d + c]
template <bool B16>      // B16: the second operand (the forward's output O) is bf16
__global__ __launch_bounds__(256) void rowdot_heads_kernel(const float* a, const void* bv, float* out, int batch, int sq, int heads, int hd,
                                                           int64_t ld) {
    const float* bb = reinterpret_cast<const float*>(bv);
    const int64_t total = (int64_t)batch * sq * heads;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int hh = (int)(i % heads);
        const int64_t t = i / heads;
        const int q = (int)(t % sq);
        const int b = (int)(t / sq);
        const float* pa = a + ((int64_t)b * sq + q) * ld + hh * hd;
        float s0 = 0.0f;
        if constexpr (B16) {
            const unsigned short* pb = reinterpret_cast<const unsigned short*>(bv) + ((int64_t)b * sq + q) * ld + hh * hd;
            for (int c = 0; c < hd; c += 4) {
                const float4 x = *reinterpret_cast<const float4*>(pa + c);
                const uint2 y = *reinterpret_cast<const uint2*>(pb + c);
                s0 += (x.x * __uint_as_float(y.x << 16) + x.y * __uint_as_float(y.x & 0xffff0000u)) +
                      (x.z * __uint_as_float(y.y << 16) + x.w * __uint_as_float(y.y & 0xffff0000u));
            }
        } else {
        const float* pb = bb + ((int64_t)b * sq + q) * ld + hh * hd;
        for (int c = 0; c < hd; c += 4) {
            const float4 x = *reinterpret_cast<const float4*>(pa + c), y = *reinterpret_cast<const float4*>(pb + c);
            s0 += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
        }
        out[((int64_t)b * heads + hh) * sq + q] = s0;
    }
}

template <int HD, bool SP = false, bool F16 = false>
void launch_attn(const AttnArgs& a, int batch, hipStream_t s) {
    dim3 grid((unsigned)(((a.sq + 127) / 128) * a.heads * batch));
    hipLaunchKernelGGL((attn_fwd_kernel<HD, true, SP, F16>), grid, dim3(256), 0, s, a);
}

// fp32 [rows][ld] (first `cols` columns) -> two fp16 planes of the same layout: hi toward zero, lo = x - hi
__global__ __launch_bounds__(256) void split_halves_kernel(const float* x, unsigned short* hi, unsigned short* lo, int64_t n4) {
    float amax = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        amax = mf_amax3(mf_amax3(amax, v.x, v.y), v.z, v.w);
        uint2 h, l;
        mf_split_f16x2(v.x, v.y, h.x, l.x);
        mf_split_f16x2(v.z, v.w, h.y, l.y);
        reinterpret_cast<uint2*>(hi)[i] = h;
        reinterpret_cast<uint2*>(lo)[i] = l;
    }
    mf_raise_if_over(&g_split_ovf_attn, amax);
}

}  // namespace

unsigned* mf_ovf_flag_attention() {
    unsigned* p = nullptr;
    (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_split_ovf_attn));
    return p;
}

extern "C" int mf_attention_bf16_lse(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                                     void* out, int64_t ldo, float* lse, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                                     int32_t head_dim, float scale, void* stream);
extern "C" int mf_attention_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                                 void* out, int64_t ldo, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                                 int32_t head_dim, float scale, void* stream) {
    return mf_attention_bf16_lse(q, ldq, k, ldk, vt, ldvt, out, ldo, nullptr, batch, heads, sq, skv, head_dim, scale, stream);
}

static int attention_16(bool f16, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                        void* out, int64_t ldo, float* lse, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                        int32_t head_dim, float scale, void* stream);

extern "C" int mf_attention_bf16_lse(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                                     void* out, int64_t ldo, float* lse, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                                     int32_t head_dim, float scale, void* stream) {
    return attention_16(false, q, ldq, k, ldk, vt, ldvt, out, ldo, lse, batch, heads, sq, skv, head_dim, scale, stream);
}

extern "C" int mf_attention_f16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                                void* out, int64_t ldo, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                                int32_t head_dim, float scale, void* stream) {
    return attention_16(true, q, ldq, k, ldk, vt, ldvt, out, ldo, nullptr, batch, heads, sq, skv, head_dim, scale, stream);
}

static int attention_16(bool f16, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                        void* out, int64_t ldo, float* lse, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                        int32_t head_dim, float scale, void* stream) {
    MF_CHECK_ARG(q && k && vt && out, "mf_attention_bf16: null pointer");
    MF_CHECK_ARG(batch >= 1 && heads >= 1 && sq >= 1 && skv >= 1, "mf_attention_bf16: bad sizes");
    MF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0 && ldvt >= skv,
                 "mf_attention_bf16: leading dims must be multiples of 8 and ldvt >= skv");
    if (!mf_aligned16(q) || !mf_aligned16(k) || !mf_aligned16(vt) || !mf_aligned16(out)) {
        mf_set_error("mf_attention_bf16: pointers must be 16-byte aligned");
        return MF_EALIGN;
    }
    AttnArgs a{};
    a.q = (const char*)q; a.ldq = ldq; a.k = (const char*)k; a.ldk = ldk; a.vt = (const char*)vt; a.ldvt = ldvt;
    a.out = (char*)out; a.ldo = ldo; a.heads = heads; a.sq = sq; a.skv = skv; a.batch = batch;
    a.lse = lse;
    a.c = scale * 1.44269504088896340736f;
    { static const bool off = getenv("MFHIP_ATTN_NOXCD") != nullptr; a.no_xcd_order = off; }
    hipStream_t s = (hipStream_t)stream;
    switch (f16 ? -head_dim : head_dim) {
        case 8: launch_attn<8>(a, batch, s); break;
        case 40: launch_attn<40>(a, batch, s); break;
        case 64: launch_attn<64>(a, batch, s); break;
        case 80: launch_attn<80>(a, batch, s); break;
        case 160: launch_attn<160>(a, batch, s); break;
        case -8: launch_attn<8, false, true>(a, batch, s); break;
        case -40: launch_attn<40, false, true>(a, batch, s); break;
        case -64: launch_attn<64, false, true>(a, batch, s); break;
        case -80: launch_attn<80, false, true>(a, batch, s); break;
        case -160: launch_attn<160, false, true>(a, batch, s); break;
        default:
            mf_set_error("mf_attention_bf16: unsupported head_dim %d (have 8, 40, 64, 80, 160)", head_dim);
            return MF_EINVAL;
    }
    MF_CHECK_LAUNCH("mf_attention_bf16");
    return MF_OK;
}

extern "C" int mf_split_halves(const float* x, void* hi, void* lo, int64_t n, void* stream) {
    MF_CHECK_ARG(x && hi && lo && n >= 0 && n % 4 == 0, "mf_split_halves: null pointer or n not a multiple of 4");
    if (!mf_aligned16(x) || (((uintptr_t)hi) & 7) || (((uintptr_t)lo) & 7)) {
        mf_set_error("mf_split_halves: x must be 16-byte and hi / lo 8-byte aligned");
        return MF_EALIGN;
    }
    if (n == 0) return MF_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_halves_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)hi, (unsigned short*)lo, n / 4);
    MF_CHECK_LAUNCH("mf_split_halves");
    return MF_OK;
}

extern "C" int mf_attention_f16x3_lse(const void* q_hi, const void* q_lo, int64_t ldq, const void* k_hi, const void* k_lo, int64_t ldk,
                                      const void* vt_hi, const void* vt_lo, int64_t ldvt, float* out, int64_t ldo, float* lse, int32_t batch,
                                      int32_t heads, int32_t sq, int32_t skv, int32_t head_dim, float scale, void* stream);

extern "C" int mf_attention_f16x3(const void* q_hi, const void* q_lo, int64_t ldq, const void* k_hi, const void* k_lo, int64_t ldk,
                                  const void* vt_hi, const void* vt_lo, int64_t ldvt, float* out, int64_t ldo, int32_t batch,
                                  int32_t heads, int32_t sq, int32_t skv, int32_t head_dim, float scale, void* stream) {
    return mf_attention_f16x3_lse(q_hi, q_lo, ldq, k_hi, k_lo, ldk, vt_hi, vt_lo, ldvt, out, ldo, nullptr, batch, heads, sq, skv, head_dim, scale,
                                  stream);
}

extern "C" int mf_attention_f16x3_lse(const void* q_hi, const void* q_lo, int64_t ldq, const void* k_hi, const void* k_lo, int64_t ldk,
                                      const void* vt_hi, const void* vt_lo, int64_t ldvt, float* out, int64_t ldo, float* lse, int32_t batch,
                                      int32_t heads, int32_t sq, int32_t skv, int32_t head_dim, float scale, void* stream) {
    MF_CHECK_ARG(q_hi && q_lo && k_hi && k_lo && vt_hi && vt_lo && out, "mf_attention_f16x3: null pointer");
    MF_CHECK_ARG(batch >= 1 && heads >= 1 && sq >= 1 && skv >= 1, "mf_attention_f16x3: bad sizes");
    MF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && ldvt >= skv,
                 "mf_attention_f16x3: leading dims must be multiples of 8 (ldo: 4) and ldvt >= skv");
    if (!mf_aligned16(q_hi) || !mf_aligned16(q_lo) || !mf_aligned16(k_hi) || !mf_aligned16(k_lo) || !mf_aligned16(vt_hi) ||
        !mf_aligned16(vt_lo) || !mf_aligned16(out)) {
        mf_set_error("mf_attention_f16x3: pointers must be 16-byte aligned");
        return MF_EALIGN;
    }
    AttnArgs a{};
    a.q = (const char*)q_hi; a.q2 = (const char*)q_lo; a.ldq = ldq;
    a.k = (const char*)k_hi; a.k2 = (const char*)k_lo; a.ldk = ldk;
    a.vt = (const char*)vt_hi; a.vt2 = (const char*)vt_lo; a.ldvt = ldvt;
    a.out = (char*)out; a.ldo = ldo; a.heads = heads; a.sq = sq; a.skv = skv; a.batch = batch;
    a.lse = lse;
    a.c = scale * 1.44269504088896340736f;
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 8: launch_attn<8, true>(a, batch, s); break;
        case 40: launch_attn<40, true>(a, batch, s); break;
        case 64: launch_attn<64, true>(a, batch, s); break;
        case 80: launch_attn<80, true>(a, batch, s); break;
        default:
            mf_set_error("mf_attention_f16x3: unsupported head_dim %d (have 8, 40, 64, 80)", head_dim);
            return MF_EINVAL;
    }
    MF_CHECK_LAUNCH("mf_attention_f16x3");
    return MF_OK;
}

template <int HD, bool B16 = false>
static void launch_attn_bwd(const mf_attn_bwd_desc* d, hipStream_t s) {
    const float c = d->scale * 1.44269504088896340736f;
    int sh = 0;
    while (!B16 && sh < 8 && (2 << sh) <= d->skv) ++sh;        // 2^sh <= skv, at most 2^8: P * 2^sh <= 256, far inside fp16 (bf16: none)
    const float pshift = (float)sh, inv_pscale = 1.0f / (float)(1 << sh);
    {   // dK, dV: a wave owns 32 keys and streams the query tiles
        AttnBwdArgs a{};
        a.c1[0] = (const char*)d->k_hi; a.c1[1] = (const char*)d->k_lo; a.ldc1 = d->ldk;
        a.c2[0] = (const char*)d->v_hi; a.c2[1] = (const char*)d->v_lo; a.ldc2 = d->ldv;
        a.r1[0] = (const char*)d->q_hi; a.r1[1] = (const char*)d->q_lo; a.ldr1 = d->ldq;
        a.r2[0] = (const char*)d->do_hi; a.r2[1] = (const char*)d->do_lo; a.ldr2 = d->lddo;
        a.t1[0] = (const char*)d->qt_hi; a.t1[1] = (const char*)d->qt_lo; a.ldt1 = d->ldqt;
        a.t2[0] = (const char*)d->dot_hi; a.t2[1] = (const char*)d->dot_lo; a.ldt2 = d->lddot;
        a.lse = d->lse; a.dd = d->dd; a.out1 = (float*)d->dk; a.out2 = (float*)d->dv; a.ldo = d->ldo;
        a.heads = d->heads; a.sc = d->skv; a.sr = d->sq; a.batch = d->batch; a.sq = d->sq; a.c = c; a.scale = d->scale;
        a.pshift = pshift; a.inv_pscale = inv_pscale; a.out16 = B16 && d->out_dtype == MF_BF16;
        dim3 grid((unsigned)(((d->skv + BWD_NW * 32 - 1) / (BWD_NW * 32)) * d->heads * d->batch));
        hipLaunchKernelGGL((attn_bwd_kernel<HD, true, B16>), grid, dim3(BWD_NW * 64), 0, s, a);
    }
    {   // dQ: a wave owns 32 queries and streams the key tiles
        AttnBwdArgs a{};
        a.c1[0] = (const char*)d->q_hi; a.c1[1] = (const char*)d->q_lo; a.ldc1 = d->ldq;
        a.c2[0] = (const char*)d->do_hi; a.c2[1] = (const char*)d->do_lo; a.ldc2 = d->lddo;
        a.r1[0] = (const char*)d->k_hi; a.r1[1] = (const char*)d->k_lo; a.ldr1 = d->ldk;
        a.r2[0] = (const char*)d->v_hi; a.r2[1] = (const char*)d->v_lo; a.ldr2 = d->ldv;
        a.t1[0] = (const char*)d->kt_hi; a.t1[1] = (const char*)d->kt_lo; a.ldt1 = d->ldkt;
        a.t2[0] = a.t1[0]; a.t2[1] = a.t1[1]; a.ldt2 = d->ldkt;
        a.lse = d->lse; a.dd = d->dd; a.out1 = (float*)d->dq; a.out2 = nullptr; a.ldo = d->ldo;
        a.heads = d->heads; a.sc = d->sq; a.sr = d->skv; a.batch = d->batch; a.sq = d->sq; a.c = c; a.scale = d->scale;
        a.pshift = pshift; a.inv_pscale = inv_pscale; a.out16 = B16 && d->out_dtype == MF_BF16;
        dim3 grid((unsigned)(((d->sq + BWD_NW * 32 - 1) / (BWD_NW * 32)) * d->heads * d->batch));
        hipLaunchKernelGGL((attn_bwd_kernel<HD, false, B16>), grid, dim3(BWD_NW * 64), 0, s, a);
    }
}

extern "C" int mf_sizeof_attn_bwd_desc(void) { return (int)sizeof(mf_attn_bwd_desc); }

extern "C" int mf_attention_bwd_f16x3(const mf_attn_bwd_desc* d, void* stream) {
    MF_CHECK_ARG(d && d->q_hi && d->q_lo && d->k_hi && d->k_lo && d->v_hi && d->v_lo && d->do_hi && d->do_lo && d->qt_hi && d->qt_lo && d->kt_hi &&
                     d->kt_lo && d->dot_hi && d->dot_lo && d->lse && d->dd && d->dq && d->dk && d->dv,
                 "mf_attention_bwd_f16x3: null pointer");
    MF_CHECK_ARG(d->batch >= 1 && d->heads >= 1 && d->sq >= 1 && d->skv >= 1, "mf_attention_bwd_f16x3: bad sizes");
    MF_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->lddo % 8 == 0 && d->ldqt % 8 == 0 && d->ldkt % 8 == 0 &&
                     d->lddot % 8 == 0 && d->ldo % 4 == 0 && d->ldqt >= d->sq && d->lddot >= d->sq && d->ldkt >= d->skv && d->sq % 4 == 0,
                 "mf_attention_bwd_f16x3: leading dims must be multiples of 8 (ldo: 4), transposed rows at least as long as the sequence, sq %% 4 == 0");
    const void* ptrs[] = {d->q_hi, d->q_lo, d->k_hi, d->k_lo, d->v_hi, d->v_lo, d->do_hi, d->do_lo, d->qt_hi, d->qt_lo, d->kt_hi, d->kt_lo,
                          d->dot_hi, d->dot_lo, d->dq, d->dk, d->dv, d->lse, d->dd};
    for (const void* q : ptrs)
        if (!mf_aligned16(q)) {
            mf_set_error("mf_attention_bwd_f16x3: pointers must be 16-byte aligned");
            return MF_EALIGN;
        }
    hipStream_t s = (hipStream_t)stream;
    switch (d->head_dim) {
        case 8: launch_attn_bwd<8>(d, s); break;
        case 40: launch_attn_bwd<40>(d, s); break;
        default:       // 64 / 80 would need 204 KB of LDS for the double-buffered dK/dV pass; their layers (<= 1024 tokens) keep the unfused backward
            mf_set_error("mf_attention_bwd_f16x3: unsupported head_dim %d (have 8, 40)", d->head_dim);
            return MF_EINVAL;
    }
    MF_CHECK_LAUNCH("mf_attention_bwd_f16x3");
    return MF_OK;
}

extern "C" int mf_attention_bwd_bf16(const mf_attn_bwd_desc* d, void* stream) {
    MF_CHECK_ARG(d && d->q_hi && d->k_hi && d->v_hi && d->do_hi && d->qt_hi && d->kt_hi && d->dot_hi && d->lse && d->dd && d->dq && d->dk && d->dv,
                 "mf_attention_bwd_bf16: null pointer");
    MF_CHECK_ARG(d->batch >= 1 && d->heads >= 1 && d->sq >= 1 && d->skv >= 1, "mf_attention_bwd_bf16: bad sizes");
    MF_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->lddo % 8 == 0 && d->ldqt % 8 == 0 && d->ldkt % 8 == 0 &&
                     d->lddot % 8 == 0 && d->ldo % 4 == 0 && d->ldqt >= d->sq && d->lddot >= d->sq && d->ldkt >= d->skv && d->sq % 4 == 0,
                 "mf_attention_bwd_bf16: leading dims must be multiples of 8 (ldo: 4), transposed rows at least as long as the sequence, sq %% 4 == 0");
    const void* ptrs[] = {d->q_hi, d->k_hi, d->v_hi, d->do_hi, d->qt_hi, d->kt_hi, d->dot_hi, d->dq, d->dk, d->dv, d->lse, d->dd};
    for (const void* q : ptrs)
        if (!mf_aligned16(q)) {
            mf_set_error("mf_attention_bwd_bf16: pointers must be 16-byte aligned");
            return MF_EALIGN;
        }
    hipStream_t s = (hipStream_t)stream;
    switch (d->head_dim) {
        case 8: launch_attn_bwd<8, true>(d, s); break;
        case 40: launch_attn_bwd<40, true>(d, s); break;
        case 80: launch_attn_bwd<80, true>(d, s); break;      // single planes only: the split form's tiles do not fit in LDS at 80
        default:
            mf_set_error("mf_attention_bwd_bf16: unsupported head_dim %d (have 8, 40, 80)", d->head_dim);
            return MF_EINVAL;
    }
    MF_CHECK_LAUNCH("mf_attention_bwd_bf16");
    return MF_OK;
}

extern "C" int mf_rowdot_heads(const float* a, const float* b, float* out, int32_t batch, int32_t sq, int32_t heads, int32_t head_dim, int64_t ld,
                               void* stream) {
    MF_CHECK_ARG(a && b && out && batch >= 1 && sq >= 1 && heads >= 1 && head_dim >= 4 && head_dim % 4 == 0 && ld % 4 == 0 && ld >= heads * head_dim,
                 "mf_rowdot_heads: bad arguments (head_dim and ld multiples of 4)");
    if (!mf_aligned16(a) || !mf_aligned16(b)) {
        mf_set_error("mf_rowdot_heads: pointers must be 16-byte aligned");
        return MF_EALIGN;
    }
    const int64_t total = (int64_t)batch * sq * heads;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rowdot_heads_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (const void*)b, out, batch, sq, heads, head_dim, ld);
    MF_CHECK_LAUNCH("mf_rowdot_heads");
    return MF_OK;
}

// D = rowdot(dO, O16) AND dO16 = bf16(dO) from one read of dO (the backward needs both).  A thread owns 8 channels of one row: two
// 16-byte loads of dO, one of O16, one 16-byte store of dO16; the head_dim / 8 partial products of a head are added in lane order.
__global__ __launch_bounds__(256) void rowdot_cast_kernel(const float* __restrict__ a, const unsigned short* __restrict__ b16,
                                                          unsigned short* __restrict__ a16, float* __restrict__ out, int batch, int sq, int heads,
                                                          int tph, int c8, int rpp) {
    __shared__ float part[256];
    const int rl = threadIdx.x / c8, cl = threadIdx.x - rl * c8;
    const int64_t rows = (int64_t)batch * sq;
    for (int64_t r0 = (int64_t)blockIdx.x * rpp; r0 < rows; r0 += (int64_t)gridDim.x * rpp) {
        const int64_t row = r0 + rl;
        float pv = 0.0f;
        if (rl < rpp && row < rows) {
            const int64_t e = (row * c8 + cl) * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(a + e), x1 = *reinterpret_cast<const float4*>(a + e + 4);
            const uint4 y = *reinterpret_cast<const uint4*>(b16 + e);
            *reinterpret_cast<uint4*>(a16 + e) = uint4{pack_bf16x2(x0.x, x0.y), pack_bf16x2(x0.z, x0.w), pack_bf16x2(x1.x, x1.y), pack_bf16x2(x1.z, x1.w)};
            pv = ((x0.x * __uint_as_float(y.x << 16) + x0.y * __uint_as_float(y.x & 0xffff0000u)) +
                  (x0.z * __uint_as_float(y.y << 16) + x0.w * __uint_as_float(y.y & 0xffff0000u))) +
                 ((x1.x * __uint_as_float(y.z << 16) + x1.y * __uint_as_float(y.z & 0xffff0000u)) +
                  (x1.z * __uint_as_float(y.w << 16) + x1.w * __uint_as_float(y.w & 0xffff0000u)));
        }
        part[threadIdx.x] = pv;
        __syncthreads();
        if (rl < rpp && row < rows && cl < heads) {
            float sv = 0.0f;
            for (int j = 0; j < tph; ++j) sv += part[rl * c8 + cl * tph + j];
            const int64_t b = row / sq, q = row - b * sq;
            out[(b * heads + cl) * sq + q] = sv;
        }
        __syncthreads();
    }
}

extern "C" int mf_rowdot_heads_cast(const float* a, const void* b16, void* a16, float* out, int32_t batch, int32_t sq, int32_t heads,
                                    int32_t head_dim, void* stream) {
    MF_CHECK_ARG(a && b16 && a16 && out && batch >= 1 && sq >= 1 && heads >= 1 && head_dim >= 8 && head_dim % 8 == 0 && heads * head_dim <= 2048,
                 "mf_rowdot_heads_cast: bad arguments (head_dim a multiple of 8, heads * head_dim <= 2048, contiguous rows)");
    if (!mf_aligned16(a) || !mf_aligned16(b16) || !mf_aligned16(a16)) {
        mf_set_error("mf_rowdot_heads_cast: pointers must be 16-byte aligned");
        return MF_EALIGN;
    }
    const int c8 = heads * head_dim / 8;
    const int rpp = 256 / c8;
    const int64_t rows = (int64_t)batch * sq;
    int64_t blocks = (rows + rpp - 1) / rpp;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rowdot_cast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (const unsigned short*)b16,
                       (unsigned short*)a16, out, batch, sq, heads, head_dim / 8, c8, rpp);
    MF_CHECK_LAUNCH("mf_rowdot_heads_cast");
    return MF_OK;
}

extern "C" int mf_rowdot_heads_bf16(const float* a, const void* b, float* out, int32_t batch, int32_t sq, int32_t heads, int32_t head_dim, int64_t ld,
                                    void* stream) {
    MF_CHECK_ARG(a && b && out && batch >= 1 && sq >= 1 && heads >= 1 && head_dim >= 4 && head_dim % 4 == 0 && ld % 4 == 0 && ld >= heads * head_dim,
                 "mf_rowdot_heads_bf16: bad arguments (head_dim and ld multiples of 4)");
    if (!mf_aligned16(a) || (((uintptr_t)b) & 7) != 0) {
        mf_set_error("mf_rowdot_heads_bf16: a must be 16-byte, b 8-byte aligned");
        return MF_EALIGN;
    }
    const int64_t total = (int64_t)batch * sq * heads;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rowdot_heads_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, batch, sq, heads, head_dim, ld);
    MF_CHECK_LAUNCH("mf_rowdot_heads_bf16");
    return MF_OK;
}
