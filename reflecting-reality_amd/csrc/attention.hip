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
#include "mf_common.h"

namespace {

struct AttnArgs {
    const char* q; int64_t ldq;
    const char* k; int64_t ldk;
    const char* vt; int64_t ldvt;
    const char* q2; const char* k2; const char* vt2;   // SP: the low-half planes (same layout as q / k / vt)
    char* out; int64_t ldo;
    int heads, sq, skv, batch;
    float c;   // softmax scale * log2(e)
    int no_xcd_order;   // A/B switch (MFHIP_ATTN_NOXCD=1): keep the hardware's round-robin block order
};

__device__ __forceinline__ uint4 ldg16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

__device__ __forceinline__ f32x16_t mfma16(const uint4& a, const uint4& b, const f32x16_t& c, bool f16) {
    return f16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0)
               : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int HD, bool DB, bool SP = false>
__global__ __launch_bounds__(256, (HD <= 80 && !SP) ? 3 : (SP && HD > 64 ? 1 : 2)) void attn_fwd_kernel(const AttnArgs p) {
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
            *reinterpret_cast<uint32_t*>(smem + (i >> 5) * BUF_BYTES + V0 + ONES_ROW * RBV + (i & 31) * 4) = SP ? 0x3C003C00u : 0x3F803F80u;   // 1.0 (hi plane only)
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
                for (int e = 0; e < 4; ++e)
                    w[e] = pack_bf16x2(__uint_as_float(w[e] << 16) * p.c, __uint_as_float(w[e] & 0xffff0000u) * p.c);
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
    const unsigned ones_x = (MJ && h) ? 0x3F803F80u : 0u, ones_y = (MJ && h) ? 0x00003F80u : 0u;
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
                    st[tt] = mfma16(am, qf[0][ks], ks == 0 ? zero16 : st[tt], false);
                } else {
                    st[tt] = mfma16(a, qf[0][ks], ks == 0 ? zero16 : st[tt], false);
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
                // the offset the matrix pipe can subtract exactly: three bf16 pieces
                const unsigned b0 = pack_bf16x2(want, 0.0f) & 0xffffu;
                const float r1 = want - __uint_as_float(b0 << 16);
                const unsigned b1 = pack_bf16x2(r1, 0.0f) & 0xffffu;
                const float r2 = r1 - __uint_as_float(b1 << 16);
                const unsigned b2 = pack_bf16x2(r2, 0.0f) & 0xffffu;
                const float m_rep = __uint_as_float(b0 << 16) + __uint_as_float(b1 << 16) + __uint_as_float(b2 << 16);
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
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[tt][e], p.c, -m_new));
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
                    // P = hi + lo in fp16 halves (hi toward zero, lo = P - hi exact in fp32); P <= 1, and fp16
                    // subnormals survive both v_cvt_pkrtz_f16_f32 and the matrix pipe (tools/micro/f16_denorm.hip)
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = st[tt][8 * s + 2 * e], a1 = st[tt][8 * s + 2 * e + 1];
                        const auto hh = __builtin_amdgcn_cvt_pkrtz(a0, a1);
                        const auto ll = __builtin_amdgcn_cvt_pkrtz(a0 - (float)hh[0], a1 - (float)hh[1]);
                        hw[e] = __builtin_bit_cast(unsigned, hh);
                        lw[e] = __builtin_bit_cast(unsigned, ll);
                    }
                    pf[0][2 * tt + s] = uint4{hw[0], hw[1], hw[2], hw[3]};
                    pf[1][2 * tt + s] = uint4{lw[0], lw[1], lw[2], lw[3]};
                } else {
                    uint4 u;
                    u.x = pack_bf16x2(st[tt][8 * s + 0], st[tt][8 * s + 1]);
                    u.y = pack_bf16x2(st[tt][8 * s + 2], st[tt][8 * s + 3]);
                    u.z = pack_bf16x2(st[tt][8 * s + 4], st[tt][8 * s + 5]);
                    u.w = pack_bf16x2(st[tt][8 * s + 6], st[tt][8 * s + 7]);
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
                    o[d] = mfma16(a, pf[0][kst], o[d], false);
                }
            }
        wait_vmcnt<0>();      // this wave's share of tile t+1 has landed ...
        __syncthreads();      // ... and so has everybody else's; every wave is done reading tile t
    }

    // ---- epilogue: normalise, transpose through LDS, row-contiguous stores ----
    if (ONES) l = __shfl(o[L_D][L_E], (lane & 31) + 32 * L_H, 64);      // the half-wave that holds accumulator row ONES_ROW
    else l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
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
                u.x = pack_bf16x2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
                u.y = pack_bf16x2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
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


template <int HD, bool SP = false>
void launch_attn(const AttnArgs& a, int batch, hipStream_t s) {
    dim3 grid((unsigned)(((a.sq + 127) / 128) * a.heads * batch));
    hipLaunchKernelGGL((attn_fwd_kernel<HD, true, SP>), grid, dim3(256), 0, s, a);
}

// fp32 [rows][ld] (first `cols` columns) -> two fp16 planes of the same layout: hi toward zero, lo = x - hi
__device__ unsigned g_split_ovf_attn;     // raised when a value handed to mf_split_halves exceeded the fp16 range (mf_common.h)

__global__ __launch_bounds__(256) void split_halves_kernel(const float* x, unsigned short* hi, unsigned short* lo, int64_t n4) {
    float amax = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        amax = mf_amax3(mf_amax3(amax, v.x, v.y), v.z, v.w);
        const auto h0 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h1 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
        const auto l0 = __builtin_amdgcn_cvt_pkrtz(v.x - (float)h0[0], v.y - (float)h0[1]);
        const auto l1 = __builtin_amdgcn_cvt_pkrtz(v.z - (float)h1[0], v.w - (float)h1[1]);
        reinterpret_cast<uint2*>(hi)[i] = uint2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
        reinterpret_cast<uint2*>(lo)[i] = uint2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
    }
    mf_raise_if_over(&g_split_ovf_attn, amax);
}

}  // namespace

unsigned* mf_ovf_flag_attention() {
    unsigned* p = nullptr;
    (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_split_ovf_attn));
    return p;
}

extern "C" int mf_attention_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                                 void* out, int64_t ldo, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
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
    a.c = scale * 1.44269504088896340736f;
    { static const bool off = getenv("MFHIP_ATTN_NOXCD") != nullptr; a.no_xcd_order = off; }
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 8: launch_attn<8>(a, batch, s); break;
        case 40: launch_attn<40>(a, batch, s); break;
        case 64: launch_attn<64>(a, batch, s); break;
        case 80: launch_attn<80>(a, batch, s); break;
        case 160: launch_attn<160>(a, batch, s); break;
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

extern "C" int mf_attention_f16x3(const void* q_hi, const void* q_lo, int64_t ldq, const void* k_hi, const void* k_lo, int64_t ldk,
                                  const void* vt_hi, const void* vt_lo, int64_t ldvt, float* out, int64_t ldo, int32_t batch,
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
