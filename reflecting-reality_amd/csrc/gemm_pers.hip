// Persistent 1x1 GEMM with the epilogue UNDER the next tile's main loop (tile 70, round 6): the 128-row successor of tile 69.
//
// Why.  The 1x1 GEMMs of the transformer blocks (attention.py:1188-1201 FeedForward, attention_processor.py:190-205 to_q / to_out,
// transformer_2d.py:162,225 proj_in / proj_out) have K = 320 ... 1280: a block of gemm_conv_kernel spends as long in its epilogue as
// in its main loop, and because a launch is one wave of blocks that all start together, every CU reaches its epilogue at the same
// time — the memory system idles through the main loops and then takes a burst of all the stores (and residual loads) at once
// (profiles/r05_gemm_phase_stamps.txt: 15 us of epilogue behind 47 us of main loop on the 64 x 64 convs, 12 behind 7 on the 1x1s).
// Here a block keeps a RANGE of 160-column output tiles of its 128 rows, and dedicated epilogue waves turn the slab of tile j into
// output rows while the compute waves multiply tile j + 1: stores trickle out under the main loop, nothing bursts.
//
// Roles (1024 threads = 16 waves, four per SIMD, <= 128 VGPRs each):
//   waves 0-7   compute: 4 x 2 wave tiles of 32 x 80 = 2 x 5 tiles of 16 x 16 (v_mfma_f32_16x16x32), two compute waves per SIMD so
//               that one's fragment reads fly under the other's MFMAs (tile 69 had one and lost for it);
//   waves 8-11  staging: LDS-DMA of the A (128 x 128 B) and W (160 x 128 B) tiles into a 2-deep ring (the fp32 slab of a whole
//               128 x 160 tile takes 82 KB of the 160, so the ring cannot be three deep), running through all the block's output
//               tiles without a drain;
//   waves 12-15 epilogue: LayerNorm fold, bias, residual, SiLU / GEGLU, cast, 16-byte stores (gemm_conv_kernel's epilogue_store8)
//               of slab chunks BETWEEN the barriers of the next tile's K steps.  Everything the epilogue reads from global memory
//               (residual vectors, bias and column-sum rows) is fetched one tile ahead, right after the previous tile's stores were
//               issued: loads and stores share vmcnt and return in order, so a load issued between stores would wait for them.
//
// Every wave arrives at every barrier.  Barrier #g (g = 0 .. G - 1, G = tiles x K tiles) means "K tile g has landed and K tile g - 1
// is no longer read"; the slab of output tile j is written before barrier #(last(j) + 1) and read between barriers #(last(j) + 1)
// and #(last(j) + nkt) — the compute waves write the next slab only after the latter.  After barrier #G all waves share the last slab.
#include <hip/hip_runtime.h>

#include "gemm_conv_kernel.h"

namespace mfgemm {

namespace {

constexpr int PB_BM = 128, PB_BN = 160, PB_STAGES = 2;
constexpr int PB_STAGE_BYTES = (PB_BM + PB_BN) * 128;               // 36 KB: [A tile 128 rows][W tile 160 rows], 128 bytes of K per row
constexpr int PB_SLAB_RS = (PB_BN + 4) * 4;                         // slab row stride (bytes): 160 fp32 + 16 bytes (bank spread)
constexpr int PB_SLAB_OFF = PB_STAGES * PB_STAGE_BYTES;             // 73728
constexpr int PB_LNST_OFF = PB_SLAB_OFF + PB_BM * PB_SLAB_RS;       // + 83968
constexpr int PB_EROW_OFF = PB_LNST_OFF + PB_BM * 8;                // + 1024: per epilogue wave [bias row][column-sum row], 160 fp32 each
constexpr int PB_EROW_BYTES = 2 * PB_BN * 4;
constexpr int PB_SMEM = PB_EROW_OFF + 4 * PB_EROW_BYTES;            // 163840 = all 160 KB
static_assert(PB_SMEM <= 160 * 1024, "LDS");
constexpr int PB_CPR = PB_BN / 8;                                   // 20 (row, 8-column) items per row
constexpr int PB_CHUNKS = PB_BM * PB_CPR / 64;                      // 40 chunks of 64 items per output tile
constexpr int PB_CPW = PB_CHUNKS / 4;                               // 10 per epilogue wave

// The epilogue of one (row, 8 columns) item with NO global load in it — not even on a path that is never taken: hipcc places the wait
// for a conditionally loaded value behind the join of the branches, as vmcnt(0), and since loads and stores share that counter every
// item would then wait for the previous item's store to come back (measured: 2.5 us per chunk, 740 us for a 125 us GEMM, with
// epilogue_store8's `if (p.rs) ... if (p.temb) ...` ladder in the chunk code).  Bias comes from the wave's LDS row (zeros when the
// call has none), the 16-bit residual from registers fetched a tile ahead; the host refuses everything else for this tile.
template <bool F16, bool RES, bool GEGLU, bool LN>
__device__ __forceinline__ void pers_store8(const GemmArgs& p, int m, int n, const char* sp, const char* eb, const char* ec, float2 st, const uint4& q0) {
    // sp: the item's 8 fp32 accumulators in the slab; eb / ec: its 8 bias values / 8 LayerNorm column sums in the wave's LDS rows;
    // st: (mean, rstd) of its row.  Worked in two halves of four columns so that at most a dozen values are live beside the
    // prefetched residual vectors (the RES variants sit at the 128-register budget of a sixteen-wave block).
    float v[8];
    uint32_t pk[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 a = *reinterpret_cast<const float4*>(sp + 16 * h);
        const float4 b = *reinterpret_cast<const float4*>(eb + 16 * h);
        float x[4] = {a.x, a.y, a.z, a.w};
        if constexpr (LN) {                                          // rstd * (acc - mean * colsum)
            const float4 c = *reinterpret_cast<const float4*>(ec + 16 * h);
            x[0] = st.y * (x[0] - st.x * c.x); x[1] = st.y * (x[1] - st.x * c.y); x[2] = st.y * (x[2] - st.x * c.z); x[3] = st.y * (x[3] - st.x * c.w);
        }
        x[0] = (x[0] + b.x) * p.alpha; x[1] = (x[1] + b.y) * p.alpha; x[2] = (x[2] + b.z) * p.alpha; x[3] = (x[3] + b.w) * p.alpha;
        if constexpr (RES) {
            float r0, r1, r2, r3;
            unpack_h2<F16>(h ? q0.z : q0.x, r0, r1);
            unpack_h2<F16>(h ? q0.w : q0.y, r2, r3);
            x[0] += r0; x[1] += r1; x[2] += r2; x[3] += r3;
        }
        if (!GEGLU && p.act == MF_ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = silu_precise(x[j]);
        }
        if constexpr (GEGLU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * h + j] = x[j];
        } else if (p.out_dt == MF_F32) {
            *reinterpret_cast<float4*>(p.out + ((int64_t)m * p.ldc + n + 4 * h) * 4) = make_float4(x[0], x[1], x[2], x[3]);
        } else {
            pk[2 * h] = pack_h2<F16>(x[0], x[1]);
            pk[2 * h + 1] = pack_h2<F16>(x[2], x[3]);
        }
    }
    if constexpr (GEGLU) {
        // weight rows are interleaved [4 values | 4 gates]: out[n/2 + j] = v[j] * gelu_erf(v[4 + j])  (activations.py:100-103); erf by
        // Abramowitz-Stegun 7.1.26 for a 16-bit output, erff for fp32: epilogue_store8's arithmetic, instruction for instruction
        float g[4];
        if (p.out_dt != MF_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = v[4 + j];
                const float z = fabsf(x) * 0.70710678118654752440f;
                const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
                const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
                const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
                g[j] = v[j] * (0.5f * x + 0.5f * fabsf(x) * e);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = v[j] * (0.5f * v[4 + j] * (1.0f + erff(v[4 + j] * 0.70710678118654752440f)));
        }
        const int64_t o = (int64_t)m * p.ldc + (n >> 1);
        if (p.out_dt == MF_F32) {
            *reinterpret_cast<float4*>(p.out + o * 4) = make_float4(g[0], g[1], g[2], g[3]);
        } else {
            uint2 u;
            u.x = pack_h2<F16>(g[0], g[1]);
            u.y = pack_h2<F16>(g[2], g[3]);
            *reinterpret_cast<uint2*>(p.out + o * 2) = u;
        }
    } else if (p.out_dt != MF_F32) {
        *reinterpret_cast<uint4*>(p.out + ((int64_t)m * p.ldc + n) * 2) = uint4{pk[0], pk[1], pk[2], pk[3]};
    }
}

// RES / GEGLU / LN: compile-time properties of the call (16-bit residual fetched ahead; GEGLU epilogue; folded LayerNorm).  One kernel
// with run-time flags keeps the live ranges of all three in one register allocation — 40 registers of residual vectors beside the
// GEGLU temporaries spilled, and a spill reload in the chunk code is a scratch LOAD behind the stores: the wait this file is about.
template <int DT, bool RES, bool GEGLU, bool LN>
__global__ __launch_bounds__(1024) void gemm_pers_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool F16 = DT == MF_F16;
    const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int role = wave < 8 ? 0 : (wave < 12 ? 1 : 2), rw = role == 0 ? wave : (wave & 3);
    // XCD-aware order (blocks b and b + 8 share an XCD): an XCD gets a contiguous run of the logical order, which is range-major, so
    // the blocks of one XCD walk the same few column ranges and that slice of W stays in its L2
    int bid = (int)blockIdx.x;
    {
        const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int range = bid / p.tiles_m, tile_m = bid - range * p.tiles_m;
    const int m0 = tile_m * PB_BM, nt0 = range * p.nloop;            // first row, first output tile of this block
    const int nkt = p.nkt, G = p.nloop * nkt;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
    float2* lnst = reinterpret_cast<float2*>(smem + PB_LNST_OFF);    // (mean, rstd) of the block's rows (folded LayerNorm)

    if (role == 1) {
        // ---- staging waves: A tile 16 DMAs of 8 rows (four per wave), W tile 20 (five per wave), swizzle applied to the SOURCE chunk ----
        const srd_t sA = make_srd(p.a0, (unsigned)((int64_t)p.M * p.ld0b));
        const srd_t sW = make_srd(p.w, (unsigned)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2));
        const int r8 = lane >> 3, slot = lane & 7;
        unsigned offA[4], offW[5];                                   // byte offsets of this lane's 16 bytes at K tile 0 / output tile 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (4 * rw + i) * 8 + r8;
            offA[i] = (unsigned)(m0 + r) * (unsigned)p.ld0b + (unsigned)((slot ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int r = (5 * rw + i) * 8 + r8;
            offW[i] = (unsigned)(nt0 * PB_BN + r) * (unsigned)(p.ldw * 2) + (unsigned)((slot ^ ((r >> 1) & 7)) << 4);
        }
        const unsigned wstep = (unsigned)PB_BN * (unsigned)(p.ldw * 2);           // one output tile further
        int i_kt = 0, i_st = 0;
        unsigned wbase = 0;
        const bool dbg_nodma = (p.dbg_epi & 4) != 0;                 // developer switch (MFHIP_DBG_EPI=4): no DMA — what the loop costs without its loads
        auto issue = [&]() {
            const unsigned ldsS = lds0 + i_st * PB_STAGE_BYTES;
            if (dbg_nodma) { i_st ^= 1; if (++i_kt == nkt) { i_kt = 0; wbase += wstep; } return; }
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16_buf(offA[i] + i_kt * 128, sA, ldsS + ((4 * rw + i) * 8) * 128);
#pragma unroll
            for (int i = 0; i < 5; ++i) dma16_buf(offW[i] + wbase + i_kt * 128, sW, ldsS + PB_BM * 128 + ((5 * rw + i) * 8) * 128);
            i_st ^= 1;
            if (++i_kt == nkt) { i_kt = 0; wbase += wstep; }
        };
        issue();                                                     // K tile 0
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g + 1 < G; ++g) {
            issue();                                                 // K tile g + 1 into the stage tile g - 1 has left
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
        }
        __builtin_amdgcn_s_barrier();                                // #G
    } else if (role == 0) {
        // ---- compute waves: 4 x 2 wave tiles of 32 x 80 = 2 x 5 tiles of 16 x 16 -------------------------------------------------------
        const int wm = rw >> 1, wn = rw & 1, r16 = lane & 15, kg = lane >> 4;
        const int key16 = (r16 >> 1) & 7;                            // tile bases are multiples of 16 rows: one key for every fragment
        f32x4_t acc[2][5];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        int st = 0, kt = 0;
        __builtin_amdgcn_s_barrier();                                // #0
        const bool dbg_nomma = (p.dbg_epi & 2) != 0;                 // developer switch (MFHIP_DBG_EPI=2): no fragment reads / MFMAs
        for (int g = 0; g < G; ++g) {
            if (dbg_nomma) {
                st ^= 1;
                if (++kt == nkt) kt = 0;
                __builtin_amdgcn_s_barrier();
                continue;
            }
            const char* A16 = smem + st * PB_STAGE_BYTES + (wm * 32 + r16) * 128;
            const char* B16 = smem + st * PB_STAGE_BYTES + PB_BM * 128 + (wn * 80 + r16) * 128;
            uint4 fa[2][2], fb[2][5];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[ks][a] = *reinterpret_cast<const uint4*>(A16 + a * 16 * 128 + (((4 * ks + kg) ^ key16) << 4));
#pragma unroll
                for (int b = 0; b < 5; ++b) fb[ks][b] = *reinterpret_cast<const uint4*>(B16 + b * 16 * 128 + (((4 * ks + kg) ^ key16) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 5; ++b)
                        acc[a][b] = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, fa[ks][a]), __builtin_bit_cast(bf16x8_t, fb[ks][b]), acc[a][b], 0, 0, 0);
            st ^= 1;
            if (++kt == nkt) {
                // output tile finished: accumulators -> slab (element r of acc[a][b]: row 16 a + 4 kg + r, column 16 b + r16 of the wave tile).
                // The epilogue waves finished the previous slab before they arrived at barrier #g (see the header).
                kt = 0;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 5; ++b) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            *reinterpret_cast<float*>(smem + PB_SLAB_OFF + (wm * 32 + 16 * a + 4 * kg + r) * PB_SLAB_RS + (wn * 80 + 16 * b + r16) * 4) = acc[a][b][r];
                        acc[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
                    }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): fragment reads (and slab writes) of this step are complete
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
        }
    } else {
        // ---- epilogue waves ------------------------------------------------------------------------------------------------------------
        char* erow = smem + PB_EROW_OFF + rw * PB_EROW_BYTES;        // this wave's own [bias][column sums] rows of the tile it works on
        constexpr bool pre = RES;                                    // 16-bit residual vectors are fetched a tile ahead
        uint4 q0[RES ? PB_CPW : 1];
        float4 pb = make_float4(0, 0, 0, 0), pc = make_float4(0, 0, 0, 0);   // lanes 0-39: four bias / column-sum values of the NEXT tile
        // requests for output tile `jt` (relative to nt0): residual vectors of this wave's ten chunks, bias and column-sum rows
        auto prefetch = [&](int jt) {
            const int n0 = (nt0 + jt) * PB_BN;
            if constexpr (pre) {
#pragma unroll
                for (int i = 0; i < PB_CPW; ++i) {
                    int it = (rw + 4 * i) * 64 + lane;
                    asm volatile("" : "+v"(it));                     // opaque: keeps the ten address chains out of the loop-invariant hoist (they spilled)
                    const int row = it / PB_CPR, cg = it - row * PB_CPR;
                    q0[i] = *reinterpret_cast<const uint4*>(p.res0 + ((int64_t)(m0 + row) * p.ld_res0 + n0 + cg * 8) * 2);
                }
            }
            if (lane < PB_BN / 4) {
                if (p.bias) pb = *reinterpret_cast<const float4*>(p.bias + n0 + lane * 4);
                if constexpr (LN) pc = *reinterpret_cast<const float4*>(p.ln_cs + n0 + lane * 4);
            }
        };
        // the prefetched rows of the tile about to be worked on -> this wave's LDS rows (wave-private: no barrier)
        auto publish_rows = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the prefetch (and, behind it in program order, nothing else) has landed
            // ... and the compiler is told so: it tracks a load's destination registers as pending until a wait IT placed, and a wait
            // before the first use of q0[i] inside chunk i would be a vmcnt(0) behind chunk i - 1's store.  Passing the registers
            // through an empty asm makes them values defined here.
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < PB_CPW; ++i) asm volatile("" : "+v"(q0[i].x), "+v"(q0[i].y), "+v"(q0[i].z), "+v"(q0[i].w));
            }
            if (lane < PB_BN / 4) {
                *reinterpret_cast<float4*>(erow + lane * 16) = pb;
                *reinterpret_cast<float4*>(erow + PB_BN * 4 + lane * 16) = pc;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        };
        auto do_chunk = [&](int i, int n0) {
            int it = (rw + 4 * i) * 64 + lane;
            asm volatile("" : "+v"(it));                             // opaque (see prefetch)
            const int row = it / PB_CPR, cg = it - row * PB_CPR;
            const int m = m0 + row, n = n0 + cg * 8;
            const char* sp = smem + PB_SLAB_OFF + row * PB_SLAB_RS + cg * 32;
            float2 st = make_float2(0.0f, 1.0f);
            if constexpr (LN) st = lnst[row];
            uint4 q = q0[RES ? i : 0];
            if constexpr (RES) asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));   // opaque: the bf16 -> fp32 unpacking of all ten vectors was hoisted out of the chunk loop (80 registers)
            pers_store8<F16, RES, GEGLU, LN>(p, m, n, sp, erow + cg * 32, erow + PB_BN * 4 + cg * 32, st, q);
        };
        if (p.nloop > 1) prefetch(0);
        const int cpi = (PB_CPW + nkt - 2) / (nkt - 1);              // chunks per barrier interval (nkt >= 2: host check)
        int pending_n0 = -1, done = 0, kt = 0, jt = 0;
        __builtin_amdgcn_s_barrier();                                // #0
        // folded LayerNorm: (sum, sum of squares) of the block's 128 rows, taken from the A tiles as they pass through LDS during the
        // first output tile's K steps (no global load in these waves): 32 rows per wave, two lanes per row, 64 bytes each per K tile
        float ls1 = 0.0f, ls2 = 0.0f;
        const int lrow = rw * 32 + (lane >> 1), lhalf = lane & 1;
        for (int g = 0; g < G; ++g) {
            if (LN && g < nkt) {
                const char* ar = smem + (g & 1) * PB_STAGE_BYTES + lrow * 128 + lhalf * 64;      // (the swizzle permutes a row's chunks: a sum does not care)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x[8];
                    unpack_h8<F16>(*reinterpret_cast<const uint4*>(ar + c * 16), x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ls1 += x[e]; ls2 = fmaf(x[e], x[e], ls2); }
                }
                if (g == nkt - 1) {
                    ls1 += __shfl_xor(ls1, 1); ls2 += __shfl_xor(ls2, 1);
                    const float invk = 1.0f / (float)p.K, mean = ls1 * invk;
                    float var = ls2 * invk - mean * mean;
                    if (var < 0.0f) var = 0.0f;
                    if (lhalf == 0) lnst[lrow] = make_float2(mean, 1.0f / sqrtf(var + p.ln_eps));
                }
            }
            if (pending_n0 >= 0 && (p.dbg_epi & 1)) pending_n0 = -1;     // developer switch (MFHIP_DBG_EPI=1): the epilogue waves only keep the barriers
            if (pending_n0 >= 0) {
                if (done == 0) publish_rows();
#pragma unroll 1
                for (int i = 0; i < cpi && done < PB_CPW; ++i, ++done) {
                    // (a switch keeps q0[] in registers: the chunk index must be a compile-time constant)
                    switch (done) {
                        case 0: do_chunk(0, pending_n0); break; case 1: do_chunk(1, pending_n0); break;
                        case 2: do_chunk(2, pending_n0); break; case 3: do_chunk(3, pending_n0); break;
                        case 4: do_chunk(4, pending_n0); break; case 5: do_chunk(5, pending_n0); break;
                        case 6: do_chunk(6, pending_n0); break; case 7: do_chunk(7, pending_n0); break;
                        case 8: do_chunk(8, pending_n0); break; default: do_chunk(9, pending_n0); break;
                    }
                }
                if (done == PB_CPW) {
                    pending_n0 = -1;
                    if (jt < p.nloop - 1) prefetch(jt);              // operands of the tile being multiplied now, behind this tile's stores (the last tile's slab is shared by all waves below)
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // the slab reads issued so far have returned
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
            if (++kt == nkt) {                                       // the compute waves wrote output tile jt's slab before this barrier
                kt = 0;
                if (g + 1 < G) { pending_n0 = (nt0 + jt) * PB_BN; done = 0; }
                ++jt;
            }
        }
    }
    // ---- the last slab: all sixteen waves (plain loads of the epilogue operands: nothing left to hide them under) ----------------------
    {
        const int n0 = (nt0 + p.nloop - 1) * PB_BN;
        for (int c = wave; c < PB_CHUNKS; c += 16) {
            const int it = c * 64 + lane;
            const int row = it / PB_CPR, cg = it - row * PB_CPR;
            const int m = m0 + row, n = n0 + cg * 8;
            const char* sp = smem + PB_SLAB_OFF + row * PB_SLAB_RS + cg * 32;
            float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 16);
            if (p.ln_cs) {
                const float2 st = lnst[row];
                const float4 c0 = *reinterpret_cast<const float4*>(p.ln_cs + n), c1 = *reinterpret_cast<const float4*>(p.ln_cs + n + 4);
                lo.x = st.y * (lo.x - st.x * c0.x); lo.y = st.y * (lo.y - st.x * c0.y); lo.z = st.y * (lo.z - st.x * c0.z); lo.w = st.y * (lo.w - st.x * c0.w);
                hi.x = st.y * (hi.x - st.x * c1.x); hi.y = st.y * (hi.y - st.x * c1.y); hi.z = st.y * (hi.z - st.x * c1.z); hi.w = st.y * (hi.w - st.x * c1.w);
            }
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            epilogue_store8<F16>(p, 0, m, n, v);
        }
    }
}

}  // namespace

template <int DT>
static const void* pers_fn(bool res, bool geglu, bool ln) {
    if (geglu) return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, true, true>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, true, false>);
    if (res) return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, true, false, true>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, true, false, false>);
    return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, false, true>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, false, false>);
}

bool launch_pers(int dtype, const GemmArgs& a, hipStream_t s) {
    if (dtype != MF_BF16 && dtype != MF_F16) return false;
    const bool geglu = a.act == MF_ACT_GEGLU4, res = a.res0 != nullptr, ln = a.ln_cs != nullptr;
    if (geglu && res) return false;
    const int fl = dtype == MF_F16 ? 1 : 0, variant = (geglu ? 4 : (res ? 2 : 0)) + (ln ? 1 : 0);
    const void* fn = fl ? pers_fn<MF_F16>(res, geglu, ln) : pers_fn<MF_BF16>(res, geglu, ln);
    // the dynamic-LDS attribute is per function AND per device: one flag per (flavour, variant, device), the call's result checked
    static bool attr[2][6][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!attr[fl][variant][dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM) != hipSuccess) return false;
        attr[fl][variant][dev] = true;
    }
    void* args[] = {const_cast<GemmArgs*>(&a)};
    return hipLaunchKernel(fn, dim3((unsigned)(a.tiles_m * (a.tiles_n / a.nloop))), dim3(1024), args, PB_SMEM, s) == hipSuccess;
}

}  // namespace mfgemm
