// Persistent 1x1 GEMM with the epilogue UNDER the next tile's main loop (tile 70, round 6): the 128-row successor of tile 69.
//
// Why.  The 1x1 GEMMs of the transformer blocks (attention.py:1188-1201 FeedForward, attention_processor.py:190-205 to_q / to_out,
// transformer_2d.py:162,225 proj_in / proj_out) have K = 320 ... 1280: a block of gemm_conv_kernel spends as long in its epilogue as
// in its main loop, and because a launch is one wave of blocks that all start together, every CU reaches its epilogue at the same
// time — the memory system idles through the main loops and then takes a burst of all the stores (and residual loads) at once
// (profiles/r05_gemm_phase_stamps.txt: 15 us of epilogue behind 47 us of main loop on the 64 x 64 convs, 12 behind 7 on the 1x1s).
// Here a block keeps a RANGE of 160-column output tiles of its 128 rows, and the slab of tile j is turned into output rows while
// tile j + 1 is multiplied: stores trickle out under the main loop, nothing bursts.
//
// Roles (1024 threads = 16 waves, four per SIMD, <= 128 VGPRs each):
//   waves 0-7   compute: 4 x 2 wave tiles of 32 x 80 = 2 x 5 tiles of 16 x 16 (v_mfma_f32_16x16x32), two compute waves per SIMD so
//               that one's fragment reads fly under the other's MFMAs (tile 69 had one and lost for it).  When an output tile is
//               finished they apply what needs the fp32 accumulator — the LayerNorm fold rstd * (acc - mean * colsum), bias, alpha —
//               in the accumulator layout and write the tile to the slab ROUNDED TO THE 16-BIT STORAGE TYPE.  That rounding is the
//               reference's own: in its bf16 / fp16 runs a Linear's output is a 16-bit tensor before the residual add or the GEGLU
//               that follows (attention.py:1188-1201, activations.py:100-103); gemm_conv_kernel's fp32 slabs round once, later.
//               A 16-bit slab is 42 KB instead of 82: the ring beside it is THREE deep again (two K tiles in flight, as in the
//               warp-specialised tiles) — with the fp32 slab of the first version the two-deep ring ran 1540 clocks per K tile;
//   waves 8-11  staging: LDS-DMA of the A (128 x 128 B) and W (160 x 128 B) tiles, counted vmcnt, one barrier per K tile, running
//               through all the block's output tiles without a drain;
//   waves 12-15 epilogue: (row, 8 columns) items of the slab -> residual add, SiLU / GEGLU, cast, 16-byte stores.  The GEGLU of a
//               K = 320 tile is as much vector work as its main loop is matrix work (measured: 85 us of epilogue alone on four waves
//               against 59 us of main loop for the 64 x 64 feed-forward), so the compute waves take 16 of a tile's 40 chunks,
//               one per barrier interval, behind their MFMAs.
// No wave that stores ever WAITS for a global load issued after one of its stores: loads and stores share vmcnt and return in order,
// so that wait is a wait for the store's round trip — and hipcc places the wait for a conditionally loaded value behind the join of
// the branches even on the path that loaded nothing (measured on the first version: 2.5 us per chunk).  Residual vectors are fetched a
// tile ahead and handed to the compiler as asm-defined values; bias and column-sum rows come through LDS; the chunk code has no
// global load on any path.
//
// Every wave arrives at every barrier.  Barrier #g (g = 0 .. G - 1, G = tiles x K tiles) means "K tile g has landed and K tile g - 1
// is no longer read"; interval g lies between barriers #g and #(g + 1).  Output tile j is multiplied during intervals j nkt ..
// (j + 1) nkt - 1; its accumulators are written to the slab at the START of interval (j + 1) nkt (the LayerNorm statistics of the
// block's rows, gathered from the A tiles of output tile 0 as they pass through LDS, are complete by then), read during the
// nkt - 1 intervals that follow, and the next tile overwrites the slab at the start of interval (j + 2) nkt.  After barrier #G the
// compute waves write the last slab and all sixteen waves share it.
#include <hip/hip_runtime.h>

#include "gemm_conv_kernel.h"

namespace mfgemm {

namespace {

constexpr int PB_BM = 128, PB_BN = 160, PB_STAGES = 3, PB_PF = PB_STAGES - 1;
constexpr int PB_STAGE_BYTES = (PB_BM + PB_BN) * 128;               // 36 KB: [A tile 128 rows][W tile 160 rows], 128 bytes of K per row
constexpr int PB_SLAB_RS = PB_BN * 2 + 16;                          // slab row stride (bytes): 160 16-bit values + 16 bytes
constexpr int PB_SLAB_OFF = PB_STAGES * PB_STAGE_BYTES;             // 110592
constexpr int PB_SLAB_RST = PB_BM * 2 + 16;                         // transposed slab (V^T tiles): [column][128 rows] 16-bit, 272 bytes per column
constexpr int PB_SLAB_BYTES = PB_BN * PB_SLAB_RST;                  // 43520 >= 128 * 336
static_assert(PB_SLAB_BYTES >= PB_BM * PB_SLAB_RS, "the slab region holds either layout");
constexpr int PB_LNST_OFF = PB_SLAB_OFF + PB_SLAB_BYTES;            // + 43520
constexpr int PB_EROW_OFF = PB_LNST_OFF + PB_BM * 8;                // + 1024: two buffers (output tile parity) of [bias row][column-sum row], 160 fp32 each
constexpr int PB_EROW_BYTES = 2 * PB_BN * 4;
constexpr int PB_LUT_OFF = PB_EROW_OFF + 2 * PB_EROW_BYTES;         // 157696: GEGLU variants: the normal CDF on [-6, 6) in steps of 1 / 64
constexpr int PB_LUT_N = 768;                                       // entries (Phi(x_i), Phi(x_i+1) - Phi(x_i)), 8 bytes each
constexpr int PB_SMEM = PB_LUT_OFF;                                 // 157696 without the table
constexpr int PB_SMEM_GEGLU = PB_LUT_OFF + PB_LUT_N * 8;            // 163840 with it: all of the 160 KB
static_assert(PB_SMEM_GEGLU <= 160 * 1024, "LDS");
constexpr int PB_CPR = PB_BN / 8;                                   // 20 (row, 8-column) items per row
constexpr int PB_CHUNKS = PB_BM * PB_CPR / 64;                      // 40 chunks of 64 items per output tile
constexpr int PB_CE = 6, PB_CC = 2;                                 // chunks per epilogue wave / per compute wave
static_assert(4 * PB_CE + 8 * PB_CC == PB_CHUNKS, "chunk ownership covers the slab");

// One (row, 8 columns) item: 16 bytes of the slab -> [+ residual] -> [SiLU | GEGLU] -> 16 (GEGLU: 8) bytes of output.  No global load.
template <bool F16, bool RES, bool GEGLU>
__device__ __forceinline__ void pers_item(const GemmArgs& p, int m, int n, const char* sp, const uint4& q, const char* lut) {
    const uint4 s = *reinterpret_cast<const uint4*>(sp);
    if constexpr (GEGLU) {
        // weight rows are interleaved [4 values | 4 gates]: out[n/2 + j] = v[j] * gelu_erf(v[4 + j]) = v[j] * x * Phi(x)  (activations.py:100-103).
        // Phi from a table in LDS (768 steps of 1 / 64 over [-6, 6), linear interpolation: |error| <= 7.4e-6 = h^2 / 8 max |Phi''|; the gate
        // is a 16-bit value here and the product is rounded to 16 bits): nine vector instructions and one LDS read per value where the
        // erf of Abramowitz-Stegun 7.1.26 (epilogue_store8) takes twenty with two transcendentals — the GEGLU of a K = 320 tile is as
        // much vector work as its main loop is matrix work (header), so this is the launch's critical path.
        float v[8], g[4];
        unpack_h8<F16>(s, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = v[4 + j];
            const float u = fmaf(__builtin_amdgcn_fmed3f(x, -6.0f, 5.984375f), 64.0f, 384.0f);     // [0, 767]
            const float fr = __builtin_amdgcn_fractf(u);
            const float2 e = *reinterpret_cast<const float2*>(lut + (int)(u - fr) * 8);
            g[j] = v[j] * (x * fmaf(fr, e.y, e.x));
        }
        uint2 u;
        u.x = pack_h2<F16>(g[0], g[1]);
        u.y = pack_h2<F16>(g[2], g[3]);
        *reinterpret_cast<uint2*>(p.out + ((int64_t)m * p.ldc + (n >> 1)) * 2) = u;
    } else {
        uint4 o = s;
        if (RES || p.act == MF_ACT_SILU) {
            float v[8];
            unpack_h8<F16>(s, v);
            if constexpr (RES) {
                float r[8];
                unpack_h8<F16>(q, r);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += r[j];
            }
            if (p.act == MF_ACT_SILU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = silu_precise(v[j]);
            }
            o = pack_h8<F16>(v);
        }
        *reinterpret_cast<uint4*>(p.out + ((int64_t)m * p.ldc + n) * 2) = o;
    }
}

// One item of a V^T tile (mf_gemm_desc.vt_out: the V third of a fused q | k | v projection, written [image][channel][token]): 8
// consecutive tokens of one channel = 16 contiguous bytes of the TRANSPOSED slab = 16 contiguous bytes of V^T.
__device__ __forceinline__ void pers_item_vt(const GemmArgs& p, int m, int n, const char* sp) {
    const uint4 s = *reinterpret_cast<const uint4*>(sp);
    const int img = m / p.vt_tokens, tok = m - img * p.vt_tokens;
    *reinterpret_cast<uint4*>(p.vt_out + (((int64_t)img * (p.N - p.vt_n0) + (n - p.vt_n0)) * p.vt_ld + tok) * 2) = s;
}

// RES / GEGLU / LN: compile-time properties of the call (16-bit residual fetched ahead; GEGLU epilogue; folded LayerNorm): with
// run-time flags the live ranges of all three share one register allocation and the chunk code spills — a scratch reload there is
// a global load behind the stores, the very wait this file is about.
// VT: columns n >= p.vt_n0 (whole output tiles: vt_n0 % 160 == 0) are V^T tiles.
template <int DT, bool RES, bool GEGLU, bool LN, bool VT>
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
    const int dbg = p.dbg_epi;      // developer switches (MFHIP_DBG_EPI): 1 no chunk work, 2 no fragment reads / MFMAs, 4 no DMA — garbage results, same barriers

    if constexpr (GEGLU) {
        // the normal CDF table (see pers_item): entry i = (Phi(x_i), Phi(x_i + 1/64) - Phi(x_i)), x_i = -6 + i / 64; before barrier #0
        if (tid < PB_LUT_N) {
            const float x0 = -6.0f + (float)tid * 0.015625f, x1 = x0 + 0.015625f;
            const float p0 = 0.5f * erfcf(-x0 * 0.70710678118654752440f), p1 = 0.5f * erfcf(-x1 * 0.70710678118654752440f);
            *reinterpret_cast<float2*>(smem + PB_LUT_OFF + tid * 8) = make_float2(p0, p1 - p0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // written before this wave arrives at barrier #0
    }

    // ---- chunks of the slab: who owns which, and the work on one ------------------------------------------------------------------
    // chunk c = 64 items, item it = 64 c + lane = (row it / 20, column group it % 20).  Epilogue wave e owns chunks 6 e .. 6 e + 5,
    // compute wave w owns 24 + 2 w and 25 + 2 w.  A wave's chunks of output tile j are worked on in the intervals after the slab was
    // published; their 16-bit residual vectors were requested one tile earlier.
    auto item_coords = [&](int c, int& row, int& cg) {
        int it = c * 64 + lane;
        asm volatile("" : "+v"(it));                                 // opaque: keeps the per-chunk address chains out of the loop-invariant hoist (they spilled)
        row = it / PB_CPR;
        cg = it - row * PB_CPR;
    };
    auto do_chunk = [&](int c, int n0, const uint4& qv) {
        if (VT && n0 >= p.vt_n0) {                                   // a V^T tile: item = (column it / 16, rows 8 (it % 16) ..)
            int it = c * 64 + lane;
            asm volatile("" : "+v"(it));
            const int col = it >> 4, rg = it & 15;
            pers_item_vt(p, m0 + rg * 8, n0 + col, smem + PB_SLAB_OFF + col * PB_SLAB_RST + rg * 16);
            return;
        }
        int row, cg;
        item_coords(c, row, cg);
        uint4 q = qv;
        if constexpr (RES) asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));   // opaque: the unpacking of all prefetched vectors was hoisted out of the loop (80 registers)
        pers_item<F16, RES, GEGLU>(p, m0 + row, n0 + cg * 8, smem + PB_SLAB_OFF + row * PB_SLAB_RS + cg * 16, q, smem + PB_LUT_OFF);
    };
    auto fetch_res = [&](int c, int n0) -> uint4 {
        int row, cg;
        item_coords(c, row, cg);
        return *reinterpret_cast<const uint4*>(p.res0 + ((int64_t)(m0 + row) * p.ld_res0 + n0 + cg * 8) * 2);
    };

    // developer switches (MFHIP_DBG_EPI bits 16 / 32 / 64): static priority 1 for the compute / staging / epilogue waves
    // GEGLU variants: the compute waves run at priority 1 — their MFMAs and fragment reads then win the issue port over the
    // epilogue waves' (and their own partners') GEGLU arithmetic, which has slack; measured 112 -> 98 us on the 64 x 64 feed-forward,
    // neutral on the other variants (profiles/r06_tile70_time_breakdown.txt)
    if ((GEGLU || (dbg & 16)) && role == 0) __builtin_amdgcn_s_setprio(1);
    if ((dbg & 32) && role == 1) __builtin_amdgcn_s_setprio(1);
    if ((dbg & 64) && role == 2) __builtin_amdgcn_s_setprio(1);
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
        const bool nodma = (dbg & 4) != 0;
        auto issue = [&]() {
            const unsigned ldsS = lds0 + i_st * PB_STAGE_BYTES;
            if (!nodma) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16_buf(offA[i] + i_kt * 128, sA, ldsS + ((4 * rw + i) * 8) * 128);
#pragma unroll
                for (int i = 0; i < 5; ++i) dma16_buf(offW[i] + wbase + i_kt * 128, sW, ldsS + PB_BM * 128 + ((5 * rw + i) * 8) * 128);
            }
            i_st = i_st == PB_STAGES - 1 ? 0 : i_st + 1;
            if (++i_kt == nkt) { i_kt = 0; wbase += wstep; }
        };
        auto wait_newer = [&](int newer) {                           // all of this wave's DMAs but those of the `newer` youngest tiles have landed
            if (newer >= 2) wait_vmcnt<18>();
            else if (newer == 1) wait_vmcnt<9>();
            else wait_vmcnt<0>();
        };
        static_assert(PB_PF <= 2, "wait_newer covers 0 .. 2 younger tiles");
        for (int k = 0; k < PB_PF && k < G; ++k) issue();
        wait_newer((G < PB_PF ? G : PB_PF) - 1);
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g + 1 < G; ++g) {
            if (g + PB_PF < G) issue();                              // K tile g + PB_PF into the stage of tile g - 1
            const int youngest = g + PB_PF < G ? g + PB_PF : G - 1;
            wait_newer(youngest - (g + 1));
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
        // The products are taken TRANSPOSED — the W fragment is the MFMA's A operand, the activation fragment its B operand (both are
        // read the same way: row r16, k chunk kg) — so that element r of acc[a][b] is row 16 a + r16, COLUMN 16 b + 4 kg + r of the wave
        // tile: a lane holds four consecutive output channels of one pixel, i.e. 8 contiguous bytes of the 16-bit slab.  (With the usual
        // order a lane holds four rows of one column and the slab write is 2-byte stores or a lane-pair exchange with selects: measured
        // 1.3 us per output tile on the two compute waves of a SIMD, half of a K = 320 tile's main loop.)
        // accumulators -> slab; `jt`: the output tile (relative to nt0) they belong to: its bias / column-sum rows sit in row buffer jt & 1.
        // V^T tiles go to the TRANSPOSED slab [column][128 rows].  The accumulators are the same (a lane: four channels of one pixel): for
        // each channel the sixteen lanes of a row group hold sixteen consecutive pixels, so neighbouring lanes (pixels P, P + 1) trade
        // halves — the even lane keeps channels 0-1 of both pixels, the odd lane channels 2-3 — and every lane writes two dwords.
        // (No second operand order in the main loop: a run-time choice between two MFMA blocks spilled the fragments.)
        auto dump_vt = [&](int jt) {
            const char* er = smem + PB_EROW_OFF + (jt & 1) * PB_EROW_BYTES;
            const float alpha = p.alpha;
            const int odd = r16 & 1;
            float2 rst[2];
            if constexpr (LN) {
#pragma unroll
                for (int a = 0; a < 2; ++a) rst[a] = lnst[wm * 32 + 16 * a + r16];
            }
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const int col = wn * 80 + 16 * b + 4 * kg;
                const float4 bias = *reinterpret_cast<const float4*>(er + col * 4);
                float4 cs = make_float4(0, 0, 0, 0);
                if constexpr (LN) cs = *reinterpret_cast<const float4*>(er + PB_BN * 4 + col * 4);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float x[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                    if constexpr (LN) {
                        x[0] = rst[a].y * (x[0] - rst[a].x * cs.x); x[1] = rst[a].y * (x[1] - rst[a].x * cs.y);
                        x[2] = rst[a].y * (x[2] - rst[a].x * cs.z); x[3] = rst[a].y * (x[3] - rst[a].x * cs.w);
                    }
                    x[0] = (x[0] + bias.x) * alpha; x[1] = (x[1] + bias.y) * alpha; x[2] = (x[2] + bias.z) * alpha; x[3] = (x[3] + bias.w) * alpha;
                    const float s0 = odd ? x[0] : x[2], s1 = odd ? x[1] : x[3];                    // what the partner takes
                    const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
                    const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
                    const uint32_t d0 = pack_h2<F16>(odd ? g0 : x[0], odd ? x[2] : g0);            // channel col + 2 odd: pixels (P & ~1, P | 1)
                    const uint32_t d1 = pack_h2<F16>(odd ? g1 : x[1], odd ? x[3] : g1);            // channel col + 2 odd + 1
                    char* sp = smem + PB_SLAB_OFF + (col + 2 * odd) * PB_SLAB_RST + (wm * 32 + 16 * a + (r16 & ~1)) * 2;
                    *reinterpret_cast<uint32_t*>(sp) = d0;
                    *reinterpret_cast<uint32_t*>(sp + PB_SLAB_RST) = d1;
                    acc[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        };
        auto dump_plain = [&](int jt) {
            const char* er = smem + PB_EROW_OFF + (jt & 1) * PB_EROW_BYTES;
            const float alpha = p.alpha;
            float2 rst[2];                                           // (mean, rstd) of this lane's two rows
            if constexpr (LN) {
#pragma unroll
                for (int a = 0; a < 2; ++a) rst[a] = lnst[wm * 32 + 16 * a + r16];
            }
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const int col = wn * 80 + 16 * b + 4 * kg;
                const float4 bias = *reinterpret_cast<const float4*>(er + col * 4);
                float4 cs = make_float4(0, 0, 0, 0);
                if constexpr (LN) cs = *reinterpret_cast<const float4*>(er + PB_BN * 4 + col * 4);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float x[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                    if constexpr (LN) {                              // rstd * (acc - mean * colsum)
                        x[0] = rst[a].y * (x[0] - rst[a].x * cs.x); x[1] = rst[a].y * (x[1] - rst[a].x * cs.y);
                        x[2] = rst[a].y * (x[2] - rst[a].x * cs.z); x[3] = rst[a].y * (x[3] - rst[a].x * cs.w);
                    }
                    x[0] = (x[0] + bias.x) * alpha; x[1] = (x[1] + bias.y) * alpha; x[2] = (x[2] + bias.z) * alpha; x[3] = (x[3] + bias.w) * alpha;
                    uint2 h;
                    h.x = pack_h2<F16>(x[0], x[1]);
                    h.y = pack_h2<F16>(x[2], x[3]);
                    *reinterpret_cast<uint2*>(smem + PB_SLAB_OFF + (wm * 32 + 16 * a + r16) * PB_SLAB_RS + col * 2) = h;
                    acc[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        };
        auto is_vt = [&](int jt) -> bool { return VT && (nt0 + jt) * PB_BN >= p.vt_n0; };
        auto dump = [&](int jt) {
            if (is_vt(jt)) dump_vt(jt);
            else dump_plain(jt);
        };
        uint4 q[RES ? PB_CC : 1];
        auto prefetch = [&](int jt) {                                // residual vectors of this wave's chunks of output tile jt
            if constexpr (RES) {
                const int n0 = (nt0 + jt) * PB_BN;
#pragma unroll
                for (int i = 0; i < PB_CC; ++i) q[i] = fetch_res(4 * PB_CE + PB_CC * rw + i, n0);
            }
        };
        if (p.nloop > 1) prefetch(0);
        const int avail = nkt - 1, per = (PB_CC + avail - 1) / avail;          // intervals a slab is readable for (nkt >= 2: host check); chunks per interval
        int st = 0, kt = 0, jt = 0, pend_n0 = -1, done = 0;          // jt: the output tile interval g belongs to
        const bool nomma = (dbg & 2) != 0;
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g < G; ++g) {
            const bool dumping = kt == 0 && g > 0;                   // the first interval of output tile jt: tile jt - 1's accumulators leave
            if (dumping && !(dbg & 8)) dump(jt - 1);
            if (!nomma) {
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
                            acc[a][b] = mfma16x32<DT>(__builtin_bit_cast(bf16x8_t, fb[ks][b]), __builtin_bit_cast(bf16x8_t, fa[ks][a]), acc[a][b], 0, 0, 0);   // transposed: see dump_plain
            }
            st = st == PB_STAGES - 1 ? 0 : st + 1;
            // this wave's share of the previous tile's slab, behind the MFMAs (issued, not yet retired: the vector work overlaps them)
            if (pend_n0 >= 0 && !dumping) {
                if (done == 0) {
                    if constexpr (RES) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the prefetch has landed (this wave's last store is a tile old) ...
#pragma unroll
                        for (int i = 0; i < PB_CC; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w));   // ... and the compiler need not wait for it again, behind a store
                    }
                }
#pragma unroll 1
                for (int i = 0; i < per && done < PB_CC; ++i, ++done) {
                    if (dbg & 1) continue;
                    if (done == 0) do_chunk(4 * PB_CE + PB_CC * rw, pend_n0, q[0]);
                    else do_chunk(4 * PB_CE + PB_CC * rw + 1, pend_n0, q[RES ? 1 : 0]);
                }
                if (done == PB_CC) {
                    pend_n0 = -1;
                    if (jt < p.nloop - 1) prefetch(jt);              // of the tile being multiplied now; behind this tile's stores, waited for a tile later
                }
            }
            if (dumping) { pend_n0 = (nt0 + jt - 1) * PB_BN; done = 0; }          // readable from the next interval on
            if (++kt == nkt) { kt = 0; ++jt; }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): fragment reads, slab writes and slab reads of this step are complete
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
        }
        dump(p.nloop - 1);                                           // the last tile: its LayerNorm statistics / rows are in place since barrier #G
        __builtin_amdgcn_s_waitcnt(0xc07f);
    } else {
        // ---- epilogue waves ------------------------------------------------------------------------------------------------------------
        uint4 q[RES ? PB_CE : 1];
        float4 pb = make_float4(0, 0, 0, 0), pc = make_float4(0, 0, 0, 0);   // wave 12, lanes 0-39: four bias / column-sum values of a later tile
        constexpr bool rows_duty = true;                             // (every epilogue wave fetches the rows: see fetch_rows)
        auto prefetch = [&](int jt) {                                // residual vectors of this wave's chunks of output tile jt
            if constexpr (RES) {
                const int n0 = (nt0 + jt) * PB_BN;
#pragma unroll
                for (int i = 0; i < PB_CE; ++i) q[i] = fetch_res(PB_CE * rw + i, n0);
            }
        };
        // bias and column-sum rows of output tile jt -> registers.  Unconditional, by every lane of every epilogue wave (lanes past the row
        // re-read its start, a call without bias reads the weight's first bytes and publishes zeros): a load under a lane mask or a branch
        // is merged with the old value by a register copy, and hipcc waits for the load — vmcnt(0), behind this wave's stores — right there
        // (measured: wave 12 stalled every block ~1.5 us per output tile).  Only wave 12 publishes.
        const float* brow = p.bias ? p.bias : reinterpret_cast<const float*>(p.w);
        const int rlane = lane < PB_BN / 4 ? lane : lane - PB_BN / 4;          // 0 .. 39
        auto fetch_rows = [&](int jt) {
            const int n0 = (nt0 + jt) * PB_BN;
            pb = *reinterpret_cast<const float4*>(brow + n0 + rlane * 4);
            if constexpr (LN) pc = *reinterpret_cast<const float4*>(p.ln_cs + n0 + rlane * 4);
        };
        auto publish_rows = [&](int jt) {                            // (wave 12) ... -> row buffer jt & 1 (the caller has waited for them)
            if (rw == 0 && lane < PB_BN / 4) {
                char* er = smem + PB_EROW_OFF + (jt & 1) * PB_EROW_BYTES;
                *reinterpret_cast<float4*>(er + lane * 16) = p.bias ? pb : make_float4(0, 0, 0, 0);
                *reinterpret_cast<float4*>(er + PB_BN * 4 + lane * 16) = pc;
            }
        };
        // rows of output tiles 0 and 1 before the first barrier (the first dump is nkt barriers away); tile j + 2's rows are published
        // while tile j's slab is worked on (its dump has read buffer j & 1 by then) and requested one tile before that
        fetch_rows(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish_rows(0);
        if (p.nloop > 1) {
            fetch_rows(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish_rows(1);
            prefetch(0);
        }
        fetch_rows(p.nloop > 2 ? 2 : p.nloop - 1);
        // folded LayerNorm: (sum, sum of squares) of the block's 128 rows, taken from the A tiles as they pass through LDS during the
        // first output tile's K steps (no global load in these waves): 32 rows per wave, two lanes per row, 64 bytes each per K tile
        float ls1 = 0.0f, ls2 = 0.0f;
        const int lrow = rw * 32 + (lane >> 1), lhalf = lane & 1;
        const int avail = nkt - 1, per = (PB_CE + avail - 1) / avail;
        int pend_n0 = -1, pend_jt = 0, done = 0, kt = 0, jt = 0, st = 0;       // jt: the output tile interval g belongs to
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g < G; ++g) {
            if (LN && g < nkt) {
                const char* ar = smem + st * PB_STAGE_BYTES + lrow * 128 + lhalf * 64;      // (the swizzle permutes a row's chunks: a sum does not care)
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
            st = st == PB_STAGES - 1 ? 0 : st + 1;
            if (pend_n0 >= 0) {
                if (done == 0) {
                    if (RES || rows_duty) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // residual vectors / rows requested a tile ago (this wave's last store is older still)
                    if constexpr (RES) {
#pragma unroll
                        for (int i = 0; i < PB_CE; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w));
                    }
                    asm volatile("" : "+v"(pb.x), "+v"(pb.y), "+v"(pb.z), "+v"(pb.w), "+v"(pc.x), "+v"(pc.y), "+v"(pc.z), "+v"(pc.w));
                    if (pend_jt + 2 < p.nloop) publish_rows(pend_jt + 2);     // rows of tile pend_jt + 2 into the buffer tile pend_jt's dump has read
                    fetch_rows(pend_jt + 3 < p.nloop ? pend_jt + 3 : p.nloop - 1);      // (unconditional: see fetch_rows)
                }
#pragma unroll 1
                for (int i = 0; i < per && done < PB_CE; ++i, ++done) {
                    if (dbg & 1) continue;
                    // (a switch keeps q[] in registers: the chunk index must be a compile-time constant)
                    switch (done) {
                        case 0: do_chunk(PB_CE * rw + 0, pend_n0, q[0]); break;
                        case 1: do_chunk(PB_CE * rw + 1, pend_n0, q[RES ? 1 : 0]); break;
                        case 2: do_chunk(PB_CE * rw + 2, pend_n0, q[RES ? 2 : 0]); break;
                        case 3: do_chunk(PB_CE * rw + 3, pend_n0, q[RES ? 3 : 0]); break;
                        case 4: do_chunk(PB_CE * rw + 4, pend_n0, q[RES ? 4 : 0]); break;
                        default: do_chunk(PB_CE * rw + 5, pend_n0, q[RES ? 5 : 0]); break;
                    }
                }
                if (done == PB_CE) {
                    pend_n0 = -1;
                    if (pend_jt + 1 < p.nloop - 1) prefetch(pend_jt + 1);      // behind this tile's stores, waited for a tile later
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // the LDS reads / writes issued so far are complete
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
            // the compute waves wrote output tile jt - 1's slab during this interval when it was the first of tile jt: readable from the next on
            if (kt == 0 && jt > 0) { pend_n0 = (nt0 + jt - 1) * PB_BN; pend_jt = jt - 1; done = 0; }
            if (++kt == nkt) { kt = 0; ++jt; }
        }
    }
    // ---- the last slab: all sixteen waves (plain loads of the residual: nothing left to hide them under) --------------------------------
    __builtin_amdgcn_s_barrier();                                    // the compute waves' last dump is in the slab
    {
        const int n0 = (nt0 + p.nloop - 1) * PB_BN;
        for (int c = wave; c < PB_CHUNKS; c += 16) {
            if (VT && n0 >= p.vt_n0) {
                if (!(dbg & 1)) do_chunk(c, n0, uint4{0, 0, 0, 0});
                continue;
            }
            int row, cg;
            item_coords(c, row, cg);
            uint4 q = uint4{0, 0, 0, 0};
            if constexpr (RES) q = *reinterpret_cast<const uint4*>(p.res0 + ((int64_t)(m0 + row) * p.ld_res0 + n0 + cg * 8) * 2);
            if (!(dbg & 1)) pers_item<F16, RES, GEGLU>(p, m0 + row, n0 + cg * 8, smem + PB_SLAB_OFF + row * PB_SLAB_RS + cg * 16, q, smem + PB_LUT_OFF);
        }
    }
}

}  // namespace

template <int DT>
static const void* pers_fn(bool res, bool geglu, bool ln, bool vt) {
    if (vt) return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, false, true, true>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, false, false, true>);
    if (geglu) return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, true, true, false>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, true, false, false>);
    if (res) return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, true, false, true, false>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, true, false, false, false>);
    return ln ? reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, false, true, false>) : reinterpret_cast<const void*>(&gemm_pers_kernel<DT, false, false, false, false>);
}

bool launch_pers(int dtype, const GemmArgs& a, hipStream_t s) {
    if (dtype != MF_BF16 && dtype != MF_F16) return false;
    const bool geglu = a.act == MF_ACT_GEGLU4, res = a.res0 != nullptr, ln = a.ln_cs != nullptr, vt = a.vt_out != nullptr;
    if ((geglu && res) || (vt && (geglu || res))) return false;
    const int fl = dtype == MF_F16 ? 1 : 0, variant = (vt ? 6 : geglu ? 4 : (res ? 2 : 0)) + (ln ? 1 : 0);
    const void* fn = fl ? pers_fn<MF_F16>(res, geglu, ln, vt) : pers_fn<MF_BF16>(res, geglu, ln, vt);
    // the dynamic-LDS attribute is per function AND per device: one flag per (flavour, variant, device), the call's result checked
    static bool attr[2][8][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!attr[fl][variant][dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM_GEGLU) != hipSuccess) return false;
        attr[fl][variant][dev] = true;
    }
    void* args[] = {const_cast<GemmArgs*>(&a)};
    return hipLaunchKernel(fn, dim3((unsigned)(a.tiles_m * (a.tiles_n / a.nloop))), dim3(1024), args, geglu ? PB_SMEM_GEGLU : PB_SMEM, s) == hipSuccess;
}

}  // namespace mfgemm
