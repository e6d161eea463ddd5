// Resident-patch 3x3 convolution kernels (tiles 16-19 of mf_gemm_conv): see gemm_conv_kernel.h for the family.
#include "gemm_conv_kernel.h"

namespace mfgemm {
namespace {

// =====================================================================================================
// 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS ("halo" tiles), bf16.
//
// The implicit-GEMM kernel above re-stages the A tile for each of the 9 taps, so a 3x3 conv moves 9 shifted copies
// of the same pixels L2 -> LDS, and the chip-wide L2 -> LDS rate (~12-13 TB/s measured) is what bounds it.  Here an
// M tile is a TH x 16 rectangle of output pixels of one image: per 32-channel chunk the (TH+2) x 18 input patch
// is DMA'd ONCE (double buffered) and the 9 taps read it at shifted row offsets; only the weights stream per
// tap, through a ring of SW thin stages (BN rows x 64 B), SW-1 of them in flight.  L2 -> LDS bytes per flop drop
// ~2x for square-ish tiles and the deeper ring keeps more bytes in flight per CU for the same LDS footprint.
//
// LDS rows are 64 B (32 bf16 channels).  16-byte chunk c of row r sits at physical chunk c ^ ((r >> 2) & 3); with
// ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS) sixteen rows whose
// indices are distinct mod 16 are conflict-free.  MFMA fragment row f of fragment I is output pixel
//     ty = 2 I + (f >> 4),  tx = ((f & 15) + 14 (f >> 4)) & 15
// (the odd image row of a fragment is rotated by two pixels) so that the patch rows a lane group reads,
// p0 + {0-3, 12-15} and p0 + 18 + {2-9}, stay distinct mod 16 for every tap shift p0.  The epilogue applies the same map.
template <int TH, int BN, int WAVES_M, int WAVES_N, int SW>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2)
void conv3x3_halo_kernel(const GemmArgs p) {
    constexpr int TW = 16, PW = TW + 2, PH = TH + 2, PPIX = PH * PW;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = TH * TW;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    // Wave roles for the DMAs: the last wave fetches the input patches, the others stream the weights.  vmcnt is per
    // wave and retires in order, so a patch (served from the Infinity Cache / HBM, needed once per 9 tiles) issued by
    // a weight wave would have to land within the weight ring's 2-3 tiles of slack; on its own wave it has all 9.
    constexpr int A_INSTR = (PPIX + 15) / 16;                  // wave-instructions (16 rows x 64 B each) per patch
    constexpr int NWW = NW - 1, AW = NW - 1;
    constexpr int W_INSTR = (BN + 15) / 16, WP = (W_INSTR + NWW - 1) / NWW;
    constexpr int A_BYTES = A_INSTR * 1024, W_BYTES = WP * NWW * 1024;
    constexpr int EP_RS = (WN + 4) * 4;
    constexpr int SMEM = 2 * A_BYTES + SW * W_BYTES;
    constexpr int SR = (NW * 32 * EP_RS <= SMEM) ? 32 : 16;
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TH % 2 == 0, "wave tile must be a multiple of 32x32");
    static_assert(NW * SR * EP_RS <= SMEM, "epilogue slabs must fit in the staging LDS");
    static_assert(SW >= 3 && SW <= 8 && (SW - 1) * WP < 64, "ring depth");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tile_m = bid / p.tiles_n;
    const int tile_n = bid - tile_m * p.tiles_n;
    const int n0 = tile_n * BN;
    const int tiles_x = p.Win / TW;
    const int tiles_img = (p.Hin / TH) * tiles_x;
    const int img = tile_m / tiles_img;
    const int tr = tile_m - img * tiles_img;
    const int y0 = (tr / tiles_x) * TH, x0 = (tr - (tr / tiles_x) * tiles_x) * TW;
    const int ksplit = blockIdx.z;

    const int c_begin = ksplit * p.kt_per_split;              // 32-channel chunks of this split
    int c_end = c_begin + p.kt_per_split;
    if (c_end > p.nkt) c_end = p.nkt;
    const int nchunks = c_end - c_begin;
    const int nt = nchunks * 9;

    const int npix = (p.M / p.HoWo) * p.Hin * p.Win;
    const srd_t srdA0 = make_srd(p.a0, (unsigned)((npix - 1) * p.ld0b + p.C0 * 2));
    const srd_t srdA1 = make_srd(p.a1 ? p.a1 : p.a0, (unsigned)((npix - 1) * p.ld1b + (p.Ctot - p.C0) * 2));
    const srd_t srdW = make_srd(p.w, (unsigned)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);

    // ---- DMA coordinates --------------------------------------------------------------------
    unsigned aoff[A_INSTR], woff[WP];
    srd_t srdCur = srdA0;
    int ai_c = c_begin * 32;                                      // first channel of the next patch to issue
    auto a_retarget = [&]() {                                      // patch wave only; runs at most twice per kernel
        const bool seg = ai_c >= p.C0;
        const int cin = seg ? ai_c - p.C0 : ai_c;
        const int ldb = seg ? p.ld1b : p.ld0b;
        srdCur = seg ? srdA1 : srdA0;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int pr = i * 16 + (lane >> 2);                   // patch row this lane fills
            const int py = pr / PW, px = pr - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = pr < PPIX && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int lc = (lane & 3) ^ ((pr >> 2) & 3);
            aoff[i] = ok ? (unsigned)(((img * p.Hin + iy) * p.Win + ix) * ldb + (cin + lc * 8) * 2) : 0x80000000u;
        }
    };
    auto issue_A = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            dma16_buf(aoff[i], srdCur, lds0 + buf * A_BYTES + i * 1024);
            aoff[i] += 64;
        }
        ai_c += 32;
        if (ai_c == p.C0 && p.Ctot > p.C0) a_retarget();      // the next patch comes from the second tensor
    };
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave + NWW * i) * 16 + (lane >> 2);       // weight row (output channel within the tile)
        const int lc = (lane & 3) ^ ((r >> 2) & 3);
        const int n = n0 + r;
        woff[i] = (wave != AW && r < BN && n < p.N) ? (unsigned)(((int64_t)n * p.ldw) * 2 + lc * 16) : 0x80000000u;
    }
    int wi_tap = 0;
    unsigned wi_k = (unsigned)c_begin * 64u;                      // byte offset of the next weight tile inside a row
    auto issue_W_at = [&](int stage, bool last_tap) {             // last_tap: the tile issued is tap 8 of its chunk
#pragma unroll
        for (int i = 0; i < WP; ++i)
            dma16_buf(woff[i] + wi_k, srdW, lds0 + 2 * A_BYTES + stage * W_BYTES + (wave + NWW * i) * 1024);
        wi_k += (unsigned)p.Ctot * 2u;
        if (last_tap) wi_k -= (unsigned)p.Ctot * 18u - 64u;
    };
    auto issue_W = [&](int stage) {
        const bool last = wi_tap == 8;
        wi_tap = last ? 0 : wi_tap + 1;
        issue_W_at(stage, last);
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;
    int pr00[MT];                                                  // patch row of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ty = 2 * (wm * MT + i) + (frow >> 4);
        const int tx = ((frow & 15) + 14 * (frow >> 4)) & 15;
        pr00[i] = ty * PW + tx;
    }
    const int bkey = (frow >> 2) & 3;
    const char* Wfrag = smem + 2 * A_BYTES + (wn * WN + frow) * 64;

    // Fragment reads run one tile ahead of the MFMAs: while tile t is multiplied from registers, the ds_reads of
    // tile t+1 are in flight, so neither the LDS latency nor the burst of reads after a barrier sits in front of the
    // matrix pipe.  Two register sets alternate roles (the loop below is unrolled by two).
    auto ldfrag = [&](int abuf, int wstage, int tapoff, uint4 (&fa)[2][MT], uint4 (&fb)[2][NT]) {
        const char* Ab = smem + abuf * A_BYTES;
        const char* Wb = Wfrag + wstage * W_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int pr = pr00[i] + tapoff;
                fa[ks][i] = *reinterpret_cast<const uint4*>(Ab + pr * 64 + ((((2 * ks + fh) ^ (pr >> 2)) & 3) << 4));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
                fb[ks][j] = *reinterpret_cast<const uint4*>(Wb + j * 32 * 64 + (((2 * ks + fh) ^ bkey) << 4));
        }
    };
    auto mma = [&](const uint4 (&fa)[2][MT], const uint4 (&fb)[2][NT]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[ks][i]),
                                                                        __builtin_bit_cast(bf16x8_t, fb[ks][j]), acc[i][j], 0, 0, 0);
    };

    // ---- main loop: tile t = (chunk, tap).  Ring invariant at the barrier of iteration t: tile t is in registers,
    // W(t+1) is resident, W(t+2) .. W(t+SW-1) are in flight and slot t % SW is free (its ds_reads retired: lgkmcnt(0)
    // precedes the barrier) for W(t+SW).  Weight waves wait with a counted vmcnt (their loads retire in order: all
    // but the SW-2 youngest tiles); the patch wave waits for its patch only in the iteration that first reads it.
    if (nt > 0) {
        uint4 fa0[2][MT], fb0[2][NT], fa1[2][MT], fb1[2][NT];
        const bool is_aw = wave == AW;
        auto wait_w = [&](int nw) {                               // at most nw weight tiles of this wave still in flight
            if (nw >= SW - 1) wait_vmcnt<(SW - 1) * WP>();
            else if (nw == SW - 2) wait_vmcnt<(SW - 2) * WP>();
            else if (SW > 3 && nw == SW - 3) wait_vmcnt<(SW > 3 ? SW - 3 : 0) * WP>();
            else if (SW > 4 && nw == SW - 4) wait_vmcnt<(SW > 4 ? SW - 4 : 0) * WP>();
            else if (nw >= 1 && nw < SW - 4) wait_vmcnt<WP>();
            else wait_vmcnt<0>();
        };
        if (is_aw) {
            a_retarget();
            issue_A(0);
            wait_vmcnt<0>();
        } else {
            for (int s0 = 0; s0 < SW; ++s0)
                if (s0 < nt) issue_W(s0);
            wait_w(nt - 1 < SW - 1 ? nt - 1 : SW - 1);
        }
        __builtin_amdgcn_s_barrier();
        ldfrag(0, 0, 0, fa0, fb0);
        // state of tile t (being multiplied) and of tile t+1 (being read)
        int tap = 0, chunk = 0, slot = 0;
        int n_abuf = 0, n_wst = 0, n_tapoff = 0, n_dx = 0, n_tap = 0;
        auto iteration = [&](int t, const uint4 (&ca)[2][MT], const uint4 (&cb)[2][NT], uint4 (&na)[2][MT], uint4 (&nb)[2][NT]) {
            const bool has_next = t + 1 < nt;
            if (has_next) {
                if (is_aw) {
                    if (tap == 8) wait_vmcnt<0>();               // tile t+1 is the first to read the next patch
                } else {
                    const int rem = nt - 2 - t;
                    wait_w(rem < SW - 2 ? rem : SW - 2);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wave's reads of tile t have retired
            __builtin_amdgcn_s_barrier();
            if (is_aw) {
                if (tap == 0 && chunk < nchunks - 1) issue_A((chunk & 1) ^ 1);
            } else if (t + SW < nt) {
                issue_W(slot);
            }
            if (has_next) {
                n_wst = n_wst == SW - 1 ? 0 : n_wst + 1;
                ++n_tap; ++n_dx; ++n_tapoff;
                if (n_dx == 3) { n_dx = 0; n_tapoff += PW - 3; }
                if (n_tap == 9) { n_tap = 0; n_tapoff = 0; n_abuf ^= 1; }
                ldfrag(n_abuf, n_wst, n_tapoff, na, nb);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == SW - 1 ? 0 : slot + 1;
            if (++tap == 9) { tap = 0; ++chunk; }
        };
        // Steady state (every chunk but the last, two chunks = 18 tiles per trip so the register sets keep their
        // roles): tap, wait counts, patch offsets and "is there a next tile" are compile-time; only the ring slot
        // and the patch buffer chunk & 1 are scalar registers.
        auto steady = [&](auto tapc, const uint4 (&ca)[2][MT], const uint4 (&cb)[2][NT], uint4 (&na)[2][MT], uint4 (&nb)[2][NT]) {
            constexpr int TAP = decltype(tapc)::value;
            constexpr int NTAP = (TAP + 1) % 9;
            if (is_aw) {
                if constexpr (TAP == 8) wait_vmcnt<0>();
            } else {
                wait_vmcnt<(SW - 2) * WP>();
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (is_aw) {
                if constexpr (TAP == 0) issue_A((chunk & 1) ^ 1);
            } else {
                issue_W_at(slot, (TAP + SW) % 9 == 8);
            }
            const int nslot = slot == SW - 1 ? 0 : slot + 1;
            ldfrag(TAP == 8 ? (chunk & 1) ^ 1 : (chunk & 1), nslot, (NTAP / 3) * PW + NTAP % 3, na, nb);
            __builtin_amdgcn_sched_barrier(0);
            mma(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
        };
        auto steady_chunk = [&](uint4 (&a0)[2][MT], uint4 (&b0)[2][NT], uint4 (&a1)[2][MT], uint4 (&b1)[2][NT]) {
            steady(std::integral_constant<int, 0>{}, a0, b0, a1, b1);
            steady(std::integral_constant<int, 1>{}, a1, b1, a0, b0);
            steady(std::integral_constant<int, 2>{}, a0, b0, a1, b1);
            steady(std::integral_constant<int, 3>{}, a1, b1, a0, b0);
            steady(std::integral_constant<int, 4>{}, a0, b0, a1, b1);
            steady(std::integral_constant<int, 5>{}, a1, b1, a0, b0);
            steady(std::integral_constant<int, 6>{}, a0, b0, a1, b1);
            steady(std::integral_constant<int, 7>{}, a1, b1, a0, b0);
            steady(std::integral_constant<int, 8>{}, a0, b0, a1, b1);
            ++chunk;
        };
        while (chunk + 2 < nchunks) {
            steady_chunk(fa0, fb0, fa1, fb1);
            steady_chunk(fa1, fb1, fa0, fb0);
        }
        // tail (the last one or two chunks, and every short split): the generic iteration with run-time checks
        int t = chunk * 9;
        n_abuf = chunk & 1; n_wst = slot;
        wi_tap = SW % 9;                                           // tap of W(t + SW): t is a multiple of 9
        for (; t + 1 < nt; t += 2) {
            iteration(t, fa0, fb0, fa1, fb1);
            iteration(t + 1, fa1, fb1, fa0, fb0);
        }
        if (t < nt) iteration(t, fa0, fb0, fa1, fb1);
    }
    __syncthreads();

    // ---- epilogue (same slab scheme as gemm_conv_kernel; rows go through the pixel map above) ------------------
    char* slab = smem + wave * (SR * EP_RS);
    constexpr int CPR = WN / 8;
    constexpr int ITEMS = SR * CPR;
    float* ws = p.splitk > 1 ? p.ws + (int64_t)ksplit * (int64_t)p.M * p.N : nullptr;
#pragma unroll
    for (int ih = 0; ih < MT * (32 / SR); ++ih) {
        const int i = ih / (32 / SR), half = ih % (32 / SR);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = half * (SR / 2); e < half * (SR / 2) + SR / 2; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * fh - half * SR;
                *reinterpret_cast<float*>(slab + row * EP_RS + (j * 32 + frow) * 4) = acc[i][j][e];
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it0 = 0; it0 < ITEMS; it0 += 64) {
            const int it = it0 + lane;
            if (ITEMS % 64 != 0 && it >= ITEMS) continue;
            const int row = it / CPR, ec = (it - row * CPR) * 8;
            const int f = half * SR + row;
            const int ty = 2 * (wm * MT + i) + (f >> 4);
            const int tx = ((f & 15) + 14 * (f >> 4)) & 15;
            const int m = (img * p.Hin + y0 + ty) * p.Win + x0 + tx;
            const int n = n0 + wn * WN + ec;
            float v[8];
            const float4 lo = *reinterpret_cast<const float4*>(slab + row * EP_RS + ec * 4);
            const float4 hi = *reinterpret_cast<const float4*>(slab + row * EP_RS + ec * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            if (n < p.N) {
                if (ws) {
                    if (n + 8 <= p.N && (p.N & 3) == 0) {
                        *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) = lo;
                        *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n + 4) = hi;
                    } else {
                        for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) ws[(int64_t)m * p.N + n + jj] = v[jj];
                    }
                } else if (p.vec_ok && n + 8 <= p.N) {
                    epilogue_store8<false>(p, 0, m, n, v);
                } else {
                    for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) epilogue_store(p, 0, m, n + jj, v[jj]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// =====================================================================================================
// conv3x3_pingpong_kernel: the halo idea on a 256 x 160 tile with ONE 8-wave block per CU and a two-group ping-pong
// schedule (cdna_hip_programming.md, 8-wave GEMM template).  The per-CU DMA path (vector-memory issue -> TA/TCP ->
// LDS) costs ~30 % of the 128 x 160 kernels whatever the L2 hit rate ("What bounds the convs"); this tile
// moves 2.9x fewer DMA'd bytes per flop: 16 x 16 output pixels share one 18 x 18 x 64-channel patch (full 128-byte rows)
// per 9 taps, and 256 pixels share every 160 x 64 weight stage.
//
// Waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the four SIMDs.  A tile (one tap x 64 channels) is two
// sub-tiles of two k-steps; time is cut into intervals by s_barrier:
//     interval 2s   : group 0 multiplies sub-tile s from registers | group 1 reads the fragments of sub-tile s
//     interval 2s+1 : group 0 reads the fragments of sub-tile s+1  | group 1 multiplies sub-tile s
// so each SIMD's matrix pipe always has exactly one wave in its MFMA phase (10 MFMAs), and every wave does its DMA
// issue, LDS reads and bookkeeping while its SIMD partner multiplies.  One sub-tile's fragments (12 x 16 B) live in
// registers (a whole tile's 96 VGPRs beside the 80 accumulators spilled).  Ring: stage t is read in intervals 4t-1 ..
// 4t+2; a group refills it with its share of W(t+3) when it starts reading tile t+1, and every weight wave makes
// sure its share of W(T) has landed at the end of interval 4T-2: 7-8 intervals of slack.  Weight DMAs come from waves
// 0-6 (3 x 1 KiB each per stage); wave 7 fetches the patches (41 x 1 KiB per chunk, 4 per read phase of taps 0..5 so
// the address arithmetic is spread out) and therefore only ever waits for a patch at a chunk boundary.
template <int BN, int SW>
__global__ __launch_bounds__(512, 2)
void conv3x3_pingpong_kernel(const GemmArgs p) {
    constexpr int TH = 16, TW = 16, PW = TW + 2, PH = TH + 2, PPIX = PH * PW;
    constexpr int NW = 8, AW = 7, NWW = 7;
    constexpr int NT = BN / 32, KS = 4;
    constexpr int A_INSTR = (PPIX + 7) / 8;                     // 8 rows x 128 B per wave-instruction
    constexpr int A_PER_TAP = (A_INSTR + 5) / 6;                // issued during taps 0..5 of the previous chunk
    constexpr int W_INSTR = (BN + 7) / 8, WP = (W_INSTR + NWW - 1) / NWW;
    constexpr int A_BYTES = A_INSTR * 1024, W_BYTES = WP * NWW * 1024;
    constexpr int WN = BN, EP_RS = (WN + 4) * 4;
    constexpr int SMEM = 2 * A_BYTES + SW * W_BYTES;
    constexpr int SR = 16;
    static_assert(BN % 32 == 0 && SW == 3, "tile");
    static_assert(NW * SR * EP_RS <= SMEM && SMEM <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;

    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tile_m = bid / p.tiles_n;
    const int tile_n = bid - tile_m * p.tiles_n;
    const int n0 = tile_n * BN;
    const int tiles_x = p.Win / TW;
    const int tiles_img = (p.Hin / TH) * tiles_x;
    const int img = tile_m / tiles_img;
    const int tr = tile_m - img * tiles_img;
    const int y0 = (tr / tiles_x) * TH, x0 = (tr - (tr / tiles_x) * tiles_x) * TW;
    const int ksplit = blockIdx.z;

    const int c_begin = ksplit * p.kt_per_split;              // 64-channel chunks of this split
    int c_end = c_begin + p.kt_per_split;
    if (c_end > p.nkt) c_end = p.nkt;
    const int nchunks = c_end - c_begin;
    const int nt = nchunks * 9;

    const int npix = (p.M / p.HoWo) * p.Hin * p.Win;
    const srd_t srdA0 = make_srd(p.a0, (unsigned)((npix - 1) * p.ld0b + p.C0 * 2));
    const srd_t srdA1 = make_srd(p.a1 ? p.a1 : p.a0, (unsigned)((npix - 1) * p.ld1b + (p.Ctot - p.C0) * 2));
    const srd_t srdW = make_srd(p.w, (unsigned)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);

    // ---- weight DMA (waves 0..6) ----------------------------------------------------------------
    unsigned woff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave + NWW * i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        const int n = n0 + r;
        woff[i] = (wave != AW && r < BN && n < p.N) ? (unsigned)(((int64_t)n * p.ldw) * 2 + lc * 16) : 0x80000000u;
    }
    int wi_tap = 0;
    unsigned wi_k = (unsigned)c_begin * 128u;
    auto issue_W = [&](int stage) {
#pragma unroll
        for (int i = 0; i < WP; ++i)
            dma16_buf(woff[i] + wi_k, srdW, lds0 + 2 * A_BYTES + stage * W_BYTES + (wave + NWW * i) * 1024);
        wi_k += (unsigned)p.Ctot * 2u;
        if (++wi_tap == 9) { wi_tap = 0; wi_k -= (unsigned)p.Ctot * 18u - 128u; }
    };
    // ---- patch DMA (wave 7): instruction q of the patch of the chunk starting at channel c ---------------
    // The pixel index of every (instruction, lane) of a patch is chunk invariant: a table in LDS (built once by the
    // whole block) keeps wave 7's read phases down to one ds_read + 4 VALU per DMA instruction.
    int* pix_tab = reinterpret_cast<int*>(smem + SMEM);
    for (int e = tid; e < A_INSTR * 64; e += 512) {
        const int pr = (e >> 6) * 8 + ((e & 63) >> 3);
        const int py = pr / PW, px = pr - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = pr < PPIX && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        pix_tab[e] = ok ? (img * p.Hin + iy) * p.Win + ix : -1;
    }
    __syncthreads();
    auto issue_A_range = [&](int buf, int cstart, int q_lo, int q_hi) {
        const bool seg = cstart >= p.C0;
        const int cin = seg ? cstart - p.C0 : cstart;
        const int ldb = seg ? p.ld1b : p.ld0b;
        const srd_t srd = seg ? srdA1 : srdA0;
        for (int q = q_lo; q < q_hi; ++q) {
            const int pix = pix_tab[q * 64 + lane];
            const int lc = (lane & 7) ^ ((q * 4 + (lane >> 4)) & 7);          // (pr >> 1) & 7 with pr = 8 q + lane / 8
            const unsigned off = pix >= 0 ? (unsigned)(pix * ldb + (cin + lc * 8) * 2) : 0x80000000u;
            dma16_buf(off, srd, lds0 + buf * A_BYTES + q * 1024);
        }
    };

    f32x16_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;
    const int pr00 = (2 * wave + (frow >> 4)) * PW + (((frow & 15) + 14 * (frow >> 4)) & 15);
    const int bkey = (frow >> 1) & 7;
    const char* Wfrag = smem + 2 * A_BYTES + frow * 128;

    uint4 fa[2], fb[2][NT];
    auto ldfrag = [&](int abuf, int wstage, int tapoff, int half) {
        const int pr = pr00 + tapoff;
        const char* Ap = smem + abuf * A_BYTES + pr * 128;
        const int akey = (pr >> 1) & 7;
        const char* Wb = Wfrag + wstage * W_BYTES;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int c = 4 * half + 2 * k2 + fh;             // 16-byte chunk = k-step (2 half + k2), lane half fh
            fa[k2] = *reinterpret_cast<const uint4*>(Ap + ((c ^ akey) << 4));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                fb[k2][j] = *reinterpret_cast<const uint4*>(Wb + j * 32 * 128 + ((c ^ bkey) << 4));
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[k2]),
                                                                 __builtin_bit_cast(bf16x8_t, fb[k2][j]), acc[j], 0, 0, 0);
    };

    if (nt > 0) {
        const bool is_aw = wave == AW;
        constexpr int A_PER_PHASE = (A_INSTR + 11) / 12;       // patch instructions per read phase over taps 0..5
        // prologue: patch 0, stages 0..2; group 0 reads sub-tile 0 in "interval -1"
        if (is_aw) {
            issue_A_range(0, c_begin * 64, 0, A_INSTR);
            wait_vmcnt<0>();
        } else {
            for (int s0 = 0; s0 < SW; ++s0)
                if (s0 < nt) issue_W(s0);
            if (nt >= 3) wait_vmcnt<2 * WP>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        // per-group state of the sub-tile it reads next (group 0 starts with sub-tile 1 = tile 0 / half 1, group 1 with 0)
        int r_t = 0, r_half = 0, r_tap = 0, r_tapoff = 0, r_dx = 0, r_abuf = 0, r_slot = 0, r_chunk = 0;
        auto read_phase = [&]() {
            if (r_t < nt) {
                ldfrag(r_abuf, r_slot, r_tapoff, r_half);
                if (!is_aw) {
                    // starting tile r_t: tile r_t - 1 has been read by both groups, its stage takes W(r_t - 1 + SW)
                    if (r_half == 0 && r_t >= 1 && r_t - 1 + SW < nt) issue_W(r_slot == 0 ? SW - 1 : r_slot - 1);
                } else if (r_chunk + 1 < nchunks && r_tap < 6) {
                    const int q_lo = (2 * r_tap + r_half) * A_PER_PHASE;
                    int q_hi = q_lo + A_PER_PHASE;
                    if (q_hi > A_INSTR) q_hi = A_INSTR;
                    issue_A_range((r_chunk & 1) ^ 1, (c_begin + r_chunk + 1) * 64, q_lo, q_hi);
                }
            }
            if (r_half == 0) {
                r_half = 1;
            } else {
                r_half = 0;
                ++r_t;
                r_slot = r_slot == SW - 1 ? 0 : r_slot + 1;
                ++r_tap; ++r_dx; ++r_tapoff;
                if (r_dx == 3) { r_dx = 0; r_tapoff += PW - 3; }
                if (r_tap == 9) { r_tap = 0; r_tapoff = 0; r_abuf ^= 1; ++r_chunk; }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): fragments are in registers, the stage may be refilled
        };
        auto end_even = [&](int s) {       // end of interval 2s
            if (!is_aw) {
                if (s & 1) {               // interval 4T-2, T = (s+1)/2: every wave's share of W(T) must be in before 4T-1
                    if (((s + 1) >> 1) + 1 < nt) wait_vmcnt<WP>();
                    else wait_vmcnt<0>();
                }
            } else if (r_tap == 0 && r_half == 0) {
                wait_vmcnt<0>();           // wave 7 has just read the last sub-tile of a chunk: the next one opens the next patch
            }
            __builtin_amdgcn_s_barrier();
        };
        const int ns = 2 * nt;
        if (grp == 0) {
            read_phase();                                // sub-tile 0
            for (int sidx = 0; sidx < ns; ++sidx) {
                __builtin_amdgcn_sched_barrier(0);
                mma();                                   // interval 2s
                __builtin_amdgcn_sched_barrier(0);
                end_even(sidx);
                read_phase();                            // interval 2s+1: sub-tile s+1
                __builtin_amdgcn_s_barrier();
            }
        } else {
            for (int sidx = 0; sidx < ns; ++sidx) {
                read_phase();                            // interval 2s
                end_even(sidx);
                __builtin_amdgcn_sched_barrier(0);
                mma();                                   // interval 2s+1
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    __syncthreads();

    // ---- epilogue (slab scheme of gemm_conv_kernel; rows go through the pixel map) ------------------------------
    char* slab = smem + wave * (SR * EP_RS);
    constexpr int CPR = WN / 8;
    constexpr int ITEMS = SR * CPR;
    float* ws = p.splitk > 1 ? p.ws + (int64_t)ksplit * (int64_t)p.M * p.N : nullptr;
#pragma unroll
    for (int half = 0; half < 32 / SR; ++half) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = half * (SR / 2); e < half * (SR / 2) + SR / 2; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * fh - half * SR;
                *reinterpret_cast<float*>(slab + row * EP_RS + (j * 32 + frow) * 4) = acc[j][e];
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it0 = 0; it0 < ITEMS; it0 += 64) {
            const int it = it0 + lane;
            if (ITEMS % 64 != 0 && it >= ITEMS) continue;
            const int row = it / CPR, ec = (it - row * CPR) * 8;
            const int f = half * SR + row;
            const int ty = 2 * wave + (f >> 4);
            const int tx = ((f & 15) + 14 * (f >> 4)) & 15;
            const int m = (img * p.Hin + y0 + ty) * p.Win + x0 + tx;
            const int n = n0 + ec;
            float v[8];
            const float4 lo = *reinterpret_cast<const float4*>(slab + row * EP_RS + ec * 4);
            const float4 hi = *reinterpret_cast<const float4*>(slab + row * EP_RS + ec * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            if (n < p.N) {
                if (ws) {
                    if (n + 8 <= p.N && (p.N & 3) == 0) {
                        *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) = lo;
                        *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n + 4) = hi;
                    } else {
                        for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) ws[(int64_t)m * p.N + n + jj] = v[jj];
                    }
                } else if (p.vec_ok && n + 8 <= p.N) {
                    epilogue_store8<false>(p, 0, m, n, v);
                } else {
                    for (int jj = 0; jj < 8 && n + jj < p.N; ++jj) epilogue_store(p, 0, m, n + jj, v[jj]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int TH, int BN, int WMv, int WNv, int SW>
void launch_halo(const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int NW = WMv * WNv, PPIX = (TH + 2) * 18;
    constexpr int WP = ((BN + 15) / 16 + NW - 2) / (NW - 1);
    constexpr int smem = 2 * ((PPIX + 15) / 16) * 1024 + SW * WP * (NW - 1) * 1024;
    static_assert(smem <= 80 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<TH, BN, WMv, WNv, SW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<TH, BN, WMv, WNv, SW>), grid, dim3(NW * 64), smem, s, a);
}

template <int BN>
void launch_pingpong(const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int smem = 2 * ((18 * 18 + 7) / 8) * 1024 + 3 * (((BN + 7) / 8 + 6) / 7) * 7 * 1024 + ((18 * 18 + 7) / 8) * 256;
    static_assert(smem <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pingpong_kernel<BN, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_pingpong_kernel<BN, 3>), grid, dim3(512), smem, s, a);
}

}  // namespace

bool launch_halo_family(int tile, const GemmArgs& a, dim3 hgrid, hipStream_t hs) {
    if (tile == 19) launch_pingpong<160>(a, hgrid, hs);
    else if (tile == 16) launch_halo<8, 160, 4, 1, 4>(a, hgrid, hs);
    else if (tile == 17) launch_halo<8, 128, 2, 2, 4>(a, hgrid, hs);
    else if (tile == 18) launch_halo<8, 128, 2, 2, 6>(a, hgrid, hs);
    else return false;
    return true;
}

}  // namespace mfgemm
