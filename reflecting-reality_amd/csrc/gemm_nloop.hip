// Persistent short-K GEMM (tile 69): one block keeps its 64 rows of A and walks a RANGE of 160-column output tiles, so that the
// per-block costs of the implicit-GEMM kernel — argument loads and address set-up, the first DMA round trip, block turnover — are paid
// once per range instead of once per tile, and the epilogue of tile j runs UNDER the main loop of tile j + 1.
//
// Why: the feed-forward input projection of a transformer block (attention.py:1188-1201 → GEGLU, activations.py:85-103: K = 320 / 640 /
// 1280, N = 8 K) is 4096 output tiles of five K tiles each at 64 x 64 latents; a block of gemm_conv_kernel lives 11 us for 2.3 us of
// main loop there (profiles/r05_gemm_phase_stamps.txt) and the launch runs at 430 TF/s.
//
// Roles (768 threads = 12 waves, three per SIMD): waves 0-3 multiply (2 x 2 wave tiles of 32 x 80 on v_mfma_f32_16x16x32), waves 4-7
// stage (LDS-DMA of the A and W tiles into a 3-deep ring, the counted-vmcnt / one-barrier-per-K-tile protocol of the warp-specialised
// tiles of gemm_conv_kernel.h, running through all the block's output tiles without a drain), waves 8-11 run the epilogue: when the
// compute waves have finished output tile j they write their accumulators to an fp32 slab in LDS (its own 42 KB, not the ring) and go
// on with tile j + 1; the epilogue waves turn the slab into output rows — LayerNorm fold, bias, residual, SiLU / GEGLU, cast, 16-byte
// stores: gemm_conv_kernel's own epilogue_store8 — between the barriers of tile j + 1's K tiles.  DMA waves and storing waves are
// different waves on purpose: loads and stores share vmcnt but complete out of order with respect to each other, which would break the
// counted waits of the ring.
//
// Every wave arrives at every barrier.  Barrier #g (g = 0 .. G - 1, G = tiles x K tiles) means "K tile g has landed and K tile g - 1 is
// no longer read"; the slab of output tile j is written before barrier #(last(j) + 1) and read between barriers #(last(j) + 1) and
// #(last(j) + nkt) — the compute waves write the next slab only after the latter.  After barrier #G all twelve waves share the last slab.
#include <hip/hip_runtime.h>

#include "gemm_conv_kernel.h"

namespace mfgemm {

namespace {

constexpr int NL_BM = 64, NL_BN = 160, NL_STAGES = 3, NL_PF = NL_STAGES - 1;   // ring depth: K tiles g + 1 .. g + NL_PF in flight while tile g is multiplied
constexpr int NL_STAGE_BYTES = (NL_BM + NL_BN) * 128;               // 28 KB: [A tile 64 rows][W tile 160 rows], 128 bytes of K per row
constexpr int NL_SLAB_RS = (NL_BN + 4) * 4;                         // slab row stride (bytes): 160 fp32 + 16 bytes (bank spread)
constexpr int NL_SLAB_OFF = NL_STAGES * NL_STAGE_BYTES;             // 86016
constexpr int NL_LNST_OFF = NL_SLAB_OFF + NL_BM * NL_SLAB_RS;       // + 41984
constexpr int NL_SMEM = NL_LNST_OFF + NL_BM * 8;                    // + 512 = 128512 bytes
static_assert(NL_SMEM <= 160 * 1024, "LDS");
constexpr int NL_CHUNKS = NL_BM * (NL_BN / 8) / 64;                 // 20 chunks of 64 (row, 8-column) items per output tile

template <int DT>
__global__ __launch_bounds__(768, 1) void gemm_nloop_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool F16 = DT == MF_F16;
    const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int role = wave >> 2, rw = wave & 3;                       // 0 compute, 1 staging, 2 epilogue; wave of its role
    const int nranges = p.tiles_n / p.nloop;                         // column ranges per row tile
    const int tile_m = (int)blockIdx.x / nranges, range = (int)blockIdx.x - tile_m * nranges;
    const int m0 = tile_m * NL_BM, nt0 = range * p.nloop;            // first row, first output tile of this block
    const int nkt = p.nkt, G = p.nloop * nkt;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
    float2* lnst = reinterpret_cast<float2*>(smem + NL_LNST_OFF);    // (mean, rstd) of the block's rows (folded LayerNorm)

    // ---- the epilogue of one slab chunk: 64 items of (row, 8 columns) ---------------------------------------------------------------
    auto do_chunk = [&](int c, int n0) {
        const int it = c * 64 + lane;
        const int row = it / (NL_BN / 8), cg = it - row * (NL_BN / 8);
        const int m = m0 + row, n = n0 + cg * 8;
        const char* sp = smem + NL_SLAB_OFF + row * NL_SLAB_RS + cg * 32;
        float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 16);
        if (p.ln_cs) {                                               // rstd * (acc - mean * colsum)
            const float2 st = lnst[row];
            const float4 c0 = *reinterpret_cast<const float4*>(p.ln_cs + n), c1 = *reinterpret_cast<const float4*>(p.ln_cs + n + 4);
            lo.x = st.y * (lo.x - st.x * c0.x); lo.y = st.y * (lo.y - st.x * c0.y); lo.z = st.y * (lo.z - st.x * c0.z); lo.w = st.y * (lo.w - st.x * c0.w);
            hi.x = st.y * (hi.x - st.x * c1.x); hi.y = st.y * (hi.y - st.x * c1.y); hi.z = st.y * (hi.z - st.x * c1.z); hi.w = st.y * (hi.w - st.x * c1.w);
        }
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        epilogue_store8<F16>(p, 0, m, n, v);
    };

    if (role == 1) {
        // ---- staging waves: A tile 8 DMAs of 8 rows (two per wave), W tile 20 (five per wave), swizzle applied to the SOURCE chunk ----
        const srd_t sA = make_srd(p.a0, (unsigned)((int64_t)p.M * p.ld0b));
        const srd_t sW = make_srd(p.w, (unsigned)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2));
        const int r8 = lane >> 3, slot = lane & 7;
        unsigned offA[2], offW[5];                                   // byte offsets of this lane's 16 bytes at K tile 0 / output tile 0
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (2 * rw + i) * 8 + r8;
            offA[i] = (unsigned)(m0 + r) * (unsigned)p.ld0b + (unsigned)((slot ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int r = (5 * rw + i) * 8 + r8;
            offW[i] = (unsigned)(nt0 * NL_BN + r) * (unsigned)(p.ldw * 2) + (unsigned)((slot ^ ((r >> 1) & 7)) << 4);
        }
        const unsigned wstep = (unsigned)NL_BN * (unsigned)(p.ldw * 2);           // one output tile further
        int i_kt = 0, i_st = 0;
        unsigned wbase = 0;
        auto issue = [&]() {
            const unsigned ldsS = lds0 + i_st * NL_STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_buf(offA[i] + i_kt * 128, sA, ldsS + ((2 * rw + i) * 8) * 128);
#pragma unroll
            for (int i = 0; i < 5; ++i) dma16_buf(offW[i] + wbase + i_kt * 128, sW, ldsS + NL_BM * 128 + ((5 * rw + i) * 8) * 128);
            i_st = i_st == NL_STAGES - 1 ? 0 : i_st + 1;
            if (++i_kt == nkt) { i_kt = 0; wbase += wstep; }
        };
        auto wait_newer = [&](int newer) {                           // all of this wave's DMAs but those of the `newer` youngest tiles have landed
            if (newer >= 2) wait_vmcnt<14>();
            else if (newer == 1) wait_vmcnt<7>();
            else wait_vmcnt<0>();
        };
        static_assert(NL_PF <= 3, "wait_newer covers 0 .. 2 younger tiles");      // (a 4-deep ring measured no faster: the loop is not DMA-latency bound)
        for (int k = 0; k < NL_PF && k < G; ++k) issue();
        wait_newer((G < NL_PF ? G : NL_PF) - 1);
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g + 1 < G; ++g) {
            if (g + NL_PF < G) issue();                              // K tile g + NL_PF into the stage of tile g - 1
            const int youngest = g + NL_PF < G ? g + NL_PF : G - 1;
            wait_newer(youngest - (g + 1));
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
        }
        __builtin_amdgcn_s_barrier();                                // #G
    } else if (role == 0) {
        // ---- compute waves: 2 x 2 wave tiles of 32 x 80 = 2 x 5 tiles of 16 x 16 ------------------------------------------------------
        const int wm = rw >> 1, wn = rw & 1, r16 = lane & 15, kg = lane >> 4;
        const int key16 = (r16 >> 1) & 7;                            // tile bases are multiples of 16 rows: one key for every fragment
        f32x4_t acc[2][5];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        int st = 0, kt = 0;
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g < G; ++g) {
            const char* A16 = smem + st * NL_STAGE_BYTES + (wm * 32 + r16) * 128;
            const char* B16 = smem + st * NL_STAGE_BYTES + NL_BM * 128 + (wn * 80 + r16) * 128;
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
            st = st == NL_STAGES - 1 ? 0 : st + 1;
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
                            *reinterpret_cast<float*>(smem + NL_SLAB_OFF + (wm * 32 + 16 * a + 4 * kg + r) * NL_SLAB_RS + (wn * 80 + 16 * b + r16) * 4) = acc[a][b][r];
                        acc[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
                    }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): fragment reads (and slab writes) of this step are complete
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
        }
    } else {
        // ---- epilogue waves ------------------------------------------------------------------------------------------------------------
        if (p.ln_cs) {
            // (mean, rstd) of the block's 64 rows over K: 16 rows per wave, four lanes per row, 16-byte loads
            const int row = rw * 16 + (lane >> 2), q = lane & 3;
            const char* ap = p.a0 + (int64_t)(m0 + row) * p.ld0b;
            float s1 = 0.0f, s2 = 0.0f;
            for (int c = q; c < p.K / 8; c += 4) {
                float x[8];
                unpack_h8<F16>(*reinterpret_cast<const uint4*>(ap + c * 16), x);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1 += x[e]; s2 = fmaf(x[e], x[e], s2); }
            }
            s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
            s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
            const float invk = 1.0f / (float)p.K, mean = s1 * invk;
            float var = s2 * invk - mean * mean;
            if (var < 0.0f) var = 0.0f;
            if (q == 0) lnst[row] = make_float2(mean, 1.0f / sqrtf(var + p.ln_eps));
        }
        const int per_wave = NL_CHUNKS / 4;                          // 5 chunks of a slab per epilogue wave
        const int cpi = (per_wave + nkt - 2) / (nkt - 1);            // chunks per barrier interval (nkt >= 2: host check)
        int pending_n0 = -1, done = 0, kt = 0, jt = 0;
        __builtin_amdgcn_s_barrier();                                // #0
        for (int g = 0; g < G; ++g) {
            if (pending_n0 >= 0) {
                for (int i = 0; i < cpi && done < per_wave; ++i, ++done) do_chunk(rw + 4 * done, pending_n0);
                if (done == per_wave) pending_n0 = -1;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // the slab reads issued so far have returned
            __builtin_amdgcn_s_barrier();                            // #(g + 1)
            if (++kt == nkt) {                                       // the compute waves wrote output tile jt's slab before this barrier
                kt = 0;
                if (g + 1 < G) { pending_n0 = (nt0 + jt) * NL_BN; done = 0; }
                ++jt;
            }
        }
    }
    // ---- the last slab: all twelve waves ----------------------------------------------------------------------------------------------
    const int n0_last = (nt0 + p.nloop - 1) * NL_BN;
    for (int c = wave; c < NL_CHUNKS; c += 12) do_chunk(c, n0_last);
}

}  // namespace

bool launch_nloop(int dtype, const GemmArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.tiles_m * (a.tiles_n / a.nloop)));
    if (dtype != MF_BF16 && dtype != MF_F16) return false;
    // the dynamic-LDS attribute is per function AND per device: one flag per (flavour, device), the call's result checked
    static bool attr[2][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const int fl = dtype == MF_F16 ? 1 : 0;
    if (!attr[fl][dev]) {
        const void* fn = fl ? reinterpret_cast<const void*>(&gemm_nloop_kernel<MF_F16>) : reinterpret_cast<const void*>(&gemm_nloop_kernel<MF_BF16>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NL_SMEM) != hipSuccess) return false;
        attr[fl][dev] = true;
    }
    if (fl) hipLaunchKernelGGL(gemm_nloop_kernel<MF_F16>, grid, dim3(768), NL_SMEM, s, a);
    else hipLaunchKernelGGL(gemm_nloop_kernel<MF_BF16>, grid, dim3(768), NL_SMEM, s, a);
    return true;
}

}  // namespace mfgemm
