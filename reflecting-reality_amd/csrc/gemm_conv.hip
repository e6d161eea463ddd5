// Host entry point of the implicit-GEMM family (mf_gemm_conv): argument checks, tile / split-K choice, dispatch to the tile
// groups (gemm_*.hip, conv_halo.hip).  Kernel template and design notes: gemm_conv_kernel.h.
#include "gemm_conv_kernel.h"

namespace mfgemm {
namespace {

__device__ unsigned g_split_ovf_gemm;     // raised when an MF_F16X3 operand exceeded the fp16 range (mf_common.h); kernels get its address in GemmArgs::ovf
#ifdef MF_STAMPS
unsigned long long* g_stamps_host = nullptr;
#endif

template <bool F16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int64_t mn = (int64_t)p.M * p.N;
    if (p.vec_ok) {           // N % 8 == 0: 8 channels per thread, 16/32-byte accesses everywhere
        const int64_t total8 = mn * p.nz / 8;
        for (int64_t i8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i8 < total8;
             i8 += (int64_t)gridDim.x * blockDim.x) {
            const int64_t idx = i8 * 8;
            const int z = (int)(idx / mn);
            const int64_t r = idx - (int64_t)z * mn;
            const int m = (int)(r / p.N);
            const int n = (int)(r - (int64_t)m * p.N);
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            // four slabs' loads in flight at a time (a loop of one load + one add per slab is a chain of `splitk` fabric round trips:
            // the slabs were written by other XCDs); the additions keep the slab order, so the sums are bit-identical
            const int64_t sstride = (int64_t)p.nz * mn;
            const float* src0 = p.ws + (int64_t)z * mn + r;
            int s = 0;
            for (; s + 4 <= p.splitk; s += 4) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const float4*>(src0 + (s + u) * sstride);
                    b[u] = *reinterpret_cast<const float4*>(src0 + (s + u) * sstride + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
                }
            }
            for (; s < p.splitk; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(src0 + s * sstride);
                const float4 b = *reinterpret_cast<const float4*>(src0 + s * sstride + 4);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
            const int zq = z / p.zdiv, zr = z - zq * p.zdiv;
            epilogue_store8<F16>(p, zq * p.o_zs_o + zr * p.o_zs_i, m, n, v, false, uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}, zq);
        }
        return;
    }
    const int64_t total = mn * p.nz;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(idx / mn);
        const int64_t r = idx - (int64_t)z * mn;
        const int m = (int)(r / p.N);
        const int n = (int)(r - (int64_t)m * p.N);
        float v = 0.0f;
        for (int s = 0; s < p.splitk; ++s) v += p.ws[((int64_t)s * p.nz + z) * mn + r];
        const int zq = z / p.zdiv, zr = z - zq * p.zdiv;
        epilogue_store(p, zq * p.o_zs_o + zr * p.o_zs_i, m, n, v, zq);
    }
}

// mf_gemm_desc.gn_part for launches whose epilogue cannot produce it (split-K reduce, resident-patch and persistent tiles, image
// sizes the tile's rows do not divide): per-channel (sum, sum of squares) of every block of R rows of the STORED output.  grid
// (M / R, ceil(N / 256)); a thread owns one column (coalesced 2- / 4-byte loads across the block), fixed order.
__global__ __launch_bounds__(256) void gn_colsum_kernel(const char* out, int out_dt, int64_t ldc, int M, int N, int R, float2* part) {
    const int n = blockIdx.y * 256 + threadIdx.x, rb = blockIdx.x;
    if (n >= N) return;
    float a = 0.0f, b = 0.0f;
    const int m1 = (rb + 1) * R < M ? (rb + 1) * R : M;
    for (int m = rb * R; m < m1; ++m) {
        const float x = load_as_f32(out, out_dt, (int64_t)m * ldc + n);
        a += x; b = fmaf(x, x, b);
    }
    part[(int64_t)rb * N + n] = make_float2(a, b);
}

// ... the same for a 16-bit output with 16-byte rows: 32 column groups of 8 channels x 8 row lanes per block, 16-byte loads, the row
// lanes combined through LDS in lane order (the scalar form above walks R rows with one 2-byte load each: 38 us where this takes ~5)
// ET: 0 bf16, 1 fp16, 2 fp32 (the split-precision modes keep fp32 activations: two 16-byte loads per 8 channels)
template <int ET>
__global__ __launch_bounds__(256) void gn_colsum8_kernel(const char* out, int64_t ldc, int M, int N, int R, float2* part) {
    __shared__ float2 red[8][32][8];
    const int cgl = threadIdx.x & 31, rl = threadIdx.x >> 5, rb = blockIdx.x;
    const int n = (blockIdx.y * 32 + cgl) * 8;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.0f; q[e] = 0.0f; }
    if (n < N) {
        const int m1 = (rb + 1) * R < M ? (rb + 1) * R : M;
        for (int m = rb * R + rl; m < m1; m += 8) {
            float x[8];
            if constexpr (ET == 2) {
                const float4* src = reinterpret_cast<const float4*>(out + ((int64_t)m * ldc + n) * 4);
                const float4 lo = src[0], hi = src[1];
                x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
            } else {
                unpack_h8<ET == 1>(*reinterpret_cast<const uint4*>(out + ((int64_t)m * ldc + n) * 2), x);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += x[e]; q[e] = fmaf(x[e], x[e], q[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cgl][e] = make_float2(s[e], q[e]);
    __syncthreads();
    // thread (cgl, e = rl): the eight row lanes of one channel, in lane order
    if (n < N) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float2 v = red[r][cgl][rl]; a += v.x; b += v.y; }
        part[(int64_t)rb * N + n + rl] = make_float2(a, b);
    }
}

struct TileCfg { int bm, bn, threads, stages, halo, dxr, fx; };   // fx: MF_FX_* bits of the instantiation (gemm_conv_kernel.h)   // halo: rows of output pixels per tile of conv3x3_halo_kernel (0 = implicit GEMM)
// keep in sync with launch_tile()
const TileCfg kTiles[] = {
    {128, 128, 256, 2},  // 1
    {128, 64, 256, 2},   // 2
    {64, 64, 256, 2},    // 3
    {256, 64, 256, 2},   // 4
    {256, 128, 512, 2},  // 5
    {64, 128, 256, 2},   // 6
    {128, 128, 256, 3},  // 7   3-stage ring variants (tiles t+1, t+2 in flight)
    {128, 64, 256, 3},   // 8
    {64, 64, 256, 3},    // 9
    {256, 64, 256, 3},   // 10
    {256, 128, 512, 3},  // 11
    {64, 128, 256, 3},   // 12
    {192, 128, 256, 2},  // 13  96x64 per wave, 80 KB: still two blocks per CU
    {128, 160, 256, 2},  // 14  4x1 waves, 32x160 per wave: exact fit for N = 320 / 640 / 1280
    {128, 192, 256, 2},  // 15  64x96 per wave
    {128, 160, 256, 4, 8},  // 16  conv3x3_halo_kernel: 8x16 output pixels x 160 channels, 4x1 waves
    {128, 128, 256, 4, 8},  // 17  conv3x3_halo_kernel: 8x16 x 128, 2x2 waves
    {128, 128, 256, 6, 8},  // 18  same with a 6-deep weight ring
    {256, 160, 512, 3, 16}, // 19  conv3x3_pingpong_kernel: 16x16 pixels x 160 channels, 8 waves, one block per CU
    {128, 160, 256, 2, 0, 1},  // 20  gemm_conv_kernel with dx-tap reuse of the A window (3x3 / stride 1 convs), 4x1 waves
    {128, 128, 256, 2, 0, 1},  // 21  same, 2x2 waves
    {64, 128, 256, 2, 0, 1},   // 22
    {128, 64, 256, 2, 0, 1},   // 23
    {192, 128, 256, 2, 0, 1},  // 24
    // 25-30: the 16x16x32 MFMA form (bf16 only) of tiles 1, 14, 20, 21, 6, 2
    {128, 128, 256, 2},        // 25
    {128, 160, 256, 2},        // 26
    {128, 160, 256, 2, 0, 1},  // 27
    {128, 128, 256, 2, 0, 1},  // 28
    {64, 128, 256, 2},         // 29
    {128, 64, 256, 2},         // 30
    // 31-36: deeper LDS rings for the tiles of small-M / short-K calls, where a block walks its K loop at the pace of one
    // L2 -> LDS round trip (~1 us) per tile in flight: more tiles in flight per block instead of more blocks per CU
    {128, 128, 256, 4},        // 31
    {128, 64, 256, 4},         // 32
    {64, 128, 256, 4},         // 33
    {64, 64, 256, 4},          // 34
    {64, 128, 256, 6},         // 35
    {64, 64, 256, 6},          // 36
    // 37-38: warp-specialised dx-reuse conv (bf16): 4 compute waves + 4 staging waves per block, 3-deep W ring, one block
    // per CU.  `threads` = the staging threads (the A window is BM + threads / 8 rows)
    {256, 160, 256, 2, 0, 1},  // 37  8x1 compute waves of 32x160 (+ 4 staging): three waves per SIMD
    {128, 160, 256, 2, 0, 1},  // 38  4x1 compute waves of 32x160 (+ 4 staging): for grids of <= 256 tiles of 128 rows
    {256, 160, 256, 2, 0, 1},  // 39  = 37 on 16x16x32 MFMAs
    {128, 160, 256, 2, 0, 1},  // 40  = 38 on 16x16x32 MFMAs
    // 41-46: the warp-specialised form of the plain ring (any call of the implicit-GEMM kernel: 1x1, strided, upsampled)
    {128, 160, 256, 3},        // 41  4 + 4 waves
    {256, 160, 256, 3},        // 42  8 + 4 waves, 16x16x32 MFMAs
    {128, 160, 256, 3},        // 43  = 41 on 16x16x32 MFMAs
    {128, 128, 256, 3},        // 44  4 (2x2) + 4 waves
    {256, 128, 256, 3},        // 45  8 (4x2) + 4 waves
    {256, 128, 256, 3},        // 46  = 45 on 16x16x32 MFMAs
    {128, 160, 256, 2, 0, 1},  // 47  = 38 / 40 with EIGHT compute waves (4x2 of 32x80, 16x16x32 MFMAs only) + 4 staging
    {128, 160, 256, 3},        // 48  = 41 / 43 with eight compute waves of 32x80 + 4 staging
    // 49-52 (round 5): 64x80 wave tiles on 16x16x32 MFMAs — 9 fragment reads per 20 MFMAs instead of 12 (32x160) or 14 (32x80)
    {256, 160, 256, 2, 0, 1},  // 49  dx-reuse conv, 4x2 compute waves of 64x80 + 4 staging (the successor of 37 / 39)
    {256, 160, 256, 3},        // 50  plain ring, 4x2 compute waves of 64x80 + 4 staging (the successor of 42)
    {128, 160, 256, 2, 0, 1},  // 51  dx-reuse conv, 2x2 compute waves of 64x80 + 4 staging
    {128, 160, 256, 3},        // 52  plain ring, 2x2 compute waves of 64x80 + 4 staging
    // 53-62 (round 5): the round-5 loop forms (MF_FX_ALL: cross-tile fragment pipeline, staged epilogue rows, early A window) of
    // 47, 48, 40, 43, 44, 46, 49, 50, 51, 52.  Separate numbers because the forms cost 20-80 VGPRs (a wave per SIMD on most tiles):
    // faster alone, not always faster beside the other stream's blocks — the step tuner decides per call site
    {128, 160, 256, 2, 0, 1, 7},  // 53  = 47
    {128, 160, 256, 3, 0, 0, 7},  // 54  = 48
    {128, 160, 256, 2, 0, 1, 7},  // 55  = 40
    {128, 160, 256, 3, 0, 0, 7},  // 56  = 43
    {128, 128, 256, 3, 0, 0, 7},  // 57  = 44
    {256, 128, 256, 3, 0, 0, 7},  // 58  = 46
    {256, 160, 256, 2, 0, 1, 7},  // 59  = 49
    {256, 160, 256, 3, 0, 0, 7},  // 60  = 50
    {128, 160, 256, 2, 0, 1, 7},  // 61  = 51
    {128, 160, 256, 3, 0, 0, 7},  // 62  = 52
    // 63-66 (round 5): FOUR compute waves of 64x160 / 64x128 (32x32x16 MFMAs, 160 / 128 accumulators per lane) + four staging waves,
    // two waves per SIMD at 256 registers, with the cross-tile pipeline at k16 granularity (MF_FX_XQ): the wave tile whose LDS reads
    // fit under its MFMAs (gemm_conv_kernel.h, XQ)
    {256, 160, 256, 2, 0, 1, 14},  // 63  dx-reuse conv, 4x1 compute waves of 64x160
    {256, 160, 256, 3, 0, 0, 14},  // 64  plain ring, 4x1 compute waves of 64x160
    {256, 128, 256, 2, 0, 1, 14},  // 65  dx-reuse conv, 4x1 compute waves of 64x128
    {256, 128, 256, 3, 0, 0, 14},  // 66  plain ring, 4x1 compute waves of 64x128
    // 67-68 (round 5): the two-blocks-per-CU forms of 14 / 20 with 2x2 waves of 64x80 on 16x16x32 MFMAs (9 fragment reads per 20 MFMAs
    // where the 4x1 waves of 32x160 read 12): no staging waves, two-stage ring, 74 KB
    {128, 160, 256, 2},            // 67
    {128, 160, 256, 2, 0, 1},      // 68
    // 69 (round 5): the persistent short-K GEMM of gemm_nloop.hip — 64 rows of A per block, a range of 160-column output tiles per
    // block, the epilogue of tile j under the main loop of tile j + 1 (4 compute + 4 staging + 4 epilogue waves)
    {64, 160, 256, 3},             // 69
    // 70 (round 6): gemm_pers.hip — 128 rows of A per block, a range of 160-column output tiles, EIGHT compute waves (two per SIMD) of
    // 32 x 80 + four staging + four epilogue waves on a two-deep ring beside a whole fp32 slab: the epilogue of tile j under tile j + 1
    {128, 160, 256, 2},            // 70
    // (round 3: FOUR-deep rings of 48 / 41 — 147 KB, three K tiles in flight — were built, parity-tested and offered to the tuner
    // over the whole step: picked for none of 100 shapes, gpurun_out/r03e/tune_user.json; removed again)
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
inline bool tile_ws(int tile) { return tile >= 37 && tile <= 66; }                                 // compute waves + four staging waves
constexpr int kNloopTile = 69;
constexpr int kPersTile = 70;
inline bool tile_ws_ring(int tile) { return tile_ws(tile) && kTiles[tile - 1].dxr == 0; }           // ... of the plain ring (any call)

// tiles whose kernels have an in-launch split-K combine (keep in sync with the launch_skf cases below)
bool tile_has_skf(int tile, int dtype, int w_split, bool a_f32) {
    if (a_f32) return false;
    const bool base = tile == 1 || tile == 2 || tile == 3 || tile == 6;
    if (dtype == MF_BF16) return base || tile == 41 || tile == 43 || tile == 44 || tile == 48;   // (45 / 46, 256 x 128: the tail spills there)
    if (dtype == MF_F16X3) return base || (w_split && (tile == 41 || tile == 44));
    return false;
}


inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Heuristic tile choice (mf_gemm_desc.tile overrides it; the Python host autotunes per shape).
int pick_tile(int M, int N, int nz, int splitk, int max_tile = kNumTiles, bool split = false) {
    int best = 1;
    double best_cost = 1e300;
    for (int t = 1; t <= max_tile; ++t) {
        const TileCfg& c = kTiles[t - 1];
        if (split && t != 1 && t != 2 && t != 3 && t != 6) continue;     // instantiated for every split variant
        const double tiles = (double)cdiv(M, c.bm) * cdiv(N, c.bn) * nz * (splitk > 1 ? splitk : 1);
        if (c.stages != 2 || c.halo || c.dxr) continue;                            // the ring / halo variants are picked by the host autotuner
        const int bpc = (2 * (c.bm + c.bn) * 128 <= 80 * 1024) ? 2 : 1;   // blocks per CU that fit in LDS
        const double rounds = (double)(int64_t)((tiles + 256.0 * bpc - 1) / (256.0 * bpc));
        double per_cu = tiles / 256.0;
        if (per_cu < 1.0) per_cu = 1.0;
        if (per_cu > bpc) per_cu = bpc;
        const double inten = (double)c.bm * c.bn / (c.bm + c.bn);
        const double eff = inten / (inten + 24.0);
        const double cost = rounds * per_cu * c.bm * c.bn / eff;
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

}  // namespace
}  // namespace mfgemm
using namespace mfgemm;

unsigned* mf_ovf_flag_gemm() {
    unsigned* p = nullptr;
    (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_split_ovf_gemm));
    return p;
}

#ifdef MF_STAMPS
extern "C" int mf_debug_set_stamps(void* ptr) {
    g_stamps_host = (unsigned long long*)ptr;      // handed to every later launch in GemmArgs::stamps
    return 0;
}
#endif

extern "C" int mf_gemm_num_tiles(void) { return kNumTiles; }
extern "C" int mf_gemm_tile_table_version(void) { return 1; }
extern "C" int mf_gemm_tile_shape(int tile, int* bm, int* bn) {
    if (tile < 1 || tile > kNumTiles) return MF_EINVAL;
    *bm = kTiles[tile - 1].bm;
    *bn = kTiles[tile - 1].bn;
    return MF_OK;
}

extern "C" int mf_gemm_conv(const mf_gemm_desc* d, void* stream) {
    MF_CHECK_ARG(d != nullptr, "mf_gemm_conv: null descriptor");
    MF_CHECK_ARG(d->dtype == MF_F32 || d->dtype == MF_BF16 || d->dtype == MF_F16X3 || d->dtype == MF_BF16X3 || d->dtype == MF_FP8 ||
                     d->dtype == MF_BF16X1 || d->dtype == MF_F16,
                 "mf_gemm_conv: bad dtype %d", d->dtype);
    const bool split = d->dtype == MF_F16X3 || d->dtype == MF_BF16X3 || d->dtype == MF_BF16X1;
    MF_CHECK_ARG(d->dtype != MF_BF16X1 || d->w_split == 0, "mf_gemm_conv: MF_BF16X1 takes the raw fp32 weight");
    MF_CHECK_ARG((d->dtype == MF_FP8) == (d->a_dtype == MF_FP8), "mf_gemm_conv: fp8 compute takes fp8 activations (and only those)");
    MF_CHECK_ARG(!split || d->a_dtype == MF_F32, "mf_gemm_conv: the split codes take fp32 activations");
    MF_CHECK_ARG(d->w_split == 0 || (d->w_split == 1 && split && d->ldw % 32 == 0),
                 "mf_gemm_conv: w_split needs a split compute code and rows padded to a multiple of 32 k");
    const int es = mf_dtype_size(d->dtype);
    const int vec = 16 / es;
    MF_CHECK_ARG(d->a0 && d->w && d->out, "mf_gemm_conv: null a0/w/out");
    MF_CHECK_ARG(d->a_dtype == d->dtype || (d->a_dtype == MF_F32 && d->dtype != MF_F32),
                 "mf_gemm_conv: a_dtype %d incompatible with compute dtype %d", d->a_dtype, d->dtype);
    // fp32 activations are converted on load only by the bf16 kernels (a_f32 below): the fp16 instantiations have no converting
    // path and would read the fp32 bytes as fp16 pairs
    MF_CHECK_ARG(d->dtype != MF_F16 || d->a_dtype == MF_F16, "mf_gemm_conv: MF_F16 compute takes fp16 activations (a_dtype %d): cast first", d->a_dtype);
    MF_CHECK_ARG(d->c0 > 0 && d->c1 >= 0 && (d->a1 != nullptr) == (d->c1 > 0), "mf_gemm_conv: bad c0/c1/a1");
    MF_CHECK_ARG(d->c0 % vec == 0 && d->c1 % vec == 0, "mf_gemm_conv: channels (%d,%d) must be multiples of %d",
                 d->c0, d->c1, vec);
    MF_CHECK_ARG(d->lda0 % vec == 0 && d->lda1 % vec == 0 && d->ldw % vec == 0,
                 "mf_gemm_conv: lda/ldw must be multiples of %d elements", vec);
    MF_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->batch >= 1 && d->h_in >= 1 && d->w_in >= 1 &&
                     d->h_out >= 1 && d->w_out >= 1 && d->n >= 1,
                 "mf_gemm_conv: bad geometry");
    MF_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "mf_gemm_conv: upsample must be 0/1");
    MF_CHECK_ARG(d->nz >= 1 && d->zdiv >= 1, "mf_gemm_conv: nz/zdiv must be >= 1");
    if (!mf_aligned16(d->a0) || !mf_aligned16(d->w) || (d->a1 && !mf_aligned16(d->a1))) {
        mf_set_error("mf_gemm_conv: a0/a1/w must be 16-byte aligned");
        return MF_EALIGN;
    }
    MF_CHECK_ARG((d->a_zs_o % vec) == 0 && (d->a_zs_i % vec) == 0 && (d->w_zs_o % vec) == 0 && (d->w_zs_i % vec) == 0,
                 "mf_gemm_conv: batch strides must be multiples of %d elements", vec);

    GemmArgs a{};
    a.howo_sh = a.wo_sh = a.ho_sh = -1;       // (set with the geometry below; the resident-patch tiles return before that)
    {
        static unsigned* ovf = mf_ovf_flag_gemm();
        a.ovf = ovf;
    }
#ifdef MF_STAMPS
    a.stamps = g_stamps_host;
#endif
    const bool a_f32 = (d->a_dtype == MF_F32 && d->dtype == MF_BF16);
    const int aes = a_f32 ? 4 : es;
    a.a0 = (const char*)d->a0; a.a1 = (const char*)d->a1;
    a.C0 = d->c0; a.Ctot = d->c0 + d->c1;
    MF_CHECK_ARG(d->lda0 * aes < (1ll << 31) && d->lda1 * aes < (1ll << 31), "mf_gemm_conv: pixel stride too large");
    MF_CHECK_ARG((int64_t)d->batch * d->h_in * d->w_in < (1ll << 31), "mf_gemm_conv: too many input pixels");
    a.ld0b = (int)(d->lda0 * aes); a.ld1b = (int)(d->lda1 * aes);
    a.Hin = d->h_in; a.Win = d->w_in; a.Ho = d->h_out; a.Wo = d->w_out; a.HoWo = d->h_out * d->w_out;
    a.KW = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.ups = d->upsample;
    {
        auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (v > 0 && (1 << s) == v) ? s : -1; };
        a.howo_sh = lg(a.HoWo); a.wo_sh = lg(a.Wo); a.ho_sh = lg(a.Ho);
    }
    a.w = (const char*)d->w; a.ldw = d->ldw;
    const int64_t M64 = (int64_t)d->batch * d->h_out * d->w_out;
    MF_CHECK_ARG(M64 < (1ll << 31), "mf_gemm_conv: M too large");
    a.M = (int)M64; a.N = d->n; a.K = d->kh * d->kw * a.Ctot;
    MF_CHECK_ARG(d->ldw >= a.K, "mf_gemm_conv: ldw %lld < K %d", (long long)d->ldw, a.K);
    a.zdiv = d->zdiv; a.nz = d->nz;
    a.a_zs_o = d->a_zs_o; a.a_zs_i = d->a_zs_i; a.w_zs_o = d->w_zs_o; a.w_zs_i = d->w_zs_i;
    a.o_zs_o = d->o_zs_o; a.o_zs_i = d->o_zs_i;
    a.bias = d->bias; a.bias_mode = d->bias_mode; a.temb = d->temb; a.ld_temb = d->ld_temb;
    a.rs = d->a_scale; a.cs = d->w_scale; a.rs_zs = d->a_scale_zs; a.cs_zs = d->w_scale_zs;
    a.res0 = (const char*)d->res0; a.res0_dt = d->res0_dtype; a.ld_res0 = d->ld_res0;
    a.res1 = (const char*)d->res1; a.res1_dt = d->res1_dtype; a.ld_res1 = d->ld_res1;
    a.res1_rows = (d->res1 && d->res1_rows > 0 && d->res1_rows < a.M) ? d->res1_rows : 0;
    MF_CHECK_ARG(a.res1_rows == 0 || a.M % a.res1_rows == 0, "mf_gemm_conv: res1_rows=%d must divide M=%d", d->res1_rows, a.M);
    a.alpha = d->alpha; a.act = d->act;
    a.out = (char*)d->out; a.out_dt = d->out_dtype; a.ldc = d->ldc;
    {   // the 16-bit flavour is a compile-time property of the kernels (DT == MF_F16): fp16 results / residuals only in the fp16 mode
        const int r0 = d->res0 ? d->res0_dtype : -1, r1 = d->res1 ? d->res1_dtype : -1;
        MF_CHECK_ARG(d->dtype == MF_F16 ? !mf_any_bf16(d->out_dtype, r0, r1) : !mf_any_f16(d->out_dtype, r0, r1),
                     "mf_gemm_conv: fp16 outputs / residuals go with dtype MF_F16 and bf16 ones with the other modes (dtype %d, out %d, res %d %d)",
                     d->dtype, d->out_dtype, r0, r1);
    }
    a.ln_cs = d->ln_colsum; a.ln_eps = d->ln_eps;
    a.vt_out = (char*)d->vt_out; a.vt_n0 = d->vt_n0; a.vt_tokens = d->vt_tokens; a.vt_ld = d->vt_ld;
    if (d->ln_colsum || d->vt_out) {
        // served by the warp-specialised ring tiles only (41-46, 48): the staging waves gather the row statistics
        MF_CHECK_ARG(mf_is16(d->dtype) && d->a_dtype == d->dtype && d->kh == 1 && d->kw == 1 && d->c1 == 0 && d->nz == 1 &&
                         (d->splitk == 0 || d->splitk == 1) && d->a_scale == nullptr && d->w_scale == nullptr,
                     "mf_gemm_conv: ln_colsum / vt_out need a plain bf16 / fp16 1x1 GEMM (one A segment, no batching, no split-K, no scales)");
        MF_CHECK_ARG(d->tile == 0 || tile_ws_ring(d->tile) || (d->tile == kNloopTile && !d->vt_out) || d->tile == kPersTile,
                     "mf_gemm_conv: tile %d does not serve ln_colsum / vt_out (the warp-specialised ring tiles 41-46, 48, 50, 52, 54, 56-58, 60, 62 do)", d->tile);
        MF_CHECK_ARG(!d->ln_colsum || (mf_aligned16(d->ln_colsum) && d->n % 8 == 0 && d->ln_eps > 0.0f), "mf_gemm_conv: ln_colsum must be 16-byte aligned, n %% 8 == 0, ln_eps > 0");
        if (d->vt_out) {
            MF_CHECK_ARG(mf_is16(d->out_dtype) && mf_aligned16(d->vt_out) && d->vt_tokens > 0 && d->vt_tokens % 8 == 0 && a.M % d->vt_tokens == 0 &&
                             d->vt_ld % 8 == 0 && d->vt_ld >= d->vt_tokens && d->vt_n0 > 0 && d->vt_n0 < d->n && d->vt_n0 % 160 == 0 &&
                             d->vt_n0 % 128 == 0 && d->res0 == nullptr && d->res1 == nullptr && d->temb == nullptr && d->act == MF_ACT_NONE &&
                             d->bias_mode == 0,
                         "mf_gemm_conv: vt_out needs bf16 output, tokens %% 8 == 0 dividing M, vt_n0 a multiple of 640 inside (0, n), no residual / temb / activation");
        }
    }
    MF_CHECK_ARG(d->act != MF_ACT_GEGLU4 || (d->n % 8 == 0 && d->ldc % 4 == 0 && d->res0 == nullptr && d->res1 == nullptr &&
                                             (d->splitk == 0 || d->splitk == 1)),
                 "mf_gemm_conv: the GEGLU epilogue needs n %% 8 == 0, ldc %% 4 == 0, no residuals and no forced split-K");
    MF_CHECK_ARG(d->nz == 1 || (d->res0 == nullptr && d->res1 == nullptr && d->temb == nullptr),
                 "mf_gemm_conv: residual/temb epilogue is not defined for batched (nz > 1) calls");
    // GroupNorm partial sums of the output (mf_gemm_desc.gn_part): R rows per block, in the epilogue or by gn_colsum_kernel
    const int gn_hw = d->h_out * d->w_out;
    const int gn_r_fallback = gn_hw % 128 == 0 ? 128 : (gn_hw % 64 == 0 ? 64 : 32);
    if (d->gn_part) {
        MF_CHECK_ARG(d->gn_part_rows != nullptr && d->n % 8 == 0 && d->nz == 1 && gn_hw % 32 == 0 && d->act != MF_ACT_GEGLU4 && !d->vt_out &&
                         d->ldc >= d->n && d->o_zs_o == 0 && d->o_zs_i == 0 && mf_aligned16(d->gn_part) &&
                         d->gn_part_floats >= 2ll * d->n * (M64 / 32),
                     "mf_gemm_conv: gn_part needs gn_part_rows, n %% 8 == 0, nz == 1, h_out * w_out %% 32 == 0, no GEGLU / vt_out, and "
                     "2 * n * (M / 32) = %lld floats (got %lld)", (long long)(2ll * d->n * (M64 / 32)), (long long)d->gn_part_floats);
    }
    if (d->gn_grouped) *d->gn_grouped = 0;
    auto gn_fallback = [&](hipStream_t st) -> int {     // after the launch(es) that wrote `out`
        if ((mf_is16(d->out_dtype) || d->out_dtype == MF_F32) && d->ldc % 8 == 0 && mf_aligned16(d->out)) {
            const dim3 g8((unsigned)(a.M / gn_r_fallback), (unsigned)cdiv(a.N, 256));
            if (d->out_dtype == MF_F16) hipLaunchKernelGGL(gn_colsum8_kernel<1>, g8, dim3(256), 0, st, (const char*)d->out, d->ldc, a.M, a.N, gn_r_fallback, (float2*)d->gn_part);
            else if (d->out_dtype == MF_F32) hipLaunchKernelGGL(gn_colsum8_kernel<2>, g8, dim3(256), 0, st, (const char*)d->out, d->ldc, a.M, a.N, gn_r_fallback, (float2*)d->gn_part);
            else hipLaunchKernelGGL(gn_colsum8_kernel<0>, g8, dim3(256), 0, st, (const char*)d->out, d->ldc, a.M, a.N, gn_r_fallback, (float2*)d->gn_part);
        } else
        hipLaunchKernelGGL(gn_colsum_kernel, dim3((unsigned)(a.M / gn_r_fallback), (unsigned)cdiv(a.N, 256)), dim3(256), 0, st,
                           (const char*)d->out, d->out_dtype, d->ldc, a.M, a.N, gn_r_fallback, (float2*)d->gn_part);
        *d->gn_part_rows = gn_r_fallback;
        MF_CHECK_LAUNCH("mf_gemm_conv(gn_part column sums)");
        return MF_OK;
    };
    // the 8-wide vector epilogue needs 8-channel-aligned rows and 16-byte aligned bases everywhere
    a.vec_ok = (d->n % 8 == 0) && (d->ldc % (d->act == MF_ACT_GEGLU4 ? 4 : 8) == 0) && mf_aligned16(d->out) && (d->o_zs_o % 8 == 0) &&
               (d->o_zs_i % 8 == 0) &&
               (!d->bias || d->bias_mode == 1 || mf_aligned16(d->bias)) &&
               (!d->w_scale || (mf_aligned16(d->w_scale) && d->w_scale_zs % 4 == 0)) &&
               (!d->temb || (mf_aligned16(d->temb) && d->ld_temb % 4 == 0)) &&
               (!d->res0 || (mf_aligned16(d->res0) && d->ld_res0 % 8 == 0)) &&
               (!d->res1 || (mf_aligned16(d->res1) && d->ld_res1 % 8 == 0));

    int tile = d->tile;
    if (tile <= 0 || tile > kNumTiles) tile = (d->ln_colsum || d->vt_out) ? 48 : pick_tile(a.M, a.N, a.nz, d->splitk, a_f32 ? 6 : 24, split || d->dtype == MF_FP8);
    if (a_f32) {
        // the converting path only exists for the 2-stage tiles 1..6; 7..12 are the same shapes with a deeper ring.
        // Resolve the EFFECTIVE tile before the grid is derived from it (a 192x128 grid on a 128x128 kernel would leave
        // rows unwritten); anything else does not apply and is refused, never rerouted.
        if (tile >= 7 && tile <= 12) tile -= 6;
        MF_CHECK_ARG(tile <= 6, "mf_gemm_conv: tile %d does not apply to fp32 activations with bf16 compute (tiles 1-12 do)", tile);
    }
    MF_CHECK_ARG(tile < 25 || tile > 30 || (mf_is16(d->dtype) && !a_f32), "mf_gemm_conv: tile %d (16x16x32 MFMA form) does not apply: bf16 / fp16 only", tile);
    const bool split_ws = d->dtype == MF_F16X3 && d->w_split == 1 && (tile == 37 || tile == 38 || tile == 41 || tile == 44);
    MF_CHECK_ARG(tile < 31 || split_ws || (!a_f32 && !split && d->dtype != MF_FP8), "mf_gemm_conv: tile %d (deep ring) does not apply to this precision", tile);
    MF_CHECK_ARG(tile < 37 || mf_is16(d->dtype) || split_ws,
                 "mf_gemm_conv: tile %d (warp-specialised) does not apply: bf16, or f16x3 with a pre-split weight on tiles 37 / 38 / 41 / 44", tile);
    const TileCfg& tc = kTiles[tile - 1];
    if (tc.halo) {
        // conv3x3_halo_kernel: bf16, 3x3 / stride 1 / pad 1, whole TH x 16 tiles, 32-channel chunks
        const int64_t npix = (int64_t)d->batch * d->h_in * d->w_in;
        const int64_t ext_a = (npix - 1) * (int64_t)(a.ld0b > a.ld1b ? a.ld0b : a.ld1b) + (int64_t)a.Ctot * 2;
        const int64_t ext_w = ((int64_t)(a.N - 1) * a.ldw + a.K) * 2;
        const bool ok = d->dtype == MF_BF16 && !a_f32 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 &&
                        d->pad_l == 1 && !d->upsample && d->h_out == d->h_in && d->w_out == d->w_in && d->nz == 1 &&
                        d->h_in % tc.halo == 0 && d->w_in % 16 == 0 && a.C0 % (tc.halo == 16 ? 64 : 32) == 0 &&
                        a.Ctot % (tc.halo == 16 ? 64 : 32) == 0 &&
                        ext_a < (1ll << 31) - (1 << 20) && ext_w < (1ll << 31) - (1 << 20) && d->act != MF_ACT_GEGLU4 &&
                        d->o_zs_o == 0 && d->o_zs_i == 0;
        MF_CHECK_ARG(ok, "mf_gemm_conv: tile %d (3x3 halo kernel) does not apply to this call", tile);
        a.nkt = a.Ctot / (tc.halo == 16 ? 64 : 32);
        const int64_t tiles = (int64_t)(a.M / tc.bm) * cdiv(a.N, tc.bn);
        int sk = d->splitk;
        if (sk == 0) {
            sk = 1;
            if (tiles < 384 && d->ws != nullptr) {
                sk = (int)((512 + tiles - 1) / tiles);
                if (sk > a.nkt / 2) sk = a.nkt / 2;
                if ((int64_t)sk * a.M * a.N > d->ws_floats) sk = (int)(d->ws_floats / ((int64_t)a.M * a.N));
            }
        }
        if (sk < 1) sk = 1;
        if (sk > a.nkt) sk = a.nkt;
        a.kt_per_split = cdiv(a.nkt, sk);
        a.splitk = cdiv(a.nkt, a.kt_per_split);
        a.ws = d->ws;
        MF_CHECK_ARG(a.splitk == 1 || (a.ws != nullptr && (int64_t)a.splitk * a.M * a.N <= d->ws_floats),
                     "mf_gemm_conv: split-K=%d needs a workspace of %lld floats", a.splitk, (long long)a.splitk * a.M * a.N);
        a.tiles_n = cdiv(a.N, tc.bn);
        a.nblk = (int)tiles;
        dim3 hgrid((unsigned)tiles, 1, (unsigned)a.splitk);
        hipStream_t hs = (hipStream_t)stream;
        MF_CHECK_ARG(launch_halo_family(tile, a, hgrid, hs), "mf_gemm_conv: tile %d is not a resident-patch tile", tile);
        MF_CHECK_LAUNCH("mf_gemm_conv(halo)");
        if (a.splitk > 1) {
            const int64_t total = (int64_t)a.M * a.N / (a.vec_ok ? 8 : 1);
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(d->dtype == MF_F16 ? splitk_reduce_kernel<true> : splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, hs, a);
            MF_CHECK_LAUNCH("mf_gemm_conv(split-K reduce)");
        }
        if (d->gn_part) return gn_fallback(hs);
        return MF_OK;
    }
    const int bk = 128 / es;
    a.nkt = cdiv(a.K, bk);
    if (tc.dxr) {
        const int64_t npix = (int64_t)d->batch * d->h_in * d->w_in;
        const int64_t ext_a = (npix - 1) * (int64_t)(a.ld0b > a.ld1b ? a.ld0b : a.ld1b) + (int64_t)a.Ctot * aes;
        const int64_t ext_w = ((int64_t)(a.N - 1) * a.ldw + a.K) * es;
        const bool ok = !a_f32 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && !d->upsample &&
                        d->h_out == d->h_in && d->w_out == d->w_in && d->nz == 1 && a.C0 % bk == 0 && a.Ctot % bk == 0 &&
                        ext_a < (1ll << 31) - (1 << 20) && ext_w < (1ll << 31) - (1 << 20) &&
                        ((d->w_in <= tc.bm && tc.bm % d->w_in == 0 && (tc.bm / d->w_in) * (d->w_in + 2) <= tc.bm + tc.threads / 8) ||
                         (d->w_in > tc.bm && d->w_in % tc.bm == 0));
        MF_CHECK_ARG(ok, "mf_gemm_conv: tile %d (dx-tap reuse) does not apply to this call", tile);
    }
    const int64_t tiles_mn = (int64_t)cdiv(a.M, tc.bm) * cdiv(a.N, tc.bn) * a.nz;
    int splitk = d->splitk;
    if (splitk == 0) {
        // heuristic: fill the 256 CUs when the output grid alone cannot, keeping >= 4 K-tiles per split
        splitk = 1;
        if (tiles_mn <= 192 && a.nkt >= 8 && d->ws != nullptr && d->act != MF_ACT_GEGLU4 && !d->ln_colsum && !d->vt_out && tile != kNloopTile && tile != kPersTile) {
            splitk = (int)((384 + tiles_mn - 1) / tiles_mn);
            if (splitk > a.nkt / 4) splitk = a.nkt / 4;
            const int64_t per_split = (int64_t)a.nz * a.M * a.N;
            if ((int64_t)splitk * per_split > d->ws_floats) splitk = (int)(d->ws_floats / per_split);
            if (splitk < 1) splitk = 1;
        }
    }
    a.splitk = splitk > 1 ? splitk : 1;
    if (a.splitk > a.nkt) a.splitk = a.nkt;
    a.kt_per_split = cdiv(a.nkt, a.splitk);
    if (tc.dxr) a.kt_per_split = (a.kt_per_split + 2) / 3 * 3;       // whole (ky, chunk) groups of three taps
    a.splitk = cdiv(a.nkt, a.kt_per_split);   // no empty splits
    a.ws = d->ws;
    MF_CHECK_ARG(a.splitk == 1 || (a.ws != nullptr && (int64_t)a.splitk * a.nz * a.M * a.N <= d->ws_floats),
                 "mf_gemm_conv: split-K=%d needs a workspace of %lld floats", a.splitk,
                 (long long)a.splitk * a.nz * a.M * a.N);
    // in-launch combine: one ticket per output tile; a grid with more tiles than tickets keeps the reduce launch
    a.sk_tickets = (a.splitk > 1 && d->sk_tickets != nullptr && tiles_mn <= (int64_t)d->sk_ticket_cap &&
                    tile_has_skf(tile, d->dtype, d->w_split, a_f32)) ? (unsigned*)d->sk_tickets : nullptr;
    {   // fast staging: every K tile inside one (tap, segment) and every operand addressable with 31-bit offsets
        const int64_t npix = (int64_t)d->batch * d->h_in * d->w_in;
        const int64_t ext_a = (npix - 1) * (int64_t)(a.ld0b > a.ld1b ? a.ld0b : a.ld1b) + (int64_t)a.Ctot * aes;
        const int64_t ext_w = ((int64_t)(a.N - 1) * a.ldw + a.K) * es;
        a.fast = (a.C0 % bk == 0) && (a.Ctot % bk == 0) && ext_a < (1ll << 31) - (1 << 20) && ext_w < (1ll << 31) - (1 << 20);
    }
    MF_CHECK_ARG(d->act != MF_ACT_GEGLU4 || a.vec_ok, "mf_gemm_conv: GEGLU epilogue needs 16-byte aligned bias/out");
    a.pointwise = d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && !d->upsample && d->h_out == d->h_in &&
                  d->w_out == d->w_in;
    a.tiles_n = cdiv(a.N, tc.bn);
    a.tiles_m = cdiv(a.M, tc.bm);
    {   // division-free prologue: shifts for the power-of-two extents, a magic multiplier for the dx-reuse window pitch
        const int weff = a.Wo < tc.bm ? a.Wo : tc.bm;
        a.wfr_magic = (unsigned)((1u << 20) / (unsigned)(weff + 2)) + 1u;
    }
    {   // staged epilogue rows (GemmArgs::epb): warp-specialised tiles, the vector epilogue, per-column bias, no split-K slabs;
        // a time embedding needs Ho * Wo a power of two and the tile's images inside the rows the tile reserves
        static const bool off = getenv("MFHIP_NO_EPB") != nullptr;          // A/B switch
        const int nimg_max = (tc.fx & MF_FX_EPB) ? epb_nimg(tc.bm, tc.bn, 3, tc.dxr != 0, true) : 0;
        bool ok = !off && nimg_max > 0 && a.vec_ok && a.bias_mode == 0 && a.splitk == 1 && (a.bias || a.temb || a.ln_cs);
        a.epb_sh = 31;
        if (ok && a.temb) {
            const int hw = a.HoWo;
            int sh = 0;
            while ((1 << sh) < hw) ++sh;
            const int nimg = hw >= tc.bm ? (hw % tc.bm == 0 ? 1 : 0) : (tc.bm % hw == 0 ? tc.bm / hw : 0);
            ok = (1 << sh) == hw && nimg >= 1 && nimg <= nimg_max;
            a.epb_sh = sh;
        }
        a.epb = ok ? 1 : 0;
    }
    const int64_t nblk = (int64_t)a.tiles_m * a.tiles_n * a.splitk;
    MF_CHECK_ARG(nblk < (1ll << 31) && a.nz < 65536, "mf_gemm_conv: grid too large");
    a.nblk = (int)nblk;
    a.ord_mfast = 0;
    a.ord_pw = a.tiles_n;
    // wide 1x1 GEMMs (FF projections, N = 2560 ... 10240): panels of 8 column tiles, so the ~64 tiles an XCD runs at once
    // are 8 x 8 and the panel's W stays in its L2 (tools/bench_order.py: 8192 x 5120 x 640 101 -> 93 us, 2048 x 10240 x
    // 1280 85 -> 82 us, 16384 x 5120 x 640 197 -> 185 us, fetch 426 -> ~115 MB; the 3x3 convs do not care: their W
    // re-reads are served by the MALL, every order measured within 3 %)
    if (d->kh == 1 && d->kw == 1 && a.tiles_n > 8) a.ord_pw = 8;
    {   // A/B switch: MFHIP_ORD="mfast,pw" forces the tile order (pw 0 = no panels)
        static const char* ord = getenv("MFHIP_ORD");
        if (ord) {
            int mf = 0, pw = 0;
            if (sscanf(ord, "%d,%d", &mf, &pw) == 2) {
                a.ord_mfast = mf != 0;
                const int li = a.ord_mfast ? a.tiles_m : a.tiles_n;
                a.ord_pw = (pw > 0 && pw < li) ? pw : li;
            }
        }
    }
    { static const bool off = getenv("MFHIP_NO_RES_PRE") != nullptr; a.dbg_no_res_pre = off; }     // A/B switch
    { static const int e = getenv("MFHIP_DBG_EPI") ? atoi(getenv("MFHIP_DBG_EPI")) : 0; a.dbg_epi = e; }
    if (tile == kNloopTile) {
        const bool ok = mf_is16(d->dtype) && !a_f32 && a.pointwise && d->c1 == 0 && a.nz == 1 && a.splitk == 1 && a.M % 64 == 0 && a.N % 160 == 0 &&
                        a.K % 64 == 0 && a.nkt >= 2 && a.vec_ok && a.bias_mode == 0 && !a.temb && !a.vt_out && !a.rs && !a.cs && a.fast &&
                        (int64_t)a.M * a.ld0b < (1ll << 31) - (1 << 20);
        MF_CHECK_ARG(ok, "mf_gemm_conv: tile %d (persistent short-K GEMM) takes a 1x1 bf16 / fp16 call with M %% 64 == 0, N %% 160 == 0, K %% 64 == 0, "
                         "K >= 128, one A segment, per-column bias, no time embedding / scales / transposed columns / split-K", tile);
        // column ranges per row tile: enough blocks for the chip, as many output tiles per block as that leaves
        static const int forced = getenv("MFHIP_NLOOP_RANGES") ? atoi(getenv("MFHIP_NLOOP_RANGES")) : 0;      // developer sweep
        int ranges = 1;
        while (a.tiles_m * ranges < 256 && a.tiles_n % (ranges * 2) == 0) ranges *= 2;
        if (forced > 0 && a.tiles_n % forced == 0) ranges = forced;
        a.nloop = a.tiles_n / ranges;
        MF_CHECK_ARG(launch_nloop(d->dtype, a, (hipStream_t)stream), "mf_gemm_conv: tile %d is not instantiated for dtype %d", tile, d->dtype);
        MF_CHECK_LAUNCH("mf_gemm_conv(nloop)");
        if (d->gn_part) return gn_fallback((hipStream_t)stream);
        return MF_OK;
    }
    // statistics in the epilogue: the final values of a block's rows are in its LDS slabs; one image per block of rows
    const bool gn_fused = d->gn_part != nullptr && a.splitk == 1 && a.vec_ok && gn_hw % tc.bm == 0 && a.M % tc.bm == 0;
    a.gn_part = gn_fused ? (float2*)d->gn_part : nullptr;
    // per-group sums too: the consumer's groups are whole inside a tile's columns (and the LDS of the smallest tiles has room for
    // one more row of column totals: (WAVES_M + 1) * BN float pairs)
    bool gn_grouped = false;
    if (gn_fused && d->gn_groups > 0 && a.N % d->gn_groups == 0) {
        const int cpg = a.N / d->gn_groups;
        gn_grouped = tc.bn % cpg == 0 && tc.bm >= 64 && (int64_t)2 * (a.N + d->gn_groups) * (a.M / tc.bm) <= d->gn_part_floats;
        if (gn_grouped) { a.gn_grp = a.gn_part + (int64_t)a.N * (a.M / tc.bm); a.gn_cpg = cpg; }
    }
    if (d->gn_grouped) *d->gn_grouped = gn_grouped ? 1 : 0;
    if (tile == kPersTile) {
        const bool ok = mf_is16(d->dtype) && !a_f32 && a.pointwise && d->c1 == 0 && a.nz == 1 && a.splitk == 1 && a.M % 128 == 0 && a.N % 160 == 0 &&
                        a.K % 64 == 0 && a.nkt >= 2 && a.vec_ok && a.bias_mode == 0 && !a.temb && !a.rs && !a.cs && a.fast && !a.res1 && (!a.res0 || a.res0_dt != MF_F32) && a.out_dt != MF_F32 &&
                        (!a.vt_out || (a.vt_n0 % 160 == 0 && a.vt_tokens % 8 == 0 && a.act == MF_ACT_NONE && !a.res0)) &&
                        (int64_t)a.M * a.ld0b < (1ll << 31) - (1 << 20);
        MF_CHECK_ARG(ok, "mf_gemm_conv: tile %d (persistent 128-row GEMM) takes a 1x1 bf16 / fp16 call with M %% 128 == 0, N %% 160 == 0, K %% 64 == 0, "
                         "K >= 128, one A segment, 16-bit output (and residual), per-column bias, one residual at most, no time embedding / scales / transposed columns / split-K", tile);
        // column ranges per row tile: enough blocks for the chip (one block per CU: the kernel takes all of its LDS), as many
        // output tiles per block as that leaves — the epilogue of every tile but a block's last runs under the next main loop
        static const int forced = getenv("MFHIP_PERS_RANGES") ? atoi(getenv("MFHIP_PERS_RANGES")) : 0;        // developer sweep
        int ranges = 1;
        while (a.tiles_m * ranges < 256 && a.tiles_n % (ranges * 2) == 0) ranges *= 2;
        if (forced > 0 && a.tiles_n % forced == 0) ranges = forced;
        a.nloop = a.tiles_n / ranges;
        MF_CHECK_ARG(launch_pers(d->dtype, a, (hipStream_t)stream), "mf_gemm_conv: tile %d is not instantiated for dtype %d", tile, d->dtype);
        MF_CHECK_LAUNCH("mf_gemm_conv(pers)");
        if (d->gn_part) return gn_fallback((hipStream_t)stream);
        return MF_OK;
    }
    dim3 grid((unsigned)nblk, 1, (unsigned)a.nz);
    hipStream_t s = (hipStream_t)stream;
    bool launched;
    if (d->dtype == MF_FP8) launched = launch_fp8(tile, a, grid, s);
    else if (d->dtype == MF_BF16X1) launched = launch_bf16x1(tile, a, grid, s);
    else if (d->dtype == MF_F16X3) launched = (d->w_split && tile >= 37) ? launch_f16x3_ws(tile, a, grid, s) : launch_f16x3(tile, a, grid, s, d->w_split != 0);
    else if (d->dtype == MF_BF16X3) launched = launch_bf16x3(tile, a, grid, s, d->w_split != 0);
    else if (d->dtype == MF_F16) {
        launched = tile <= 6    ? launch_f16_a(tile, a, grid, s, false)
                   : tile <= 24 ? launch_f16_b(tile, a, grid, s)
                   : (tile <= 36 || !tile_ws(tile)) ? launch_f16_c(tile, a, grid, s)
                   : tc.dxr     ? launch_f16_ws_dx(tile, a, grid, s)
                                : launch_f16_ws_ring(tile, a, grid, s);
    } else if (d->dtype == MF_BF16) {
        launched = (a_f32 || tile <= 6) ? launch_bf16_a(tile, a, grid, s, a_f32)
                   : tile <= 24         ? launch_bf16_b(tile, a, grid, s)
                   : (tile <= 36 || !tile_ws(tile)) ? launch_bf16_c(tile, a, grid, s)
                   : tc.dxr             ? launch_bf16_ws_dx(tile, a, grid, s)
                                        : launch_bf16_ws_ring(tile, a, grid, s);
    } else {
        launched = tile <= 12 ? launch_f32_a(tile, a, grid, s) : launch_f32_b(tile, a, grid, s);
    }
    MF_CHECK_ARG(launched, "mf_gemm_conv: tile %d is not instantiated for dtype %d (w_split=%d)", tile, d->dtype, d->w_split);
    MF_CHECK_LAUNCH("mf_gemm_conv");
    if (d->deferred_splits) *d->deferred_splits = 0;
    if (a.splitk > 1 && a.sk_tickets == nullptr && d->defer_reduce && d->deferred_splits && a.vec_ok && a.nz == 1 && !d->res0 && !d->res1 &&
        d->act == MF_ACT_NONE && !d->a_scale && !d->w_scale && d->bias_mode == 0 && !d->ln_colsum && !d->vt_out && !d->gn_part && d->ldc == d->n) {
        // the consumer (mf_groupnorm, sk_ws) sums the slabs: no reduce launch, `out` stays unwritten
        *d->deferred_splits = a.splitk;
        return MF_OK;
    }
    if (a.splitk > 1 && a.sk_tickets == nullptr) {
        const int64_t total = (int64_t)a.M * a.N * a.nz / (a.vec_ok ? 8 : 1);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(d->dtype == MF_F16 ? splitk_reduce_kernel<true> : splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
        MF_CHECK_LAUNCH("mf_gemm_conv(split-K reduce)");
    }
    if (d->gn_part) {
        if (!gn_fused) return gn_fallback(s);
        *d->gn_part_rows = tc.bm;
    }
    return MF_OK;
}
