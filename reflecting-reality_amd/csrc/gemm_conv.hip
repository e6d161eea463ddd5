// Implicit-GEMM convolution / linear / strided-batched NT GEMM for gfx950 (MI355X) on MFMA.
//
// One kernel family serves every contraction of the MirrorFusion hot path (see include/mfhip.h,
// mf_gemm_conv): conv3x3 (stride 1/2, symmetric or asymmetric zero padding, optional fused
// nearest-2x upsample, optional two-tensor channel concat), conv1x1 / nn.Linear, and the batched
// QK^T / PV products of the unfused attention path.
//
// Design (CDNA4):
//   * M = batch*Ho*Wo pixels, N = Cout, K = kh*kw*Cin.  Activations NHWC, weights [N][K].
//   * A block stages BM x 128 B of A and BN x 128 B of W per K-tile into LDS (double buffered).
//     A K-tile is 128 bytes of K per row in BOTH precisions (64 bf16 / 32 f32), so the staging,
//     swizzle and fragment addressing are byte-identical for bf16 and f32; only the MFMA differs:
//       bf16: one v_mfma_f32_32x32x16_bf16 per 16-byte fragment (k = 8h + j),
//       f32 : four v_mfma_f32_32x32x2_f32 per 16-byte fragment (element e covers k = 4h + e of the
//             8-wide step; A and W use the same k permutation so the dot product is unchanged).
//   * LDS image [row][8 x 16 B], chunk index XOR ((row >> 1) & 7): conflict-free ds_read_b128 for
//     the 32x32 fragment pattern (16 distinct rows per lane group -> 16 distinct 16-B slots of the
//     256-B bank row); ds_write_b128 writes whole 128-B rows per 8 lanes (conflict-free).
//   * Register-staged global->LDS pipeline: tile t+2 is in flight in registers while tile t+1 sits
//     in LDS and tile t is being multiplied (write after the barrier, re-issue immediately).
//   * fp32 accumulate; fused epilogue: alpha*(acc + bias + temb) + res0 + res1, SiLU, dtype cast.
//   * split-K (deterministic fp32 slabs + reduce kernel that runs the same epilogue) for the
//     small-spatial layers (8x8 / 16x16 latents) that cannot fill 256 CUs otherwise.
#include "mf_common.h"

namespace {

struct GemmArgs {
    const char* a0; const char* a1;
    int C0, Ctot;
    int64_t lda0, lda1;
    int a_f32;
    int Hin, Win, Ho, Wo, HoWo, KW, stride, pad_t, pad_l, ups;
    const char* w; int64_t ldw;
    int M, N, K;
    int zdiv; int64_t a_zs_o, a_zs_i, w_zs_o, w_zs_i, o_zs_o, o_zs_i;
    int splitk, kt_per_split, nkt, nz;
    float* ws;
    const float* bias; int bias_mode;
    const float* temb; int64_t ld_temb;
    const char* res0; int res0_dt; int64_t ld_res0;
    const char* res1; int res1_dt; int64_t ld_res1;
    float alpha; int act;
    char* out; int out_dt; int64_t ldc;
    int tiles_n;
};

// Final epilogue for one output element (all branches are wave-uniform).
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int64_t zo, int m, int n, float v) {
    if (p.bias) v += p.bias_mode ? p.bias[m] : p.bias[n];
    if (p.temb) v += p.temb[(int64_t)(m / p.HoWo) * p.ld_temb + n];
    v *= p.alpha;
    if (p.res0) v += load_as_f32(p.res0, p.res0_dt, (int64_t)m * p.ld_res0 + n);
    if (p.res1) v += load_as_f32(p.res1, p.res1_dt, (int64_t)m * p.ld_res1 + n);
    if (p.act == MF_ACT_SILU) v = silu_precise(v);
    store_from_f32(p.out, p.out_dt, zo + (int64_t)m * p.ldc + n, v);
}

__device__ __forceinline__ uint4 ld16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

// 8 consecutive fp32 -> 8 bf16 (RNE) packed in 16 bytes
__device__ __forceinline__ uint4 ld8f_to_bf16(const char* p) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 16);
    uint4 r;
    r.x = (uint32_t)f32_to_bf16(lo.x) | ((uint32_t)f32_to_bf16(lo.y) << 16);
    r.y = (uint32_t)f32_to_bf16(lo.z) | ((uint32_t)f32_to_bf16(lo.w) << 16);
    r.z = (uint32_t)f32_to_bf16(hi.x) | ((uint32_t)f32_to_bf16(hi.y) << 16);
    r.w = (uint32_t)f32_to_bf16(hi.z) | ((uint32_t)f32_to_bf16(hi.w) << 16);
    return r;
}

template <int DT, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_conv_kernel(const GemmArgs p) {
    constexpr int NTHR = WAVES_M * WAVES_N * 64;
    constexpr int ES = (DT == MF_F32) ? 4 : 2;   // element size of the compute dtype
    constexpr int VEC = 16 / ES;                 // elements per 16-byte vector
    constexpr int BK = 128 / ES;                 // K elements per tile (128 bytes per row)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int RPP = NTHR / 8;                // rows staged per pass
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tile_m = blockIdx.x / p.tiles_n;
    const int tile_n = blockIdx.x - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z / p.splitk;
    const int ksplit = blockIdx.z - z * p.splitk;
    const int zq = z / p.zdiv, zr = z - zq * p.zdiv;

    const int aes = p.a_f32 ? 4 : ES;            // A storage element size
    const char* a0 = p.a0 + (zq * p.a_zs_o + zr * p.a_zs_i) * aes;
    const char* a1 = p.a1 ? p.a1 + (zq * p.a_zs_o + zr * p.a_zs_i) * aes : nullptr;
    const char* wbase = p.w + (zq * p.w_zs_o + zr * p.w_zs_i) * ES;

    // ---- per-thread staging coordinates -------------------------------------------------
    const int chunk = tid & 7;       // which 16-B chunk of the 128-B K-tile row
    const int lrow = tid >> 3;       // row within a staging pass
    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + lrow + i * RPP;
        if (m < p.M) {
            const int b = m / p.HoWo;
            const int r = m - b * p.HoWo;
            const int oy = r / p.Wo;
            const int ox = r - oy * p.Wo;
            a_pix[i] = b * p.Hin * p.Win;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
        } else {
            a_pix[i] = 0;
            a_iy0[i] = -(1 << 28);   // forces the bounds test to fail -> zero fill
            a_ix0[i] = 0;
        }
    }
    const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;

    const int kt_begin = ksplit * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > p.nkt) kt_end = p.nkt;
    const int nt = kt_end - kt_begin;

    int kk = kt_begin * BK + chunk * VEC;        // this thread's K element index in the current tile
    int c, ky, kx;
    {
        const int tap = kk / p.Ctot;
        c = kk - tap * p.Ctot;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
    }

    uint4 ra[A_IT], rb[B_IT];

    auto load_tile = [&]() {
        const bool kvalid = kk < p.K;
        const char* base; int64_t ld; int cc;
        if (c < p.C0) { base = a0; ld = p.lda0; cc = c; }
        else { base = a1; ld = p.lda1; cc = c - p.C0; }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            const bool ok = kvalid && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            if (ok) {
                const int64_t off = (int64_t)(a_pix[i] + (iy >> p.ups) * p.Win + (ix >> p.ups)) * ld + cc;
                if (DT == MF_BF16 && p.a_f32) ra[i] = ld8f_to_bf16(base + off * 4);
                else ra[i] = ld16(base + off * ES);
            } else {
                ra[i] = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + lrow + i * RPP;
            if (kvalid && n < p.N) rb[i] = ld16(wbase + ((int64_t)n * p.ldw + kk) * ES);
            else rb[i] = make_uint4(0, 0, 0, 0);
        }
        // advance to the next K tile
        kk += BK;
        c += BK;
        while (c >= p.Ctot) {
            c -= p.Ctot;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    auto store_tile = [&](int stage) {
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + BM * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int row = lrow + i * RPP;
            *reinterpret_cast<uint4*>(As + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = lrow + i * RPP;
            *reinterpret_cast<uint4*>(Bs + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = rb[i];
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int frow = lane & 31;              // fragment row (A: m, W: n) within a 32-row tile
    const int fh = lane >> 5;                // which half of the k-step this lane holds
    const int fkey = (frow >> 1) & 7;        // swizzle key (tile bases are multiples of 32 rows)

    auto compute = [&](int stage) {
        const char* As = smem + stage * STAGE_BYTES + (wm * WM + frow) * 128;
        const char* Bs = smem + stage * STAGE_BYTES + BM * 128 + (wn * WN + frow) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (((2 * ks + fh) ^ fkey) << 4);
            uint4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (DT == MF_BF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4_t av = __builtin_bit_cast(f32x4_t, fa[i]);
                        const f32x4_t bv = __builtin_bit_cast(f32x4_t, fb[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    // ---- main loop ------------------------------------------------------------------------
    if (nt > 0) {
        load_tile();
        store_tile(0);
        if (nt > 1) load_tile();
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            compute(t & 1);
            if (t + 1 < nt) {
                store_tile((t + 1) & 1);
                if (t + 2 < nt) load_tile();
            }
            __syncthreads();
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------
    const int ncol0 = n0 + wn * WN + (lane & 31);
    if (p.splitk > 1) {
        float* ws = p.ws + ((int64_t)ksplit * p.nz + z) * (int64_t)p.M * p.N;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (m < p.M) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int n = ncol0 + j * 32;
                        if (n < p.N) ws[(int64_t)m * p.N + n] = acc[i][j][e];
                    }
                }
            }
    } else {
        const int64_t zo = zq * p.o_zs_o + zr * p.o_zs_i;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (m < p.M) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int n = ncol0 + j * 32;
                        if (n < p.N) epilogue_store(p, zo, m, n, acc[i][j][e]);
                    }
                }
            }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int64_t mn = (int64_t)p.M * p.N;
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
        epilogue_store(p, zq * p.o_zs_o + zr * p.o_zs_i, m, n, v);
    }
}

struct TileCfg { int bm, bn, threads; };
// keep in sync with launch_tile()
const TileCfg kTiles[] = {
    {128, 128, 256},  // 1
    {128, 64, 256},   // 2
    {64, 64, 256},    // 3
    {256, 64, 256},   // 4
    {256, 128, 512},  // 5
    {64, 128, 256},   // 6
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int DT, int BM, int BN, int WMv, int WNv>
void launch_one(const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_kernel<DT, BM, BN, WMv, WNv>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_conv_kernel<DT, BM, BN, WMv, WNv>), grid, dim3(WMv * WNv * 64), smem, s, a);
}

template <int DT>
void launch_tile(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    switch (tile) {
        case 1: launch_one<DT, 128, 128, 2, 2>(a, grid, s); break;
        case 2: launch_one<DT, 128, 64, 2, 2>(a, grid, s); break;
        case 3: launch_one<DT, 64, 64, 2, 2>(a, grid, s); break;
        case 4: launch_one<DT, 256, 64, 4, 1>(a, grid, s); break;
        case 5: launch_one<DT, 256, 128, 4, 2>(a, grid, s); break;
        case 6: launch_one<DT, 64, 128, 2, 2>(a, grid, s); break;
        default: break;
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Heuristic tile choice: the largest tile whose padded work is close to the minimum and whose grid
// still covers the 256 CUs; smaller tiles otherwise.
int pick_tile(int M, int N, int nz, int splitk) {
    int best = 1;
    double best_cost = 1e300;
    for (int t = 1; t <= kNumTiles; ++t) {
        const TileCfg& c = kTiles[t - 1];
        const double tiles = (double)cdiv(M, c.bm) * cdiv(N, c.bn) * nz * (splitk > 1 ? splitk : 1);
        const int bpc = (2 * (c.bm + c.bn) * 128 <= 80 * 1024) ? 2 : 1;   // blocks per CU that fit in LDS
        const double rounds = (double)(int64_t)((tiles + 256.0 * bpc - 1) / (256.0 * bpc));
        double per_cu = tiles / 256.0;
        if (per_cu < 1.0) per_cu = 1.0;
        if (per_cu > bpc) per_cu = bpc;
        // efficiency prior ~ arithmetic intensity of the tile against the LDS/L2 feed
        const double inten = (double)c.bm * c.bn / (c.bm + c.bn);
        const double eff = inten / (inten + 24.0);
        const double cost = rounds * per_cu * c.bm * c.bn / eff;
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

}  // namespace

extern "C" int mf_gemm_num_tiles(void) { return kNumTiles; }
extern "C" int mf_gemm_tile_shape(int tile, int* bm, int* bn) {
    if (tile < 1 || tile > kNumTiles) return MF_EINVAL;
    *bm = kTiles[tile - 1].bm;
    *bn = kTiles[tile - 1].bn;
    return MF_OK;
}

extern "C" int mf_gemm_conv(const mf_gemm_desc* d, void* stream) {
    MF_CHECK_ARG(d != nullptr, "mf_gemm_conv: null descriptor");
    MF_CHECK_ARG(d->dtype == MF_F32 || d->dtype == MF_BF16, "mf_gemm_conv: bad dtype %d", d->dtype);
    const int es = mf_dtype_size(d->dtype);
    const int vec = 16 / es;
    MF_CHECK_ARG(d->a0 && d->w && d->out, "mf_gemm_conv: null a0/w/out");
    MF_CHECK_ARG(d->a_dtype == d->dtype || (d->a_dtype == MF_F32 && d->dtype == MF_BF16),
                 "mf_gemm_conv: a_dtype %d incompatible with compute dtype %d", d->a_dtype, d->dtype);
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
    a.a0 = (const char*)d->a0; a.a1 = (const char*)d->a1;
    a.C0 = d->c0; a.Ctot = d->c0 + d->c1;
    a.lda0 = d->lda0; a.lda1 = d->lda1;
    a.a_f32 = (d->a_dtype == MF_F32 && d->dtype == MF_BF16) ? 1 : 0;
    a.Hin = d->h_in; a.Win = d->w_in; a.Ho = d->h_out; a.Wo = d->w_out; a.HoWo = d->h_out * d->w_out;
    a.KW = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.ups = d->upsample;
    a.w = (const char*)d->w; a.ldw = d->ldw;
    const int64_t M64 = (int64_t)d->batch * d->h_out * d->w_out;
    MF_CHECK_ARG(M64 < (1ll << 31), "mf_gemm_conv: M too large");
    a.M = (int)M64; a.N = d->n; a.K = d->kh * d->kw * a.Ctot;
    MF_CHECK_ARG(d->ldw >= a.K, "mf_gemm_conv: ldw %lld < K %d", (long long)d->ldw, a.K);
    a.zdiv = d->zdiv; a.nz = d->nz;
    a.a_zs_o = d->a_zs_o; a.a_zs_i = d->a_zs_i; a.w_zs_o = d->w_zs_o; a.w_zs_i = d->w_zs_i;
    a.o_zs_o = d->o_zs_o; a.o_zs_i = d->o_zs_i;
    a.bias = d->bias; a.bias_mode = d->bias_mode; a.temb = d->temb; a.ld_temb = d->ld_temb;
    a.res0 = (const char*)d->res0; a.res0_dt = d->res0_dtype; a.ld_res0 = d->ld_res0;
    a.res1 = (const char*)d->res1; a.res1_dt = d->res1_dtype; a.ld_res1 = d->ld_res1;
    a.alpha = d->alpha; a.act = d->act;
    a.out = (char*)d->out; a.out_dt = d->out_dtype; a.ldc = d->ldc;
    MF_CHECK_ARG(d->nz == 1 || (d->res0 == nullptr && d->res1 == nullptr && d->temb == nullptr),
                 "mf_gemm_conv: residual/temb epilogue is not defined for batched (nz > 1) calls");

    int tile = d->tile;
    if (tile <= 0 || tile > kNumTiles) tile = pick_tile(a.M, a.N, a.nz, d->splitk);
    const TileCfg& tc = kTiles[tile - 1];
    const int bk = 128 / es;
    a.nkt = cdiv(a.K, bk);
    const int64_t tiles_mn = (int64_t)cdiv(a.M, tc.bm) * cdiv(a.N, tc.bn) * a.nz;
    int splitk = d->splitk;
    if (splitk == 0) {
        // heuristic: fill the 256 CUs when the output grid alone cannot, keeping >= 4 K-tiles per split
        splitk = 1;
        if (tiles_mn < 160 && a.nkt >= 8 && d->ws != nullptr) {
            splitk = (int)((256 + tiles_mn - 1) / tiles_mn);
            if (splitk > a.nkt / 4) splitk = a.nkt / 4;
            const int64_t per_split = (int64_t)a.nz * a.M * a.N;
            if ((int64_t)splitk * per_split > d->ws_floats) splitk = (int)(d->ws_floats / per_split);
            if (splitk < 1) splitk = 1;
        }
    }
    a.splitk = splitk > 1 ? splitk : 1;
    if (a.splitk > a.nkt) a.splitk = a.nkt;
    a.kt_per_split = cdiv(a.nkt, a.splitk);
    a.splitk = cdiv(a.nkt, a.kt_per_split);   // no empty splits
    a.ws = d->ws;
    MF_CHECK_ARG(a.splitk == 1 || (a.ws != nullptr && (int64_t)a.splitk * a.nz * a.M * a.N <= d->ws_floats),
                 "mf_gemm_conv: split-K=%d needs a workspace of %lld floats", a.splitk,
                 (long long)a.splitk * a.nz * a.M * a.N);
    a.tiles_n = cdiv(a.N, tc.bn);
    const int64_t nblk = (int64_t)cdiv(a.M, tc.bm) * a.tiles_n;
    MF_CHECK_ARG(nblk < (1ll << 31) && (int64_t)a.nz * a.splitk < 65536, "mf_gemm_conv: grid too large");
    dim3 grid((unsigned)nblk, 1, (unsigned)(a.nz * a.splitk));
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == MF_BF16) launch_tile<MF_BF16>(tile, a, grid, s);
    else launch_tile<MF_F32>(tile, a, grid, s);
    MF_CHECK_LAUNCH("mf_gemm_conv");
    if (a.splitk > 1) {
        const int64_t total = (int64_t)a.M * a.N * a.nz;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
        MF_CHECK_LAUNCH("mf_gemm_conv(split-K reduce)");
    }
    return MF_OK;
}
