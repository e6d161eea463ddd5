// Upper bound on what moving the d = 40 flash loop's P.V product to v_mfma_f32_16x16x32_bf16 could buy.
//
// Per 64-key tile and wave the kernel issues (csrc/attention.hip): 6 MFMAs 32x32x16 for S^T = K.Q^T (d padded 40 -> 48) and
// 8 for O^T += V^T.P^T (O^T rows padded 40 -> 64) = 14 x 32 = 448 matrix-pipe clocks, beside 33 v_exp_f32 and ~100 other VALU
// instructions.  On 16x16x32 tiles O^T pads to 48 (12 MFMAs x 16 clocks = 192 instead of 256) — but the P^T operand of a 16x16x32
// product has the 16x16 accumulator layout, so S^T has to come from 16x16x32 products too, whose K step is 32: d pads 40 -> 64
// (16 MFMAs x 16 = 256 clocks instead of 192).  The matrix clocks are the same 448 either way; what is left to gain is the
// clock the chip holds (MI355X_MICROARCH.md, DVFS give-back (7)).  This loop runs both instruction mixes with the SAME VALU filler
// (33 v_exp_f32 + 100 v_fma_f32 per tile) at the kernel's occupancy (3 waves per SIMD): time per tile, on random data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE, int VALU>       // MODE 0: 14 x 32x32x16; 1: 28 x 16x16x32; VALU 1: with the softmax's VALU filler
__global__ __launch_bounds__(256, 3) void k(const unsigned* seed, float* out, int iters) {
    bf16x8_t a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            const unsigned s = seed[(threadIdx.x * 8 + e + i * 2048) & 8191];
            a[i][e] = (short)(0x3f00 | (s & 0xff)); b[i][e] = (short)(0xbf00 ^ ((s >> 8) & 0x80ff));      // random mantissas and signs
        }
    f32x16_t acc[2];
    f32x4_t acc4[8];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 8; ++i) acc4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float v[33];
    for (int i = 0; i < 33; ++i) v[i] = -0.001f * (float)(threadIdx.x + i);
    float w = 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 14; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i + 1) & 3], acc[i & 1], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 28; ++i) acc4[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + 1) & 3], acc4[i & 7], 0, 0, 0);
        }
        if (VALU) {
#pragma unroll
            for (int i = 0; i < 33; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.999f);                        // 33 v_exp_f32 (+ 33 v_mul)
#pragma unroll
            for (int i = 0; i < 67; ++i) w = fmaf(w, 0.9999f, v[i % 33] * 1e-9f);                            // the other ~100 VALU (mul + fma)
            for (int i = 0; i < 33; ++i) v[i] -= 0.5f;
        }
    }
    float s = w;
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += acc4[i][e];
    for (int i = 0; i < 33; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int VALU>
float run(const unsigned* seed, float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, VALU>), dim3(768), dim3(256), 0, 0, seed, out, iters);       // 3 blocks per CU = 3 waves per SIMD
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    unsigned* seed; hipMalloc(&seed, 8192 * 4);
    unsigned h[8192]; srand(7); for (int i = 0; i < 8192; ++i) h[i] = (unsigned)rand();
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int round = 0; round < 3; ++round) {        // interleaved rounds in one process
        const float a0 = run<0, 0>(seed, out, iters), a1 = run<1, 0>(seed, out, iters), b0 = run<0, 1>(seed, out, iters), b1 = run<1, 1>(seed, out, iters);
        // per "tile" (one loop iteration) per wave, in ns; 3 waves per SIMD share the pipes
        printf("round %d: MFMAs only: 14 x 32x32x16 %.1f ns / tile, 28 x 16x16x32 %.1f ns (x%.3f); with 33 v_exp + ~100 VALU: %.1f ns vs %.1f ns (x%.3f)\n", round,
               a0 * 1e6 / iters, a1 * 1e6 / iters, a0 / a1, b0 * 1e6 / iters, b1 * 1e6 / iters, b0 / b1);
    }
    return 0;
}
