// Practical MFMA ceiling: waves doing nothing but independent v_mfma_f32_32x32x16_bf16 (no memory), 1..2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(threadIdx.x * 3 + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 * iters * 5 * 32768.0;
            printf("blocks=%d (%.1f waves/SIMD) %.2f ms -> %.0f TFLOP/s\n", blocks, blocks / 256.0, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
