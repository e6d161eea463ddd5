// Shader clock under load: s_memtime (shader cycles) vs wall time for a pure-MFMA loop and an MFMA + ds_read_b128 loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <bool LDS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[65536];
    for (int i = threadIdx.x * 16; i < 65536; i += 256 * 16) *reinterpret_cast<uint4*>(sm + i) = make_uint4(i, 1, 2, 3);
    __syncthreads();
    f32x16_t acc[5];
    for (int i = 0; i < 5; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    uint4 a = make_uint4(threadIdx.x, 1, 2, 3), b[5];
    for (int j = 0; j < 5; ++j) b[j] = make_uint4(j, threadIdx.x, 2, 3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            const char* base = sm + ((threadIdx.x * 16 + it * 4096) & 32767);
            a = *reinterpret_cast<const uint4*>(base);
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = *reinterpret_cast<const uint4*>(base + 4096 * (j + 1));
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b[i]), acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 5; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMallocManaged(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lds = 0; lds < 2; ++lds)
        for (int rep = 0; rep < 4; ++rep) {
            const int iters = 60000, blocks = 512;
            (void)hipEventRecord(e0);
            if (lds) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 * iters * 5 * 32768.0;
            printf("%s: %.2f ms, %.0f TFLOP/s, block0 wave0 spent %llu counter ticks -> %.0f MHz if ticks are shader cycles (%.1f cycles per MFMA)\n",
                   lds ? "mfma+ds_read" : "mfma only", ms, flops / ms / 1e9, *cyc, *cyc / (ms * 1e3), (double)*cyc / (iters * 5.0));
        }
    return 0;
}
