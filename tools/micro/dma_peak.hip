// Practical global->CU load ceiling per CU, no compute: (0) buffer_load_dwordx4 ... lds (the LDS-DMA the GEMM/conv
// kernels stage with), (1) buffer_load_dwordx4 into VGPRs, (2) VGPRs + ds_write_b128.  Each wave keeps two batches
// of U 1-KB loads in flight (the 2-stage ring of the kernels).  Source footprint: `span` MB cycled (L2 / MALL / HBM).
//   hipcc --offload-arch=gfx950 -O3 -o dma_peak dma_peak.hip && ./dma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) int srd_t;
typedef __attribute__((address_space(3))) char* lds_ptr_t;
__device__ __forceinline__ srd_t make_srd(const char* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    srd_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int U>
__global__ __launch_bounds__(256) void k(const char* src, unsigned span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const srd_t srd = make_srd(src, span);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem) + wave * (2 * U * 1024);
    // this wave streams 1-KB chunks: chunk c of batch b -> offset ((blockIdx*4 + wave) * iters*U + b*U + c) KB, wrapped
    unsigned off = (unsigned)((((size_t)blockIdx.x * 4 + wave) * (size_t)iters * U * 1024) % span) + lane * 16;
    uint4 r[2][U];
    unsigned acc = 0;
    auto issue = [&](int b) {
#pragma unroll
        for (int c = 0; c < U; ++c) {
            if (MODE == 0) {
                const unsigned l = lds_base + (b * U + c) * 1024;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(srd), "s"(l) : "memory", "m0");
            } else {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[b][c]) : "v"(off), "s"(srd) : "memory");
            }
            off += 1024;
            if (off >= span) off -= span;
        }
    };
    auto consume = [&](int b) {
        if (MODE == 0) return;
#pragma unroll
        for (int c = 0; c < U; ++c) {
            if (MODE == 2) *reinterpret_cast<uint4*>(smem + wave * (2 * U * 1024) + (b * U + c) * 1024 + lane * 16) = r[b][c];
            else acc ^= r[b][c].x ^ r[b][c].w;
        }
    };
    issue(0);
    for (int it = 0; it < iters; it += 2) {
        issue(1);
        wait_vmcnt<U>();
        consume(0);
        issue(0);
        wait_vmcnt<U>();
        consume(1);
    }
    wait_vmcnt<0>();
    if (MODE != 1) { __syncthreads(); acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4); }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int U>
void run(const char* src, unsigned span, int blocks, unsigned* sink, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 400;
    const int smem = 4 * 2 * U * 1024;
    hipFuncSetAttribute((const void*)k<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), smem, 0, src, span, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double bytes = (double)blocks * 4 * (iters + 1) * U * 1024.0;
    printf("mode=%d U=%2d blocks=%4d (%d KB LDS/block) span=%4u MB: %.3f ms  %.1f TB/s  %.1f GB/s/CU  %.1f B/clk/CU@2.4GHz\n", MODE, U, blocks,
           smem / 1024, span >> 20, best, bytes / best / 1e9, bytes / best / 1e6 / 256, bytes / best / 1e6 / 256 / 2.4);
}

int main() {
    char* src; hipMalloc(&src, 1u << 30); hipMemset(src, 1, 1u << 30);
    unsigned* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned span_mb : {16u, 128u, 1024u - 1}) {
        const unsigned span = span_mb << 20;
        for (int blocks : {256, 512, 1024}) {
            run<0, 4>(src, span, blocks, sink, e0, e1);
            run<0, 9>(src, span, blocks, sink, e0, e1);
            run<0, 16>(src, span, blocks, sink, e0, e1);
            run<1, 4>(src, span, blocks, sink, e0, e1);
            run<1, 9>(src, span, blocks, sink, e0, e1);
            run<2, 9>(src, span, blocks, sink, e0, e1);
        }
    }
    return 0;
}
