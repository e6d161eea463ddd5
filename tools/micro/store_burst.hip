// How fast can every CU of an MI355X write an output tile at once?  (What bounds the GEMM / conv epilogues: tools/stamps.py shows
// 11 us for the bare stores of a 256 x 160 bf16 tile per CU, 80 KB, = 1.9 TB/s chip-wide.)  One block per CU writes a tile of ROWS
// rows x SEG bytes at row pitch PITCH from registers, 16 bytes per lane, consecutive lanes along a row — the epilogue's pattern —
// with no loads, no LDS and no arithmetic.  Variants: row segment (320 B = half an output row, 640 B = the whole row), threads per
// block, non-temporal / write-through stores, and a tile written in 1 .. 4 time-separated pieces by one block.
//   hipcc --offload-arch=gfx950 -O3 -o store_burst store_burst.hip && ./store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int MODE>   // 0 plain, 1 nontemporal, 2 sc1 (write-through)
__device__ __forceinline__ void st16(char* p, const uint4& v) {
    if (MODE == 0) *reinterpret_cast<uint4*>(p) = v;
    else if (MODE == 1) {
        typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
        const u4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u4_t*>(p));
    }
    else {
        typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
        const u4_t w = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
    }
}

template <int MODE>
__global__ void k(char* out, int rows, int seg, int pitch, int tiles_per_row, int reps) {
    const int tile = blockIdx.x;
    const int tm = tile / tiles_per_row, tn = tile - tm * tiles_per_row;
    char* base = out + (size_t)tm * rows * pitch + (size_t)tn * seg;
    const int vpr = seg / 16;                      // 16-byte vectors per row segment
    const int total = rows * vpr;
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int row = i / vpr, c = i - row * vpr;
            st16<MODE>(base + (size_t)row * pitch + c * 16, v);
            v.x += 1;
        }
}

template <int MODE>
float run(char* out, int blocks, int threads, int rows, int seg, int pitch, int tpr) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, rows, seg, pitch, tpr, 1);
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int t = 0; t < 7; ++t) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, rows, seg, pitch, tpr, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms * 100.0f);           // us per launch
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    char* out;
    const size_t bytes = 256ull << 20;
    hipMalloc(&out, bytes);
    hipMemset(out, 0, bytes);
    struct Case { const char* name; int blocks, threads, rows, seg, pitch, tpr; };
    const Case cases[] = {
        {"256 x 160 bf16 tiles of a 32768 x 320 output (the 64^2 convs)", 256, 768, 256, 320, 640, 2},
        {"same, 512 threads", 256, 512, 256, 320, 640, 2},
        {"same, 256 threads", 256, 256, 256, 320, 640, 2},
        {"128 x 320 bf16 tiles (whole rows), 512 threads", 256, 512, 128, 640, 640, 1},
        {"128 x 160 tiles, 512 blocks of 512 threads", 512, 512, 128, 320, 640, 2},
        {"128 x 160 tiles of a 8192 x 640 output, 256 blocks", 256, 512, 128, 320, 1280, 4},
        {"128 x 128 tiles of a 2048 x 1280 output, 160 blocks", 160, 512, 128, 256, 2560, 10},
        {"64 x 64 tiles of a 2048 x 1280 output, 640 blocks of 256", 640, 256, 64, 128, 2560, 20},
        {"84 MB: 128 x 160 tiles of a 32768 x 1280 output (GEGLU), 2048 blocks", 2048, 512, 128, 320, 2560, 8},
    };
    for (const Case& c : cases) {
        const double mb = (double)c.blocks * c.rows * c.seg / 1e6;
        const float t0 = run<0>(out, c.blocks, c.threads, c.rows, c.seg, c.pitch, c.tpr);
        const float t1 = run<1>(out, c.blocks, c.threads, c.rows, c.seg, c.pitch, c.tpr);
        const float t2 = run<2>(out, c.blocks, c.threads, c.rows, c.seg, c.pitch, c.tpr);
        printf("%-72s %6.1f MB: plain %6.2f us (%5.2f TB/s)  nt %6.2f us (%5.2f)  sc1 %6.2f us (%5.2f)\n", c.name, mb, t0, mb / t0, t1, mb / t1, t2,
               mb / t2);
    }
    return 0;
}
