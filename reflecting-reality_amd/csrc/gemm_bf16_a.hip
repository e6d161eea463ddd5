// bf16 tiles 1-6: two-stage ring, with their fp32-activation (register-staged, converting) and in-launch split-K twins
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


template <bool AF>
static bool launch_plain(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_BF16;
    switch (tile) {
        case 1: launch_one<DT, 128, 128, 2, 2, AF, 2>(a, grid, s); return true;
        case 2: launch_one<DT, 128, 64, 2, 2, AF, 2>(a, grid, s); return true;
        case 3: launch_one<DT, 64, 64, 2, 2, AF, 2>(a, grid, s); return true;
        case 4: launch_one<DT, 256, 64, 4, 1, AF, 2>(a, grid, s); return true;
        case 5: launch_one<DT, 256, 128, 4, 2, AF, 2>(a, grid, s); return true;
        case 6: launch_one<DT, 64, 128, 2, 2, AF, 2>(a, grid, s); return true;
        default: return false;
    }
}

bool launch_bf16_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool a_f32) {
    // fp32 activations converted to bf16 on load are register staged with 2 stages: mf_gemm_conv has already resolved the
    // tile to 1..6, so the grid it computed matches the kernel's BM x BN
    if (a_f32) return launch_plain<true>(tile, a, grid, s);
    constexpr int DT = MF_BF16;
    switch (tile) {
        case 1: launch_skf<DT, 128, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 2: launch_skf<DT, 128, 64, 2, 2, false, 2>(a, grid, s); return true;
        case 3: launch_skf<DT, 64, 64, 2, 2, false, 2>(a, grid, s); return true;
        case 6: launch_skf<DT, 64, 128, 2, 2, false, 2>(a, grid, s); return true;
        default: return launch_plain<false>(tile, a, grid, s);
    }
}

}  // namespace mfgemm
