// bf16 tiles 7-15 (three-stage rings, 192 / 160 / 192-wide tiles) and 20-24 (dx-tap reuse)
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_bf16_b(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_BF16;
    switch (tile) {
        case 7: launch_one<DT, 128, 128, 2, 2, false, 3>(a, grid, s); return true;
        case 8: launch_one<DT, 128, 64, 2, 2, false, 3>(a, grid, s); return true;
        case 9: launch_one<DT, 64, 64, 2, 2, false, 3>(a, grid, s); return true;
        case 10: launch_one<DT, 256, 64, 4, 1, false, 3>(a, grid, s); return true;
        case 11: launch_one<DT, 256, 128, 4, 2, false, 3>(a, grid, s); return true;
        case 12: launch_one<DT, 64, 128, 2, 2, false, 3>(a, grid, s); return true;
        case 13: launch_one<DT, 192, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 14: launch_one<DT, 128, 160, 4, 1, false, 2>(a, grid, s); return true;
        case 15: launch_one<DT, 128, 192, 2, 2, false, 2>(a, grid, s); return true;
        case 20: launch_one<DT, 128, 160, 4, 1, false, 2, true>(a, grid, s); return true;
        case 21: launch_one<DT, 128, 128, 2, 2, false, 2, true>(a, grid, s); return true;
        case 22: launch_one<DT, 64, 128, 2, 2, false, 2, true>(a, grid, s); return true;
        case 23: launch_one<DT, 128, 64, 2, 2, false, 2, true>(a, grid, s); return true;
        case 24: launch_one<DT, 192, 128, 2, 2, false, 2, true>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
