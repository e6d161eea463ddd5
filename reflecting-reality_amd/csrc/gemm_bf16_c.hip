// bf16 tiles 25-30 (the 16x16x32 MFMA form of tiles 1, 14, 20, 21, 6, 2) and 31-36 (deeper rings)
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_bf16_c(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_BF16;
    switch (tile) {
        case 25: launch_one<DT, 128, 128, 2, 2, false, 2, false, false, true>(a, grid, s); return true;
        case 26: launch_one<DT, 128, 160, 4, 1, false, 2, false, false, true>(a, grid, s); return true;
        case 27: launch_one<DT, 128, 160, 4, 1, false, 2, true, false, true>(a, grid, s); return true;
        case 28: launch_one<DT, 128, 128, 2, 2, false, 2, true, false, true>(a, grid, s); return true;
        case 29: launch_one<DT, 64, 128, 2, 2, false, 2, false, false, true>(a, grid, s); return true;
        case 30: launch_one<DT, 128, 64, 2, 2, false, 2, false, false, true>(a, grid, s); return true;
        case 31: launch_one<DT, 128, 128, 2, 2, false, 4>(a, grid, s); return true;
        case 32: launch_one<DT, 128, 64, 2, 2, false, 4>(a, grid, s); return true;
        case 33: launch_one<DT, 64, 128, 2, 2, false, 4>(a, grid, s); return true;
        case 34: launch_one<DT, 64, 64, 2, 2, false, 4>(a, grid, s); return true;
        case 35: launch_one<DT, 64, 128, 2, 2, false, 6>(a, grid, s); return true;
        case 36: launch_one<DT, 64, 64, 2, 2, false, 6>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
