// fp32 MFMA (v_mfma_f32_32x32x2_f32), tiles 13-15, 20-24, 31-36
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_f32_b(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_F32;
    switch (tile) {
        case 13: launch_one<DT, 192, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 14: launch_one<DT, 128, 160, 4, 1, false, 2>(a, grid, s); return true;
        case 15: launch_one<DT, 128, 192, 2, 2, false, 2>(a, grid, s); return true;
        case 20: launch_one<DT, 128, 160, 4, 1, false, 2, true>(a, grid, s); return true;
        case 21: launch_one<DT, 128, 128, 2, 2, false, 2, true>(a, grid, s); return true;
        case 22: launch_one<DT, 64, 128, 2, 2, false, 2, true>(a, grid, s); return true;
        case 23: launch_one<DT, 128, 64, 2, 2, false, 2, true>(a, grid, s); return true;
        case 24: launch_one<DT, 192, 128, 2, 2, false, 2, true>(a, grid, s); return true;
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
