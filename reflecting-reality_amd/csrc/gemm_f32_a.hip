// fp32 MFMA (v_mfma_f32_32x32x2_f32), tiles 1-12
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_f32_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_F32;
    switch (tile) {
        case 1: launch_one<DT, 128, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 2: launch_one<DT, 128, 64, 2, 2, false, 2>(a, grid, s); return true;
        case 3: launch_one<DT, 64, 64, 2, 2, false, 2>(a, grid, s); return true;
        case 4: launch_one<DT, 256, 64, 4, 1, false, 2>(a, grid, s); return true;
        case 5: launch_one<DT, 256, 128, 4, 2, false, 2>(a, grid, s); return true;
        case 6: launch_one<DT, 64, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 7: launch_one<DT, 128, 128, 2, 2, false, 3>(a, grid, s); return true;
        case 8: launch_one<DT, 128, 64, 2, 2, false, 3>(a, grid, s); return true;
        case 9: launch_one<DT, 64, 64, 2, 2, false, 3>(a, grid, s); return true;
        case 10: launch_one<DT, 256, 64, 4, 1, false, 3>(a, grid, s); return true;
        case 11: launch_one<DT, 256, 128, 4, 2, false, 3>(a, grid, s); return true;
        case 12: launch_one<DT, 64, 128, 2, 2, false, 3>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
