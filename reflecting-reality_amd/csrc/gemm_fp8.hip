// fp8 e4m3 (v_mfma_scale_f32_32x32x64_f8f6f4): the tiles instantiated for it
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_fp8(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    switch (tile) {
        case 1: launch_one<MF_FP8, 128, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 2: launch_one<MF_FP8, 128, 64, 2, 2, false, 2>(a, grid, s); return true;
        case 3: launch_one<MF_FP8, 64, 64, 2, 2, false, 2>(a, grid, s); return true;
        case 6: launch_one<MF_FP8, 64, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 7: launch_one<MF_FP8, 128, 128, 2, 2, false, 3>(a, grid, s); return true;
        case 13: launch_one<MF_FP8, 192, 128, 2, 2, false, 2>(a, grid, s); return true;
        case 14: launch_one<MF_FP8, 128, 160, 4, 1, false, 2>(a, grid, s); return true;
        case 15: launch_one<MF_FP8, 128, 192, 2, 2, false, 2>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
