// f16x3 with a pre-split W on the warp-specialised tiles 37, 38, 41, 44
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_f16x3_ws(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_F16X3;
    // (fp32 operands: the staging bytes are the bf16 kernels' bytes; 256 x 128 with 8 + 4 waves spills at the 170-register
    // budget of three waves per SIMD: not offered)
    switch (tile) {
        case 37: launch_one<DT, 256, 160, 8, 1, false, 3, true, true, false, true>(a, grid, s); return true;
        case 38: launch_one<DT, 128, 160, 4, 1, false, 3, true, true, false, true>(a, grid, s); return true;
        case 41: launch_skf<DT, 128, 160, 4, 1, false, 3, false, true, false, true>(a, grid, s); return true;
        case 44: launch_skf<DT, 128, 128, 2, 2, false, 3, false, true, false, true>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
