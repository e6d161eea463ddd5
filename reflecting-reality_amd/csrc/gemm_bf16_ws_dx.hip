// bf16 tiles 37-40, 47, 49, 51: warp-specialised dx-reuse 3x3 convs (compute waves + four staging waves)
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_bf16_ws_dx(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_BF16;
    switch (tile) {
        case 37: launch_one<DT, 256, 160, 8, 1, false, 3, true, false, false, true>(a, grid, s); return true;
        case 38: launch_one<DT, 128, 160, 4, 1, false, 3, true, false, false, true>(a, grid, s); return true;
        case 39: launch_one<DT, 256, 160, 8, 1, false, 3, true, false, true, true>(a, grid, s); return true;
        case 40: launch_one<DT, 128, 160, 4, 1, false, 3, true, false, true, true>(a, grid, s); return true;
        case 47: launch_one<DT, 128, 160, 4, 2, false, 3, true, false, false, true, true>(a, grid, s); return true;
        case 49: launch_one<DT, 256, 160, 4, 2, false, 3, true, false, false, true, true>(a, grid, s); return true;
        case 51: launch_one<DT, 128, 160, 2, 2, false, 3, true, false, false, true, true>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
