// bf16 tiles 41-46, 48, 50, 52: the warp-specialised form of the plain ring (1x1, strided, upsampled calls), with in-launch split-K twins
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


bool launch_bf16_ws_ring(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_BF16;
    switch (tile) {
        case 41: launch_skf<DT, 128, 160, 4, 1, false, 3, false, false, false, true>(a, grid, s); return true;
        case 42: launch_one<DT, 256, 160, 8, 1, false, 3, false, false, true, true>(a, grid, s); return true;
        case 43: launch_skf<DT, 128, 160, 4, 1, false, 3, false, false, true, true>(a, grid, s); return true;
        case 44: launch_skf<DT, 128, 128, 2, 2, false, 3, false, false, false, true>(a, grid, s); return true;
        case 45: launch_one<DT, 256, 128, 4, 2, false, 3, false, false, false, true>(a, grid, s); return true;
        case 46: launch_one<DT, 256, 128, 4, 2, false, 3, false, false, true, true>(a, grid, s); return true;
        case 48: launch_skf<DT, 128, 160, 4, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
        case 50: launch_one<DT, 256, 160, 4, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
        case 52: launch_one<DT, 128, 160, 2, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
