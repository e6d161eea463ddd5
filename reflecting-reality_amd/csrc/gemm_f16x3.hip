// f16x3 (the parity mode: fp32 operands as two fp16 halves, three MFMAs per product): tiles 1-3, 6, 7, 14, 20-22
// (one tile group of mf_gemm_conv; kernel template and design notes: gemm_conv_kernel.h)
#include "gemm_conv_kernel.h"

namespace mfgemm {


// The tiles whose register budget holds the split fragments.  WPK: the W operand was split ahead of time.
template <int DT, bool WPK>
static bool launch_split_base(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    switch (tile) {
        case 1: launch_one<DT, 128, 128, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        case 2: launch_one<DT, 128, 64, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        case 3: launch_one<DT, 64, 64, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        case 6: launch_one<DT, 64, 128, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        default: break;
    }
    if constexpr (WPK) {
        switch (tile) {
            case 7: launch_one<DT, 128, 128, 2, 2, false, 3, false, true>(a, grid, s); return true;
            case 14: launch_one<DT, 128, 160, 4, 1, false, 2, false, true>(a, grid, s); return true;
            case 20: launch_one<DT, 128, 160, 4, 1, false, 2, true, true>(a, grid, s); return true;
            case 21: launch_one<DT, 128, 128, 2, 2, false, 2, true, true>(a, grid, s); return true;
            case 22: launch_one<DT, 64, 128, 2, 2, false, 2, true, true>(a, grid, s); return true;
            default: break;
        }
    } else if constexpr (DT == MF_F16X3 || DT == MF_BF16X1) {
        // raw fp32 W (training: the optimizer rewrites the weights every step; dgrad weights): the big-conv tiles too
        switch (tile) {
            case 14: launch_one<DT, 128, 160, 4, 1, false, 2, false, false>(a, grid, s); return true;
            case 20: launch_one<DT, 128, 160, 4, 1, false, 2, true, false>(a, grid, s); return true;
            case 21: launch_one<DT, 128, 128, 2, 2, false, 2, true, false>(a, grid, s); return true;
            default: break;
        }
    }
    return false;
}

template <bool WPK>
static bool launch_f16x3_t(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DT = MF_F16X3;
    switch (tile) {     // in-launch split-K twins
        case 1: launch_skf<DT, 128, 128, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        case 2: launch_skf<DT, 128, 64, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        case 3: launch_skf<DT, 64, 64, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        case 6: launch_skf<DT, 64, 128, 2, 2, false, 2, false, WPK>(a, grid, s); return true;
        default: return launch_split_base<DT, WPK>(tile, a, grid, s);
    }
}

bool launch_f16x3(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool wpk) {
    return wpk ? launch_f16x3_t<true>(tile, a, grid, s) : launch_f16x3_t<false>(tile, a, grid, s);
}

}  // namespace mfgemm
