// Tile tables of the two 16-bit compute codes (MF_BF16, MF_F16): the same kernels, the MFMA's operand type differs.  Each launch16_*<DT>
// is instantiated in its own translation unit (gemm_bf16_*.hip, gemm_f16_*.hip).  SKF (in-launch split-K combine) twins: bf16 only.
#pragma once
#include "gemm_conv_kernel.h"

namespace mfgemm {

template <int DT, bool AF>
static bool launch16_plain(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
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

// tiles 1-6: two-stage ring (bf16: with their fp32-activation (register-staged, converting) and in-launch split-K twins)
template <int DT>
static bool launch16_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool a_f32) {
    if constexpr (DT == MF_BF16) {
        // fp32 activations converted to bf16 on load are register staged with 2 stages: mf_gemm_conv has already resolved the
        // tile to 1..6, so the grid it computed matches the kernel's BM x BN
        if (a_f32) return launch16_plain<DT, true>(tile, a, grid, s);
        switch (tile) {
            case 1: launch_skf<DT, 128, 128, 2, 2, false, 2>(a, grid, s); return true;
            case 2: launch_skf<DT, 128, 64, 2, 2, false, 2>(a, grid, s); return true;
            case 3: launch_skf<DT, 64, 64, 2, 2, false, 2>(a, grid, s); return true;
            case 6: launch_skf<DT, 64, 128, 2, 2, false, 2>(a, grid, s); return true;
            default: break;
        }
    }
    return launch16_plain<DT, false>(tile, a, grid, s);
}

// tiles 7-15 (three-stage rings, 192 / 160 / 192-wide tiles) and 20-24 (dx-tap reuse)
template <int DT>
static bool launch16_b(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
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

// tiles 25-30 (the 16x16x32 MFMA form of tiles 1, 14, 20, 21, 6, 2), 31-36 (deeper rings), 67-68 (2x2 waves of 64x80 on 16x16 tiles)
template <int DT>
static bool launch16_c(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
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
        case 67: launch_one<DT, 128, 160, 2, 2, false, 2, false, false, false, false, true>(a, grid, s); return true;
        case 68: launch_one<DT, 128, 160, 2, 2, false, 2, true, false, false, false, true>(a, grid, s); return true;
        default: return false;
    }
}

// tiles 37-40, 47, 49, 51 (+ 53, 55, 59, 61, 63, 65): warp-specialised dx-reuse 3x3 convs (compute waves + four staging waves)
template <int DT>
static bool launch16_ws_dx(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    switch (tile) {
        case 37: launch_one<DT, 256, 160, 8, 1, false, 3, true, false, false, true>(a, grid, s); return true;
        case 38: launch_one<DT, 128, 160, 4, 1, false, 3, true, false, false, true>(a, grid, s); return true;
        case 39: launch_one<DT, 256, 160, 8, 1, false, 3, true, false, true, true>(a, grid, s); return true;
        case 40: launch_one<DT, 128, 160, 4, 1, false, 3, true, false, true, true>(a, grid, s); return true;
        case 47: launch_one<DT, 128, 160, 4, 2, false, 3, true, false, false, true, true>(a, grid, s); return true;
        case 49: launch_one<DT, 256, 160, 4, 2, false, 3, true, false, false, true, true>(a, grid, s); return true;
        case 51: launch_one<DT, 128, 160, 2, 2, false, 3, true, false, false, true, true>(a, grid, s); return true;
        // the round-5 loop forms (MF_FX_ALL) of 47, 40, 49, 51
        case 53: launch_one<DT, 128, 160, 4, 2, false, 3, true, false, false, true, true, false, MF_FX_ALL>(a, grid, s); return true;
        case 55: launch_one<DT, 128, 160, 4, 1, false, 3, true, false, true, true, false, false, MF_FX_ALL>(a, grid, s); return true;
        case 59: launch_one<DT, 256, 160, 4, 2, false, 3, true, false, false, true, true, false, MF_FX_ALL>(a, grid, s); return true;
        case 61: launch_one<DT, 128, 160, 2, 2, false, 3, true, false, false, true, true, false, MF_FX_ALL>(a, grid, s); return true;
        // four compute waves of 64x160 / 64x128 on the k16-granular cross-tile pipeline
        case 63: launch_one<DT, 256, 160, 4, 1, false, 3, true, false, false, true, false, false, MF_FX_XQ | MF_FX_EPB | MF_FX_AE>(a, grid, s); return true;
        case 65: launch_one<DT, 256, 128, 4, 1, false, 3, true, false, false, true, false, false, MF_FX_XQ | MF_FX_EPB | MF_FX_AE>(a, grid, s); return true;
        default: return false;
    }
}

// tiles 41-46, 48, 50, 52 (+ 54, 56-58, 60, 62, 64, 66): the warp-specialised form of the plain ring (1x1, strided, upsampled calls)
template <int DT>
static bool launch16_ws_ring(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) {
    if constexpr (DT == MF_BF16) {
        switch (tile) {      // in-launch split-K twins
            case 41: launch_skf<DT, 128, 160, 4, 1, false, 3, false, false, false, true>(a, grid, s); return true;
            case 43: launch_skf<DT, 128, 160, 4, 1, false, 3, false, false, true, true>(a, grid, s); return true;
            case 44: launch_skf<DT, 128, 128, 2, 2, false, 3, false, false, false, true>(a, grid, s); return true;
            case 48: launch_skf<DT, 128, 160, 4, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
            default: break;
        }
    }
    switch (tile) {
        case 41: launch_one<DT, 128, 160, 4, 1, false, 3, false, false, false, true>(a, grid, s); return true;
        case 42: launch_one<DT, 256, 160, 8, 1, false, 3, false, false, true, true>(a, grid, s); return true;
        case 43: launch_one<DT, 128, 160, 4, 1, false, 3, false, false, true, true>(a, grid, s); return true;
        case 44: launch_one<DT, 128, 128, 2, 2, false, 3, false, false, false, true>(a, grid, s); return true;
        case 45: launch_one<DT, 256, 128, 4, 2, false, 3, false, false, false, true>(a, grid, s); return true;
        case 46: launch_one<DT, 256, 128, 4, 2, false, 3, false, false, true, true>(a, grid, s); return true;
        case 48: launch_one<DT, 128, 160, 4, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
        case 50: launch_one<DT, 256, 160, 4, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
        case 52: launch_one<DT, 128, 160, 2, 2, false, 3, false, false, false, true, true>(a, grid, s); return true;
        // the round-5 loop forms (MF_FX_ALL) of 48, 43, 44, 46, 50, 52
        case 54: launch_one<DT, 128, 160, 4, 2, false, 3, false, false, false, true, true, false, MF_FX_ALL>(a, grid, s); return true;
        case 56: launch_one<DT, 128, 160, 4, 1, false, 3, false, false, true, true, false, false, MF_FX_ALL>(a, grid, s); return true;
        case 57: launch_one<DT, 128, 128, 2, 2, false, 3, false, false, false, true, false, false, MF_FX_ALL>(a, grid, s); return true;
        case 58: launch_one<DT, 256, 128, 4, 2, false, 3, false, false, true, true, false, false, MF_FX_ALL>(a, grid, s); return true;
        case 60: launch_one<DT, 256, 160, 4, 2, false, 3, false, false, false, true, true, false, MF_FX_ALL>(a, grid, s); return true;
        case 62: launch_one<DT, 128, 160, 2, 2, false, 3, false, false, false, true, true, false, MF_FX_ALL>(a, grid, s); return true;
        case 64: launch_one<DT, 256, 160, 4, 1, false, 3, false, false, false, true, false, false, MF_FX_XQ | MF_FX_EPB | MF_FX_AE>(a, grid, s); return true;
        case 66: launch_one<DT, 256, 128, 4, 1, false, 3, false, false, false, true, false, false, MF_FX_XQ | MF_FX_EPB | MF_FX_AE>(a, grid, s); return true;
        default: return false;
    }
}

}  // namespace mfgemm
