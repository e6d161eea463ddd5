// bf16 tiles 1-6
// (one tile group of mf_gemm_conv; tile tables: gemm_16bit_tiles.h, kernel template: gemm_conv_kernel.h)
#include "gemm_16bit_tiles.h"

namespace mfgemm {
bool launch_bf16_a(int tile, const GemmArgs& a, dim3 grid, hipStream_t s, bool a_f32) { return launch16_a<MF_BF16>(tile, a, grid, s, a_f32); }
}  // namespace mfgemm
