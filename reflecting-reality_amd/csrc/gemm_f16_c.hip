// f16 tiles 25-36
// (one tile group of mf_gemm_conv; tile tables: gemm_16bit_tiles.h, kernel template: gemm_conv_kernel.h)
#include "gemm_16bit_tiles.h"

namespace mfgemm {
bool launch_f16_c(int tile, const GemmArgs& a, dim3 grid, hipStream_t s) { return launch16_c<MF_F16>(tile, a, grid, s); }
}  // namespace mfgemm
