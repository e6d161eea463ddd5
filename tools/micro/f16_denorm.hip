// Does the gfx950 matrix pipe keep fp16 subnormals (operands of v_mfma_f32_32x32x16_f16), and does
// v_cvt_pkrtz_f16_f32 produce them?  Decides whether the f16x3 split mode of gemm_conv needs a scaled low half.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/f16_denorm.hip -o tools/micro/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ void k(float* o, float xa, float xb) {
    auto ha = __builtin_amdgcn_cvt_pkrtz(xa, xa);
    auto hb = __builtin_amdgcn_cvt_pkrtz(xb, xb);
    f16x8_t fa, fb;
    for (int i = 0; i < 8; i += 2) { fa[i] = ha[0]; fa[i + 1] = ha[1]; fb[i] = hb[0]; fb[i + 1] = hb[1]; }
    f32x16_t acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    if (threadIdx.x == 0) { o[0] = acc[0]; o[1] = (float)ha[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    const float cases[][2] = {{ldexpf(1.f, -20), 1.f}, {ldexpf(1.f, -24), 1.f}, {ldexpf(1.f, -20), ldexpf(1.f, -4)},
                              {ldexpf(1.5f, -16), 3.f}, {1.f, 1.f}};
    int bad = 0;
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        const float want = 16.f * c[0] * c[1];
        printf("a=%g b=%g: cvt(a)=%g  mfma=%g  want=%g  %s\n", c[0], c[1], h[1], h[0], want, h[0] == want ? "ok" : "MISMATCH");
        bad += h[0] != want;
    }
    printf(bad ? "f16 subnormals are NOT preserved\n" : "f16 subnormals preserved by cvt_pkrtz and the matrix pipe\n");
    return 0;
}
