// Shared device/host helpers for libmfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mfhip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t; // 8 fp16 = one MFMA A/B fragment (split precision)
typedef __attribute__((ext_vector_type(8))) int i32x8_t;       // 32 fp8 = one f8f6f4 MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef unsigned short bf16_raw;

void mf_set_error(const char* fmt, ...);

#define MF_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            mf_set_error(__VA_ARGS__);   \
            return MF_EINVAL;            \
        }                                \
    } while (0)

#define MF_CHECK_LAUNCH(name)                                                         \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            mf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return MF_ELAUNCH;                                                        \
        }                                                                             \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_raw, b);
}

// two fp32 -> packed bf16x2 (RNE) in ONE v_cvt_pk_bf16_f32; (uint32)f32_to_bf16(a) | f32_to_bf16(b) << 16 costs two
// conversions plus an SDWA or
typedef __attribute__((ext_vector_type(2))) float mf_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 mf_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    const mf_f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mf_bf16x2_t));
}

// ---- the two 16-bit storage types (MF_BF16, MF_F16): a packed pair <-> two fp32, by dtype code (wave-uniform at every call site)
typedef __attribute__((ext_vector_type(2))) _Float16 mf_f16x2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {          // nearest-even (v_cvt_f16_f32 x 2 + pack); NOT pkrtz
    const mf_f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mf_f16x2_t));
}
// Kernels take the 16-bit flavour as a template argument (F16): a run-time test per packed pair splits the unrolled load / store
// batches of the memory-bound kernels into basic blocks (measured: GroupNorm +20 %, LayerNorm +25 %, every GEMM epilogue slower).
template <bool F16>
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    if constexpr (F16) return pack_f16x2(a, b);
    else return pack_bf16x2(a, b);
}
template <bool F16>
__device__ __forceinline__ void unpack_h2(uint32_t u, float& a, float& b) {
    if constexpr (F16) {
        const mf_f16x2_t h = __builtin_bit_cast(mf_f16x2_t, u);
        a = (float)h[0]; b = (float)h[1];
    } else {
        a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
    }
}
template <bool F16>
__device__ __forceinline__ void unpack_h8(const uint4& u, float* o) {
    unpack_h2<F16>(u.x, o[0], o[1]); unpack_h2<F16>(u.y, o[2], o[3]); unpack_h2<F16>(u.z, o[4], o[5]); unpack_h2<F16>(u.w, o[6], o[7]);
}
template <bool F16>
__device__ __forceinline__ uint4 pack_h8(const float* v) {
    return uint4{pack_h2<F16>(v[0], v[1]), pack_h2<F16>(v[2], v[3]), pack_h2<F16>(v[4], v[5]), pack_h2<F16>(v[6], v[7])};
}
// the 16-bit flavour of one launch: every 16-bit operand is fp16, or every one is bf16 (host-side check of the entry points)
static inline bool mf_any_f16(int a, int b = -1, int c = -1, int d = -1) { return a == MF_F16 || b == MF_F16 || c == MF_F16 || d == MF_F16; }
static inline bool mf_any_bf16(int a, int b = -1, int c = -1, int d = -1) { return a == MF_BF16 || b == MF_BF16 || c == MF_BF16 || d == MF_BF16; }

// dtype-generic scalar load/store (dt is wave-uniform at every call site)
__device__ __forceinline__ float load_as_f32(const void* p, int dt, int64_t i) {
    if (dt == MF_F32) return ((const float*)p)[i];
    if (dt == MF_F16) return (float)((const _Float16*)p)[i];
    return bf16_to_f32(((const bf16_raw*)p)[i]);
}
__device__ __forceinline__ void store_from_f32(void* p, int dt, int64_t i, float v) {
    if (dt == MF_F32) ((float*)p)[i] = v;
    else if (dt == MF_F16) ((_Float16*)p)[i] = (_Float16)v;
    else ((bf16_raw*)p)[i] = f32_to_bf16(v);
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): only where the result is rounded to bf16 anyway
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_precise(float x) { return x / (1.0f + expf(-x)); }

// ---- LDS-DMA helpers shared by the GEMM / conv and attention kernels ---------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int srd_t;   // buffer resource descriptor (4 SGPRs)

__device__ __forceinline__ srd_t make_srd(const char* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    srd_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));     // stride 0 (raw buffer)
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);                      // num_records: loads at offset >= this return 0
    r.w = 0x00020000;
    return r;
}

__device__ __forceinline__ void dma16_buf(unsigned voff, srd_t srd, unsigned lds_off) {
    // buffer_load ... lds: per-lane 32-bit byte offset into the descriptor, hardware range check (an offset past
    // num_records writes zeros: that IS the conv zero padding / tile tail), wave-uniform LDS destination in M0.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 ::"v"(voff), "s"(srd), "s"(lds_off) : "memory", "m0");
}

// the same with 4 bytes per lane: 64 consecutive dwords into LDS [lds_off, +256 B)
__device__ __forceinline__ void dma4_buf(unsigned voff, srd_t srd, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds"
                 ::"v"(voff), "s"(srd), "s"(lds_off) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- range guard of the fp16 split precision (MF_F16X3) ----------------------------------------------------------------
// v_cvt_pkrtz_f16_f32 SATURATES: an fp32 operand above 65504 becomes 65504, not inf, so hi + lo is silently wrong and no
// NaN / inf ever trips an isfinite() check downstream.  Every kernel that splits operands keeps a per-lane running max of
// |x| (one v_max3_f32 per operand pair) and raises its translation unit's flag once per wave at the end; the host reads
// and clears the flags with mf_split_overflow().
#define MF_F16_MAX 65504.0f
__device__ __forceinline__ float mf_amax3(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }   // v_max3_f32 |a| |b|
__device__ __forceinline__ void mf_raise_if_over(unsigned* flag, float amax) {
    if (amax > MF_F16_MAX) atomicOr(flag, 1u);
}
// (hi, lo) fp16 halves of two fp32 values: hi = x toward zero, lo = x - hi toward zero.  The low half comes from ONE
// v_fma_mix_f32 per element (fp32 fma reading the f16 half of `hi` directly: x - hi, the same single rounding as
// v_cvt_f32_f16 + v_sub_f32) — four instructions per pair instead of six; the compiler does not form it from C.
__device__ __forceinline__ void mf_split_f16x2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(b));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
unsigned* mf_ovf_flag_gemm();        // device addresses of the three translation units' flags (host functions)
unsigned* mf_ovf_flag_attention();
unsigned* mf_ovf_flag_train();

// bytes per element in memory; the split compute codes keep fp32 operands
static inline int mf_dtype_size(int dt) { return (dt == MF_BF16 || dt == MF_F16) ? 2 : (dt == MF_FP8 ? 1 : 4); }
static inline bool mf_is16(int dt) { return dt == MF_BF16 || dt == MF_F16; }
static inline bool mf_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
