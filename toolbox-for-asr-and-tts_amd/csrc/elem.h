// 16-bit element types of the encoder's MFMA operands: bf16 or IEEE fp16 (same MFMA rate on gfx950).
// Storage is uint16_t everywhere; F16 = true selects fp16.
#pragma once

#include <type_traits>
#include <utility>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asrfe {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;
typedef __attribute__((ext_vector_type(4))) uint16_t u16x4;

// fp16 conversions saturate instead of overflowing: with the FP16_OVFL bit of the wave's MODE register set, a
// conversion whose result would exceed the fp16 range gives +-65504 instead of +-inf (a true infinity stays one).
// Called once at the top of every kernel that rounds activations to fp16; bf16 has the fp32 exponent range and
// needs nothing.  hwreg(HW_REG_MODE = 1, offset 23, size 1) = 1 | 23 << 6 | (1 - 1) << 11.
template <bool F16>
__device__ __forceinline__ void saturate_fp16_conversions() {
    if constexpr (F16) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
}

template <bool F16>
__device__ __forceinline__ uint16_t to_elem(float v) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, static_cast<_Float16>(v));
    else return __builtin_bit_cast(uint16_t, static_cast<__bf16>(v));
}

template <bool F16>
__device__ __forceinline__ uint32_t pack_elem2(float a, float b) {
    // a two-element vector conversion: one v_cvt_pk_{f16,bf16}_f32 (round to nearest even)
    typedef __attribute__((ext_vector_type(2))) float f2;
    const f2 ab = {a, b};
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, h2));
    } else {
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, b2));
    }
}

// two packed elements (low half = first) -> floats
template <bool F16>
__device__ __forceinline__ void unpack_elem2(uint32_t w, float& lo, float& hi) {
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 v = __builtin_bit_cast(h2, w);
        lo = static_cast<float>(v[0]);
        hi = static_cast<float>(v[1]);
    } else {
        lo = __uint_as_float(w << 16);
        hi = __uint_as_float(w & 0xffff0000u);
    }
}

// D = A (32 x 16) * B (16 x 32) + C on the matrix cores; operands are 8 consecutive k per lane.
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const u16x8 a, const u16x8 b, const f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// compile-time loop: f(IC<0>{}), ..., f(IC<N - 1>{})
template <int I>
using IC = std::integral_constant<int, I>;

template <int... I, class Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& f) {
    (f(IC<I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

}  // namespace asrfe
