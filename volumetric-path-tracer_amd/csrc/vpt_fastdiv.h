// Division by a grid extent without the divider (to_unit, vpt_trace_common.h): host-side check of one divisor.
//
// The reference divides the index-space position by the grid extent at every look-up (render_kernel.cu:987-997); the quotient
// has to be the correctly rounded one.  For a divisor d known in advance, with r = RN(1 / d):
//     y = RN(q r);   e = q - d y  (one FMA: exact);   u = RN(y + e r)
// is the correctly rounded q / d for all q but not for all d (Brisebarre, Muller, Raina: "Accelerating correctly rounded
// floating-point division when the divisor is known in advance", IEEE TC 2004), so every divisor is CHECKED, not assumed: mul, fma and
// the division commute with scaling q by a power of two as long as nothing under- or overflows, so the 2^23 significands of one
// binade decide for all q whose intermediates stay normal (the device guards that range and divides otherwise).  ~3 ms per divisor.
#pragma once
#include <cstdint>
#include <cstring>
namespace vpt {
// r: the reciprocal the device will multiply by (RN(1 / d) in the product; a parameter so that tests can break it)
#if defined(__x86_64__)
#define VPT_FASTDIV_TARGET __attribute__((target("fma")))
#else
#define VPT_FASTDIV_TARGET
#endif
VPT_FASTDIV_TARGET inline bool fast_div_check_fma(float d, float r) {
    uint32_t bad = 0;
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        const uint32_t bits = 0x3f800000u | m;                // q in [1, 2)
        float q;
        std::memcpy(&q, &bits, 4);
        const float ref = q / d;
        float y = q * r;
        const float e = __builtin_fmaf(-d, y, q);             // (x86: one vfmadd; elsewhere the target's fma or libm's correctly rounded fmaf)
        y = __builtin_fmaf(e, r, y);
        bad |= (ref != y) ? 1u : 0u;
    }
    return bad == 0u;
}
// d: a grid extent (integer-valued, 1 <= d <= 2^16: with |q| in [2^-40, 2^40] no intermediate of the sequence is subnormal)
inline bool fast_div_ok(float d, float r) {
    if (!(d >= 1.0f && d <= 65536.0f)) return false;
#if defined(__x86_64__)
    if (!__builtin_cpu_supports("fma")) return false;         // no hardware FMA to check with in reasonable time: keep the division
#endif
    return fast_div_check_fma(d, r);
}
}  // namespace vpt
