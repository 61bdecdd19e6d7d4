// modarith.cuh -- 64-bit modular arithmetic for CDNA4 (no native 64-bit multiplier: everything
// lowers to v_mad_u64_u32 / v_mul_hi_u32 chains).  Any algorithm is admissible as long as the final
// canonical residues equal the reference's (SEAL/util/uintarithsmallmod.h:167-326).
#pragma once
#include "common.h"

namespace moai {

__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return __umul64hi(a, b);
}

// Shoup lazy product: y * w mod q in [0, 2q) for ANY 64-bit y, given wq = floor(w * 2^64 / q)
// (multiply_uint_mod_lazy, uintarithsmallmod.h:313-326).  Written as y*w + t*(-q) so that the two
// low products accumulate in one v_mad_u64_u32 chain instead of ending in a 64-bit subtract
// (v_sub_co / v_subb_co plus a VCC wait state on gfx950).
__device__ __forceinline__ uint64_t mul_shoup_lazy(uint64_t y, uint64_t w, uint64_t wq, uint64_t q)
{
    uint64_t t = mulhi64(y, wq);
    return y * w + t * (0 - q);
}

__device__ __forceinline__ uint64_t csub(uint64_t x, uint64_t m)
{
    // x >= m ? x - m : x
    return x >= m ? x - m : x;
}

// Barrett reduction of a 128-bit value (hi:lo) with cr = floor(2^128/q) (barrett_reduce_128,
// uintarithsmallmod.h:167-203).  Exact for q < 2^63.
__device__ __forceinline__ uint64_t barrett128(uint64_t lo, uint64_t hi, uint64_t q, uint64_t cr0, uint64_t cr1)
{
    // floor((hi:lo) * cr / 2^128), low 64 bits
    uint64_t carry = mulhi64(lo, cr0);
    uint64_t t_lo = lo * cr1;
    uint64_t t_hi = mulhi64(lo, cr1);
    uint64_t s = t_lo + carry;
    uint64_t tmp3 = t_hi + (s < t_lo ? 1 : 0);
    uint64_t u_lo = hi * cr0;
    uint64_t u_hi = mulhi64(hi, cr0);
    uint64_t s2 = s + u_lo;
    uint64_t c2 = u_hi + (s2 < s ? 1 : 0);
    uint64_t quot = hi * cr1 + tmp3 + c2;
    uint64_t r = lo - quot * q;
    return csub(r, q);
}

__device__ __forceinline__ uint64_t mulmod_barrett(uint64_t a, uint64_t b, uint64_t q, uint64_t cr0, uint64_t cr1)
{
    return barrett128(a * b, mulhi64(a, b), q, cr0, cr1);
}

// barrett_reduce_64 (uintarithsmallmod.h:211-230)
__device__ __forceinline__ uint64_t barrett64(uint64_t x, uint64_t q, uint64_t cr1)
{
    uint64_t t = mulhi64(x, cr1);
    return csub(x - t * q, q);
}

// Cooley-Tukey butterfly, Harvey lazy form: x, y in [0, 4q) -> [0, 4q)   (SEAL/util/ntt.h:30-61,
// dwthandler.h:110-163)
__device__ __forceinline__ void ct_bfly(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2)
{
#if defined(MOAI_ABLATE) && MOAI_ABLATE == 1
    // diagnostic build only (tools/ablate.sh): data movement without the arithmetic
    x += y + w;
    y ^= wq + q + q2;
    return;
#endif
    uint64_t u = csub(x, q2);
    uint64_t v = mul_shoup_lazy(y, w, wq, q);
    x = u + v;
    y = u + q2 - v;
}

// Cooley-Tukey butterfly without the guard: values grow by 2q per stage, so 16 stages starting below
// 4q stay below 36q.  Used when 36q < 2^64 (every prime of MOAI's chain, <= 58 bits); the pass that
// finishes the transform reduces with one Barrett step.  Same residues as ct_bfly.
__device__ __forceinline__ void ct_bfly_noguard(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2)
{
#if defined(MOAI_ABLATE) && MOAI_ABLATE == 1
    x += y + w;
    y ^= wq + q + q2;
    return;
#endif
    uint64_t v = mul_shoup_lazy(y, w, wq, q);
    uint64_t u = x;
    x = u + v;
    y = u + q2 - v;
}

template <bool NOGUARD>
__device__ __forceinline__ void ct_bfly_t(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2)
{
    if (NOGUARD)
    {
        ct_bfly_noguard(x, y, w, wq, q, q2);
    }
    else
    {
        ct_bfly(x, y, w, wq, q, q2);
    }
}

// Gentleman-Sande butterfly, lazy: x, y in [0, 2q) -> [0, 2q)   (dwthandler.h:226-250)
__device__ __forceinline__ void gs_bfly(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2)
{
#if defined(MOAI_ABLATE) && MOAI_ABLATE == 1
    x += y + w;
    y ^= wq + q + q2;
    return;
#endif
    uint64_t u = x;
    uint64_t v = y;
    x = csub(u + v, q2);
    y = mul_shoup_lazy(u + q2 - v, w, wq, q);
}

// last inverse stage with N^-1 folded in (dwthandler.h:273-314)
__device__ __forceinline__ void gs_bfly_last(uint64_t &x, uint64_t &y, const Tw &ninv, const Tw &ninv_w1, uint64_t q,
                                             uint64_t q2)
{
    uint64_t u = x;
    uint64_t v = y;
    x = mul_shoup_lazy(csub(u + v, q2), ninv.w, ninv.wq, q);
    y = mul_shoup_lazy(u + q2 - v, ninv_w1.w, ninv_w1.wq, q);
}

} // namespace moai
