// modarith.hip.h -- 64-bit modular arithmetic for CDNA4 (no native 64-bit multiplier: everything
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
    uint64_t v = mul_shoup_lazy(y, w, wq, q);
    uint64_t u = x;
    x = u + v;
    y = u + q2 - v;
}

// ---- FP64 arithmetic for primes below 2^51 ---------------------------------------------------------------
// gfx950 issues v_fma_f64 / v_mul_f64 / v_add_f64 / v_rndne_f64 at the rate of ONE 32-bit integer multiply
// (tools/valu_bench: 59 vs 56 lane-ops/clk/CU), and a 64-bit Shoup product costs ten of those multiplies.  For
// q < 2^51 the same exact integers fit the 53-bit significand: values live in the 64-bit registers as doubles
// holding (signed) integers, and
//     y * w - rint(y * (w/q)) * q      with h = RN(y*w), l = fma(y, w, -h) (the exact rounding error of h)
// is computed without any rounding: |y| < 2^52 makes the quotient estimate off by less than 1.5, so the true
// remainder r has |r| < 1.5 q, h - c*q = r - l is an integer below 2^53 (|l| <= 2^50), the fma returns it
// exactly and adding l gives r.  Eight FP64 operations per butterfly instead of ~26 integer ones (measured
// 4.5e12 against 1.7e12 butterflies/s).  The final canonical residues are those of the same exact integers,
// i.e. the reference's.  Everything here relies on -ffp-contract=off plus the explicit fma calls.
// Twiddles are {double w, double RN(w/q)} in a Tw's two words; the mode's (q, q2) arguments carry the bit
// patterns of (double q, double RN(1/q)).
enum
{
    M_LAZY8 = -2,  // integer, q < 2^60: approximate Shoup quotient, values below 8q, see ct_bfly_lazy8; plain forward NTT
    M_GUARD2 = -1, // integer, any q < 2^61: the guard of every SECOND stage only (values below 8q), plain forward NTT
    M_GUARD = 0,   // integer, reference discipline [0,4q)
    M_NOGUARD = 1, // integer, 36 q < 2^64
    M_FPN = 2,     // FP64, q < 2^52 / 25: sixteen stages need no intermediate reduction
    M_FPR = 3      // FP64, q < 2^51: the untouched operand of every butterfly is reduced first
};

__device__ __forceinline__ double u2d(uint64_t b)
{
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ uint64_t d2u(double d)
{
    return (uint64_t)__double_as_longlong(d);
}
// exact conversion of an integer below 2^53
__device__ __forceinline__ double fp_from_u64(uint64_t v)
{
    return __builtin_fma((double)(uint32_t)(v >> 32), 4294967296.0, (double)(uint32_t)v);
}
// exact conversion of an integer below 2^52 in two cheap steps: its bits are the mantissa of 2^52 + v (an OR on the high
// dword), and subtracting 2^52 is exact.  The key-switch kernels convert sixteen key residues and eight digits per
// thread and digit this way (canonical residues of primes below 2^51) instead of with two v_cvt + one fma.
__device__ __forceinline__ double fp_from_u52(uint64_t v)
{
    return u2d(v | 0x4330000000000000ull) - 4503599627370496.0;
}
// x - rint(x/q) q, |result| <= q/2 (+1): exact for integer |x| < 2^53
__device__ __forceinline__ double fp_red(double x, double q, double qinv)
{
    double c = __builtin_rint(x * qinv);
    return __builtin_fma(-c, q, x);
}
__device__ __forceinline__ double fp_mulmod(double y, double w, double wq, double q)
{
    double h = y * w;
    double l = __builtin_fma(y, w, -h);
    double c = __builtin_rint(y * wq);
    return __builtin_fma(-c, q, h) + l;
}
// y * k - rint(y k / q) q for two plain integers (no precomputed quotient): with |y| <= q/2 + 1 and
// 0 <= k < q < 2^51 the estimate RN(y k) * RN(1/q) is within 0.4 of the real quotient, |result| < 0.9 q
__device__ __forceinline__ double fp_mulmod_q(double y, double k, double q, double qinv)
{
    double h = y * k;
    double l = __builtin_fma(y, k, -h);
    double c = __builtin_rint(h * qinv);
    return __builtin_fma(-c, q, h) + l;
}
// representative in [0, q) as a 64-bit integer
__device__ __forceinline__ uint64_t fp_to_canonical(double x, double q, double qinv)
{
    double r = fp_red(x, q, qinv);
    r = r < 0.0 ? r + q : r;
    return d2u(r + 4503599627370496.0) & 0x000fffffffffffffull; // 2^52 + r has r in its low 52 bits
}

template <bool RED>
__device__ __forceinline__ void ct_bfly_fp(uint64_t &xb, uint64_t &yb, uint64_t wb, uint64_t wqb, uint64_t qb, uint64_t qinvb)
{
    const double q = u2d(qb);
    double u = u2d(xb);
    if (RED)
    {
        u = fp_red(u, q, u2d(qinvb));
    }
    double v = fp_mulmod(u2d(yb), u2d(wb), u2d(wqb), q);
    xb = d2u(u + v);
    yb = d2u(u - v);
}

// The same butterfly with the twiddle given as w alone (8 bytes instead of 16): the quotient is estimated from
// RN(y w) * RN(1/q) like fp_mulmod_q.  The estimate is off by up to 3 * 2^-53 |y|, so the product comes out with
// |v| < 2q for |y| < 2^52: sixteen stages from q/2 stay below 33 q (M_FPN needs 33 q < 2^52, true for primes
// below 2^46.9); M_FPR (q < 2^51, so 2^53 > 4q) reduces BOTH operands first, which keeps |v| < 0.88 q and the outputs below
// 1.38 q.  One stage without the reductions may follow such a stage: with |u|, |y| <= 1.38 q the estimate is off by at most
// 0.5 + 3 * 2^-53 * 1.38 q < 1.54, so |v| < 1.54 q, h - c q = r - l stays an integer below 2^52 (|l| <= 2^49) and the outputs stay
// below 2.92 q < 2^53: exact.  The key switch's contiguous pass therefore reduces at every SECOND stage (its input, the
// strided pass's output, is below 2q).
template <bool RED>
__device__ __forceinline__ void ct_bfly_fp1(uint64_t &xb, uint64_t &yb, double w, double q, double qinv)
{
    double u = u2d(xb), y = u2d(yb);
    if (RED)
    {
        u = fp_red(u, q, qinv);
        y = fp_red(y, q, qinv);
    }
    double v = fp_mulmod_q(y, w, q, qinv);
    xb = d2u(u + v);
    yb = d2u(u - v);
}
// the same with the choice as an argument: a constant once the caller's stage loop is unrolled
__device__ __forceinline__ void ct_bfly_fp1_sel(uint64_t &xb, uint64_t &yb, double w, double q, double qinv, const bool red)
{
    if (red)
    {
        ct_bfly_fp1<true>(xb, yb, w, q, qinv);
    }
    else
    {
        ct_bfly_fp1<false>(xb, yb, w, q, qinv);
    }
}

// Cooley-Tukey butterfly of M_GUARD2.  The reference subtracts 2q from x when x >= 2q in every butterfly to keep
// values below 4q (dwthandler.h:94-146); the Shoup product accepts any 64-bit y and the sums grow by at most 2q per
// stage, so for q < 2^61 one guard per TWO stages is enough: an unguarded stage takes values below 6q to below 8q,
// the next one subtracts 4q when x >= 4q (u < 4q) and is back below 6q.  Residues are unchanged.
template <bool GUARDED>
__device__ __forceinline__ void ct_bfly_guard2(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2)
{
    uint64_t u = GUARDED ? csub(x, q2 << 1) : x;
    uint64_t v = mul_shoup_lazy(y, w, wq, q);
    x = u + v;
    y = u + q2 - v;
}

// ---- M_LAZY8: the integer butterfly with fewer instructions (q < 2^60) ----------------------------------------------
// Instruction counts on gfx950 decide this kernel (PMC: the integer pipes are busy 85 % of the time), and the exact
// Shoup product costs about 20 of the 31 VALU instructions of a butterfly: four partial products plus three register moves
// and a 64-bit add for hi64(y * wq), two three-multiply low products, and a 64-bit subtract that lowers to a carry pair.
// Here:
//  * the quotient takes the two HIGH cross terms only:  t' = y1 q1 + hi32(y1 q0) + hi32(y0 q1)  (y = y1:y0, wq = q1:q0).
//    The exact t = floor(y wq / 2^64) has  t - 2 <= t' <= t  (the dropped low parts of the two cross terms and hi32(y0 q0)
//    sum to less than 3 * 2^32), so  r = y w - t' q  lies in [0, 4q) instead of [0, 2q): two mul_hi, one mad, one add;
//  * r is ONE multiply-add chain modulo 2^64:  y0 w0 + t0 n0  in a 64-bit accumulator, the four cross products into its
//    high word, with n = 2^64 - q read from the per-prime record (no subtract);
//  * every stage guards  u = x - 4q if x >= 4q  by adding 2^64 - 4q and testing the sign (values stay below 8q < 2^63), so
//    x' = u + r < 8q and y' = u + 4q - r in (0, 8q): the same residues as the reference's [0, 4q) discipline.
// The pass that finishes the transform reduces below q with three such conditional subtractions.
__device__ __forceinline__ uint64_t csub_sign(uint64_t x, uint64_t neg_m)
{
    // x - m if x >= m else x, for x < 2^63 and m <= 2^62, given 2^64 - m
    uint64_t d = x + neg_m;
    return (int64_t)d < 0 ? x : d;
}
__device__ __forceinline__ uint64_t mul_shoup_approx(uint64_t y, uint64_t w, uint64_t wq, uint64_t nq)
{
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32);
    const uint32_t q0 = (uint32_t)wq, q1 = (uint32_t)(wq >> 32);
    uint64_t t = (uint64_t)y1 * q1 + __umulhi(y1, q0);
    t += __umulhi(y0, q1);
    const uint32_t t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32);
    const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
    uint64_t p = (uint64_t)y0 * w0;
    p += (uint64_t)t0 * n0;
    const uint32_t hi = (uint32_t)(p >> 32) + y0 * w1 + y1 * w0 + t0 * n1 + t1 * n0;
    return ((uint64_t)hi << 32) | (uint32_t)p;
}
// x, y below 8q -> below 8q; nq = 2^64 - q, n4q = 2^64 - 4q
__device__ __forceinline__ void ct_bfly_lazy8(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t nq, uint64_t n4q)
{
    const uint64_t u = csub_sign(x, n4q);
    const uint64_t v = mul_shoup_approx(y, w, wq, nq);
    x = u + v;
    y = u - (v + n4q);
}

// Gentleman-Sande counterparts: x, y below 4q -> below 4q (the sum is guarded, the difference goes through the product,
// which accepts any 64-bit operand); the last stage folds N^-1 in and needs no guard at all
__device__ __forceinline__ void gs_bfly_lazy8(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t nq, uint64_t n4q)
{
    const uint64_t u = x, v = y;
    x = csub_sign(u + v, n4q);
    y = mul_shoup_approx(u - (v + n4q), w, wq, nq);
}
__device__ __forceinline__ void gs_bfly_last_lazy8(uint64_t &x, uint64_t &y, const Tw &ninv, const Tw &ninv_w1, uint64_t nq, uint64_t n4q)
{
    const uint64_t u = x, v = y;
    x = mul_shoup_approx(u + v, ninv.w, ninv.wq, nq);
    y = mul_shoup_approx(u - (v + n4q), ninv_w1.w, ninv_w1.wq, nq);
}
template <bool LZ>
__device__ __forceinline__ void gs_bfly_sel(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t a, uint64_t b);

// STAGES_LEFT = number of stages after this one (compile-time in the unrolled tiles): M_GUARD2 guards the stages
// with an even number left, so the last stage of a transform is guarded and hands over values below 6q
template <int MODE, int STAGES_LEFT = 0>
__device__ __forceinline__ void ct_bfly_t(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2)
{
    if (MODE == M_LAZY8)
    {
        ct_bfly_lazy8(x, y, w, wq, q, q2); // (q, q2) carry (2^64 - q, 2^64 - 4q)
    }
    else if (MODE == M_GUARD2)
    {
        ct_bfly_guard2<(STAGES_LEFT % 2) == 0>(x, y, w, wq, q, q2);
    }
    else if (MODE == M_FPN)
    {
        ct_bfly_fp<false>(x, y, w, wq, q, q2);
    }
    else if (MODE == M_FPR)
    {
        ct_bfly_fp<true>(x, y, w, wq, q, q2);
    }
    else if (MODE == M_NOGUARD)
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

// (a, b) = (q, 2q) for the exact butterflies, (2^64 - q, 2^64 - 4q) for the M_LAZY8 ones
template <>
__device__ __forceinline__ void gs_bfly_sel<false>(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t a, uint64_t b)
{
    gs_bfly(x, y, w, wq, a, b);
}
template <>
__device__ __forceinline__ void gs_bfly_sel<true>(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t a, uint64_t b)
{
    gs_bfly_lazy8(x, y, w, wq, a, b);
}

// ---- inverse transform in exact FP64 arithmetic (primes below 2^51; the scheme of the forward M_FPN / M_FPR modes) ----------
// Gentleman-Sande butterfly on doubles holding integers: x' = u + v, y' = (u - v) w mod q with the product reduced at once
// (fp_mulmod: exact for |u - v| < 2^52).  The sums double from stage to stage, the products come out below 1.5 q:
//   FPR  (q < 2^51, 2^53 > 4q): sum and difference are reduced in every butterfly (inputs below q, outputs below 0.75 q);
//   FPN  (q < 2^52 / 25): `redsum` reduces the sum in every FOURTH stage of a pass -- from q/2 (the load folds the input) or from
//        the ~1.2 q of a product the sums stay below 10q, the differences below 19q (below 2^52, as 25q is) -- and the last stage of the transform
//        multiplies both outputs by N^-1 (gs_bfly_last_fp).
// Same exact integers as the integer butterflies, canonical at the end (fp_to_canonical): the reference's residues.
template <bool FPR>
__device__ __forceinline__ void gs_bfly_fp(uint64_t &xb, uint64_t &yb, uint64_t wb, uint64_t wqb, uint64_t qb, uint64_t qinvb, const bool redsum)
{
    const double q = u2d(qb), qi = u2d(qinvb);
    const double u = u2d(xb), v = u2d(yb);
    double s = u + v, d = u - v;
    if (FPR)
    {
        s = fp_red(s, q, qi);
        d = fp_red(d, q, qi);
    }
    else if (redsum)
    {
        s = fp_red(s, q, qi);
    }
    xb = d2u(s);
    yb = d2u(fp_mulmod(d, u2d(wb), u2d(wqb), q));
}
// last stage: x' = (u + v) N^-1, y' = (u - v) (N^-1 w_1); the constants as plain doubles, quotient from RN(1/q)
template <bool FPR>
__device__ __forceinline__ void gs_bfly_last_fp(uint64_t &xb, uint64_t &yb, double ninv, double ninv_w1, double q, double qi)
{
    const double u = u2d(xb), v = u2d(yb);
    double s = u + v, d = u - v;
    if (FPR)
    {
        s = fp_red(s, q, qi);
        d = fp_red(d, q, qi);
    }
    xb = d2u(fp_mulmod_q(s, ninv, q, qi));
    yb = d2u(fp_mulmod_q(d, ninv_w1, q, qi));
}
// IM: 0 exact integer [0, 2q), 1 M_LAZY8, 2 FPN, 3 FPR; (a, b) = the mode's two constants
template <int IM>
__device__ __forceinline__ void gs_bfly_im(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t a, uint64_t b, const bool redsum)
{
    if (IM >= 2)
    {
        gs_bfly_fp<IM == 3>(x, y, w, wq, a, b, redsum);
    }
    else
    {
        gs_bfly_sel<IM == 1>(x, y, w, wq, a, b);
    }
}

} // namespace moai
