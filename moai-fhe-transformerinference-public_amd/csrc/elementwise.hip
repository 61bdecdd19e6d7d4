// elementwise.hip -- HBM-bound RNS polynomial arithmetic: add / sub / negate / dyadic product /
// scalar rows / ciphertext tensor product / level drop / Galois gather.
//
// Reference: SEAL/util/polyarithsmallmod.cpp:18-278 and the Evaluator methods that call them
// (cited per entry point in include/moai_hip.h).  Every kernel streams rows of N coefficients with
// 16-byte accesses, one RNS prime per block row (blockIdx.y = polynomial row), so the per-prime
// constants are wave-uniform scalars.
#include <algorithm>
#include <mutex>

#include "launch.h"
#include "modarith.hip.h"

namespace moai {

struct EwArgs
{
    const uint64_t *a;
    const uint64_t *b;
    uint64_t *out;
    const PrimeConst *pc;
    uint32_t L;          // rows per polynomial
    uint32_t n2;         // N / 2 (16-byte chunks per row)
    uint32_t b_rows;     // rows of b (n_poly_b * L); row index is taken modulo this (broadcast)
};

enum EwOp
{
    EW_ADD,
    EW_SUB,
    EW_NEG,
    EW_MUL
};

template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(EwArgs g)
{
    const uint32_t row = blockIdx.y;
    const uint32_t prime = row % g.L;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q;
    const uint64_t cr0 = pc->cr0, cr1 = pc->cr1;
    const ulonglong2 *a2 = reinterpret_cast<const ulonglong2 *>(g.a) + (size_t)row * g.n2;
    const ulonglong2 *b2 = reinterpret_cast<const ulonglong2 *>(g.b) + (size_t)(row % g.b_rows) * g.n2;
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)row * g.n2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        ulonglong2 x = a2[i];
        ulonglong2 r;
        if (OP == EW_NEG)
        {
            r.x = x.x ? q - x.x : 0;
            r.y = x.y ? q - x.y : 0;
        }
        else
        {
            ulonglong2 y = b2[i];
            if (OP == EW_ADD)
            {
                r.x = csub(x.x + y.x, q);
                r.y = csub(x.y + y.y, q);
            }
            else if (OP == EW_SUB)
            {
                r.x = x.x >= y.x ? x.x - y.x : x.x + q - y.x;
                r.y = x.y >= y.y ? x.y - y.y : x.y + q - y.y;
            }
            else
            {
                r.x = mulmod_barrett(x.x, y.x, q, cr0, cr1);
                r.y = mulmod_barrett(x.y, y.y, q, cr0, cr1);
            }
        }
        o2[i] = r;
    }
}

struct ScalarArgs
{
    const uint64_t *a;
    uint64_t *out;
    const PrimeConst *pc;
    uint32_t L;
    uint32_t n2;
    Tw s[MOAI_MAX_RNS]; // per row: reduced scalar (+ Shoup quotient)
};

template <bool MUL>
__global__ __launch_bounds__(256) void scalar_rows_kernel(ScalarArgs g)
{
    const uint32_t row = blockIdx.y;
    const uint32_t prime = row % g.L;
    const uint64_t q = g.pc[prime].q;
    const Tw s = g.s[prime];
    const ulonglong2 *a2 = reinterpret_cast<const ulonglong2 *>(g.a) + (size_t)row * g.n2;
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)row * g.n2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        ulonglong2 x = a2[i];
        ulonglong2 r;
        if (MUL)
        {
            r.x = csub(mul_shoup_lazy(x.x, s.w, s.wq, q), q);
            r.y = csub(mul_shoup_lazy(x.y, s.w, s.wq, q), q);
        }
        else
        {
            r.x = csub(x.x + s.w, q);
            r.y = csub(x.y + s.w, q);
        }
        o2[i] = r;
    }
}

// out = base + sum_t x[t] (*) s[t]   (one scalar per term and RNS row): the accumulation chain of MOAI's column-packed ct x pt
// product -- multiply_plain by a scalar plaintext, add_inplace, 768 times per output column (Ct_pt_matrix_mul.hpp:19-42) --
// for up to SCALAR_DOT_TERMS terms per launch, pointers and scalars in the kernel arguments (nothing staged through memory)
constexpr int SCALAR_DOT_TERMS = 16;
constexpr int SCALAR_DOT_WORDS = 432; // 16 terms x 27 rows; fewer terms per launch above 27 rows
struct ScalarDotArgs
{
    const uint64_t *x[SCALAR_DOT_TERMS]; // each [size][L][N]
    const uint64_t *base;                // [size][L][N] or nullptr
    uint64_t *out;                       // may be `base`
    const PrimeConst *pc;
    uint32_t L, n2, terms;
    uint64_t s[SCALAR_DOT_WORDS];        // [terms][L], canonical residues
};

__global__ __launch_bounds__(256) void scalar_dot_kernel(ScalarDotArgs g)
{
    const uint32_t row = blockIdx.y; // poly * L + prime
    const uint32_t prime = row % g.L;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const ulonglong2 *b2 = g.base ? reinterpret_cast<const ulonglong2 *>(g.base) + (size_t)row * g.n2 : nullptr;
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)row * g.n2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        // sixteen products below 2^122 on top of a base below 2^61: no overflow of the 128-bit sums
        uint64_t lx = 0, hx = 0, ly = 0, hy = 0;
        if (b2)
        {
            const ulonglong2 b = b2[i];
            lx = b.x;
            ly = b.y;
        }
        ulonglong2 v[SCALAR_DOT_TERMS];
#pragma unroll
        for (int t = 0; t < SCALAR_DOT_TERMS; ++t)
        {
            if ((uint32_t)t < g.terms)
            {
                v[t] = (reinterpret_cast<const ulonglong2 *>(g.x[t]) + (size_t)row * g.n2)[i];
            }
        }
#pragma unroll
        for (int t = 0; t < SCALAR_DOT_TERMS; ++t)
        {
            if ((uint32_t)t < g.terms)
            {
                const uint64_t sc = g.s[(uint32_t)t * g.L + prime];
                uint64_t pl = v[t].x * sc, ph = mulhi64(v[t].x, sc);
                lx += pl;
                hx += ph + (lx < pl ? 1 : 0);
                pl = v[t].y * sc;
                ph = mulhi64(v[t].y, sc);
                ly += pl;
                hy += ph + (ly < pl ? 1 : 0);
            }
        }
        ulonglong2 r;
        r.x = barrett128(lx, hx, q, cr0, cr1);
        r.y = barrett128(ly, hy, q, cr0, cr1);
        o2[i] = r;
    }
}

// out = base + sum_t x[t] (*) p[t]  with full plaintexts p[t] [L][N] back to back: the accumulation chain of MOAI's MASKED ct x pt
// products (vector-encoded weights, Ct_pt_matrix_mul.hpp:103-170) over ciphertexts that sit in separate blocks
struct VectorDotArgs
{
    const uint64_t *x[SCALAR_DOT_TERMS]; // each [size][L][N]
    const uint64_t *p;                   // [terms][L][N]
    const uint64_t *base;
    uint64_t *out;
    const PrimeConst *pc;
    uint32_t L, n2, terms;
};

__global__ __launch_bounds__(256) void vector_dot_kernel(VectorDotArgs g)
{
    const uint32_t row = blockIdx.y; // poly * L + prime
    const uint32_t prime = row % g.L;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const ulonglong2 *b2 = g.base ? reinterpret_cast<const ulonglong2 *>(g.base) + (size_t)row * g.n2 : nullptr;
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)row * g.n2;
    const ulonglong2 *p2 = reinterpret_cast<const ulonglong2 *>(g.p) + (size_t)prime * g.n2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        uint64_t lx = 0, hx = 0, ly = 0, hy = 0;
        if (b2)
        {
            const ulonglong2 b = b2[i];
            lx = b.x;
            ly = b.y;
        }
        ulonglong2 v[SCALAR_DOT_TERMS], w[SCALAR_DOT_TERMS];
#pragma unroll
        for (int t = 0; t < SCALAR_DOT_TERMS; ++t)
        {
            if ((uint32_t)t < g.terms)
            {
                v[t] = (reinterpret_cast<const ulonglong2 *>(g.x[t]) + (size_t)row * g.n2)[i];
                w[t] = p2[(size_t)t * g.L * g.n2 + i];
            }
        }
#pragma unroll
        for (int t = 0; t < SCALAR_DOT_TERMS; ++t)
        {
            if ((uint32_t)t < g.terms)
            {
                uint64_t pl = v[t].x * w[t].x, ph = mulhi64(v[t].x, w[t].x);
                lx += pl;
                hx += ph + (lx < pl ? 1 : 0);
                pl = v[t].y * w[t].y;
                ph = mulhi64(v[t].y, w[t].y);
                ly += pl;
                hy += ph + (ly < pl ? 1 : 0);
            }
        }
        ulonglong2 r;
        r.x = barrett128(lx, hx, q, cr0, cr1);
        r.y = barrett128(ly, hy, q, cr0, cr1);
        o2[i] = r;
    }
}

// out[3][L][N] = base + sum_t multiply(x[t], y[t])  for size-2 ciphertexts that sit in separate blocks: the multiply + add_inplace chain
// of MOAI's ct x ct products (Ct_ct_matrix_mul.hpp:32-41, 121-134) with the pairs' pointers in the kernel arguments
struct CtDotPtrArgs
{
    const uint64_t *x[SCALAR_DOT_TERMS]; // each [2][L][N]
    const uint64_t *y[SCALAR_DOT_TERMS];
    const uint64_t *base;                // [3][L][N] or nullptr
    uint64_t *out;                       // [3][L][N], may be `base`
    const PrimeConst *pc;
    uint32_t L, n2, terms;
};

__device__ __forceinline__ void mac128(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b);
__global__ __launch_bounds__(256) void ct_dot_ptrs_kernel(CtDotPtrArgs g)
{
    const uint32_t prime = blockIdx.y;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const size_t rs = g.n2, p1 = (size_t)g.L * rs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        const size_t at = (size_t)prime * rs + i;
        uint64_t lo[6], hi[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            lo[k] = hi[k] = 0;
        }
        if (g.base)
        {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(g.base) + at;
#pragma unroll
            for (int k = 0; k < 3; ++k)
            {
                const ulonglong2 v = b[(size_t)k * p1];
                lo[2 * k] = v.x;
                lo[2 * k + 1] = v.y;
            }
        }
        // sixteen pairs: at most 32 products below 2^122 per sum on top of a base below 2^61 -- no overflow of 128 bits
#pragma unroll 4
        for (uint32_t t = 0; t < g.terms; ++t)
        {
            const ulonglong2 *xa = reinterpret_cast<const ulonglong2 *>(g.x[t]) + at;
            const ulonglong2 *ya = reinterpret_cast<const ulonglong2 *>(g.y[t]) + at;
            const ulonglong2 a0 = xa[0], a1 = xa[p1], b0 = ya[0], b1 = ya[p1];
            mac128(lo[0], hi[0], a0.x, b0.x);
            mac128(lo[1], hi[1], a0.y, b0.y);
            mac128(lo[2], hi[2], a0.x, b1.x);
            mac128(lo[2], hi[2], a1.x, b0.x);
            mac128(lo[3], hi[3], a0.y, b1.y);
            mac128(lo[3], hi[3], a1.y, b0.y);
            mac128(lo[4], hi[4], a1.x, b1.x);
            mac128(lo[5], hi[5], a1.y, b1.y);
        }
        ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + at;
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            ulonglong2 r;
            r.x = barrett128(lo[2 * k], hi[2 * k], q, cr0, cr1);
            r.y = barrett128(lo[2 * k + 1], hi[2 * k + 1], q, cr0, cr1);
            o[(size_t)k * p1] = r;
        }
    }
}

struct CtMulArgs
{
    const uint64_t *x; // [batch][2][L][N]
    const uint64_t *y; // [batch][2][L][N] (== x for the square)
    uint64_t *out;     // [batch][3][L][N]
    const PrimeConst *pc;
    uint32_t L;
    uint32_t n2;
};

// blockIdx.y = b * L + prime
template <bool SQUARE>
__global__ __launch_bounds__(256) void ct_mul_kernel(CtMulArgs g)
{
    const uint32_t b = blockIdx.y / g.L;
    const uint32_t prime = blockIdx.y % g.L;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const size_t rs = g.n2; // row stride in chunks
    const ulonglong2 *x0 = reinterpret_cast<const ulonglong2 *>(g.x) + ((size_t)(b * 2 + 0) * g.L + prime) * rs;
    const ulonglong2 *x1 = reinterpret_cast<const ulonglong2 *>(g.x) + ((size_t)(b * 2 + 1) * g.L + prime) * rs;
    const ulonglong2 *y0 = reinterpret_cast<const ulonglong2 *>(g.y) + ((size_t)(b * 2 + 0) * g.L + prime) * rs;
    const ulonglong2 *y1 = reinterpret_cast<const ulonglong2 *>(g.y) + ((size_t)(b * 2 + 1) * g.L + prime) * rs;
    ulonglong2 *o0 = reinterpret_cast<ulonglong2 *>(g.out) + ((size_t)(b * 3 + 0) * g.L + prime) * rs;
    ulonglong2 *o1 = reinterpret_cast<ulonglong2 *>(g.out) + ((size_t)(b * 3 + 1) * g.L + prime) * rs;
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(g.out) + ((size_t)(b * 3 + 2) * g.L + prime) * rs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        ulonglong2 a0 = x0[i], a1 = x1[i];
        ulonglong2 r0, r1, r2;
        if (SQUARE)
        {
            // (c0^2, 2 c0 c1, c1^2)   SEAL/evaluator.cpp:1262-1274
            uint64_t m;
            r0.x = mulmod_barrett(a0.x, a0.x, q, cr0, cr1);
            r0.y = mulmod_barrett(a0.y, a0.y, q, cr0, cr1);
            m = mulmod_barrett(a0.x, a1.x, q, cr0, cr1);
            r1.x = csub(m + m, q);
            m = mulmod_barrett(a0.y, a1.y, q, cr0, cr1);
            r1.y = csub(m + m, q);
            r2.x = mulmod_barrett(a1.x, a1.x, q, cr0, cr1);
            r2.y = mulmod_barrett(a1.y, a1.y, q, cr0, cr1);
        }
        else
        {
            // (x0 y0, x0 y1 + x1 y0, x1 y1)   SEAL/evaluator.cpp:805-860
            ulonglong2 b0 = y0[i], b1 = y1[i];
            r0.x = mulmod_barrett(a0.x, b0.x, q, cr0, cr1);
            r0.y = mulmod_barrett(a0.y, b0.y, q, cr0, cr1);
            r1.x = csub(mulmod_barrett(a0.x, b1.x, q, cr0, cr1) + mulmod_barrett(a1.x, b0.x, q, cr0, cr1), q);
            r1.y = csub(mulmod_barrett(a0.y, b1.y, q, cr0, cr1) + mulmod_barrett(a1.y, b0.y, q, cr0, cr1), q);
            r2.x = mulmod_barrett(a1.x, b1.x, q, cr0, cr1);
            r2.y = mulmod_barrett(a1.y, b1.y, q, cr0, cr1);
        }
        o0[i] = r0;
        o1[i] = r1;
        o2[i] = r2;
    }
}

struct CtMulGeneralArgs
{
    const uint64_t *x; // [batch][sx][L][N]
    const uint64_t *y; // [batch][sy][L][N]
    uint64_t *out;     // [batch][sx + sy - 1][L][N]
    const PrimeConst *pc;
    uint32_t L, n2, sx, sy;
};

// Evaluator::ckks_multiply, the branch for dest_size != 3 (SEAL/evaluator.cpp:862-900): output polynomial k is the sum over
// i + j = k of x[i] (*) y[j], every product reduced and every partial sum reduced (the order of the terms cannot matter: each
// step is exact mod q).  blockIdx.y = (b * dest + k) * L + prime.
__global__ __launch_bounds__(256) void ct_mul_general_kernel(CtMulGeneralArgs g)
{
    const uint32_t dest = g.sx + g.sy - 1;
    const uint32_t prime = blockIdx.y % g.L;
    const uint32_t k = (blockIdx.y / g.L) % dest;
    const uint32_t b = blockIdx.y / (g.L * dest);
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const size_t rs = g.n2;
    const uint32_t x_last = k < g.sx - 1 ? k : g.sx - 1;
    const uint32_t y_first = k < g.sy - 1 ? k : g.sy - 1;
    const uint32_t x_first = k - y_first;
    const ulonglong2 *x = reinterpret_cast<const ulonglong2 *>(g.x) + ((size_t)(b * g.sx + x_first) * g.L + prime) * rs;
    const ulonglong2 *y = reinterpret_cast<const ulonglong2 *>(g.y) + ((size_t)(b * g.sy + y_first) * g.L + prime) * rs;
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + ((size_t)(b * dest + k) * g.L + prime) * rs;
    const size_t poly = (size_t)g.L * rs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        ulonglong2 acc = make_ulonglong2(0, 0);
        for (uint32_t s = 0; s <= x_last - x_first; s++)
        {
            const ulonglong2 a = x[s * poly + i], c = (y - s * poly)[i];
            acc.x = csub(acc.x + mulmod_barrett(a.x, c.x, q, cr0, cr1), q);
            acc.y = csub(acc.y + mulmod_barrett(a.y, c.y, q, cr0, cr1), q);
        }
        o[i] = acc;
    }
}

// out row (p, i<Lout) = in row (p, i): blockIdx.y = p * Lout + i
__global__ __launch_bounds__(256) void drop_rows_kernel(const uint64_t *in, uint64_t *out, uint32_t Lin, uint32_t Lout,
                                                        uint32_t n2)
{
    const uint32_t p = blockIdx.y / Lout;
    const uint32_t i = blockIdx.y % Lout;
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(in) + ((size_t)p * Lin + i) * n2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(out) + ((size_t)p * Lout + i) * n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n2; j += gridDim.x * 256u)
    {
        d[j] = s[j];
    }
}

// out[row][i] = in[row][table[i]]   (apply_galois_ntt, SEAL/util/galois.cpp:192-218)
// source row of output row r: (r / L) * src_poly_rows + r % L -- src_poly_rows = L for a dense copy, 2 L to take the first
// polynomial of every size-2 ciphertext only
__global__ __launch_bounds__(256) void galois_gather_kernel(const uint64_t *in, uint64_t *out, const uint32_t *table,
                                                            uint32_t n, uint32_t L, uint32_t src_poly_rows)
{
    const uint64_t *s = in + ((size_t)(blockIdx.y / L) * src_poly_rows + blockIdx.y % L) * n;
    uint64_t *d = out + (size_t)blockIdx.y * n;
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 2u; i < n; i += gridDim.x * 512u)
    {
        uint2 t = *reinterpret_cast<const uint2 *>(table + i);
        ulonglong2 v;
        v.x = s[t.x];
        v.y = s[t.y];
        *reinterpret_cast<ulonglong2 *>(d + i) = v;
    }
}

// table[i] = bitrev_logn(((elt * bitrev_{logn+1}(N + i)) >> 1) & (N-1))   (galois.cpp:18-51)
__global__ void galois_table_kernel(uint32_t *table, int logn, uint32_t elt)
{
    const uint32_t n = 1u << logn;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        uint32_t reversed = __brev(n + i) >> (32 - (logn + 1));
        uint64_t raw = ((uint64_t)elt * (uint64_t)reversed) >> 1;
        uint32_t idx = (uint32_t)raw & (n - 1);
        table[i] = logn ? (__brev(idx) >> (32 - logn)) : 0;
    }
}


// ---- column-packed ciphertext x plaintext-matrix product with scalar weights --------------------------
// include/source/matrix_mul/Ct_pt_matrix_mul.hpp:4-49 computes, per output column c,
//     out[c] = sum_j multiply_plain(X[j], encode(W[j][c]))          (then one rescale)
// as rows*cols separate multiply_plain + add_inplace calls.  The scalar plaintexts have constant rows
// (SEAL/ckks.cpp:131-150), so the whole product is, per (polynomial, prime, coefficient),
//     out[c] = sum_j X[j] * w[j][c] mod q          with one scalar w per (j, c, prime).
// One workgroup column handles CG output columns at once: every loaded coefficient of X[j] feeds CG
// 128-bit accumulators (weights are wave-uniform scalars), reduced with one Barrett step every 32 terms
// and at the end.  X is streamed cols/CG times instead of cols times; the kernel is VALU-bound.
struct MatmulArgs
{
    const uint64_t *x;   // [rows][size][L][N]
    const uint64_t *w;   // [L][rows][cols] canonical scalar residues under prime r
    uint64_t *out;       // [cols][size][L][N]
    const PrimeConst *pc;
    uint32_t rows, cols, size, L, n2;
    const double *wd;    // the same weights as doubles (ct_pt_matmul_fp_kernel)
};

// The same sums in exact FP64 arithmetic for primes below 2^51 (every data prime of MOAI's chain): a product of two residues
// is formed as a rounded high part and its exact remainder (fp_mulmod_q) and reduced at once; the sums are folded every
// sixteen rows below 2^52 / 25 and every row otherwise, so every intermediate is an integer below 2^53 -- exact, hence the
// same canonical residues as the integer kernel.  Seven full-rate FP64 operations per product against a 64 x 64 -> 128-bit
// integer multiply-accumulate (a dozen 32-bit multiply-class and carry instructions), and half the accumulator registers.
template <int CG>
__global__ __launch_bounds__(256) void ct_pt_matmul_fp_kernel(MatmulArgs g)
{
    const uint32_t pr = blockIdx.y; // p * L + r
    const uint32_t r = pr % g.L;
    const uint32_t c0 = blockIdx.z * CG;
    const PrimeConst *pc = g.pc + r;
    const double qd = u2d(pc->qd), qinv = u2d(pc->qinv);
    const bool every_row = !(pc->q < ((1ull << 52) / 25));
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= g.n2)
    {
        return;
    }
    const size_t poly_stride = (size_t)g.size * g.L * g.n2;
    const ulonglong2 *__restrict__ x2 = reinterpret_cast<const ulonglong2 *>(g.x) + (size_t)pr * g.n2 + i;
    const double *__restrict__ wr = g.wd + (size_t)r * g.rows * g.cols + c0;
    double ax[CG], ay[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c)
    {
        ax[c] = ay[c] = 0.0;
    }
    for (uint32_t j = 0; j < g.rows; ++j)
    {
        const ulonglong2 v = x2[(size_t)j * poly_stride];
        const double vx = fp_red(fp_from_u52(v.x), qd, qinv), vy = fp_red(fp_from_u52(v.y), qd, qinv);
        const double *__restrict__ wj = wr + (size_t)j * g.cols;
        const bool fold = every_row || (j & 15u) == 15u;
#pragma unroll
        for (int c = 0; c < CG; ++c)
        {
            if (c0 + c < g.cols)
            {
                const double wv = wj[c];
                double sx = ax[c] + fp_mulmod_q(vx, wv, qd, qinv);
                double sy = ay[c] + fp_mulmod_q(vy, wv, qd, qinv);
                if (fold)
                {
                    sx = fp_red(sx, qd, qinv);
                    sy = fp_red(sy, qd, qinv);
                }
                ax[c] = sx;
                ay[c] = sy;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
    {
        if (c0 + c < g.cols)
        {
            ulonglong2 o;
            o.x = fp_to_canonical(ax[c], qd, qinv);
            o.y = fp_to_canonical(ay[c], qd, qinv);
            reinterpret_cast<ulonglong2 *>(g.out)[((size_t)(c0 + c) * g.size * g.L + pr) * g.n2 + i] = o;
        }
    }
}

__global__ __launch_bounds__(256) void u52_to_f64_kernel(const uint64_t *in, double *out, size_t count)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < count)
    {
        out[i] = fp_from_u52(in[i]);
    }
}

template <int CG>
__global__ __launch_bounds__(256) void ct_pt_matmul_kernel(MatmulArgs g)
{
    const uint32_t pr = blockIdx.y;            // p * L + r
    const uint32_t r = pr % g.L;
    const uint32_t c0 = blockIdx.z * CG;
    const PrimeConst *pc = g.pc + r;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= g.n2)
    {
        return;
    }
    const size_t poly_stride = (size_t)g.size * g.L * g.n2; // chunks per input ciphertext
    const ulonglong2 *__restrict__ x2 = reinterpret_cast<const ulonglong2 *>(g.x) + (size_t)pr * g.n2 + i;
    const uint64_t *__restrict__ wr = g.w + (size_t)r * g.rows * g.cols + c0;
    uint64_t lox[CG], hix[CG], loy[CG], hiy[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c)
    {
        lox[c] = hix[c] = loy[c] = hiy[c] = 0;
    }
    for (uint32_t j = 0; j < g.rows; ++j)
    {
        const ulonglong2 v = x2[(size_t)j * poly_stride];
        const uint64_t *__restrict__ wj = wr + (size_t)j * g.cols;
#pragma unroll
        for (int c = 0; c < CG; ++c)
        {
            if (c0 + c < g.cols)
            {
                const uint64_t wv = wj[c];
                uint64_t pl = v.x * wv, ph = mulhi64(v.x, wv);
                lox[c] += pl;
                hix[c] += ph + (lox[c] < pl ? 1 : 0);
                pl = v.y * wv;
                ph = mulhi64(v.y, wv);
                loy[c] += pl;
                hiy[c] += ph + (loy[c] < pl ? 1 : 0);
            }
        }
        if ((j & 31u) == 31u)
        {
            // 32 products below 2^122 each: fold back below q before the accumulator can overflow
#pragma unroll
            for (int c = 0; c < CG; ++c)
            {
                lox[c] = barrett128(lox[c], hix[c], q, cr0, cr1);
                hix[c] = 0;
                loy[c] = barrett128(loy[c], hiy[c], q, cr0, cr1);
                hiy[c] = 0;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
    {
        if (c0 + c < g.cols)
        {
            ulonglong2 o;
            o.x = barrett128(lox[c], hix[c], q, cr0, cr1);
            o.y = barrett128(loy[c], hiy[c], q, cr0, cr1);
            reinterpret_cast<ulonglong2 *>(g.out)[((size_t)(c0 + c) * g.size * g.L + pr) * g.n2 + i] = o;
        }
    }
}

__device__ __forceinline__ void mac128(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b)
{
    uint64_t pl = a * b;
    uint64_t ph = mulhi64(a, b);
    lo += pl;
    hi += ph + (lo < pl ? 1 : 0);
}

// sum_j x[j] (*) y[j] with (*) = ckks_multiply of two size-2 ciphertexts (SEAL/evaluator.cpp:805-860) and the sum
// = add_inplace (:155-240): the inner loop of MOAI's ciphertext-ciphertext products
// (include/source/matrix_mul/Ct_ct_matrix_mul.hpp:33-42, 117-131).  The reference reduces every product and
// every sum; the canonical residues of the total do not depend on when the reductions happen, so the three
// components are accumulated in 128 bits and folded every 16 terms (32 products below 2^122 in the middle
// component).  HBM bound: 4 rows read per term, 3 written at the end.
struct CtDotArgs
{
    const uint64_t *x; // [count][2][L][N]
    const uint64_t *y; // [count][2][L][N]
    uint64_t *out;     // [3][L][N]
    const PrimeConst *pc;
    uint32_t count, L, n2;
};

__global__ __launch_bounds__(256) void ct_dot_kernel(CtDotArgs g)
{
    const uint32_t prime = blockIdx.y;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= g.n2)
    {
        return;
    }
    const size_t rs = g.n2;
    const size_t ct_stride = (size_t)2 * g.L * rs;
    const ulonglong2 *__restrict__ x0 = reinterpret_cast<const ulonglong2 *>(g.x) + (size_t)prime * rs + i;
    const ulonglong2 *__restrict__ y0 = reinterpret_cast<const ulonglong2 *>(g.y) + (size_t)prime * rs + i;
    const size_t p1 = (size_t)g.L * rs; // second polynomial of a ciphertext
    uint64_t lo[6], hi[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
    {
        lo[k] = hi[k] = 0;
    }
    for (uint32_t j = 0; j < g.count; ++j)
    {
        const ulonglong2 a0 = x0[j * ct_stride], a1 = x0[j * ct_stride + p1];
        const ulonglong2 b0 = y0[j * ct_stride], b1 = y0[j * ct_stride + p1];
        mac128(lo[0], hi[0], a0.x, b0.x);
        mac128(lo[1], hi[1], a0.y, b0.y);
        mac128(lo[2], hi[2], a0.x, b1.x);
        mac128(lo[2], hi[2], a1.x, b0.x);
        mac128(lo[3], hi[3], a0.y, b1.y);
        mac128(lo[3], hi[3], a1.y, b0.y);
        mac128(lo[4], hi[4], a1.x, b1.x);
        mac128(lo[5], hi[5], a1.y, b1.y);
        if ((j & 15u) == 15u)
        {
#pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                lo[k] = barrett128(lo[k], hi[k], q, cr0, cr1);
                hi[k] = 0;
            }
        }
    }
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)prime * rs + i;
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        ulonglong2 r;
        r.x = barrett128(lo[2 * k], hi[2 * k], q, cr0, cr1);
        r.y = barrett128(lo[2 * k + 1], hi[2 * k + 1], q, cr0, cr1);
        o[(size_t)k * p1] = r;
    }
}

// out[poly] = sum_t x[xi[t]][poly] (.) p[pi[t]]: multiply_plain (SEAL/evaluator.cpp:2336-2373) of several
// ciphertext operands with several NTT-form plaintexts, accumulated with add_inplace -- the inner loop of the
// baby-step / giant-step linear transforms of MOAI's bootstrapping
// (include/source/bootstrapping/Bootstrapper.cpp:2028-2046).  Operand k is the block x + k * n_poly * L * N
// (a batch of ciphertexts, all polynomials of all of them: n_poly = batch * size); every plaintext is shared by
// the whole batch.  Lazy 128-bit accumulation, folded every 32 terms; same canonical residues as the
// reference's multiply-reduce-add sequence.
constexpr int CTPT_MAX_TERMS = 64;
struct CtPtDotArgs
{
    const uint64_t *x;
    const uint64_t *p;   // [n_pt][L][N]
    uint64_t *out;       // [n_poly][L][N]
    const PrimeConst *pc;
    uint32_t terms, L, n2, n_poly;
    uint32_t xi[CTPT_MAX_TERMS]; // 32-bit entries: read per term through scalar loads (see RowMap)
    uint32_t pi[CTPT_MAX_TERMS];
    // a second sum over the leading terms2 <= terms operands with its own plaintexts (moai_ct_pt_dot2): the operands are
    // loaded once for both
    uint64_t *out2;
    uint32_t terms2;
    uint32_t pi2[CTPT_MAX_TERMS];
};

// P polynomials per thread: a plaintext value is loaded once and multiplied into P ciphertext polynomials (a plaintext row is
// shared by the whole batch -- with one polynomial per thread it was fetched n_poly times).  blockIdx.y = group * L + prime, the
// group holds polynomials group * P .. group * P + P - 1 (the last group may be short).
// The sums stay on the integer units also for primes below 2^51: the lazy 128-bit accumulation pays one Barrett step per 32 terms,
// an exact FP64 product has to be reduced term by term (fp_mulmod_q, 7 operations at the 32-bit multiply rate).  Measured in
// round 3 on the bootstrap's baby-step sums, pack 48: 3.42 ms per launch in doubles against 2.90 ms here -- not kept.
template <int P, bool TWO>
__global__ __launch_bounds__(256) void ct_pt_dot_kernel(CtPtDotArgs g)
{
    const uint32_t prime = blockIdx.y % g.L;
    const uint32_t poly0 = (blockIdx.y / g.L) * P;
    const uint32_t np = g.n_poly - poly0 < (uint32_t)P ? g.n_poly - poly0 : (uint32_t)P; // uniform over the block
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= g.n2)
    {
        return;
    }
    const size_t op_stride = (size_t)g.n_poly * g.L * g.n2;
    const size_t pt_stride = (size_t)g.L * g.n2;
    const size_t poly_stride = (size_t)g.L * g.n2;
    const size_t row0 = ((size_t)poly0 * g.L + prime) * g.n2 + i;
    const ulonglong2 *__restrict__ xb = reinterpret_cast<const ulonglong2 *>(g.x) + row0;
    const ulonglong2 *__restrict__ pb = reinterpret_cast<const ulonglong2 *>(g.p) + (size_t)prime * g.n2 + i;
    constexpr int S = TWO ? 2 : 1;
    uint64_t lo0[S][P], hi0[S][P], lo1[S][P], hi1[S][P];
#pragma unroll
    for (int s = 0; s < S; ++s)
    {
#pragma unroll
        for (int k = 0; k < P; ++k)
        {
            lo0[s][k] = hi0[s][k] = lo1[s][k] = hi1[s][k] = 0;
        }
    }
    for (uint32_t t = 0; t < g.terms; ++t)
    {
        // the term's three operand indices first (scalar loads, one wait), then its six 16-byte operands in one batch.
        // (Requesting term t + 1's operands before term t's products -- 24 more registers, three waves per SIMD instead of
        // four -- measured the same: 2976 against 2900-3017 us per launch on the bootstrap's baby-step sums.)
        const uint32_t ip = g.pi[t], ip2 = TWO ? g.pi2[t] : 0u, ix = g.xi[t];
        const ulonglong2 b = pb[(size_t)ip * pt_stride];
        const bool second = TWO && t < g.terms2; // uniform
        ulonglong2 b2 = make_ulonglong2(0, 0);
        if (second)
        {
            b2 = pb[(size_t)ip2 * pt_stride];
        }
        const ulonglong2 *xt = xb + (size_t)ix * op_stride;
        ulonglong2 a[P];
#pragma unroll
        for (int k = 0; k < P; ++k)
        {
            a[k] = (uint32_t)k < np ? xt[(size_t)k * poly_stride] : make_ulonglong2(0, 0);
        }
#pragma unroll
        for (int k = 0; k < P; ++k)
        {
            mac128(lo0[0][k], hi0[0][k], a[k].x, b.x);
            mac128(lo1[0][k], hi1[0][k], a[k].y, b.y);
            if (TWO)
            {
                mac128(lo0[S - 1][k], hi0[S - 1][k], a[k].x, b2.x); // b2 = 0 past terms2
                mac128(lo1[S - 1][k], hi1[S - 1][k], a[k].y, b2.y);
            }
        }
        if ((t & 31u) == 31u)
        {
#pragma unroll
            for (int s = 0; s < S; ++s)
            {
#pragma unroll
                for (int k = 0; k < P; ++k)
                {
                    lo0[s][k] = barrett128(lo0[s][k], hi0[s][k], q, cr0, cr1);
                    lo1[s][k] = barrett128(lo1[s][k], hi1[s][k], q, cr0, cr1);
                    hi0[s][k] = hi1[s][k] = 0;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
    {
        ulonglong2 *ob = reinterpret_cast<ulonglong2 *>(s == 0 ? g.out : g.out2) + row0;
#pragma unroll
        for (int k = 0; k < P; ++k)
        {
            if ((uint32_t)k < np)
            {
                ulonglong2 r;
                r.x = barrett128(lo0[s][k], hi0[s][k], q, cr0, cr1);
                r.y = barrett128(lo1[s][k], hi1[s][k], q, cr0, cr1);
                ob[(size_t)k * poly_stride] = r;
            }
        }
    }
}

// out = sum over ALL r < rows of x[r] (.) p[r] (and the same with a second plaintext set): the column of a ciphertext x
// plaintext matrix product whose weights are full plaintexts (include/source/matrix_mul/Ct_pt_matrix_mul.hpp:103-170), one
// multiply_plain + add_inplace per row in the reference.  The output is a handful of rows, so the parallelism comes from the
// sum: blockIdx.z takes a slice of the rows and leaves a canonical partial sum, ct_pt_rowsum_reduce adds the slices.
struct RowSumArgs
{
    const uint64_t *x;  // [rows][n_poly][L][N]
    const uint64_t *p;  // [rows][L][N]
    const uint64_t *p2; // second plaintext set or null
    uint64_t *part;     // [splits][1 or 2][n_poly][L][N]
    uint64_t *out, *out2;
    const PrimeConst *pc;
    uint32_t rows, L, n2, n_poly, splits;
};

template <bool TWO>
__global__ __launch_bounds__(256) void ct_pt_rowsum_kernel(RowSumArgs g)
{
    const uint32_t row = blockIdx.y; // poly * L + prime
    const uint32_t prime = row % g.L;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= g.n2)
    {
        return;
    }
    const uint32_t r0 = (uint32_t)(((uint64_t)g.rows * blockIdx.z) / g.splits);
    const uint32_t r1 = (uint32_t)(((uint64_t)g.rows * (blockIdx.z + 1)) / g.splits);
    const size_t x_stride = (size_t)g.n_poly * g.L * g.n2, p_stride = (size_t)g.L * g.n2;
    const ulonglong2 *__restrict__ xb = reinterpret_cast<const ulonglong2 *>(g.x) + (size_t)row * g.n2 + i;
    const ulonglong2 *__restrict__ pb = reinterpret_cast<const ulonglong2 *>(g.p) + (size_t)prime * g.n2 + i;
    const ulonglong2 *__restrict__ pb2 = TWO ? reinterpret_cast<const ulonglong2 *>(g.p2) + (size_t)prime * g.n2 + i : nullptr;
    uint64_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, mo0 = 0, mi0 = 0, mo1 = 0, mi1 = 0;
    uint32_t since_fold = 0;
#pragma unroll 4
    for (uint32_t r = r0; r < r1; ++r)
    {
        const ulonglong2 a = xb[(size_t)r * x_stride];
        const ulonglong2 b = pb[(size_t)r * p_stride];
        mac128(lo0, hi0, a.x, b.x);
        mac128(lo1, hi1, a.y, b.y);
        if (TWO)
        {
            const ulonglong2 b2 = pb2[(size_t)r * p_stride];
            mac128(mo0, mi0, a.x, b2.x);
            mac128(mo1, mi1, a.y, b2.y);
        }
        if (++since_fold == 32u) // 32 products below 2^122 fit 128 bits for primes of at most 61 bits
        {
            since_fold = 0;
            lo0 = barrett128(lo0, hi0, q, cr0, cr1);
            lo1 = barrett128(lo1, hi1, q, cr0, cr1);
            hi0 = hi1 = 0;
            if (TWO)
            {
                mo0 = barrett128(mo0, mi0, q, cr0, cr1);
                mo1 = barrett128(mo1, mi1, q, cr0, cr1);
                mi0 = mi1 = 0;
            }
        }
    }
    const size_t plane = (size_t)g.n_poly * g.L * g.n2;
    ulonglong2 *pp = reinterpret_cast<ulonglong2 *>(g.part) + (size_t)blockIdx.z * (TWO ? 2 : 1) * plane + (size_t)row * g.n2 + i;
    ulonglong2 v;
    v.x = barrett128(lo0, hi0, q, cr0, cr1);
    v.y = barrett128(lo1, hi1, q, cr0, cr1);
    pp[0] = v;
    if (TWO)
    {
        v.x = barrett128(mo0, mi0, q, cr0, cr1);
        v.y = barrett128(mo1, mi1, q, cr0, cr1);
        pp[plane] = v;
    }
}

// blockIdx.z = which output (0 / 1)
__global__ __launch_bounds__(256) void ct_pt_rowsum_reduce(RowSumArgs g)
{
    const uint32_t row = blockIdx.y;
    const uint64_t q = g.pc[row % g.L].q;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= g.n2)
    {
        return;
    }
    const uint32_t sets = g.p2 ? 2u : 1u;
    const size_t plane = (size_t)g.n_poly * g.L * g.n2;
    const ulonglong2 *pp = reinterpret_cast<const ulonglong2 *>(g.part) + (size_t)blockIdx.z * plane + (size_t)row * g.n2 + i;
    ulonglong2 acc = pp[0];
    for (uint32_t z = 1; z < g.splits; ++z)
    {
        const ulonglong2 v = pp[(size_t)z * sets * plane];
        acc.x = csub(acc.x + v.x, q);
        acc.y = csub(acc.y + v.y, q);
    }
    (reinterpret_cast<ulonglong2 *>(blockIdx.z ? g.out2 : g.out) + (size_t)row * g.n2)[i] = acc;
}

static inline dim3 row_grid(const moai_ctx *c, size_t rows, uint32_t per_thread_chunks = 1)
{
    uint32_t n2 = (uint32_t)(c->n >> 1);
    uint32_t bx = (n2 + 256u * per_thread_chunks - 1) / (256u * per_thread_chunks);
    if (bx == 0)
    {
        bx = 1;
    }
    return dim3(bx, (uint32_t)rows);
}

static int check_rows(const moai_ctx *c, size_t n_poly, size_t L)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    if (L > c->k || L > MOAI_MAX_RNS)
    {
        return set_error(MOAI_EINVAL, "L = %zu exceeds the context's %zu primes", L, c->k);
    }
    if (n_poly * L > 0x7fffffffull)
    {
        return set_error(MOAI_EINVAL, "batch too large for one launch");
    }
    return enter_device(c);
}

int galois_table(moai_ctx *c, uint32_t elt, hipStream_t s, const uint32_t **out)
{
    if (!(elt & 1u) || elt >= 2 * c->n)
    {
        return set_error(MOAI_EINVAL, "Galois element is not valid");
    }
    std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
    size_t idx = (elt - 1) >> 1; // GaloisTool::GetIndexFromElt, SEAL/util/galois.h
    if (!c->galois_tables[idx])
    {
        uint32_t *t = nullptr;
        MOAI_HIP_CHECK(hipMalloc(&t, sizeof(uint32_t) * c->n));
        // built on the NULL stream and completed before first use by any stream
        hipLaunchKernelGGL(galois_table_kernel, dim3((uint32_t)((c->n + 255) / 256)), dim3(256), 0, 0, t, c->logn, elt);
        MOAI_LAUNCH_CHECK();
        MOAI_HIP_CHECK(hipStreamSynchronize(0));
        c->galois_tables[idx] = t;
    }
    (void)s;
    *out = c->galois_tables[idx];
    return MOAI_OK;
}

int workspace(moai_ctx *c, size_t bytes, hipStream_t s, void **out)
{
    // per-stream arena; a first-time or larger request reallocates with headroom (max of 1.25 x the request and
    // 1.5 x the old size), which synchronises the device and therefore must not happen under stream capture
    // (moai_ctx_reserve[_stream] sizes the arena beforehand)
    void *p = nullptr;
    int rc = reserve_for_stream(c, (void *)s, bytes ? bytes : 256, &p, true);
    if (rc)
    {
        return rc;
    }
    *out = p;
    return MOAI_OK;
}

template <int OP>
static int ew_launch(moai_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly, size_t n_poly_b,
                     size_t L, void *stream)
{
    int rc = check_rows(c, n_poly, L);
    if (rc)
    {
        return rc;
    }
    if (n_poly == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!a || !out || (OP != EW_NEG && !b))
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    EwArgs g;
    g.a = a;
    g.b = b ? b : a;
    g.out = out;
    g.pc = c->pc;
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    g.b_rows = (uint32_t)(n_poly_b * L);
    MOAI_CHECK_GRID_ROWS(n_poly * L);
    hipLaunchKernelGGL(ew_kernel<OP>, row_grid(c, n_poly * L), dim3(256), 0, (hipStream_t)stream, g);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

} // namespace moai

using namespace moai;

extern "C" int moai_add(moai_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly, size_t L,
                        void *stream)
{
    MOAI_AUDIT(stream, a, b, out);
    trace_op("add", L, n_poly);
    return ew_launch<EW_ADD>(c, a, b, out, n_poly, n_poly, L, stream);
}

extern "C" int moai_sub(moai_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly, size_t L,
                        void *stream)
{
    MOAI_AUDIT(stream, a, b, out);
    trace_op("sub", L, n_poly);
    return ew_launch<EW_SUB>(c, a, b, out, n_poly, n_poly, L, stream);
}

extern "C" int moai_negate(moai_ctx *c, const uint64_t *a, uint64_t *out, size_t n_poly, size_t L, void *stream)
{
    MOAI_AUDIT(stream, a, out);
    trace_op("negate", L, n_poly);
    return ew_launch<EW_NEG>(c, a, nullptr, out, n_poly, n_poly, L, stream);
}

extern "C" int moai_dyadic_mul(moai_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly,
                               size_t n_poly_b, size_t L, void *stream)
{
    MOAI_AUDIT(stream, a, b, out);
    trace_op("dyadic_mul", L, n_poly);
    if (n_poly_b != n_poly && n_poly_b != 1)
    {
        return set_error(MOAI_EINVAL, "n_poly_b must be n_poly or 1");
    }
    return ew_launch<EW_MUL>(c, a, b, out, n_poly, n_poly_b, L, stream);
}

static int scalar_rows(moai_ctx *c, const uint64_t *a, const uint64_t *scalars, uint64_t *out, size_t n_poly, size_t L,
                       void *stream, bool mul)
{
    int rc = check_rows(c, n_poly, L);
    if (rc)
    {
        return rc;
    }
    if (n_poly == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!a || !out || !scalars)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    ScalarArgs g;
    g.a = a;
    g.out = out;
    g.pc = c->pc;
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    for (size_t r = 0; r < L; r++)
    {
        uint64_t q = c->primes[r];
        uint64_t s = scalars[r] % q; // barrett_reduce_64 in polyarithsmallmod.h:209-217
        g.s[r].w = s;
        g.s[r].wq = (uint64_t)((((unsigned __int128)s) << 64) / q);
    }
    if (mul)
    {
        MOAI_CHECK_GRID_ROWS(n_poly * L);
        hipLaunchKernelGGL(scalar_rows_kernel<true>, row_grid(c, n_poly * L), dim3(256), 0, (hipStream_t)stream, g);
    }
    else
    {
        MOAI_CHECK_GRID_ROWS(n_poly * L);
        hipLaunchKernelGGL(scalar_rows_kernel<false>, row_grid(c, n_poly * L), dim3(256), 0, (hipStream_t)stream, g);
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_mul_scalar_rows(moai_ctx *c, const uint64_t *a, const uint64_t *scalars, uint64_t *out,
                                    size_t n_poly, size_t L, void *stream)
{
    MOAI_AUDIT(stream, a, scalars, out);
    trace_op("mul_scalar_rows", L, n_poly);
    return scalar_rows(c, a, scalars, out, n_poly, L, stream, true);
}

extern "C" int moai_scalar_dot(moai_ctx *c, const uint64_t *const *x, const uint64_t *scalars, size_t terms, const uint64_t *base,
                               uint64_t *out, size_t size, size_t L, void *stream)
{
    MOAI_AUDIT(stream, base, out);
    for (size_t t = 0; x && t < terms; ++t)
    {
        MOAI_AUDIT(stream, x[t]);
    }
    trace_op("ct_pt_dot", L, terms * size); // what the reference does per term: multiply_plain + add_inplace
    int rc = check_rows(c, size, L);
    if (rc)
    {
        return rc;
    }
    if (size == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!out || (terms && (!x || !scalars)))
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    for (size_t t = 0; t < terms; ++t)
    {
        if (!x[t] || x[t] == out)
        {
            return set_error(MOAI_EINVAL, "null term, or a term that is the output");
        }
        for (size_t r = 0; r < L; ++r)
        {
            if (scalars[t * L + r] >= c->primes[r])
            {
                return set_error(MOAI_EINVAL, "scalar not reduced modulo its prime");
            }
        }
    }
    const size_t per = std::min<size_t>(SCALAR_DOT_TERMS, SCALAR_DOT_WORDS / L);
    MOAI_CHECK_GRID_ROWS(size * L);
    size_t t0 = 0;
    do
    {
        const size_t cnt = std::min(per, terms - t0);
        ScalarDotArgs g;
        for (size_t t = 0; t < (size_t)SCALAR_DOT_TERMS; ++t)
        {
            g.x[t] = t < cnt ? x[t0 + t] : nullptr;
        }
        g.base = t0 == 0 ? base : out;
        g.out = out;
        g.pc = c->pc;
        g.L = (uint32_t)L;
        g.n2 = (uint32_t)(c->n >> 1);
        g.terms = (uint32_t)cnt;
        for (size_t w = 0; w < cnt * L; ++w)
        {
            g.s[w] = scalars[t0 * L + w];
        }
        hipLaunchKernelGGL(scalar_dot_kernel, row_grid(c, size * L), dim3(256), 0, (hipStream_t)stream, g);
        MOAI_LAUNCH_CHECK();
        t0 += cnt;
    } while (t0 < terms);
    return MOAI_OK;
}

extern "C" int moai_ct_dot_ptrs(moai_ctx *c, const uint64_t *const *x, const uint64_t *const *y, size_t terms, const uint64_t *base, uint64_t *out,
                                size_t L, void *stream)
{
    MOAI_AUDIT(stream, base, out);
    for (size_t t = 0; x && y && t < terms; ++t)
    {
        MOAI_AUDIT(stream, x[t], y[t]);
    }
    trace_op("ct_dot", L, terms);
    int rc = check_rows(c, 3, L);
    if (rc)
    {
        return rc;
    }
    if (L == 0)
    {
        return MOAI_OK;
    }
    if (!out || (terms && (!x || !y)))
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    for (size_t t = 0; t < terms; ++t)
    {
        if (!x[t] || !y[t] || x[t] == out || y[t] == out)
        {
            return set_error(MOAI_EINVAL, "null operand, or an operand that is the output");
        }
    }
    MOAI_CHECK_GRID_ROWS(L);
    size_t t0 = 0;
    do
    {
        const size_t cnt = std::min<size_t>(SCALAR_DOT_TERMS, terms - t0);
        CtDotPtrArgs g;
        for (size_t t = 0; t < (size_t)SCALAR_DOT_TERMS; ++t)
        {
            g.x[t] = t < cnt ? x[t0 + t] : nullptr;
            g.y[t] = t < cnt ? y[t0 + t] : nullptr;
        }
        g.base = t0 == 0 ? base : out;
        g.out = out;
        g.pc = c->pc;
        g.L = (uint32_t)L;
        g.n2 = (uint32_t)(c->n >> 1);
        g.terms = (uint32_t)cnt;
        hipLaunchKernelGGL(ct_dot_ptrs_kernel, row_grid(c, L), dim3(256), 0, (hipStream_t)stream, g);
        MOAI_LAUNCH_CHECK();
        t0 += cnt;
    } while (t0 < terms);
    return MOAI_OK;
}

extern "C" int moai_vector_dot(moai_ctx *c, const uint64_t *const *x, const uint64_t *p, size_t terms, const uint64_t *base, uint64_t *out,
                               size_t size, size_t L, void *stream)
{
    MOAI_AUDIT(stream, p, base, out);
    for (size_t t = 0; x && t < terms; ++t)
    {
        MOAI_AUDIT(stream, x[t]);
    }
    trace_op("ct_pt_dot", L, terms * size);
    int rc = check_rows(c, size, L);
    if (rc)
    {
        return rc;
    }
    if (size == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!out || (terms && (!x || !p)))
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    for (size_t t = 0; t < terms; ++t)
    {
        if (!x[t] || x[t] == out)
        {
            return set_error(MOAI_EINVAL, "null term, or a term that is the output");
        }
    }
    MOAI_CHECK_GRID_ROWS(size * L);
    size_t t0 = 0;
    do
    {
        const size_t cnt = std::min<size_t>(SCALAR_DOT_TERMS, terms - t0);
        VectorDotArgs g;
        for (size_t t = 0; t < (size_t)SCALAR_DOT_TERMS; ++t)
        {
            g.x[t] = t < cnt ? x[t0 + t] : nullptr;
        }
        g.p = p + t0 * L * c->n;
        g.base = t0 == 0 ? base : out;
        g.out = out;
        g.pc = c->pc;
        g.L = (uint32_t)L;
        g.n2 = (uint32_t)(c->n >> 1);
        g.terms = (uint32_t)cnt;
        hipLaunchKernelGGL(vector_dot_kernel, row_grid(c, size * L), dim3(256), 0, (hipStream_t)stream, g);
        MOAI_LAUNCH_CHECK();
        t0 += cnt;
    } while (t0 < terms);
    return MOAI_OK;
}

extern "C" int moai_add_scalar_rows(moai_ctx *c, const uint64_t *a, const uint64_t *scalars, uint64_t *out,
                                    size_t n_poly, size_t L, void *stream)
{
    MOAI_AUDIT(stream, a, scalars, out);
    trace_op("add_scalar_rows", L, n_poly);
    return scalar_rows(c, a, scalars, out, n_poly, L, stream, false);
}

static int ct_mul(moai_ctx *c, const uint64_t *x, const uint64_t *y, uint64_t *out, size_t L, size_t batch, void *stream,
                  bool square)
{
    int rc = check_rows(c, batch * 3, L);
    if (rc)
    {
        return rc;
    }
    if (batch == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!x || !y || !out)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    if (out == x || out == y)
    {
        return set_error(MOAI_EINVAL, "out must not alias an input");
    }
    CtMulArgs g;
    g.x = x;
    g.y = y;
    g.out = out;
    g.pc = c->pc;
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    if (square)
    {
        MOAI_CHECK_GRID_ROWS(batch * L);
        hipLaunchKernelGGL(ct_mul_kernel<true>, row_grid(c, batch * L), dim3(256), 0, (hipStream_t)stream, g);
    }
    else
    {
        MOAI_CHECK_GRID_ROWS(batch * L);
        hipLaunchKernelGGL(ct_mul_kernel<false>, row_grid(c, batch * L), dim3(256), 0, (hipStream_t)stream, g);
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_ct_multiply(moai_ctx *c, const uint64_t *x, const uint64_t *y, uint64_t *out, size_t L,
                                size_t batch, void *stream)
{
    MOAI_AUDIT(stream, x, y, out);
    trace_op("ct_multiply", L, batch);
    return ct_mul(c, x, y, out, L, batch, stream, false);
}

extern "C" int moai_ct_square(moai_ctx *c, const uint64_t *x, uint64_t *out, size_t L, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, x, out);
    trace_op("ct_square", L, batch);
    return ct_mul(c, x, x, out, L, batch, stream, true);
}

extern "C" int moai_ct_multiply_general(moai_ctx *c, const uint64_t *x, size_t size_x, const uint64_t *y, size_t size_y,
                                        uint64_t *out, size_t L, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, x, y, out);
    trace_op("ct_multiply_general", L, batch);
    if (size_x < 2 || size_y < 2 || size_x > 16 || size_y > 16 || size_x + size_y - 1 > 16)
    {
        // SEAL_CIPHERTEXT_SIZE_MIN / _MAX (SEAL/util/defines.h) bound both operands and the product
        return set_error(MOAI_EINVAL, "ciphertext sizes must be 2..16 and their product at most 16 polynomials");
    }
    const size_t dest = size_x + size_y - 1;
    int rc = check_rows(c, batch * dest, L);
    if (rc)
    {
        return rc;
    }
    if (batch == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!x || !y || !out)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    if (out == x || out == y)
    {
        return set_error(MOAI_EINVAL, "out must not alias an input");
    }
    CtMulGeneralArgs g;
    g.x = x;
    g.y = y;
    g.out = out;
    g.pc = c->pc;
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    g.sx = (uint32_t)size_x;
    g.sy = (uint32_t)size_y;
    MOAI_CHECK_GRID_ROWS(batch * dest * L);
    hipLaunchKernelGGL(ct_mul_general_kernel, row_grid(c, batch * dest * L), dim3(256), 0, (hipStream_t)stream, g);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_ct_dot(moai_ctx *c, const uint64_t *x, const uint64_t *y, uint64_t *out, size_t count, size_t L,
                           void *stream)
{
    MOAI_AUDIT(stream, x, y, out);
    trace_op("ct_dot", L, count);
    int rc = check_rows(c, count * 2, L);
    if (rc)
    {
        return rc;
    }
    if (count == 0)
    {
        return set_error(MOAI_EINVAL, "empty sum");
    }
    if (!x || !y || !out)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    for (size_t r = 0; r < L; r++)
    {
        if (c->primes[r] >> 61)
        {
            return set_error(MOAI_ELOGIC, "lazy accumulation needs primes of at most 61 bits");
        }
    }
    CtDotArgs g;
    g.x = x;
    g.y = y;
    g.out = out;
    g.pc = c->pc;
    g.count = (uint32_t)count;
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    MOAI_CHECK_GRID_ROWS(L);
    hipLaunchKernelGGL(ct_dot_kernel, row_grid(c, L), dim3(256), 0, (hipStream_t)stream, g);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

static int ct_pt_dot_common(moai_ctx *c, const uint64_t *x, const uint64_t *p, uint64_t *out, uint64_t *out2, const uint32_t *x_index,
                            const uint32_t *p_index, const uint32_t *p_index2, size_t terms, size_t terms2, size_t n_poly, size_t L,
                            void *stream)
{
    int rc = check_rows(c, n_poly, L);
    if (rc)
    {
        return rc;
    }
    if (terms == 0 || terms > CTPT_MAX_TERMS || (out2 && (terms2 == 0 || terms2 > terms)))
    {
        return set_error(MOAI_EINVAL, "between 1 and 64 terms per call (the second sum over a leading part of them)");
    }
    if (n_poly == 0)
    {
        return MOAI_OK;
    }
    if (!x || !p || !out || !x_index || !p_index || (out2 && (!p_index2 || out2 == out)))
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    for (size_t r = 0; r < L; r++)
    {
        if (c->primes[r] >> 61)
        {
            return set_error(MOAI_ELOGIC, "lazy accumulation needs primes of at most 61 bits");
        }
    }
    CtPtDotArgs g;
    g.x = x;
    g.p = p;
    g.out = out;
    g.out2 = out2;
    g.pc = c->pc;
    g.terms = (uint32_t)terms;
    g.terms2 = (uint32_t)(out2 ? terms2 : 0);
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    g.n_poly = (uint32_t)n_poly;
    for (size_t t = 0; t < terms; t++)
    {
        if (x_index[t] > 0xffffu || p_index[t] > 0xffffu || (out2 && t < terms2 && p_index2[t] > 0xffffu))
        {
            return set_error(MOAI_EINVAL, "operand index out of range");
        }
        g.xi[t] = x_index[t];
        g.pi[t] = p_index[t];
        g.pi2[t] = out2 && t < terms2 ? p_index2[t] : 0;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n_poly >= 4)
    {
        const size_t groups = (n_poly + 3) / 4;
        MOAI_CHECK_GRID_ROWS(groups * L);
        if (out2)
        {
            hipLaunchKernelGGL((ct_pt_dot_kernel<4, true>), row_grid(c, groups * L), dim3(256), 0, s, g);
        }
        else
        {
            hipLaunchKernelGGL((ct_pt_dot_kernel<4, false>), row_grid(c, groups * L), dim3(256), 0, s, g);
        }
    }
    else
    {
        MOAI_CHECK_GRID_ROWS(n_poly * L);
        if (out2)
        {
            hipLaunchKernelGGL((ct_pt_dot_kernel<1, true>), row_grid(c, n_poly * L), dim3(256), 0, s, g);
        }
        else
        {
            hipLaunchKernelGGL((ct_pt_dot_kernel<1, false>), row_grid(c, n_poly * L), dim3(256), 0, s, g);
        }
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

// ---- n separate blocks <-> one packed array (the call combiner's gather and scatter) --------------------------------------
struct BlockPtrArgs
{
    uint64_t *blk[64];
    uint64_t *packed;
    uint32_t n2; // 16-byte chunks per block
};

template <bool GATHER>
__global__ __launch_bounds__(256) void block_copy_kernel(BlockPtrArgs g)
{
    ulonglong2 *b = reinterpret_cast<ulonglong2 *>(g.blk[blockIdx.y]);
    ulonglong2 *p = reinterpret_cast<ulonglong2 *>(g.packed) + (size_t)blockIdx.y * g.n2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < g.n2; i += gridDim.x * 256u)
    {
        if (GATHER)
        {
            p[i] = b[i];
        }
        else
        {
            b[i] = p[i];
        }
    }
}

static int block_copy(moai_ctx *c, uint64_t *const *blocks, uint64_t *packed, size_t n, size_t words, bool gather, void *stream)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    if (n == 0 || words == 0)
    {
        return MOAI_OK;
    }
    if (!blocks || !packed)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    if (n > 64 || (words & 1u) || (words >> 1) > 0xffffffffull)
    {
        return set_error(MOAI_EINVAL, "at most 64 blocks of an even number of words");
    }
    int rc = enter_device(c);
    if (rc)
    {
        return rc;
    }
    BlockPtrArgs g;
    for (size_t i = 0; i < 64; ++i)
    {
        g.blk[i] = blocks[i < n ? i : 0];
        if (i < n && !blocks[i])
        {
            return set_error(MOAI_EINVAL, "null block");
        }
        if (i < n)
        {
            MOAI_AUDIT(stream, blocks[i]);
        }
    }
    g.packed = packed;
    g.n2 = (uint32_t)(words >> 1);
    uint32_t bx = (g.n2 + 255u) / 256u;
    bx = bx > 64u ? 64u : bx;
    dim3 grid(bx, (uint32_t)n);
    if (gather)
    {
        hipLaunchKernelGGL(block_copy_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, g);
    }
    else
    {
        hipLaunchKernelGGL(block_copy_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, g);
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_gather_blocks(moai_ctx *c, const uint64_t *const *src, uint64_t *packed, size_t n, size_t words, void *stream)
{
    MOAI_AUDIT(stream, packed);
    return block_copy(c, const_cast<uint64_t *const *>(src), packed, n, words, true, stream);
}

extern "C" int moai_scatter_blocks(moai_ctx *c, const uint64_t *packed, uint64_t *const *dst, size_t n, size_t words, void *stream)
{
    MOAI_AUDIT(stream, packed);
    return block_copy(c, dst, const_cast<uint64_t *>(packed), n, words, false, stream);
}

extern "C" int moai_ct_pt_dot(moai_ctx *c, const uint64_t *x, const uint64_t *p, uint64_t *out, const uint32_t *x_index,
                              const uint32_t *p_index, size_t terms, size_t n_poly, size_t L, void *stream)
{
    MOAI_AUDIT(stream, x, p, out);
    trace_op("ct_pt_dot", L, n_poly * terms);
    return ct_pt_dot_common(c, x, p, out, nullptr, x_index, p_index, nullptr, terms, 0, n_poly, L, stream);
}

extern "C" int moai_ct_pt_dot2(moai_ctx *c, const uint64_t *x, const uint64_t *p, uint64_t *out, uint64_t *out2,
                               const uint32_t *x_index, const uint32_t *p_index, const uint32_t *p_index2, size_t terms, size_t terms2,
                               size_t n_poly, size_t L, void *stream)
{
    MOAI_AUDIT(stream, x, p, out, out2);
    trace_op("ct_pt_dot", L, n_poly * (terms + terms2)); // the same products as two moai_ct_pt_dot calls
    if (!out2)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    return ct_pt_dot_common(c, x, p, out, out2, x_index, p_index, p_index2, terms, terms2, n_poly, L, stream);
}

extern "C" int moai_ct_pt_dot_rows(moai_ctx *c, const uint64_t *x, const uint64_t *p, const uint64_t *p2, uint64_t *out, uint64_t *out2,
                                   size_t rows, size_t n_poly, size_t L, void *stream)
{
    MOAI_AUDIT(stream, x, p, p2, out, out2);
    trace_op("ct_pt_dot", L, n_poly * rows * (p2 ? 2 : 1)); // the products of moai_ct_pt_dot calls over the same rows
    int rc = check_rows(c, n_poly, L);
    if (rc)
    {
        return rc;
    }
    if (rows == 0 || rows > 0xffffffu)
    {
        return set_error(MOAI_EINVAL, "between 1 and 2^24 rows");
    }
    if (n_poly == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!x || !p || !out || (p2 != nullptr) != (out2 != nullptr) || out == out2)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    for (size_t r = 0; r < L; r++)
    {
        if (c->primes[r] >> 61)
        {
            return set_error(MOAI_ELOGIC, "lazy accumulation needs primes of at most 61 bits");
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n2 = (uint32_t)(c->n >> 1);
    const size_t blocks = (size_t)((n2 + 255) / 256) * n_poly * L;
    size_t splits = (8192 + blocks - 1) / blocks; // enough workgroups to fill the chip several times over
    splits = std::max<size_t>(1, std::min<size_t>({ splits, rows, 64 }));
    const size_t sets = p2 ? 2 : 1;
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, splits * sets * n_poly * L * c->n * sizeof(uint64_t), s, &wsp);
    if (rc)
    {
        return rc;
    }
    RowSumArgs g;
    g.x = x;
    g.p = p;
    g.p2 = p2;
    g.part = static_cast<uint64_t *>(wsp);
    g.out = out;
    g.out2 = out2;
    g.pc = c->pc;
    g.rows = (uint32_t)rows;
    g.L = (uint32_t)L;
    g.n2 = n2;
    g.n_poly = (uint32_t)n_poly;
    g.splits = (uint32_t)splits;
    MOAI_CHECK_GRID_ROWS(n_poly * L);
    const dim3 grid((n2 + 255) / 256, (uint32_t)(n_poly * L), (uint32_t)splits);
    if (p2)
    {
        hipLaunchKernelGGL(ct_pt_rowsum_kernel<true>, grid, dim3(256), 0, s, g);
    }
    else
    {
        hipLaunchKernelGGL(ct_pt_rowsum_kernel<false>, grid, dim3(256), 0, s, g);
    }
    hipLaunchKernelGGL(ct_pt_rowsum_reduce, dim3((n2 + 255) / 256, (uint32_t)(n_poly * L), (uint32_t)sets), dim3(256), 0, s, g);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_mod_drop(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t size, size_t L, size_t drop,
                             size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, out);
    trace_op("mod_drop", L, batch * size);
    int rc = check_rows(c, batch * size, L);
    if (rc)
    {
        return rc;
    }
    if (drop >= L)
    {
        // "end of modulus switching chain reached", SEAL/evaluator.cpp:1500-1503
        return set_error(MOAI_EINVAL, "end of modulus switching chain reached");
    }
    if (batch * size == 0)
    {
        return MOAI_OK;
    }
    if (!in || !out)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    if (in == out && batch * size > 1 && drop > 0)
    {
        return set_error(MOAI_EINVAL, "out must not alias in");
    }
    if (in == out)
    {
        return MOAI_OK; // a single polynomial keeps its leading rows in place
    }
    const size_t Lout = L - drop;
    MOAI_CHECK_GRID_ROWS(batch * size * Lout);
    hipLaunchKernelGGL(drop_rows_kernel, row_grid(c, batch * size * Lout), dim3(256), 0, (hipStream_t)stream, in, out,
                       (uint32_t)L, (uint32_t)Lout, (uint32_t)(c->n >> 1));
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_galois_permute(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t n_poly, size_t L,
                                   uint32_t galois_elt, void *stream)
{
    MOAI_AUDIT(stream, in, out);
    trace_op("galois_permute", L, n_poly);
    int rc = check_rows(c, n_poly, L);
    if (rc)
    {
        return rc;
    }
    if (in == out)
    {
        return set_error(MOAI_EINVAL, "result cannot point to the same value as operand");
    }
    const uint32_t *table;
    rc = galois_table(c, galois_elt, (hipStream_t)stream, &table);
    if (rc)
    {
        return rc;
    }
    if (n_poly * L == 0)
    {
        return MOAI_OK;
    }
    uint32_t bx = (uint32_t)((c->n + 511) / 512);
    hipLaunchKernelGGL(galois_gather_kernel, dim3(bx, (uint32_t)(n_poly * L)), dim3(256), 0, (hipStream_t)stream, in,
                       out, table, (uint32_t)c->n, (uint32_t)L, (uint32_t)L);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

namespace moai {
// out [batch][L][N] = the Galois permutation of polynomial 0 of every ciphertext of in [batch][2][L][N]
int galois_permute_c0(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t batch, size_t L, uint32_t galois_elt, hipStream_t s)
{
    const uint32_t *table;
    int rc = galois_table(c, galois_elt, s, &table);
    if (rc)
    {
        return rc;
    }
    if (batch * L == 0)
    {
        return MOAI_OK;
    }
    MOAI_CHECK_GRID_ROWS(batch * L);
    hipLaunchKernelGGL(galois_gather_kernel, dim3((uint32_t)((c->n + 511) / 512), (uint32_t)(batch * L)), dim3(256), 0, s, in, out, table,
                       (uint32_t)c->n, (uint32_t)L, (uint32_t)(2 * L));
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}
} // namespace moai

extern "C" uint32_t moai_galois_elt_from_step(const moai_ctx *c, int step)
{
    // GaloisTool::get_elt_from_step, SEAL/util/galois.cpp:53-95, generator 5 (galois.h:169)
    if (!c)
    {
        set_error(MOAI_EINVAL, "null context");
        return 0;
    }
    const uint32_t n = (uint32_t)c->n;
    const uint64_t m = (uint64_t)n * 2;
    if (step == 0)
    {
        return (uint32_t)(m - 1);
    }
    bool sign = step < 0;
    uint32_t pos = (uint32_t)(sign ? -(int64_t)step : (int64_t)step);
    if (pos >= (n >> 1))
    {
        set_error(MOAI_EINVAL, "step count too large");
        return 0;
    }
    uint32_t steps = sign ? (n >> 1) - pos : pos;
    uint64_t e = 1;
    while (steps--)
    {
        e = (e * 5) & (m - 1);
    }
    return (uint32_t)e;
}

extern "C" int moai_ct_pt_matmul(moai_ctx *c, const uint64_t *x, const uint64_t *w, uint64_t *out, size_t rows,
                                 size_t cols, size_t size, size_t L, void *stream)
{
    MOAI_AUDIT(stream, x, w, out);
    trace_op("ct_pt_matmul", L, rows * cols * size);
    int rc = check_rows(c, (rows > cols ? rows : cols) * size, L);
    if (rc)
    {
        return rc;
    }
    if (rows == 0 || cols == 0 || size == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (!x || !w || !out || x == out)
    {
        return set_error(MOAI_EINVAL, "bad pointers");
    }
    constexpr int CG = 16;
    if (size * L > 65535 || (cols + CG - 1) / CG > 65535)
    {
        return set_error(MOAI_EINVAL, "matrix too large for one launch");
    }
    MatmulArgs g;
    g.x = x;
    g.w = w;
    g.out = out;
    g.pc = c->pc;
    g.rows = (uint32_t)rows;
    g.cols = (uint32_t)cols;
    g.size = (uint32_t)size;
    g.L = (uint32_t)L;
    g.n2 = (uint32_t)(c->n >> 1);
    g.wd = nullptr;
    dim3 grid((g.n2 + 255u) / 256u, (uint32_t)(size * L), (uint32_t)((cols + CG - 1) / CG));
    // exact FP64 sums when every prime of the level is below 2^51 (MOAI_MATMUL_FP=0: the integer kernel)
    bool fp = tuning("MOAI_MATMUL_FP", 1) != 0;
    for (size_t r = 0; r < L && fp; r++)
    {
        fp = c->primes[r] < (1ull << 51);
    }
    if (fp)
    {
        hipStream_t s = (hipStream_t)stream;
        const size_t count = L * rows * cols;
        std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
        void *wsp;
        rc = workspace(c, count * sizeof(double), s, &wsp);
        if (rc)
        {
            return rc;
        }
        g.wd = static_cast<const double *>(wsp);
        hipLaunchKernelGGL(u52_to_f64_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, w, static_cast<double *>(wsp), count);
        hipLaunchKernelGGL(ct_pt_matmul_fp_kernel<CG>, grid, dim3(256), 0, s, g);
    }
    else
    {
        hipLaunchKernelGGL(ct_pt_matmul_kernel<CG>, grid, dim3(256), 0, (hipStream_t)stream, g);
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}
