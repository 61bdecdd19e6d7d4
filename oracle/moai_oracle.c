/*
 * moai_oracle.c -- CPU restatement of the reference's RNS-CKKS hot path.  See moai_oracle.h.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path and the timed CPU baseline of bench.py.
 * Never linked into, imported by or called from the product library.
 *
 * Every function cites the reference lines it restates (relative to /root/reference/;
 * SEAL/ = thirdparty/SEAL-4.1-bs/native/src/seal/).  The algorithms are kept the reference's own
 * (radix-2 Harvey butterflies with Shoup twiddles, base-2^64 Barrett, per-prime key-switch digits
 * with one special prime) so that the CPU baseline is not a strawman.
 */
#define _GNU_SOURCE /* sincos */
#include "moai_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((u128)a * b) >> 64);
}

static inline uint32_t reverse_bits32(uint32_t x, int bit_count)
{
    /* SEAL/util/common.h reverse_bits(operand, bit_count) */
    uint32_t r = 0;
    for (int i = 0; i < bit_count; i++)
    {
        r = (r << 1) | ((x >> i) & 1u);
    }
    return r;
}

static int significant_bits(uint64_t v)
{
    int n = 0;
    while (v)
    {
        n++;
        v >>= 1;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/modulus.cpp:36-77 set_value                                                            */
/* ------------------------------------------------------------------------------------------ */
void mo_modulus_init(mo_modulus *m, uint64_t value)
{
    m->value = value;
    m->bit_count = significant_bits(value);
    if (value == 0)
    {
        m->const_ratio[0] = m->const_ratio[1] = m->const_ratio[2] = 0;
        return;
    }
    /* floor(2^128 / value) and remainder, by long division of the 3-word number {0,0,1} */
    u128 hi = ((u128)1 << 64);                 /* top two words: (1, 0) */
    uint64_t q1 = (uint64_t)(hi / value);      /* quotient word 1 */
    u128 rem = hi % value;
    u128 lo = (rem << 64);                     /* append low word 0 */
    uint64_t q0 = (uint64_t)(lo / value);
    uint64_t r = (uint64_t)(lo % value);
    m->const_ratio[0] = q0;
    m->const_ratio[1] = q1;
    m->const_ratio[2] = r;
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/util/uintarithsmallmod.h                                                               */
/* ------------------------------------------------------------------------------------------ */
uint64_t mo_barrett_reduce_64(uint64_t input, const mo_modulus *m)
{
    /* :211-230 */
    uint64_t t = mulhi64(input, m->const_ratio[1]);
    uint64_t r = input - t * m->value;
    return r >= m->value ? r - m->value : r;
}

uint64_t mo_barrett_reduce_128(const uint64_t input[2], const mo_modulus *m)
{
    /* :167-203, word for word */
    uint64_t tmp1, tmp3, carry;
    u128 tmp2;
    const uint64_t *cr = m->const_ratio;

    carry = mulhi64(input[0], cr[0]);
    tmp2 = (u128)input[0] * cr[1];
    {
        u128 s = (u128)(uint64_t)tmp2 + carry;
        tmp1 = (uint64_t)s;
        tmp3 = (uint64_t)(tmp2 >> 64) + (uint64_t)(s >> 64);
    }
    tmp2 = (u128)input[1] * cr[0];
    {
        u128 s = (u128)tmp1 + (uint64_t)tmp2;
        tmp1 = (uint64_t)s;
        carry = (uint64_t)(tmp2 >> 64) + (uint64_t)(s >> 64);
    }
    tmp1 = input[1] * cr[1] + tmp3 + carry;
    tmp3 = input[0] - tmp1 * m->value;
    return tmp3 >= m->value ? tmp3 - m->value : tmp3;
}

uint64_t mo_multiply_uint_mod(uint64_t a, uint64_t b, const mo_modulus *m)
{
    u128 z = (u128)a * b;
    uint64_t w[2] = { (uint64_t)z, (uint64_t)(z >> 64) };
    return mo_barrett_reduce_128(w, m);
}

void mo_mulop_set(mo_mulop *y, uint64_t operand, const mo_modulus *m)
{
    /* :255-286: quotient = floor(operand * 2^64 / q) */
    y->operand = operand;
    y->quotient = (uint64_t)((((u128)operand) << 64) / m->value);
}

uint64_t mo_multiply_uint_mod_lazy(uint64_t x, mo_mulop y, const mo_modulus *m)
{
    /* :313-326 */
    uint64_t t = mulhi64(x, y.quotient);
    return y.operand * x - t * m->value;
}

uint64_t mo_multiply_uint_mod_op(uint64_t x, mo_mulop y, const mo_modulus *m)
{
    /* :292-306 */
    uint64_t r = mo_multiply_uint_mod_lazy(x, y, m);
    return r >= m->value ? r - m->value : r;
}

uint64_t mo_add_uint_mod(uint64_t a, uint64_t b, const mo_modulus *m)
{
    uint64_t s = a + b;
    return s >= m->value ? s - m->value : s;
}

uint64_t mo_sub_uint_mod(uint64_t a, uint64_t b, const mo_modulus *m)
{
    uint64_t d = a - b;
    return (a < b) ? d + m->value : d;
}

uint64_t mo_negate_uint_mod(uint64_t a, const mo_modulus *m)
{
    return a ? m->value - a : 0;
}

uint64_t mo_exponentiate_uint_mod(uint64_t a, uint64_t e, const mo_modulus *m)
{
    /* SEAL/util/uintarithsmallmod.cpp exponentiate_uint_mod: square and multiply */
    if (e == 0)
    {
        return 1;
    }
    uint64_t power = a, result = 1;
    while (1)
    {
        if (e & 1)
        {
            result = mo_multiply_uint_mod(power, result, m);
        }
        e >>= 1;
        if (!e)
        {
            break;
        }
        power = mo_multiply_uint_mod(power, power, m);
    }
    return result;
}

int mo_try_invert_uint_mod(uint64_t a, uint64_t modulus, uint64_t *result)
{
    /* SEAL/util/numth.h try_invert_uint_mod via extended gcd */
    if (a == 0)
    {
        return 0;
    }
    __int128 r0 = (__int128)modulus, r1 = (__int128)(a % modulus), t0 = 0, t1 = 1;
    if (modulus == 1)
    {
        return 0;
    }
    while (r1 != 0)
    {
        __int128 q = r0 / r1;
        __int128 r2 = r0 - q * r1;
        __int128 t2 = t0 - q * t1;
        r0 = r1;
        r1 = r2;
        t0 = t1;
        t1 = t2;
    }
    if (r0 != 1)
    {
        return 0;
    }
    if (t0 < 0)
    {
        t0 += (__int128)modulus;
    }
    *result = (uint64_t)t0;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/util/numth.cpp                                                                         */
/* ------------------------------------------------------------------------------------------ */
int mo_is_prime(uint64_t value)
{
    /*
     * numth.cpp:176-276 runs Miller-Rabin with base 2 plus 39 random bases; that is a
     * probabilistic statement of "value is prime".  Restated with the fixed base set
     * {2,3,5,7,11,13,17,19,23,29,31,37}, which is a proven-exact primality test for all
     * 64-bit integers, so the predicate (and hence every generated prime) is identical.
     */
    static const uint64_t small[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    if (value < 2)
    {
        return 0;
    }
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++)
    {
        if (value == small[i])
        {
            return 1;
        }
        if (value % small[i] == 0)
        {
            return 0;
        }
    }
    mo_modulus m;
    mo_modulus_init(&m, value);
    uint64_t d = value - 1;
    uint64_t r = 0;
    while (!(d & 1))
    {
        d >>= 1;
        r++;
    }
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++)
    {
        uint64_t a = small[i];
        uint64_t x = mo_exponentiate_uint_mod(a, d, &m);
        if (x == 1 || x == value - 1)
        {
            continue;
        }
        uint64_t count = 0;
        do
        {
            x = mo_multiply_uint_mod(x, x, &m);
            count++;
        } while (x != value - 1 && count < r - 1);
        if (x != value - 1)
        {
            return 0;
        }
    }
    return 1;
}

int mo_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out)
{
    /* numth.cpp:278-311: descend from (2^bits - 1)/factor*factor + 1 in steps of factor */
    uint64_t value = (((uint64_t)1 << bit_size) - 1) / factor * factor + 1;
    uint64_t lower_bound = (uint64_t)1 << (bit_size - 1);
    size_t found = 0;
    while (found < count && value > lower_bound)
    {
        if (mo_is_prime(value))
        {
            out[found++] = value;
        }
        value -= factor;
    }
    return found == count ? 0 : -1;
}

int mo_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out)
{
    /*
     * modulus.cpp:142-183: per distinct bit size, get_primes(2n, bits, multiplicity) in
     * descending order; hand them out with back()/pop_back(), i.e. SMALLEST first in the
     * order the sizes appear.
     */
    uint64_t factor = 2 * (uint64_t)n;
    size_t mult[64] = { 0 };
    uint64_t *tables[64] = { 0 };
    int rc = 0;
    for (size_t i = 0; i < count; i++)
    {
        if (bit_sizes[i] < 2 || bit_sizes[i] > 61)
        {
            return -2;
        }
        mult[bit_sizes[i]]++;
    }
    for (int b = 0; b < 64 && rc == 0; b++)
    {
        if (mult[b])
        {
            tables[b] = (uint64_t *)malloc(sizeof(uint64_t) * mult[b]);
            rc = mo_get_primes(factor, b, mult[b], tables[b]);
        }
    }
    if (rc == 0)
    {
        for (size_t i = 0; i < count; i++)
        {
            int b = bit_sizes[i];
            out[i] = tables[b][--mult[b]];
        }
    }
    for (int b = 0; b < 64; b++)
    {
        free(tables[b]);
    }
    return rc;
}

int mo_is_primitive_root(uint64_t root, uint64_t degree, const mo_modulus *m)
{
    /* numth.cpp:313-338 */
    if (root == 0)
    {
        return 0;
    }
    return mo_exponentiate_uint_mod(root, degree >> 1, m) == m->value - 1;
}

int mo_try_minimal_primitive_root(uint64_t degree, const mo_modulus *m, uint64_t *out)
{
    /*
     * numth.cpp:340-413.  try_primitive_root draws a random element and powers it by
     * (q-1)/degree; try_minimal_primitive_root then scans all odd powers and keeps the minimum,
     * so the result does not depend on the random start.  Restated with a deterministic start
     * (candidates 2,3,4,...).
     */
    uint64_t size_entire_group = m->value - 1;
    uint64_t size_quotient_group = size_entire_group / degree;
    if (size_entire_group - size_quotient_group * degree != 0)
    {
        return 0;
    }
    uint64_t root = 0;
    int ok = 0;
    for (uint64_t cand = 2; cand < 2 + 1000 && cand < m->value; cand++)
    {
        root = mo_exponentiate_uint_mod(cand, size_quotient_group, m);
        if (mo_is_primitive_root(root, degree, m))
        {
            ok = 1;
            break;
        }
    }
    if (!ok)
    {
        return 0;
    }
    uint64_t generator_sq = mo_multiply_uint_mod(root, root, m);
    uint64_t current = root;
    for (uint64_t i = 0; i < degree; i += 2)
    {
        if (current < root)
        {
            root = current;
        }
        current = mo_multiply_uint_mod(current, generator_sq, m);
    }
    *out = root;
    return 1;
}

int mo_naf(int value, int *out, int cap)
{
    /* SEAL/util/numth.h:22-41 */
    int sign = value < 0;
    int cnt = 0;
    value = abs(value);
    for (int i = 0; value; i++)
    {
        int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi)
        {
            if (cnt < cap)
            {
                out[cnt] = (sign ? -zi : zi) * (1 << i);
            }
            cnt++;
        }
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/util/ntt.cpp:241-300 NTTTables::initialize                                             */
/* ------------------------------------------------------------------------------------------ */
int mo_ntt_tables_init(mo_ntt_tables *t, int coeff_count_power, uint64_t modulus)
{
    memset(t, 0, sizeof(*t));
    t->coeff_count_power = coeff_count_power;
    t->coeff_count = (size_t)1 << coeff_count_power;
    mo_modulus_init(&t->modulus, modulus);
    const mo_modulus *m = &t->modulus;
    size_t n = t->coeff_count;

    if (!mo_try_minimal_primitive_root(2 * (uint64_t)n, m, &t->root))
    {
        return -1;
    }
    if (!mo_try_invert_uint_mod(t->root, m->value, &t->inv_root))
    {
        return -1;
    }
    t->root_powers = (mo_mulop *)malloc(sizeof(mo_mulop) * n);
    t->inv_root_powers = (mo_mulop *)malloc(sizeof(mo_mulop) * n);

    mo_mulop root;
    mo_mulop_set(&root, t->root, m);
    uint64_t power = t->root;
    for (size_t i = 1; i < n; i++)
    {
        mo_mulop_set(&t->root_powers[reverse_bits32((uint32_t)i, coeff_count_power)], power, m);
        power = mo_multiply_uint_mod_op(power, root, m);
    }
    mo_mulop_set(&t->root_powers[0], 1, m);

    mo_mulop_set(&root, t->inv_root, m);
    power = t->inv_root;
    for (size_t i = 1; i < n; i++)
    {
        mo_mulop_set(&t->inv_root_powers[reverse_bits32((uint32_t)(i - 1), coeff_count_power) + 1], power, m);
        power = mo_multiply_uint_mod_op(power, root, m);
    }
    mo_mulop_set(&t->inv_root_powers[0], 1, m);

    uint64_t inv_n;
    if (!mo_try_invert_uint_mod((uint64_t)n, m->value, &inv_n))
    {
        return -1;
    }
    mo_mulop_set(&t->inv_degree_modulo, inv_n, m);
    return 0;
}

void mo_ntt_tables_free(mo_ntt_tables *t)
{
    free(t->root_powers);
    free(t->inv_root_powers);
    t->root_powers = t->inv_root_powers = NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/util/ntt.h:21-67 Arithmetic<uint64_t, MultiplyUIntModOperand, ...> +                   */
/* SEAL/util/dwthandler.h:94-191 transform_to_rev / :202-356 transform_from_rev                */
/* ------------------------------------------------------------------------------------------ */
void mo_ntt_negacyclic_harvey_lazy(uint64_t *values, const mo_ntt_tables *t)
{
    const uint64_t q = t->modulus.value;
    const uint64_t two_q = q << 1;
    const size_t n = t->coeff_count;
    const mo_mulop *roots = t->root_powers;
    size_t gap = n >> 1;
    size_t m = 1;
    size_t root_idx = 0;

    for (; m <= (n >> 1); m <<= 1)
    {
        size_t offset = 0;
        for (size_t i = 0; i < m; i++)
        {
            const mo_mulop r = roots[++root_idx];
            uint64_t *x = values + offset;
            uint64_t *y = x + gap;
            for (size_t j = 0; j < gap; j++)
            {
                /* guard: x >= 2q ? x - 2q : x ; mul_root: lazy Shoup in [0,2q) */
                uint64_t u = *x >= two_q ? *x - two_q : *x;
                uint64_t v = r.operand * *y - mulhi64(*y, r.quotient) * q;
                *x++ = u + v;
                *y++ = u + two_q - v;
            }
            offset += gap << 1;
        }
        gap >>= 1;
    }
}

void mo_ntt_negacyclic_harvey(uint64_t *values, const mo_ntt_tables *t)
{
    /* ntt.cpp:408-437: lazy transform, then [0,4q) -> [0,q) */
    mo_ntt_negacyclic_harvey_lazy(values, t);
    const uint64_t q = t->modulus.value;
    const uint64_t two_q = q << 1;
    for (size_t i = 0; i < t->coeff_count; i++)
    {
        uint64_t v = values[i];
        if (v >= two_q)
        {
            v -= two_q;
        }
        if (v >= q)
        {
            v -= q;
        }
        values[i] = v;
    }
}

void mo_inverse_ntt_negacyclic_harvey_lazy(uint64_t *values, const mo_ntt_tables *t)
{
    /* dwthandler.h:202-356 with scalar = inv_degree_modulo (ntt.cpp:447-449) */
    const mo_modulus *mod = &t->modulus;
    const uint64_t q = mod->value;
    const uint64_t two_q = q << 1;
    const size_t n = t->coeff_count;
    const mo_mulop *roots = t->inv_root_powers;
    size_t gap = 1;
    size_t m = n >> 1;
    size_t root_idx = 0;

    for (; m > 1; m >>= 1)
    {
        size_t offset = 0;
        for (size_t i = 0; i < m; i++)
        {
            const mo_mulop r = roots[++root_idx];
            uint64_t *x = values + offset;
            uint64_t *y = x + gap;
            for (size_t j = 0; j < gap; j++)
            {
                uint64_t u = *x;
                uint64_t v = *y;
                uint64_t s = u + v;
                *x++ = s >= two_q ? s - two_q : s;
                uint64_t d = u + two_q - v;
                *y++ = r.operand * d - mulhi64(d, r.quotient) * q;
            }
            offset += gap << 1;
        }
        gap <<= 1;
    }
    {
        /* last stage with the scalar folded in (dwthandler.h:273-314) */
        const mo_mulop scalar = t->inv_degree_modulo;
        const mo_mulop r = roots[++root_idx];
        mo_mulop scaled_r;
        /* mul_root_scalar: scaled_r = r * scalar mod q (ntt.h:53-58) */
        mo_mulop_set(&scaled_r, mo_multiply_uint_mod_op(r.operand, scalar, mod), mod);
        uint64_t *x = values;
        uint64_t *y = x + gap;
        for (size_t j = 0; j < gap; j++)
        {
            uint64_t u = *x >= two_q ? *x - two_q : *x;
            uint64_t v = *y;
            uint64_t s = u + v;
            s = s >= two_q ? s - two_q : s;
            *x++ = scalar.operand * s - mulhi64(s, scalar.quotient) * q;
            uint64_t d = u + two_q - v;
            *y++ = scaled_r.operand * d - mulhi64(d, scaled_r.quotient) * q;
        }
    }
}

void mo_inverse_ntt_negacyclic_harvey(uint64_t *values, const mo_ntt_tables *t)
{
    /* ntt.cpp:453-475: lazy transform then [0,2q) -> [0,q) */
    mo_inverse_ntt_negacyclic_harvey_lazy(values, t);
    const uint64_t q = t->modulus.value;
    for (size_t i = 0; i < t->coeff_count; i++)
    {
        if (values[i] >= q)
        {
            values[i] -= q;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/util/polyarithsmallmod.{h,cpp}                                                         */
/* ------------------------------------------------------------------------------------------ */
void mo_modulo_poly_coeffs(const uint64_t *poly, size_t n, const mo_modulus *m, uint64_t *result)
{
    /* :18-41 */
    for (size_t i = 0; i < n; i++)
    {
        result[i] = mo_barrett_reduce_64(poly[i], m);
    }
}

void mo_add_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const mo_modulus *m, uint64_t *r)
{
    /* :43-86 */
    const uint64_t q = m->value;
    for (size_t i = 0; i < n; i++)
    {
        uint64_t s = a[i] + b[i];
        r[i] = s >= q ? s - q : s;
    }
}

void mo_sub_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const mo_modulus *m, uint64_t *r)
{
    /* :88-133 */
    const uint64_t q = m->value;
    for (size_t i = 0; i < n; i++)
    {
        uint64_t d = a[i] - b[i];
        r[i] = a[i] < b[i] ? d + q : d;
    }
}

void mo_negate_poly_coeffmod(const uint64_t *a, size_t n, const mo_modulus *m, uint64_t *r)
{
    /* polyarithsmallmod.h:77-106 (0 -> 0) */
    for (size_t i = 0; i < n; i++)
    {
        r[i] = a[i] ? m->value - a[i] : 0;
    }
}

void mo_add_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t scalar, const mo_modulus *m, uint64_t *r)
{
    /* :135-164 */
    for (size_t i = 0; i < n; i++)
    {
        r[i] = mo_add_uint_mod(a[i], scalar, m);
    }
}

void mo_multiply_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t scalar, const mo_modulus *m,
                                      uint64_t *r)
{
    /* polyarithsmallmod.h:209-217 (reduce scalar, build Shoup operand) + .cpp:197-224 */
    mo_mulop s;
    mo_mulop_set(&s, mo_barrett_reduce_64(scalar, m), m);
    for (size_t i = 0; i < n; i++)
    {
        r[i] = mo_multiply_uint_mod_op(a[i], s, m);
    }
}

void mo_dyadic_product_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const mo_modulus *m, uint64_t *r)
{
    /* :226-278: 128-bit product then inline Barrett-128 */
    for (size_t i = 0; i < n; i++)
    {
        u128 z = (u128)a[i] * b[i];
        uint64_t w[2] = { (uint64_t)z, (uint64_t)(z >> 64) };
        r[i] = mo_barrett_reduce_128(w, m);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SEAL/util/galois.cpp                                                                        */
/* ------------------------------------------------------------------------------------------ */
uint32_t mo_galois_elt_from_step(int coeff_count_power, int step, uint32_t generator, int *err)
{
    /* :53-95 */
    uint32_t n = (uint32_t)1 << coeff_count_power;
    uint32_t m32 = n * 2;
    uint64_t m = m32;
    if (err)
    {
        *err = 0;
    }
    if (step == 0)
    {
        return (uint32_t)(m - 1);
    }
    int sign = step < 0;
    uint32_t pos_step = (uint32_t)abs(step);
    if (pos_step >= (n >> 1))
    {
        if (err)
        {
            *err = 1; /* "step count too large" */
        }
        return 0;
    }
    pos_step &= m32 - 1;
    if (sign)
    {
        step = (int)(n >> 1) - (int)pos_step;
    }
    else
    {
        step = (int)pos_step;
    }
    uint64_t gen = generator;
    uint64_t galois_elt = 1;
    while (step--)
    {
        galois_elt *= gen;
        galois_elt &= m - 1;
    }
    return (uint32_t)galois_elt;
}

int mo_galois_elts_all(int coeff_count_power, uint32_t generator, uint32_t *out)
{
    /* :106-131 */
    uint32_t m = (uint32_t)(((uint64_t)1 << coeff_count_power) << 1);
    int cnt = 0;
    out[cnt++] = m - 1;
    uint64_t pos_power = generator;
    uint64_t neg_power = 0;
    mo_try_invert_uint_mod(generator, m, &neg_power);
    for (int i = 0; i < coeff_count_power - 1; i++)
    {
        out[cnt++] = (uint32_t)pos_power;
        pos_power *= pos_power;
        pos_power &= (m - 1);
        out[cnt++] = (uint32_t)neg_power;
        neg_power *= neg_power;
        neg_power &= (m - 1);
    }
    return cnt;
}

void mo_galois_table_ntt(int coeff_count_power, uint32_t galois_elt, uint32_t *table)
{
    /* :18-51 */
    size_t n = (size_t)1 << coeff_count_power;
    uint32_t mask = (uint32_t)n - 1;
    for (size_t i = n; i < (n << 1); i++)
    {
        uint32_t reversed = reverse_bits32((uint32_t)i, coeff_count_power + 1);
        uint64_t index_raw = ((uint64_t)galois_elt * (uint64_t)reversed) >> 1;
        index_raw &= (uint64_t)mask;
        *table++ = reverse_bits32((uint32_t)index_raw, coeff_count_power);
    }
}

void mo_apply_galois_ntt(const uint64_t *operand, const uint32_t *table, size_t n, uint64_t *result)
{
    /* :192-218 */
    for (size_t i = 0; i < n; i++)
    {
        result[i] = operand[table[i]];
    }
}

void mo_apply_galois(const uint64_t *operand, int coeff_count_power, uint32_t galois_elt, const mo_modulus *m,
                     uint64_t *result)
{
    /* :147-190 (coefficient form; only used to check the reference's KAT) */
    uint64_t n1 = ((uint64_t)1 << coeff_count_power) - 1;
    uint64_t index_raw = 0;
    for (uint64_t i = 0; i <= n1; i++, index_raw += galois_elt)
    {
        uint64_t index = index_raw & n1;
        uint64_t v = operand[i];
        if ((index_raw >> coeff_count_power) & 1)
        {
            v = v ? m->value - v : 0;
        }
        result[index] = v;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                     */
/* ------------------------------------------------------------------------------------------ */
mo_context *mo_context_create(int coeff_count_power, const uint64_t *primes, size_t k)
{
    mo_context *c = (mo_context *)calloc(1, sizeof(mo_context));
    c->coeff_count_power = coeff_count_power;
    c->n = (size_t)1 << coeff_count_power;
    c->k = k;
    c->mods = (mo_modulus *)calloc(k, sizeof(mo_modulus));
    c->tables = (mo_ntt_tables *)calloc(k, sizeof(mo_ntt_tables));
    int bad = 0;
#pragma omp parallel for schedule(dynamic) reduction(| : bad)
    for (size_t i = 0; i < k; i++)
    {
        mo_modulus_init(&c->mods[i], primes[i]);
        if (mo_ntt_tables_init(&c->tables[i], coeff_count_power, primes[i]) != 0)
        {
            bad |= 1;
        }
    }
    if (bad)
    {
        mo_context_destroy(c);
        return NULL;
    }
    return c;
}

void mo_context_destroy(mo_context *c)
{
    if (!c)
    {
        return;
    }
    for (size_t i = 0; i < c->k; i++)
    {
        mo_ntt_tables_free(&c->tables[i]);
    }
    free(c->mods);
    free(c->tables);
    free(c);
}

void mo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
    {
        omp_set_num_threads(n);
    }
#else
    (void)n;
#endif
}

int mo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* scheme-level                                                                                */
/* ------------------------------------------------------------------------------------------ */
void mo_ntt_rns(const mo_context *c, uint64_t *data, size_t npoly, size_t L, const uint32_t *prime_index,
                int inverse)
{
    /* SEAL/evaluator.cpp:2468-2561 transform_{to,from}_ntt_inplace: every (poly, prime) row */
    for (size_t p = 0; p < npoly; p++)
    {
        for (size_t i = 0; i < L; i++)
        {
            const mo_ntt_tables *t = &c->tables[prime_index ? prime_index[i] : i];
            uint64_t *row = data + (p * L + i) * c->n;
            if (inverse)
            {
                mo_inverse_ntt_negacyclic_harvey(row, t);
            }
            else
            {
                mo_ntt_negacyclic_harvey(row, t);
            }
        }
    }
}

void mo_batch_ntt(const mo_context *c, uint64_t *data, size_t npoly, size_t L, const uint32_t *prime_index,
                  int inverse)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npoly; p++)
    {
        mo_ntt_rns(c, data + p * L * c->n, 1, L, prime_index, inverse);
    }
}

void mo_divide_and_round_q_last_ntt_inplace(const mo_context *c, uint64_t *poly, size_t L)
{
    /* SEAL/util/rns.cpp:830-901 (SEAL_USER_MOD_BIT_COUNT_MAX = 60 branch, defines.h:40) */
    const size_t n = c->n;
    uint64_t *last_input = poly + (L - 1) * n;
    const mo_modulus *last_modulus = &c->mods[L - 1];

    mo_inverse_ntt_negacyclic_harvey(last_input, &c->tables[L - 1]);

    uint64_t half = last_modulus->value >> 1;
    mo_add_poly_scalar_coeffmod(last_input, n, half, last_modulus, last_input);

    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i + 1 < L; i++)
    {
        const mo_modulus *qi = &c->mods[i];
        uint64_t *ci = poly + i * n;
        /* inv_q_last_mod_q_ (rns.cpp:769-775) */
        uint64_t inv_q_last;
        mo_try_invert_uint_mod(mo_barrett_reduce_64(last_modulus->value, qi), qi->value, &inv_q_last);

        if (qi->value < last_modulus->value)
        {
            mo_modulo_poly_coeffs(last_input, n, qi, temp);
        }
        else
        {
            memcpy(temp, last_input, sizeof(uint64_t) * n);
        }
        uint64_t neg_half_mod = qi->value - mo_barrett_reduce_64(half, qi);
        for (size_t j = 0; j < n; j++)
        {
            temp[j] += neg_half_mod;
        }
        uint64_t qi_lazy = qi->value << 2;
        mo_ntt_negacyclic_harvey_lazy(temp, &c->tables[i]);
        for (size_t j = 0; j < n; j++)
        {
            ci[j] += qi_lazy - temp[j];
        }
        mo_multiply_poly_scalar_coeffmod(ci, n, inv_q_last, qi, ci);
    }
    free(temp);
}

void mo_rescale_to_next(const mo_context *c, const uint64_t *in, size_t size, size_t L, uint64_t *out)
{
    /* SEAL/evaluator.cpp:1402-1481 (CKKS branch :1447-1455, copy :1458-1463) */
    const size_t n = c->n;
    uint64_t *copy = (uint64_t *)malloc(sizeof(uint64_t) * L * n);
    for (size_t p = 0; p < size; p++)
    {
        memcpy(copy, in + p * L * n, sizeof(uint64_t) * L * n);
        mo_divide_and_round_q_last_ntt_inplace(c, copy, L);
        memcpy(out + p * (L - 1) * n, copy, sizeof(uint64_t) * (L - 1) * n);
    }
    free(copy);
}

void mo_mod_switch_drop(const mo_context *c, const uint64_t *in, size_t size, size_t L, size_t drop, uint64_t *out)
{
    /* SEAL/evaluator.cpp:1483-1546 applied `drop` times */
    const size_t n = c->n;
    for (size_t p = 0; p < size; p++)
    {
        memmove(out + p * (L - drop) * n, in + p * L * n, sizeof(uint64_t) * (L - drop) * n);
    }
}

void mo_ckks_multiply(const mo_context *c, uint64_t *x, const uint64_t *y, size_t L)
{
    /* SEAL/evaluator.cpp:805-860: x = (x0*y0, x0*y1 + x1*y0, x1*y1) */
    const size_t n = c->n;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < L; i++)
    {
        const mo_modulus *q = &c->mods[i];
        uint64_t *x0 = x + (0 * L + i) * n, *x1 = x + (1 * L + i) * n, *x2 = x + (2 * L + i) * n;
        const uint64_t *y0 = y + (0 * L + i) * n, *y1 = y + (1 * L + i) * n;
        mo_dyadic_product_coeffmod(x1, y1, n, q, x2);
        mo_dyadic_product_coeffmod(x1, y0, n, q, temp);
        mo_dyadic_product_coeffmod(x0, y1, n, q, x1);
        mo_add_poly_coeffmod(x1, temp, n, q, x1);
        mo_dyadic_product_coeffmod(x0, y0, n, q, x0);
    }
    free(temp);
}

void mo_ckks_multiply_general(const mo_context *c, const uint64_t *x, size_t size_x, const uint64_t *y, size_t size_y, size_t L,
                              uint64_t *out)
{
    /* SEAL/evaluator.cpp:862-900 (dest_size != 3): out[k] = sum over i + j = k of x[i] (*) y[j], the terms in increasing i,
     * each product reduced and added into a zeroed accumulator. out: size_x + size_y - 1 polynomials, not an operand. */
    const size_t n = c->n, dest = size_x + size_y - 1;
    uint64_t *prod = (uint64_t *)malloc(sizeof(uint64_t) * n);
    memset(out, 0, sizeof(uint64_t) * dest * L * n);
    for (size_t k = 0; k < dest; k++)
    {
        const size_t x_last = k < size_x - 1 ? k : size_x - 1;
        const size_t y_first = k < size_y - 1 ? k : size_y - 1;
        const size_t x_first = k - y_first;
        for (size_t s = 0; s <= x_last - x_first; s++)
        {
            const uint64_t *xp = x + (x_first + s) * L * n, *yp = y + (y_first - s) * L * n;
            for (size_t i = 0; i < L; i++)
            {
                uint64_t *acc = out + (k * L + i) * n;
                mo_dyadic_product_coeffmod(xp + i * n, yp + i * n, n, &c->mods[i], prod);
                mo_add_poly_coeffmod(prod, acc, n, &c->mods[i], acc);
            }
        }
    }
    free(prod);
}

void mo_ckks_square(const mo_context *c, uint64_t *x, size_t L)
{
    /* SEAL/evaluator.cpp:1262-1274 */
    const size_t n = c->n;
    for (size_t i = 0; i < L; i++)
    {
        const mo_modulus *q = &c->mods[i];
        uint64_t *x0 = x + (0 * L + i) * n, *x1 = x + (1 * L + i) * n, *x2 = x + (2 * L + i) * n;
        mo_dyadic_product_coeffmod(x1, x1, n, q, x2);
        mo_dyadic_product_coeffmod(x0, x1, n, q, x1);
        mo_add_poly_coeffmod(x1, x1, n, q, x1);
        mo_dyadic_product_coeffmod(x0, x0, n, q, x0);
    }
}

void mo_multiply_plain(const mo_context *c, uint64_t *ct, size_t size, size_t L, const uint64_t *plain)
{
    /* SEAL/evaluator.cpp:2336-2373 */
    const size_t n = c->n;
    for (size_t p = 0; p < size; p++)
    {
        for (size_t i = 0; i < L; i++)
        {
            uint64_t *row = ct + (p * L + i) * n;
            mo_dyadic_product_coeffmod(row, plain + i * n, n, &c->mods[i], row);
        }
    }
}

void mo_ct_add(const mo_context *c, const uint64_t *a, const uint64_t *b, size_t size, size_t L, uint64_t *r)
{
    for (size_t p = 0; p < size; p++)
    {
        for (size_t i = 0; i < L; i++)
        {
            size_t o = (p * L + i) * c->n;
            mo_add_poly_coeffmod(a + o, b + o, c->n, &c->mods[i], r + o);
        }
    }
}

void mo_ct_sub(const mo_context *c, const uint64_t *a, const uint64_t *b, size_t size, size_t L, uint64_t *r)
{
    for (size_t p = 0; p < size; p++)
    {
        for (size_t i = 0; i < L; i++)
        {
            size_t o = (p * L + i) * c->n;
            mo_sub_poly_coeffmod(a + o, b + o, c->n, &c->mods[i], r + o);
        }
    }
}

void mo_ct_negate(const mo_context *c, const uint64_t *a, size_t size, size_t L, uint64_t *r)
{
    for (size_t p = 0; p < size; p++)
    {
        for (size_t i = 0; i < L; i++)
        {
            size_t o = (p * L + i) * c->n;
            mo_negate_poly_coeffmod(a + o, c->n, &c->mods[i], r + o);
        }
    }
}

void mo_switch_key_inplace(const mo_context *c, uint64_t *ct, const uint64_t *target, const uint64_t *key, size_t L)
{
    /* SEAL/evaluator.cpp:2724-3020, CKKS branch.  Names follow the reference. */
    const size_t n = c->n;
    const size_t decomp_modulus_size = L;
    const size_t key_modulus_size = c->k;
    const size_t rns_modulus_size = decomp_modulus_size + 1;
    const size_t key_component_count = 2;
    const size_t key_poly_stride = key_modulus_size * n;         /* one poly of one PublicKey */
    const size_t key_digit_stride = key_component_count * key_poly_stride;

    /* :2804-2812 t_target = INTT(copy of target) */
    uint64_t *t_target = (uint64_t *)malloc(sizeof(uint64_t) * decomp_modulus_size * n);
    memcpy(t_target, target, sizeof(uint64_t) * decomp_modulus_size * n);
    for (size_t j = 0; j < decomp_modulus_size; j++)
    {
        mo_inverse_ntt_negacyclic_harvey(t_target + j * n, &c->tables[j]);
    }

    /* :2815 t_poly_prod [key_component_count][rns_modulus_size][n] */
    uint64_t *t_poly_prod = (uint64_t *)calloc(key_component_count * rns_modulus_size * n, sizeof(uint64_t));
    u128 *t_poly_lazy = (u128 *)malloc(sizeof(u128) * key_component_count * n);
    uint64_t *t_ntt = (uint64_t *)malloc(sizeof(uint64_t) * n);

    for (size_t I = 0; I < rns_modulus_size; I++)
    {
        size_t key_index = (I == decomp_modulus_size ? key_modulus_size - 1 : I);
        const mo_modulus *qI = &c->mods[key_index];
        const size_t bound = 256; /* SEAL_MULTIPLY_ACCUMULATE_USER_MOD_MAX, defines.h:66 */
        size_t lazy_reduction_counter = bound;
        memset(t_poly_lazy, 0, sizeof(u128) * key_component_count * n);

        for (size_t J = 0; J < decomp_modulus_size; J++)
        {
            const uint64_t *t_operand;
            if (I == J)
            {
                t_operand = target + J * n; /* :2836-2839 */
            }
            else
            {
                if (c->mods[J].value <= qI->value)
                {
                    memcpy(t_ntt, t_target + J * n, sizeof(uint64_t) * n); /* :2846-2849 */
                }
                else
                {
                    mo_modulo_poly_coeffs(t_target + J * n, n, qI, t_ntt); /* :2851-2854 */
                }
                mo_ntt_negacyclic_harvey_lazy(t_ntt, &c->tables[key_index]); /* :2856 */
                t_operand = t_ntt;
            }
            for (size_t K = 0; K < key_component_count; K++)
            {
                const uint64_t *kp = key + J * key_digit_stride + K * key_poly_stride + key_index * n;
                u128 *acc = t_poly_lazy + K * n;
                if (!lazy_reduction_counter)
                {
                    for (size_t l = 0; l < n; l++)
                    {
                        u128 s = (u128)t_operand[l] * kp[l] + acc[l];
                        uint64_t w[2] = { (uint64_t)s, (uint64_t)(s >> 64) };
                        acc[l] = mo_barrett_reduce_128(w, qI);
                    }
                }
                else
                {
                    for (size_t l = 0; l < n; l++)
                    {
                        acc[l] += (u128)t_operand[l] * kp[l];
                    }
                }
            }
            if (!--lazy_reduction_counter)
            {
                lazy_reduction_counter = bound;
            }
        }
        /* :2891-2910 final reduction into t_poly_prod[K][I] */
        for (size_t K = 0; K < key_component_count; K++)
        {
            uint64_t *dst = t_poly_prod + (K * rns_modulus_size + I) * n;
            const u128 *acc = t_poly_lazy + K * n;
            if (lazy_reduction_counter == bound)
            {
                for (size_t l = 0; l < n; l++)
                {
                    dst[l] = (uint64_t)acc[l];
                }
            }
            else
            {
                for (size_t l = 0; l < n; l++)
                {
                    uint64_t w[2] = { (uint64_t)acc[l], (uint64_t)(acc[l] >> 64) };
                    dst[l] = mo_barrett_reduce_128(w, qI);
                }
            }
        }
    }

    /* :2913-3018 modulus switching with scaling (CKKS else-branch :2963-3018) */
    const mo_modulus *qk_mod = &c->mods[key_modulus_size - 1];
    const uint64_t qk = qk_mod->value;
    const uint64_t qk_half = qk >> 1;
    for (size_t K = 0; K < key_component_count; K++)
    {
        uint64_t *t_last = t_poly_prod + (K * rns_modulus_size + decomp_modulus_size) * n;
        mo_inverse_ntt_negacyclic_harvey_lazy(t_last, &c->tables[key_modulus_size - 1]);
        for (size_t l = 0; l < n; l++)
        {
            t_last[l] = mo_barrett_reduce_64(t_last[l] + qk_half, qk_mod);
        }
        for (size_t J = 0; J < decomp_modulus_size; J++)
        {
            const mo_modulus *qj = &c->mods[J];
            const uint64_t qi = qj->value;
            uint64_t *prod = t_poly_prod + (K * rns_modulus_size + J) * n;
            uint64_t *dst = ct + (K * decomp_modulus_size + J) * n;
            /* modswitch_factors = inv_q_last_mod_q of the key level (rns.cpp:769-775) */
            uint64_t inv_qk;
            mo_try_invert_uint_mod(mo_barrett_reduce_64(qk, qj), qi, &inv_qk);

            if (qk > qi)
            {
                mo_modulo_poly_coeffs(t_last, n, qj, t_ntt);
            }
            else
            {
                memcpy(t_ntt, t_last, sizeof(uint64_t) * n);
            }
            uint64_t fix = qi - mo_barrett_reduce_64(qk_half, qj);
            for (size_t l = 0; l < n; l++)
            {
                t_ntt[l] += fix;
            }
            uint64_t qi_lazy = qi << 2; /* SEAL_USER_MOD_BIT_COUNT_MAX <= 60 branch :3005-3008 */
            mo_ntt_negacyclic_harvey_lazy(t_ntt, &c->tables[J]);
            for (size_t l = 0; l < n; l++)
            {
                prod[l] += qi_lazy - t_ntt[l];
            }
            mo_multiply_poly_scalar_coeffmod(prod, n, inv_qk, qj, prod);
            mo_add_poly_coeffmod(prod, dst, n, qj, dst);
        }
    }

    free(t_ntt);
    free(t_poly_lazy);
    free(t_poly_prod);
    free(t_target);
}

void mo_batch_switch_key(const mo_context *c, uint64_t *cts, const uint64_t *targets, const uint64_t *key, size_t L,
                         size_t batch)
{
#pragma omp parallel for schedule(dynamic)
    for (size_t b = 0; b < batch; b++)
    {
        mo_switch_key_inplace(c, cts + b * 2 * L * c->n, targets + b * L * c->n, key, L);
    }
}

void mo_relinearize(const mo_context *c, uint64_t *ct3, const uint64_t *relin_key, size_t L)
{
    /* SEAL/evaluator.cpp:1345-1400 for size 3 -> 2: switch_key(ct, c2, relin_keys[get_index(2)=0]) */
    mo_switch_key_inplace(c, ct3, ct3 + 2 * L * c->n, relin_key, L);
}

void mo_relinearize_general(const mo_context *c, uint64_t *ct, size_t size, size_t dest_size, const uint64_t *const *relin_keys, size_t L)
{
    /* SEAL/evaluator.cpp:1385-1393: while the size exceeds dest_size, the last polynomial is switched with
     * relin_keys[get_index(size - 1)] = relin_keys[size - 3] into (c0, c1) and dropped. relin_keys[t]: the key for s^(t+2). */
    while (size > dest_size && size > 2)
    {
        mo_switch_key_inplace(c, ct, ct + (size - 1) * L * c->n, relin_keys[size - 3], L);
        size--;
    }
}

void mo_apply_galois_inplace(const mo_context *c, uint64_t *ct, size_t L, uint32_t galois_elt,
                             const uint64_t *galois_key)
{
    /* SEAL/evaluator.cpp:2631-2665 */
    const size_t n = c->n;
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * L * n);
    mo_galois_table_ntt(c->coeff_count_power, galois_elt, table);
    for (size_t i = 0; i < L; i++)
    {
        mo_apply_galois_ntt(ct + i * n, table, n, temp + i * n);
    }
    memcpy(ct, temp, sizeof(uint64_t) * L * n);
    for (size_t i = 0; i < L; i++)
    {
        mo_apply_galois_ntt(ct + (L + i) * n, table, n, temp + i * n);
    }
    memset(ct + L * n, 0, sizeof(uint64_t) * L * n);
    mo_switch_key_inplace(c, ct, temp, galois_key, L);
    free(temp);
    free(table);
}

void mo_modraise(const mo_context *c, const uint64_t *in, size_t Lout, uint64_t *out)
{
    /* include/source/bootstrapping/Bootstrapper.cpp:2938-2992 */
    const size_t n = c->n;
    const uint64_t q0 = c->mods[0].value;
    uint64_t *src = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t p = 0; p < 2; p++)
    {
        memcpy(src, in + p * n, sizeof(uint64_t) * n);
        mo_inverse_ntt_negacyclic_harvey(src, &c->tables[0]);
        for (size_t j = 0; j < Lout; j++)
        {
            const uint64_t q = c->mods[j].value;
            const uint64_t minus_q0 = (j == 0) ? 0 : q - q0 % q;
            uint64_t *dst = out + (p * Lout + j) * n;
            for (size_t i = 0; i < n; i++)
            {
                uint64_t v = src[i] % q;
                if (src[i] > (q0 >> 1))
                {
                    v += minus_q0;
                    v -= (v >= q) ? q : 0;
                }
                dst[i] = v;
            }
            mo_ntt_negacyclic_harvey(dst, &c->tables[j]);
        }
    }
    free(src);
}

/* ==== SEAL/ckks.{h,cpp}: CKKSEncoder ========================================================= */
#include <math.h>

void mo_complex_get_root(size_t degree, const double *roots8, size_t index, double out[2])
{
    /* SEAL/util/croots.cpp:44-75: 8-fold symmetry of the degree-th roots of unity */
    double t[2];
    index &= degree - 1;
    if (index <= degree / 8)
    {
        out[0] = roots8[2 * index];
        out[1] = roots8[2 * index + 1];
    }
    else if (index <= degree / 4)
    {
        /* mirror(a) = (imag, real) */
        out[0] = roots8[2 * (degree / 4 - index) + 1];
        out[1] = roots8[2 * (degree / 4 - index)];
    }
    else if (index <= degree / 2)
    {
        /* -conj(z) = (-re, im) */
        mo_complex_get_root(degree, roots8, degree / 2 - index, t);
        out[0] = -t[0];
        out[1] = -(-t[1]);
    }
    else if (index <= 3 * degree / 4)
    {
        mo_complex_get_root(degree, roots8, index - degree / 2, t);
        out[0] = -t[0];
        out[1] = -t[1];
    }
    else
    {
        mo_complex_get_root(degree, roots8, degree - index, t);
        out[0] = t[0];
        out[1] = -t[1];
    }
}

mo_ckks_tables *mo_ckks_tables_create(int logn)
{
    /* SEAL/ckks.cpp:13-76 */
    if (logn < 2 || logn > 20)
    {
        return NULL;
    }
    mo_ckks_tables *t = (mo_ckks_tables *)calloc(1, sizeof(*t));
    const size_t n = (size_t)1 << logn;
    t->logn = logn;
    t->n = n;
    t->slots = n >> 1;
    t->index_map = (uint32_t *)malloc(sizeof(uint32_t) * n);
    t->root_powers = (double *)calloc(2 * n, sizeof(double));
    t->inv_root_powers = (double *)calloc(2 * n, sizeof(double));
    const uint64_t m = (uint64_t)n << 1;
    uint64_t gen = 5, pos = 1;
    for (size_t i = 0; i < t->slots; i++)
    {
        uint64_t index1 = (pos - 1) >> 1;
        uint64_t index2 = (m - pos - 1) >> 1;
        t->index_map[i] = reverse_bits32((uint32_t)index1, logn);
        t->index_map[t->slots | i] = reverse_bits32((uint32_t)index2, logn);
        pos *= gen;
        pos &= (m - 1);
    }
    if (m >= 8)
    {
        /* SEAL/util/croots.cpp:18-42: polar(1.0, 2 * PI * i / degree) for i <= degree / 8 */
        const double PI_ = 3.1415926535897932384626433832795028842;
        const size_t cnt = (size_t)(m / 8 + 1);
        double *roots8 = (double *)malloc(sizeof(double) * 2 * cnt);
        for (size_t i = 0; i < cnt; i++)
        {
            /* std::polar(1.0, theta) = (cos(theta), sin(theta)); GCC at -O2 and above (the reference's
             * Release build) merges the pair into ONE sincos call, and glibc's sincos does not return
             * sin()'s bits for every argument (1 of 2049 entries differs at N = 2^13 with glibc 2.35),
             * so call sincos explicitly rather than leave it to the optimiser */
            double theta = 2 * PI_ * (double)i / (double)m, sn, cs;
            sincos(theta, &sn, &cs);
            roots8[2 * i] = 1.0 * cs;
            roots8[2 * i + 1] = 1.0 * sn;
        }
        for (size_t i = 1; i < n; i++)
        {
            double z[2];
            mo_complex_get_root((size_t)m, roots8, reverse_bits32((uint32_t)i, logn), &t->root_powers[2 * i]);
            mo_complex_get_root((size_t)m, roots8, (size_t)reverse_bits32((uint32_t)(i - 1), logn) + 1, z);
            t->inv_root_powers[2 * i] = z[0];
            t->inv_root_powers[2 * i + 1] = -z[1];
        }
        free(roots8);
    }
    else
    {
        t->root_powers[2] = 0;
        t->root_powers[3] = 1;
        t->inv_root_powers[2] = 0;
        t->inv_root_powers[3] = -1;
    }
    return t;
}

void mo_ckks_tables_free(mo_ckks_tables *t)
{
    if (t)
    {
        free(t->index_map);
        free(t->root_powers);
        free(t->inv_root_powers);
        free(t);
    }
}

/* std::complex<double> a * r, as libstdc++/libgcc evaluate it for finite operands */
static inline void cmul(const double a[2], const double r[2], double out[2])
{
    double ac = a[0] * r[0], bd = a[1] * r[1], ad = a[0] * r[1], bc = a[1] * r[0];
    out[0] = ac - bd;
    out[1] = ad + bc;
}

void mo_fft_transform_from_rev(double *values, int log_n, const double *roots, const double *scalar)
{
    /* SEAL/util/dwthandler.h:202-356 with Arithmetic<complex<double>, complex<double>, double>
     * (SEAL/ckks.h:46-81).  The reference's 4-way unrolling does not change any operation. */
    const size_t n = (size_t)1 << log_n;
    size_t gap = 1, m = n >> 1, root = 0;
    double d[2];
    for (; m > 1; m >>= 1)
    {
        size_t offset = 0;
        for (size_t i = 0; i < m; i++)
        {
            const double *r = roots + 2 * (++root);
            double *x = values + 2 * offset, *y = x + 2 * gap;
            for (size_t j = 0; j < gap; j++, x += 2, y += 2)
            {
                double u0 = x[0], u1 = x[1], v0 = y[0], v1 = y[1];
                x[0] = u0 + v0;
                x[1] = u1 + v1;
                d[0] = u0 - v0;
                d[1] = u1 - v1;
                cmul(d, r, y);
            }
            offset += gap << 1;
        }
        gap <<= 1;
    }
    const double *r = roots + 2 * (++root);
    double *x = values, *y = x + 2 * gap;
    if (scalar)
    {
        const double s = *scalar;
        const double scaled_r[2] = { r[0] * s, r[1] * s };
        for (size_t j = 0; j < gap; j++, x += 2, y += 2)
        {
            double u0 = x[0], u1 = x[1], v0 = y[0], v1 = y[1];
            x[0] = (u0 + v0) * s;
            x[1] = (u1 + v1) * s;
            d[0] = u0 - v0;
            d[1] = u1 - v1;
            cmul(d, scaled_r, y);
        }
    }
    else
    {
        for (size_t j = 0; j < gap; j++, x += 2, y += 2)
        {
            double u0 = x[0], u1 = x[1], v0 = y[0], v1 = y[1];
            x[0] = u0 + v0;
            x[1] = u1 + v1;
            d[0] = u0 - v0;
            d[1] = u1 - v1;
            cmul(d, r, y);
        }
    }
}

void mo_fft_transform_to_rev(double *values, int log_n, const double *roots)
{
    /* SEAL/util/dwthandler.h:94-191, no scalar */
    const size_t n = (size_t)1 << log_n;
    size_t gap = n >> 1, m = 1, root = 0;
    double v[2];
    for (; m <= (n >> 1); m <<= 1)
    {
        size_t offset = 0;
        for (size_t i = 0; i < m; i++)
        {
            const double *r = roots + 2 * (++root);
            double *x = values + 2 * offset, *y = x + 2 * gap;
            for (size_t j = 0; j < gap; j++, x += 2, y += 2)
            {
                double u0 = x[0], u1 = x[1];
                cmul(y, r, v);
                x[0] = u0 + v[0];
                x[1] = u1 + v[1];
                y[0] = u0 - v[0];
                y[1] = u1 - v[1];
            }
            offset += gap << 1;
        }
        gap >>= 1;
    }
}

#define MO_MAX_WORDS 64

/* value mod q of a little-endian multi-word integer: what RNSBase::decompose (SEAL/util/rns.cpp,
 * via modulo_uint, SEAL/util/uintarithsmallmod.h) leaves in each residue */
static uint64_t modulo_words(const uint64_t *w, size_t count, const mo_modulus *m)
{
    if (count == 0)
    {
        return 0;
    }
    if (count == 1)
    {
        return mo_barrett_reduce_64(w[0], m);
    }
    uint64_t acc = mo_barrett_reduce_64(w[count - 1], m);
    for (size_t i = count - 1; i-- > 0;)
    {
        uint64_t t[2] = { w[i], acc };
        acc = mo_barrett_reduce_128(t, m);
    }
    return acc;
}

/* ckks.h:549-629 / ckks.cpp:129-211: residues of round()-ed coefficient `coeffd_in` */
static void decompose_coeff(const mo_context *c, double coeffd_in, int bit_count, size_t L,
                            const uint32_t *prime_index, uint64_t *out, size_t stride)
{
    const double two_pow_64 = pow(2.0, 64);
    double coeffd = round(coeffd_in);
    int is_negative = signbit(coeffd);
    coeffd = fabs(coeffd);
    for (size_t j = 0; j < L; j++)
    {
        const mo_modulus *q = &c->mods[prime_index ? prime_index[j] : j];
        uint64_t r;
        if (bit_count <= 64)
        {
            r = mo_barrett_reduce_64((uint64_t)coeffd, q);
        }
        else if (bit_count <= 128)
        {
            uint64_t w[2] = { (uint64_t)fmod(coeffd, two_pow_64), (uint64_t)(coeffd / two_pow_64) };
            r = mo_barrett_reduce_128(w, q);
        }
        else
        {
            uint64_t w[MO_MAX_WORDS];
            size_t cnt = 0;
            double cd = coeffd;
            while (cd >= 1 && cnt < MO_MAX_WORDS)
            {
                w[cnt++] = (uint64_t)fmod(cd, two_pow_64);
                cd /= two_pow_64;
            }
            r = modulo_words(w, cnt, q);
        }
        out[j * stride] = is_negative ? mo_negate_uint_mod(r, q) : r;
    }
}

int mo_ckks_encode(const mo_context *c, const mo_ckks_tables *t, const double *values, int is_complex,
                   size_t count, size_t L, const uint32_t *prime_index, double scale, int total_bits,
                   uint64_t *dst, int *max_coeff_bit_count)
{
    /* SEAL/ckks.h:457-637 */
    const size_t n = t->n, slots = t->slots;
    if (count > slots)
    {
        return -1;
    }
    if (scale <= 0 || ((int)log2(scale) + 1 >= total_bits))
    {
        return -2;
    }
    double *conj_values = (double *)calloc(2 * n, sizeof(double));
    for (size_t i = 0; i < count; i++)
    {
        double re = is_complex ? values[2 * i] : values[i];
        double im = is_complex ? values[2 * i + 1] : 0.0;
        conj_values[2 * t->index_map[i]] = re;
        conj_values[2 * t->index_map[i] + 1] = im;
        conj_values[2 * t->index_map[i + slots]] = re;
        conj_values[2 * t->index_map[i + slots] + 1] = -im;
    }
    double fix = scale / (double)n;
    mo_fft_transform_from_rev(conj_values, t->logn, t->inv_root_powers, &fix);

    double max_coeff = 0;
    for (size_t i = 0; i < n; i++)
    {
        double a = fabs(conj_values[2 * i]);
        max_coeff = a > max_coeff ? a : max_coeff;
    }
    int bits = (int)ceil(log2(max_coeff > 1.0 ? max_coeff : 1.0)) + 1;
    if (max_coeff_bit_count)
    {
        *max_coeff_bit_count = bits;
    }
    if (bits >= total_bits)
    {
        free(conj_values);
        return -3;
    }
    for (size_t i = 0; i < n; i++)
    {
        decompose_coeff(c, conj_values[2 * i], bits, L, prime_index, dst + i, n);
    }
    free(conj_values);
    for (size_t j = 0; j < L; j++)
    {
        mo_ntt_negacyclic_harvey(dst + j * n, &c->tables[prime_index ? prime_index[j] : j]);
    }
    return 0;
}

int mo_ckks_encode_scalar(const mo_context *c, double value, size_t L, const uint32_t *prime_index,
                          double scale, int total_bits, uint64_t *rows)
{
    /* SEAL/ckks.cpp:77-216 */
    if (scale <= 0 || ((int)log2(scale) >= total_bits))
    {
        return -2;
    }
    value *= scale;
    int coeff_bit_count = (int)log2(fabs(value)) + 2;
    if (coeff_bit_count >= total_bits)
    {
        return -3;
    }
    decompose_coeff(c, value, coeff_bit_count, L, prime_index, rows, 1);
    return 0;
}
