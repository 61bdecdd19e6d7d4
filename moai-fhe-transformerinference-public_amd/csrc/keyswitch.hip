// keyswitch.hip -- rescale, key switching (relinearize / apply_galois) and modraise.
//
// Reference: RNSTool::divide_and_round_q_last_ntt_inplace (SEAL/util/rns.cpp:830-901),
// Evaluator::switch_key_inplace (SEAL/evaluator.cpp:2724-3020), relinearize_internal (:1345-1400),
// apply_galois_inplace (:2563-2665), Bootstrapper::modraise_inplace
// (include/source/bootstrapping/Bootstrapper.cpp:2938-2992).
//
// Structure on the device (all batched over the leading ciphertext index):
//   rescale:     last = INTT(c_last) -> u_i = [last + q_last/2]_{q_last} mod q_i + fix_i -> NTT_{q_i}(u_i)
//                -> out_i = (c_i - u_i) * q_last^-1
//   switch_key:  t = INTT(target); for each output modulus I in {q_0..q_{L-1}, p}:
//                    ops_J = NTT_{q_I}(t_J mod q_I) for every digit J (for I == J this reproduces
//                    target_J exactly, so no special case is needed),
//                    acc_{K,I} = sum_J ops_J (*) key[J][K][I]   (128-bit accumulate, one Barrett);
//                then the same "divide by the last modulus and round" step with p as the last
//                modulus, accumulated into the ciphertext.
// The loop over I is outermost so that the 2*L key rows of modulus I are read once per batch.
#include <mutex>

#include <cstdlib>
#include <cstring>

#include "launch.h"
#include <vector>

#include "keyswitch_kernels.hip.h"

namespace moai {

// out row y (group = y / rows_per_group, w = y % rows_per_group) = in row group * in_stride + in_off + w;
// with zero_from >= 0, rows whose w >= zero_from are written as zero instead
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint64_t *in, uint64_t *out, uint32_t rows_per_group,
                                                        uint32_t in_stride, uint32_t in_off, uint32_t zero_from,
                                                        uint32_t n2)
{
    const uint32_t grp = blockIdx.y / rows_per_group;
    const uint32_t w = blockIdx.y % rows_per_group;
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(in) + ((size_t)grp * in_stride + in_off + w) * n2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(out) + (size_t)blockIdx.y * n2;
    const bool zero = w >= zero_from;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n2; j += gridDim.x * 256u)
    {
        ulonglong2 v;
        v.x = 0;
        v.y = 0;
        d[j] = zero ? v : s[j];
    }
}

static const uint32_t NO_ZERO = 0xffffffffu;

// out row y = sum over `splits` partial copies of in row (y * in_stride + in_off), mod the one prime
__global__ __launch_bounds__(256) void sum_rows_kernel(const uint64_t *in, uint64_t *out, uint32_t in_stride, uint32_t in_off,
                                                       uint32_t splits, size_t split_stride, const PrimeConst *pc,
                                                       uint32_t prime, uint32_t n2)
{
    const uint64_t q = pc[prime].q;
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(in) + ((size_t)blockIdx.y * in_stride + in_off) * n2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(out) + (size_t)blockIdx.y * n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n2; j += gridDim.x * 256u)
    {
        ulonglong2 v = s[j];
        for (uint32_t sp = 1; sp < splits; ++sp)
        {
            ulonglong2 y = s[j + sp * (split_stride >> 1)];
            v.x = csub(v.x + y.x, q);
            v.y = csub(v.y + y.y, q);
        }
        d[j] = v;
    }
}

struct ExpandLastArgs
{
    const uint64_t *last;  // [P][N], canonical under prime_last
    uint64_t *u;           // [P][Lout][N]
    const PrimeConst *pc;
    uint32_t prime_last;
    uint32_t Lout;
    uint32_t n2;
};

// u[p][i] = ([last + q_last/2] mod q_last) mod q_i + (q_i - (q_last/2 mod q_i))     in [0, 2 q_i)
// (rns.cpp:858-878 / evaluator.cpp:2964-2992)
__global__ __launch_bounds__(256) void expand_last_kernel(ExpandLastArgs g)
{
    const uint32_t p = blockIdx.y / g.Lout;
    const uint32_t i = blockIdx.y % g.Lout;
    const uint64_t ql = g.pc[g.prime_last].q;
    const uint64_t half = ql >> 1;
    const PrimeConst *pc = g.pc + i;
    const uint64_t q = pc->q, cr1 = pc->cr1;
    const uint64_t fix = q - barrett64(half, q, cr1);
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(g.last) + (size_t)p * g.n2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(g.u) + (size_t)blockIdx.y * g.n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < g.n2; j += gridDim.x * 256u)
    {
        ulonglong2 v = s[j];
        ulonglong2 r;
        r.x = barrett64(csub(v.x + half, ql), q, cr1) + fix;
        r.y = barrett64(csub(v.y + half, ql), q, cr1) + fix;
        d[j] = r;
    }
}

struct FinalizeArgs
{
    const uint64_t *acc;   // row (p, i) at acc + (p * acc_stride + i) * N
    const uint64_t *u;     // [P][Lout][N] canonical
    uint64_t *out;         // [P][Lout][N]
    const PrimeConst *pc;
    const Tw *inv_last;    // inv_qlast + prime_last * k : q_last^-1 mod q_i
    uint32_t acc_stride;
    uint32_t Lout;
    uint32_t n2;
    const uint64_t *addend; // see ModDownArgs
    const uint64_t *addend2;
    uint32_t addend_bstride;
    int add_mode;
};

// (acc_i - u_i) * q_last^-1 mod q_i      (rns.cpp:892-897 / evaluator.cpp:3012-3017)
__global__ __launch_bounds__(256) void moddown_finalize_kernel(FinalizeArgs g)
{
    const uint32_t p = blockIdx.y / g.Lout;
    const uint32_t i = blockIdx.y % g.Lout;
    const uint64_t q = g.pc[i].q;
    const Tw inv = g.inv_last[i];
    const ulonglong2 *a = reinterpret_cast<const ulonglong2 *>(g.acc) + ((size_t)p * g.acc_stride + i) * g.n2;
    const ulonglong2 *u = reinterpret_cast<const ulonglong2 *>(g.u) + (size_t)blockIdx.y * g.n2;
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)blockIdx.y * g.n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < g.n2; j += gridDim.x * 256u)
    {
        ulonglong2 x = a[j], y = u[j], r;
        r.x = csub(mul_shoup_lazy(x.x + q - y.x, inv.w, inv.wq, q), q);
        r.y = csub(mul_shoup_lazy(x.y + q - y.y, inv.w, inv.wq, q), q);
        if (g.add_mode == 1 || (g.add_mode == 2 && !(p & 1u)))
        {
            ulonglong2 c = (reinterpret_cast<const ulonglong2 *>(g.addend) +
                            ((size_t)(p >> 1) * g.addend_bstride + (size_t)(p & 1u) * g.Lout + i) * g.n2)[j];
            r.x = csub(r.x + c.x, q);
            r.y = csub(r.y + c.y, q);
        }
        if (g.addend2)
        {
            ulonglong2 c = (reinterpret_cast<const ulonglong2 *>(g.addend2) + (size_t)blockIdx.y * g.n2)[j];
            r.x = csub(r.x + c.x, q);
            r.y = csub(r.y + c.y, q);
        }
        o[j] = r;
    }
}

// ops[b][J] = t[b][J] mod q_I   (evaluator.cpp:2844-2854); all rows under one prime
__global__ __launch_bounds__(256) void reduce_rows_kernel(const uint64_t *t, uint64_t *ops, const PrimeConst *pc,
                                                          uint32_t prime, uint32_t n2)
{
    const uint64_t q = pc[prime].q, cr1 = pc[prime].cr1;
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(t) + (size_t)blockIdx.y * n2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(ops) + (size_t)blockIdx.y * n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n2; j += gridDim.x * 256u)
    {
        ulonglong2 v = s[j];
        v.x = barrett64(v.x, q, cr1);
        v.y = barrett64(v.y, q, cr1);
        d[j] = v;
    }
}

struct MacArgs
{
    const uint64_t *ops;   // [B][L][N] NTT form under prime I
    const uint64_t *key;   // [k-1][2][k][N]
    uint64_t *acc;         // [B][2][L+1][N]
    const PrimeConst *pc;
    uint32_t L;
    uint32_t k;            // rows per key polynomial in the key's layout
    uint32_t krow;         // the key row of prime I in that layout (the special prime's is the last)
    uint32_t prime;        // I (context prime index)
    uint32_t slot;         // row of acc to write (I, or L for the special prime)
    uint32_t n2;
};

__device__ __forceinline__ void mac128(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b)
{
    uint64_t pl = a * b;
    uint64_t ph = mulhi64(a, b);
    lo += pl;
    hi += ph + (lo < pl ? 1 : 0);
}

// acc[b][K][slot] = sum_J ops[b][J] (*) key[J][K][prime]  mod q      (evaluator.cpp:2858-2910)
// blockIdx.y = b
__global__ __launch_bounds__(256) void keyswitch_mac_kernel(MacArgs g)
{
    const PrimeConst *pc = g.pc + g.prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const uint32_t b = blockIdx.y;
    const size_t n2 = g.n2;
    const ulonglong2 *ops = reinterpret_cast<const ulonglong2 *>(g.ops) + (size_t)b * g.L * n2;
    const ulonglong2 *key = reinterpret_cast<const ulonglong2 *>(g.key);
    ulonglong2 *acc0 = reinterpret_cast<ulonglong2 *>(g.acc) + ((size_t)(b * 2 + 0) * (g.L + 1) + g.slot) * n2;
    ulonglong2 *acc1 = reinterpret_cast<ulonglong2 *>(g.acc) + ((size_t)(b * 2 + 1) * (g.L + 1) + g.slot) * n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < g.n2; j += gridDim.x * 256u)
    {
        uint64_t l0x = 0, h0x = 0, l0y = 0, h0y = 0, l1x = 0, h1x = 0, l1y = 0, h1y = 0;
        for (uint32_t J = 0; J < g.L; ++J)
        {
            ulonglong2 o = ops[(size_t)J * n2 + j];
            ulonglong2 k0 = key[((size_t)(J * 2 + 0) * g.k + g.krow) * n2 + j];
            ulonglong2 k1 = key[((size_t)(J * 2 + 1) * g.k + g.krow) * n2 + j];
            mac128(l0x, h0x, o.x, k0.x);
            mac128(l0y, h0y, o.y, k0.y);
            mac128(l1x, h1x, o.x, k1.x);
            mac128(l1y, h1y, o.y, k1.y);
        }
        ulonglong2 r0, r1;
        r0.x = barrett128(l0x, h0x, q, cr0, cr1);
        r0.y = barrett128(l0y, h0y, q, cr0, cr1);
        r1.x = barrett128(l1x, h1x, q, cr0, cr1);
        r1.y = barrett128(l1y, h1y, q, cr0, cr1);
        acc0[j] = r0;
        acc1[j] = r1;
    }
}

struct RaiseArgs
{
    const uint64_t *src;  // [P][N] coefficient form, canonical mod q0
    uint64_t *out;        // [P][Lout][N]
    const PrimeConst *pc;
    uint32_t Lout;
    uint32_t n2;
};

// Bootstrapper.cpp:2964-2988: dest = src mod q_j, plus (q_j - q0 mod q_j) when src > q0/2
__global__ __launch_bounds__(256) void modraise_kernel(RaiseArgs g)
{
    const uint32_t p = blockIdx.y / g.Lout;
    const uint32_t jrow = blockIdx.y % g.Lout;
    const uint64_t q0 = g.pc[0].q;
    const uint64_t q = g.pc[jrow].q, cr1 = g.pc[jrow].cr1;
    const uint64_t minus_q0 = jrow == 0 ? 0 : q - barrett64(q0, q, cr1);
    const uint64_t half = q0 >> 1;
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(g.src) + (size_t)p * g.n2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(g.out) + (size_t)blockIdx.y * g.n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < g.n2; j += gridDim.x * 256u)
    {
        ulonglong2 v = s[j], r;
        r.x = barrett64(v.x, q, cr1);
        r.y = barrett64(v.y, q, cr1);
        if (v.x > half)
        {
            r.x = csub(r.x + minus_q0, q);
        }
        if (v.y > half)
        {
            r.y = csub(r.y + minus_q0, q);
        }
        d[j] = r;
    }
}

static inline dim3 rgrid(const moai_ctx *c, size_t rows)
{
    uint32_t n2 = (uint32_t)(c->n >> 1);
    uint32_t bx = (n2 + 255u) / 256u;
    return dim3(bx ? bx : 1, (uint32_t)rows);
}

static inline size_t align256(size_t x)
{
    return (x + 255) & ~(size_t)255;
}

// Shared tail of rescale and key switch: divide rows [0, Lout) of `acc` by the modulus `prime_last`
// whose NTT-form row is `last_rows` ([P][N], overwritten), rounding to nearest.
//   acc row (p, i) = acc + (p * acc_stride + i) * N ; out [P][Lout][N]
// scratch: u [P][Lout][N]
//   addend / addend_bstride / add_mode: what is added to the quotient (ModDownArgs); add_mode 0 for rescale
//   addend2: a second summand with the output's layout (may be `out`), or null
static int moddown(moai_ctx *c, uint64_t *last_rows, const uint64_t *acc, uint32_t acc_stride, uint64_t *u,
                   uint64_t *out, size_t P, size_t Lout, uint32_t prime_last, const uint64_t *addend, uint32_t addend_bstride,
                   int add_mode, hipStream_t s, uint32_t acc_splits = 1, size_t acc_split_stride = 0, const Tw *scal = nullptr,
                   const uint64_t *addend2 = nullptr)
{
    RowMap rm;
    uint32_t pl = prime_last;
    int rc = make_rowmap(c, 1, &pl, &rm);
    if (rc)
    {
        return rc;
    }
    rc = ntt_launch(c, last_rows, P, 1, rm, true, s);
    if (rc)
    {
        return rc;
    }
    if (c->logn >= 12)
    {
        // fused: the expand rides on the strided pass's loads, the division on the contiguous pass's stores
        ModDownArgs a;
        a.last = last_rows;
        a.u = u;
        a.acc = acc;
        a.out = out;
        a.tw = c->fwd_tw;
        a.twb = c->fwd_twb;
        a.pc = c->pc;
        a.inv_last = c->inv_qlast + (size_t)prime_last * c->k;
        a.prime_last = prime_last;
        a.acc_stride = acc_stride;
        a.Lout = (uint32_t)Lout;
        a.P = (uint32_t)P;
        a.addend = addend;
        a.addend2 = addend2;
        a.addend_bstride = addend_bstride;
        a.add_mode = add_mode;
        a.has_scal = scal ? 1 : 0;
        if (scal)
        {
            for (size_t i = 0; i < Lout; ++i)
            {
                a.scal[i] = scal[i];
            }
        }
        a.acc_splits = acc_splits;
        a.acc_split_stride = acc_split_stride;
        // one pair of launches per arithmetic mode present among the output moduli (ntt_mode); below
        // MOAI_MD_FP_MIN_ROWS rows the extra launches cost more than the FP64 butterflies save (a single
        // ciphertext at MOAI's top level: 162 vs 147 ms per bootstrap; packs of 16: 66.0 vs 67.1 ms)
        const long fp_min_rows = tuning("MOAI_MD_FP_MIN_ROWS", 256);
        const bool allow_fp = (long)(P * Lout) >= fp_min_rows;
        for (int mode = M_GUARD; mode <= M_FPR; ++mode)
        {
            a.Lsel = 0;
            for (size_t i = 0; i < Lout; ++i)
            {
                int m = ntt_mode(c, (uint32_t)i);
                if (m >= M_FPN && !allow_fp)
                {
                    m = noguard_ok(c->primes[i]) ? M_NOGUARD : M_GUARD;
                }
                if (m == mode)
                {
                    a.sel.idx[a.Lsel++] = (uint32_t)i;
                }
            }
            if (!a.Lsel)
            {
                continue;
            }
            a.total_work = (uint32_t)(P * a.Lsel * (c->n >> 12));
            a.tw = mode >= M_FPN ? c->fwd_twf : c->fwd_tw;
            a.twb = mode >= M_FPN ? c->fwd_twfb : c->fwd_twb;
#define MOAI_MD_MODE(LG, MD)                                                                         \
    hipLaunchKernelGGL((moddown_strided<LG, MD>), dim3(a.total_work), dim3(256), 0, s, a);           \
    hipLaunchKernelGGL((moddown_contig<LG, MD>), dim3(a.total_work), dim3(256), 0, s, a);
#define MOAI_MD_CASE(LG)                                     \
    case LG:                                                 \
        switch (mode)                                        \
        {                                                    \
        case M_GUARD: MOAI_MD_MODE(LG, M_GUARD) break;       \
        case M_NOGUARD: MOAI_MD_MODE(LG, M_NOGUARD) break;   \
        case M_FPN: MOAI_MD_MODE(LG, M_FPN) break;           \
        default: MOAI_MD_MODE(LG, M_FPR) break;              \
        }                                                    \
        break;
            switch (c->logn)
            {
                MOAI_MD_CASE(12)
                MOAI_MD_CASE(13)
                MOAI_MD_CASE(14)
                MOAI_MD_CASE(15)
                MOAI_MD_CASE(16)
            }
#undef MOAI_MD_CASE
#undef MOAI_MD_MODE
        }
        MOAI_LAUNCH_CHECK();
        return MOAI_OK;
    }
    if (scal)
    {
        return set_error(MOAI_ELOGIC, "fused scalar product needs the tiled transform");
    }
    ExpandLastArgs e;
    e.last = last_rows;
    e.u = u;
    e.pc = c->pc;
    e.prime_last = prime_last;
    e.Lout = (uint32_t)Lout;
    e.n2 = (uint32_t)(c->n >> 1);
    MOAI_CHECK_GRID_ROWS(P * Lout);
    hipLaunchKernelGGL(expand_last_kernel, rgrid(c, P * Lout), dim3(256), 0, s, e);
    MOAI_LAUNCH_CHECK();
    rc = make_rowmap(c, Lout, nullptr, &rm);
    if (rc)
    {
        return rc;
    }
    rc = ntt_launch(c, u, P, Lout, rm, false, s);
    if (rc)
    {
        return rc;
    }
    FinalizeArgs f;
    f.acc = acc;
    f.u = u;
    f.out = out;
    f.pc = c->pc;
    f.inv_last = c->inv_qlast + (size_t)prime_last * c->k;
    f.acc_stride = acc_stride;
    f.Lout = (uint32_t)Lout;
    f.n2 = (uint32_t)(c->n >> 1);
    f.addend = addend;
    f.addend2 = addend2;
    f.addend_bstride = addend_bstride;
    f.add_mode = add_mode;
    MOAI_CHECK_GRID_ROWS(P * Lout);
    hipLaunchKernelGGL(moddown_finalize_kernel, rgrid(c, P * Lout), dim3(256), 0, s, f);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

static int check_level(const moai_ctx *c, size_t L, size_t polys)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    if (L == 0 || L > c->k)
    {
        return set_error(MOAI_EINVAL, "L = %zu out of range for a context of %zu primes", L, c->k);
    }
    if (polys * (L + 1) > 0x7fffffffull)
    {
        return set_error(MOAI_EINVAL, "batch too large for one launch");
    }
    return enter_device(c);
}

struct KsTarget
{
    const uint64_t *ptr; // NTT-form target rows
    uint32_t stride_rows, off_rows;
    uint32_t key_rows;   // rows per key polynomial in the key's layout (key_rows_for)
};

// rows per key polynomial of `key` (k for the reference's layout, levels + 1 for a key moai_key_trim produced), after checking
// that the key holds what a switch at L data primes reads: digits J < L and rows {0 .. L-1, special} (SEAL/evaluator.cpp:2818,2831)
static int key_rows_for(moai_ctx *c, const uint64_t *key, size_t L, uint32_t *rows)
{
    std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
    auto it = c->key_layouts.find(key);
    if (it == c->key_layouts.end())
    {
        *rows = (uint32_t)c->k;
        return MOAI_OK;
    }
    if (L > it->second.digits || L + 1 > it->second.rows)
    {
        return set_error(MOAI_ERANGE, "the key was trimmed to %u levels and cannot switch a ciphertext of %zu data primes",
                         it->second.digits, L);
    }
    *rows = it->second.rows;
    return MOAI_OK;
}

template <int LOGN>
static int ks_fused_group(moai_ctx *c, const uint64_t *t, uint64_t *tmp, const uint64_t *key, uint64_t *acc, size_t L,
                          size_t batch, const KsGroup &grp, size_t G, uint32_t splits, int mode, const KsTarget &tg, hipStream_t s);

// number of output moduli whose digits are in flight at once: bounded by the scratch budget
// (MOAI_KS_TMP_MB, default 8192 MiB) so that small batches expose (L+1) x 16 tiles of parallelism in
// one launch while large batches stay within a few GiB of workspace
static size_t ks_group_size(const moai_ctx *c, size_t L, size_t batch)
{
    long budget_mb = tuning("MOAI_KS_TMP_MB", 8192);
    if (budget_mb < 1)
    {
        budget_mb = 1;
    }
    const size_t per_modulus = batch * L * c->n * sizeof(uint64_t);
    size_t g = ((size_t)budget_mb << 20) / (per_modulus ? per_modulus : 1);
    if (g < 1)
    {
        g = 1;
    }
    if (g > L + 1)
    {
        g = L + 1;
    }
    return g;
}

// Few ciphertexts leave the MAC kernel with (L+1)*16*batch long-running workgroups, barely more than the
// chip holds at two per CU, so its second round runs almost empty.  Splitting the digit range S ways gives
// S times more, S times shorter workgroups; the partial sums are added where they are consumed.
static uint32_t ks_splits(const moai_ctx *c, size_t L, size_t batch)
{
    if (c->logn < 12)
    {
        return 1;
    }
    const size_t wgs = batch * (L + 1) * (c->n >> 12);
    const size_t want = (size_t)c->num_cu * 8;
    size_t s = (want + wgs - 1) / wgs;
    if (s > 8)
    {
        s = 8;
    }
    if (s > L)
    {
        s = L;
    }
    if (s < 1)
    {
        s = 1;
    }
    // no empty split: with chunks of ceil(L/s) digits, ceil(L/chunk) splits cover the range exactly
    const size_t chunk = (L + s - 1) / s;
    return (uint32_t)((L + chunk - 1) / chunk);
}

static size_t ks_tmp_rows(const moai_ctx *c, size_t L, size_t batch)
{
    size_t fused = c->logn >= 12 ? batch * ks_group_size(c, L, batch) * L : 0;
    size_t plain = 2 * batch * L; // ops [B][L] (unfused path) and u [2B][L] (mod-down)
    return fused > plain ? fused : plain;
}

static size_t switch_key_ws_bytes(const moai_ctx *c, size_t L, size_t batch)
{
    const size_t row_bytes = c->n * sizeof(uint64_t);
    return align256(batch * L * row_bytes) + align256(ks_tmp_rows(c, L, batch) * row_bytes) +
           align256(ks_splits(c, L, batch) * batch * 2 * (L + 1) * row_bytes) + align256(batch * 2 * row_bytes);
}

// which arithmetic the fused kernels use for output modulus `prime` (keyswitch_kernels.hip.h): the forward
// transform's mode, with the integer no-guard form only when the lazy digit may also enter the MAC unreduced
static int ks_mode(const moai_ctx *c, uint32_t prime, size_t L, bool allow_fp)
{
    int m = ntt_mode(c, prime);
    if (m >= M_FPN && !allow_fp)
    {
        m = noguard_ok(c->primes[prime]) ? M_NOGUARD : M_GUARD;
    }
    if (m != M_NOGUARD)
    {
        return m;
    }
    // 36 q * q * L < 2^128
    const uint64_t q = c->primes[prime];
    unsigned __int128 lim = (unsigned __int128)q * q;
    return lim < ((~(unsigned __int128)0) / (36 * (unsigned __int128)(L ? L : 1))) ? M_NOGUARD : M_GUARD;
}

template <int LOGN, int MODE>
static int ks_fused_group_mode(moai_ctx *c, const uint64_t *t, uint64_t *tmp, const uint64_t *key, uint64_t *acc, size_t L,
                               size_t batch, const KsGroup &grp, size_t G, uint32_t splits, const KsTarget &tg, hipStream_t s)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    KsP1Args p1;
    p1.t = t;
    p1.tmp = tmp;
    p1.tw = MODE >= M_FPN ? c->fwd_twf : c->fwd_tw;
    p1.pc = c->pc;
    p1.grp = grp;
    p1.L = (uint32_t)L;
    p1.G = (uint32_t)G;
    p1.total_work = (uint32_t)(batch * G * L * TPR);
    p1.tw1 = c->fwd_twf1;
    // N = 2^16, FP64 modes, enough work to fill the chip that way: eight tiles per workgroup, software-pipelined
    // (fwd_strided_tiles; MOAI_KS_P1_ITEMS=1: one tile per workgroup)
    constexpr int P1_ITEMS = (LOGN == 16 && MODE >= M_FPN) ? 8 : 1;
    if (P1_ITEMS > 1 && p1.total_work >= 8u * 2048u && tuning("MOAI_KS_P1_ITEMS", 8) > 1)
    {
        hipLaunchKernelGGL((ks_fwd_strided<LOGN, MODE, false, P1_ITEMS>), dim3(p1.total_work / P1_ITEMS), dim3(256), 0, s, p1);
    }
    else if (MODE == M_FPN && tuning("MOAI_KS_P1_PRE", 0))
    {
        hipLaunchKernelGGL((ks_fwd_strided<LOGN, MODE, MODE == M_FPN>), dim3(p1.total_work), dim3(256), 0, s, p1);
    }
    else
    {
        hipLaunchKernelGGL((ks_fwd_strided<LOGN, MODE>), dim3(p1.total_work), dim3(256), 0, s, p1);
    }
    MOAI_LAUNCH_CHECK();
    KsP2Args p2;
    p2.tmp = tmp;
    p2.tgt = tg.ptr;
    p2.tgt_stride = tg.stride_rows;
    p2.tgt_off = tg.off_rows;
    p2.key = key;
    p2.acc = acc;
    p2.tw = MODE >= M_FPN ? c->fwd_twf : c->fwd_tw;
    p2.tw1 = c->fwd_twf1;
    p2.pc = c->pc;
    p2.grp = grp;
    p2.L = (uint32_t)L;
    p2.G = (uint32_t)G;
    p2.k = tg.key_rows;
    p2.B = (uint32_t)batch;
    p2.S = splits;
    p2.jchunk = (uint32_t)((L + splits - 1) / splits);
    p2.split_stride = batch * 2 * (L + 1) * c->n;
    p2.total_work = (uint32_t)(batch * G * TPR * 2 * splits); // 2048-coefficient tiles
    // where the MAC fetches its key residues (keyswitch_kernels.hip.h): all eight loads at the head of the MAC by default --
    // same box, same hour, l = 35 / 15, batch 64: 0.572 / 0.134 ms per ciphertext with the compiler's placement (0), 0.532 / 0.121
    // with 1, 0.547 / 0.124 with 2 (profiles/r03_b_ks_kernel_variants_ab.txt)
    const long pf = MODE >= M_FPN ? tuning("MOAI_KS_MAC_PF", 1) : 0;
    if (pf == 1)
    {
        hipLaunchKernelGGL((ks_contig_mac8<LOGN, MODE, (MODE >= M_FPN ? 1 : 0)>), dim3(p2.total_work), dim3(256), 0, s, p2);
    }
    else if (pf == 2)
    {
        hipLaunchKernelGGL((ks_contig_mac8<LOGN, MODE, (MODE >= M_FPN ? 2 : 0)>), dim3(p2.total_work), dim3(256), 0, s, p2);
    }
    else
    {
        hipLaunchKernelGGL((ks_contig_mac8<LOGN, MODE>), dim3(p2.total_work), dim3(256), 0, s, p2);
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

template <int LOGN>
static int ks_fused_group(moai_ctx *c, const uint64_t *t, uint64_t *tmp, const uint64_t *key, uint64_t *acc, size_t L,
                          size_t batch, const KsGroup &grp, size_t G, uint32_t splits, int mode, const KsTarget &tg, hipStream_t s)
{
    switch (mode)
    {
    case M_FPN: return ks_fused_group_mode<LOGN, M_FPN>(c, t, tmp, key, acc, L, batch, grp, G, splits, tg, s);
    case M_FPR: return ks_fused_group_mode<LOGN, M_FPR>(c, t, tmp, key, acc, L, batch, grp, G, splits, tg, s);
    case M_NOGUARD: return ks_fused_group_mode<LOGN, M_NOGUARD>(c, t, tmp, key, acc, L, batch, grp, G, splits, tg, s);
    default: return ks_fused_group_mode<LOGN, M_GUARD>(c, t, tmp, key, acc, L, batch, grp, G, splits, tg, s);
    }
}

// target row block of ciphertext b starts at target + (b * target_stride_rows + target_off_rows) * N.
// wsp: switch_key_ws_bytes() bytes of scratch.
// ct [batch][2][L][N] = addend (add_mode, ModDownArgs) + key switch of `target`; ct may be the addend itself
static int switch_key_impl(moai_ctx *c, uint64_t *ct, const uint64_t *target, size_t target_stride_rows,
                           size_t target_off_rows, const uint64_t *key, size_t L, size_t batch, void *wsp,
                           hipStream_t s, const uint64_t *addend, size_t addend_bstride, int add_mode, const uint64_t *addend2 = nullptr)
{
    const size_t n = c->n;
    const size_t k = c->k;
    if (k < 2)
    {
        return set_error(MOAI_ELOGIC, "keyswitching is not supported by the context");
    }
    if (L > k - 1)
    {
        return set_error(MOAI_EINVAL, "L exceeds the key's decomposition size");
    }
    uint32_t key_rows = 0;
    {
        const int krc = key_rows_for(c, key, L, &key_rows);
        if (krc)
        {
            return krc;
        }
    }
    const size_t row_bytes = n * sizeof(uint64_t);
    const size_t sz_t = align256(batch * L * row_bytes);
    const size_t sz_ops = align256(ks_tmp_rows(c, L, batch) * row_bytes); // digits in flight; later u [2B][L][N]
    const uint32_t splits = ks_splits(c, L, batch);
    const size_t split_stride = batch * 2 * (L + 1) * n; // words
    const size_t sz_acc = align256(splits * split_stride * sizeof(uint64_t));
    uint64_t *t = static_cast<uint64_t *>(wsp);
    uint64_t *ops = reinterpret_cast<uint64_t *>(static_cast<char *>(wsp) + sz_t);
    uint64_t *acc = reinterpret_cast<uint64_t *>(static_cast<char *>(wsp) + sz_t + sz_ops);
    uint64_t *last = reinterpret_cast<uint64_t *>(static_cast<char *>(wsp) + sz_t + sz_ops + sz_acc);
    const uint32_t n2 = (uint32_t)(n >> 1);

    // 1. t = INTT(target)      (evaluator.cpp:2804-2812)
    RowMap rm;
    int rc = make_rowmap(c, L, nullptr, &rm);
    if (rc)
    {
        return rc;
    }
    rc = ntt_launch(c, t, batch, L, rm, true, s, target, target_stride_rows, target_off_rows);
    if (rc)
    {
        return rc;
    }
    // 2. inner products per output modulus    (evaluator.cpp:2817-2911)
    if (c->logn >= 12)
    {
        // fused: "mod q_I" rides on the strided pass's loads, the key MAC on the contiguous pass
        const size_t G = ks_group_size(c, L, batch);
        if (batch * G * L * (n >> 11) > 0x7fffffffull)
        {
            return set_error(MOAI_EINVAL, "batch too large for one launch");
        }
        // output moduli (I = L stands for the special prime) ordered by arithmetic mode; a launch covers up to
        // G of them, all of one mode
        // every mode is one more pair of launches: a few ciphertexts at a low level are launch-bound and stay
        // on the single integer group (measured: FP64 pays from about 16 digit rows per call)
        const bool allow_fp = (long)(batch * L) >= tuning("MOAI_KS_FP_MIN_ROWS", 16);
        KsTarget tg;
        tg.ptr = target;
        tg.stride_rows = (uint32_t)target_stride_rows;
        tg.off_rows = (uint32_t)target_off_rows;
        tg.key_rows = key_rows;
        std::vector<uint32_t> order;
        std::vector<int> order_mode;
        for (int mode = M_FPR; mode >= M_GUARD; --mode)
        {
            for (size_t Iidx = 0; Iidx <= L; ++Iidx)
            {
                const uint32_t prime = (uint32_t)(Iidx == L ? k - 1 : Iidx);
                if (ks_mode(c, prime, L, allow_fp) == mode)
                {
                    order.push_back((uint32_t)Iidx);
                    order_mode.push_back(mode);
                }
            }
        }
        for (size_t o0 = 0; o0 < order.size();)
        {
            const int mode = order_mode[o0];
            size_t g = 0;
            while (o0 + g < order.size() && g < G && order_mode[o0 + g] == mode)
            {
                ++g;
            }
            KsGroup grp;
            for (size_t i = 0; i < MOAI_MAX_RNS; ++i)
            {
                size_t Iidx = order[o0 + (i < g ? i : 0)];
                grp.prime[i] = (uint32_t)(Iidx == L ? k - 1 : Iidx);
                grp.slot[i] = (uint32_t)Iidx;
            }
            switch (c->logn)
            {
            case 12:
                rc = ks_fused_group<12>(c, t, ops, key, acc, L, batch, grp, g, splits, mode, tg, s);
                break;
            case 13:
                rc = ks_fused_group<13>(c, t, ops, key, acc, L, batch, grp, g, splits, mode, tg, s);
                break;
            case 14:
                rc = ks_fused_group<14>(c, t, ops, key, acc, L, batch, grp, g, splits, mode, tg, s);
                break;
            case 15:
                rc = ks_fused_group<15>(c, t, ops, key, acc, L, batch, grp, g, splits, mode, tg, s);
                break;
            default:
                rc = ks_fused_group<16>(c, t, ops, key, acc, L, batch, grp, g, splits, mode, tg, s);
                break;
            }
            if (rc)
            {
                return rc;
            }
            o0 += g;
        }
    }
    else
    {
        // small transforms (N <= 2048): separate reduce -> NTT -> MAC launches
        for (size_t Iidx = 0; Iidx <= L; ++Iidx)
        {
            const uint32_t prime = (uint32_t)(Iidx == L ? k - 1 : Iidx);
            MOAI_CHECK_GRID_ROWS(batch * L);
            hipLaunchKernelGGL(reduce_rows_kernel, rgrid(c, batch * L), dim3(256), 0, s, t, ops, c->pc, prime, n2);
            MOAI_LAUNCH_CHECK();
            for (size_t r = 0; r < L; ++r)
            {
                rm.idx[r] = (uint32_t)prime;
            }
            rc = ntt_launch(c, ops, batch, L, rm, false, s);
            if (rc)
            {
                return rc;
            }
            MacArgs m;
            m.ops = ops;
            m.key = key;
            m.acc = acc;
            m.pc = c->pc;
            m.L = (uint32_t)L;
            m.k = key_rows;
            m.krow = Iidx == L ? key_rows - 1 : prime;
            m.prime = prime;
            m.slot = (uint32_t)Iidx;
            m.n2 = n2;
            MOAI_CHECK_GRID_ROWS(batch);
            hipLaunchKernelGGL(keyswitch_mac_kernel, rgrid(c, batch), dim3(256), 0, s, m);
            MOAI_LAUNCH_CHECK();
        }
    }
    // 3. mod-down by the special prime, accumulated into ct   (evaluator.cpp:2913-3018)
    MOAI_CHECK_GRID_ROWS(batch * 2);
    hipLaunchKernelGGL(sum_rows_kernel, rgrid(c, batch * 2), dim3(256), 0, s, acc, last, (uint32_t)(L + 1), (uint32_t)L, splits,
                       split_stride, c->pc, (uint32_t)(k - 1), n2);
    MOAI_LAUNCH_CHECK();
    return moddown(c, last, acc, (uint32_t)(L + 1), ops, ct, batch * 2, L, (uint32_t)(k - 1), addend, (uint32_t)addend_bstride, add_mode, s, splits,
                   split_stride, nullptr, addend2);
}

} // namespace moai

using namespace moai;

namespace moai {
// rows[r][:] = rows[r][:] * s mod q, canonical   (the dropped row of a fused scalar product + rescale)
__global__ __launch_bounds__(256) void scale_rows_kernel(uint64_t *rows, Tw s, uint64_t q, uint32_t n2)
{
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(rows) + (size_t)blockIdx.y * n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n2; j += gridDim.x * 256u)
    {
        ulonglong2 x = d[j];
        x.x = csub(mul_shoup_lazy(x.x, s.w, s.wq, q), q);
        x.y = csub(mul_shoup_lazy(x.y, s.w, s.wq, q), q);
        d[j] = x;
    }
}
} // namespace moai

// rescale_to_next (scalars == nullptr) or a scalar plaintext product followed by it, plus an optional addend of the output's
// shape ([batch * size][L - 1][N]; may be `out` itself) that the last kernel adds to the quotient
static int rescale_common(moai_ctx *c, const uint64_t *in, const uint64_t *scalars, const uint64_t *addend, uint64_t *out, size_t size,
                          size_t L, size_t batch, void *stream)
{
    const size_t P = batch * size;
    int rc = check_level(c, L, P);
    if (rc)
    {
        return rc;
    }
    if (L < 2)
    {
        // SEAL/evaluator.cpp:1693-1696
        return set_error(MOAI_EINVAL, "end of modulus switching chain reached");
    }
    if (P == 0)
    {
        return MOAI_OK;
    }
    if (!in || !out || in == out || in == addend)
    {
        return set_error(MOAI_EINVAL, "bad in/out pointers");
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = c->n * sizeof(uint64_t);
    if (scalars && c->logn < 12)
    {
        // small transforms take the two separate steps
        void *tmp = nullptr;
        rc = moai_malloc(&tmp, P * L * row_bytes);
        if (rc)
        {
            return rc;
        }
        rc = moai_mul_scalar_rows(c, in, scalars, static_cast<uint64_t *>(tmp), P, L, stream);
        if (!rc)
        {
            rc = rescale_common(c, static_cast<const uint64_t *>(tmp), nullptr, addend, out, size, L, batch, stream);
        }
        hipError_t e = hipStreamSynchronize(s);
        moai_free(tmp);
        return rc ? rc : (e == hipSuccess ? MOAI_OK : set_error(MOAI_EHIP, "%s", hipGetErrorString(e)));
    }
    Tw sc[MOAI_MAX_RNS];
    if (scalars)
    {
        for (size_t r = 0; r < L; r++)
        {
            const uint64_t q = c->primes[r];
            const uint64_t v = scalars[r] % q; // barrett_reduce_64, polyarithsmallmod.h:209-217
            sc[r].w = v;
            sc[r].wq = (uint64_t)((((unsigned __int128)v) << 64) / q);
        }
    }
    const size_t sz_last = align256(P * row_bytes);
    const size_t sz_u = align256(P * (L - 1) * row_bytes);
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, sz_last + sz_u, s, &wsp);
    if (rc)
    {
        return rc;
    }
    uint64_t *last = static_cast<uint64_t *>(wsp);
    uint64_t *u = reinterpret_cast<uint64_t *>(static_cast<char *>(wsp) + sz_last);
    MOAI_CHECK_GRID_ROWS(P);
    hipLaunchKernelGGL(copy_rows_kernel, rgrid(c, P), dim3(256), 0, s, in, last, 1u, (uint32_t)L, (uint32_t)(L - 1), NO_ZERO,
                       (uint32_t)(c->n >> 1));
    if (scalars)
    {
        hipLaunchKernelGGL(scale_rows_kernel, rgrid(c, P), dim3(256), 0, s, last, sc[L - 1], c->primes[L - 1], (uint32_t)(c->n >> 1));
    }
    MOAI_LAUNCH_CHECK();
    // the addend has the output's layout: polynomial p starts (L - 1) rows after polynomial p - 1 (ModDownArgs: pairs of
    // polynomials 2 (L - 1) rows apart)
    return moddown(c, last, in, (uint32_t)L, u, out, P, L - 1, (uint32_t)(L - 1), addend, (uint32_t)(2 * (L - 1)), addend ? 1 : 0, s, 1, 0,
                   scalars ? sc : nullptr);
}

extern "C" int moai_rescale(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t size, size_t L, size_t batch,
                            void *stream)
{
    MOAI_AUDIT(stream, in, out);
    trace_op("rescale", L, batch * size);
    return rescale_common(c, in, nullptr, nullptr, out, size, L, batch, stream);
}

extern "C" int moai_rescale_add(moai_ctx *c, const uint64_t *in, const uint64_t *addend, uint64_t *out, size_t size, size_t L,
                                size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, addend, out);
    trace_op("rescale_add", L, batch * size);
    if (!addend)
    {
        return set_error(MOAI_EINVAL, "null addend");
    }
    return rescale_common(c, in, nullptr, addend, out, size, L, batch, stream);
}

extern "C" int moai_mul_scalar_rescale(moai_ctx *c, const uint64_t *in, const uint64_t *scalars, uint64_t *out, size_t size,
                                       size_t L, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, scalars, out);
    trace_op("mul_scalar_rescale", L, batch * size);
    if (!scalars)
    {
        return set_error(MOAI_EINVAL, "bad pointers");
    }
    return rescale_common(c, in, scalars, nullptr, out, size, L, batch, stream);
}

extern "C" int moai_mul_scalar_rescale_add(moai_ctx *c, const uint64_t *in, const uint64_t *scalars, const uint64_t *addend,
                                           uint64_t *out, size_t size, size_t L, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, scalars, addend, out);
    trace_op("mul_scalar_rescale_add", L, batch * size);
    if (!scalars || !addend)
    {
        return set_error(MOAI_EINVAL, "bad pointers");
    }
    return rescale_common(c, in, scalars, addend, out, size, L, batch, stream);
}

extern "C" int moai_switch_key(moai_ctx *c, uint64_t *ct, const uint64_t *target, const uint64_t *key, size_t L,
                               size_t batch, void *stream)
{
    MOAI_AUDIT(stream, ct, target, key);
    trace_op("switch_key", L, batch);
    int rc = check_level(c, L, batch * 2);
    if (rc)
    {
        return rc;
    }
    if (batch == 0)
    {
        return MOAI_OK;
    }
    if (!ct || !target || !key)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, switch_key_ws_bytes(c, L, batch), (hipStream_t)stream, &wsp);
    if (rc)
    {
        return rc;
    }
    return switch_key_impl(c, ct, target, L, 0, key, L, batch, wsp, (hipStream_t)stream, ct, 2 * L, 1);
}

extern "C" int moai_relinearize(moai_ctx *c, const uint64_t *ct3, const uint64_t *relin_key, uint64_t *out, size_t L,
                                size_t batch, void *stream)
{
    MOAI_AUDIT(stream, ct3, relin_key, out);
    trace_op("relinearize", L, batch);
    int rc = check_level(c, L, batch * 3);
    if (rc)
    {
        return rc;
    }
    if (batch == 0)
    {
        return MOAI_OK;
    }
    if (!ct3 || !relin_key || !out || out == ct3)
    {
        return set_error(MOAI_EINVAL, "bad pointers");
    }
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, switch_key_ws_bytes(c, L, batch), s, &wsp);
    if (rc)
    {
        return rc;
    }
    // out = (c0, c1) + switch_key(c2, relin_keys[0])      (evaluator.cpp:1383-1392): c0 and c1 are read from ct3 by
    // the last kernel of the key switch, no copy first
    return switch_key_impl(c, out, ct3, 3 * L, 2 * L, relin_key, L, batch, wsp, s, ct3, 3 * L, 1);
}

extern "C" int moai_apply_galois_to(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t L, uint32_t galois_elt,
                                    const uint64_t *galois_key, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, out, galois_key);
    trace_op("apply_galois_to", L, batch);
    int rc = check_level(c, L, batch * 2);
    if (rc)
    {
        return rc;
    }
    if (batch == 0)
    {
        return MOAI_OK;
    }
    if (!in || !out || !galois_key)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = c->n * sizeof(uint64_t);
    const size_t sz_tmp = align256(batch * 2 * L * row_bytes);
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, sz_tmp + switch_key_ws_bytes(c, L, batch), s, &wsp);
    if (rc)
    {
        return rc;
    }
    uint64_t *tmp = static_cast<uint64_t *>(wsp);
    void *ks_ws = static_cast<char *>(wsp) + sz_tmp;
    // tmp = galois(in) for both polynomials; out = (tmp0, 0) + switch_key(tmp1)  (evaluator.cpp:2631-2654): tmp0 is
    // added by the last kernel of the key switch, so `out` is written once and may be `in`
    rc = moai_galois_permute(c, in, tmp, batch * 2, L, galois_elt, stream);
    if (rc)
    {
        return rc;
    }
    return switch_key_impl(c, out, tmp, 2 * L, L, galois_key, L, batch, ks_ws, s, tmp, 2 * L, 2);
}

extern "C" int moai_apply_galois(moai_ctx *c, uint64_t *ct, size_t L, uint32_t galois_elt, const uint64_t *galois_key,
                                 size_t batch, void *stream)
{
    MOAI_AUDIT(stream, ct, galois_key);
    return moai_apply_galois_to(c, ct, ct, L, galois_elt, galois_key, batch, stream);
}

extern "C" int moai_apply_galois_acc(moai_ctx *c, const uint64_t *in, uint64_t *acc, size_t L, uint32_t galois_elt,
                                     const uint64_t *galois_key, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, acc, galois_key);
    trace_op("apply_galois_to", L, batch); // the same key switch; the addition that follows it is what is saved
    int rc = check_level(c, L, batch * 2);
    if (rc)
    {
        return rc;
    }
    if (batch == 0)
    {
        return MOAI_OK;
    }
    if (!in || !acc || !galois_key || in == acc)
    {
        return set_error(MOAI_EINVAL, "null argument, or the sum is the input");
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = c->n * sizeof(uint64_t);
    const size_t sz_tmp = align256(batch * 2 * L * row_bytes);
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, sz_tmp + switch_key_ws_bytes(c, L, batch), s, &wsp);
    if (rc)
    {
        return rc;
    }
    uint64_t *tmp = static_cast<uint64_t *>(wsp);
    void *ks_ws = static_cast<char *>(wsp) + sz_tmp;
    rc = moai_galois_permute(c, in, tmp, batch * 2, L, galois_elt, stream);
    if (rc)
    {
        return rc;
    }
    // acc = acc + (tmp0, 0) + switch_key(tmp1): both summands are added by the key switch's last kernel
    return switch_key_impl(c, acc, tmp, 2 * L, L, galois_key, L, batch, ks_ws, s, tmp, 2 * L, 2, acc);
}

extern "C" int moai_modraise(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t L_out, size_t batch, void *stream)
{
    MOAI_AUDIT(stream, in, out);
    trace_op("modraise", L_out, batch);
    const size_t P = batch * 2;
    int rc = check_level(c, L_out, P);
    if (rc)
    {
        return rc;
    }
    if (batch == 0)
    {
        return MOAI_OK;
    }
    if (!in || !out || in == out)
    {
        return set_error(MOAI_EINVAL, "bad pointers");
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = c->n * sizeof(uint64_t);
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, align256(P * row_bytes), s, &wsp);
    if (rc)
    {
        return rc;
    }
    uint64_t *src = static_cast<uint64_t *>(wsp);
    MOAI_HIP_CHECK(hipMemcpyAsync(src, in, P * row_bytes, hipMemcpyDeviceToDevice, s));
    RowMap rm;
    rc = make_rowmap(c, 1, nullptr, &rm);
    if (rc)
    {
        return rc;
    }
    rc = ntt_launch(c, src, P, 1, rm, true, s);
    if (rc)
    {
        return rc;
    }
    RaiseArgs g;
    g.src = src;
    g.out = out;
    g.pc = c->pc;
    g.Lout = (uint32_t)L_out;
    g.n2 = (uint32_t)(c->n >> 1);
    MOAI_CHECK_GRID_ROWS(P * L_out);
    hipLaunchKernelGGL(modraise_kernel, rgrid(c, P * L_out), dim3(256), 0, s, g);
    MOAI_LAUNCH_CHECK();
    rc = make_rowmap(c, L_out, nullptr, &rm);
    if (rc)
    {
        return rc;
    }
    return ntt_launch(c, out, P, L_out, rm, false, s);
}

// ---- hoisted rotations (keyswitch_kernels.hip.h explains the identity) --------------------------------------------
namespace moai {

template <int LOGN>
static void hoist_contig_finish(moai_ctx *c, uint64_t *tmp, size_t L, size_t batch, const KsGroup &grp, size_t G, int mode, hipStream_t s)
{
    // the strided pass left [B][G][L][N] in the mode's lazy form: the plain contiguous pass finishes every row of group
    // member g under that member's prime and leaves canonical residues
    constexpr uint32_t tpr = 1u << (LOGN - 12);
    for (size_t g = 0; g < G; ++g)
    {
        NttArgs a;
        memset(&a, 0, sizeof(a));
        a.data = tmp;
        a.tw = mode >= M_FPN ? c->fwd_twf : c->fwd_tw;
        a.twb = mode >= M_FPN ? c->fwd_twfb : c->fwd_twb;
        a.pc = c->pc;
        a.L = (uint32_t)(G * L);
        a.n_poly = (uint32_t)batch;
        a.lds_twiddles = tuning("MOAI_NTT_LDSTW", 1) ? 1u : 0u;
        a.Lsel = 0;
        for (size_t J = 0; J < L; ++J)
        {
            if (grp.slot[g] == J)
            {
                continue; // the strided pass skipped it: the digit under its own prime is read from the ciphertext
            }
            a.sel.idx[a.Lsel] = (uint32_t)(g * L + J);
            a.selp.idx[a.Lsel++] = grp.prime[g];
        }
        a.total_work = a.n_poly * a.Lsel * tpr;
        switch (mode)
        {
        case M_FPN: hipLaunchKernelGGL((ntt_fwd_contig<LOGN, M_FPN>), dim3(a.total_work), dim3(256), 0, s, a); break;
        case M_FPR: hipLaunchKernelGGL((ntt_fwd_contig<LOGN, M_FPR>), dim3(a.total_work), dim3(256), 0, s, a); break;
        case M_NOGUARD: hipLaunchKernelGGL((ntt_fwd_contig<LOGN, M_NOGUARD>), dim3(a.total_work), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((ntt_fwd_contig<LOGN, M_GUARD>), dim3(a.total_work), dim3(256), 0, s, a); break;
        }
    }
}

template <int LOGN>
static int hoist_group(moai_ctx *c, const uint64_t *t, uint64_t *tmp, const uint64_t *in, size_t L, size_t batch, const KsGroup &grp,
                       size_t G, int mode, const uint32_t *const *tables, const uint32_t *const *itables, const uint64_t *const *keys,
                       const uint32_t *key_rows, const uint64_t *const *corrs, uint64_t *acc, size_t acc_stride_words, size_t R, hipStream_t s)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    KsP1Args p1;
    p1.t = t;
    p1.tmp = tmp;
    p1.tw = mode >= M_FPN ? c->fwd_twf : c->fwd_tw;
    p1.tw1 = c->fwd_twf1;
    p1.pc = c->pc;
    p1.grp = grp;
    p1.L = (uint32_t)L;
    p1.G = (uint32_t)G;
    p1.total_work = (uint32_t)(batch * G * L * TPR);
    switch (mode)
    {
    case M_FPN: hipLaunchKernelGGL((ks_fwd_strided<LOGN, M_FPN>), dim3(p1.total_work), dim3(256), 0, s, p1); break;
    case M_FPR: hipLaunchKernelGGL((ks_fwd_strided<LOGN, M_FPR>), dim3(p1.total_work), dim3(256), 0, s, p1); break;
    case M_NOGUARD: hipLaunchKernelGGL((ks_fwd_strided<LOGN, M_NOGUARD>), dim3(p1.total_work), dim3(256), 0, s, p1); break;
    default: hipLaunchKernelGGL((ks_fwd_strided<LOGN, M_GUARD>), dim3(p1.total_work), dim3(256), 0, s, p1); break;
    }
    MOAI_LAUNCH_CHECK();
    hoist_contig_finish<LOGN>(c, tmp, L, batch, grp, G, mode, s);
    MOAI_LAUNCH_CHECK();
    // FP64 modes: four or two rotations per pass over the digits (ks_hoisted_mac2); MOAI_KS_HOIST_PAIR=0 keeps one per pass,
    // 2 at most two
    const long pair = (mode == M_FPN || mode == M_FPR) ? tuning("MOAI_KS_HOIST_PAIR", 4) : 0;
    for (size_t r = 0; r < R; ++r)
    {
        const size_t nr = pair >= 4 && r + 3 < R ? 4 : (pair >= 1 && r + 1 < R ? 2 : 1);
        if (nr > 1)
        {
            HoistMac2Args m2;
            m2.dig = tmp;
            m2.ct = in;
            for (size_t h = 0; h < 4; ++h)
            {
                const size_t rr = r + (h < nr ? h : 0);
                m2.itable[h] = itables[rr];
                m2.key[h] = keys[rr];
                m2.krows[h] = key_rows[rr];
                m2.corr[h] = corrs[rr];
                m2.acc[h] = acc + rr * acc_stride_words;
            }
            m2.pc = c->pc;
            m2.grp = grp;
            m2.L = (uint32_t)L;
            m2.G = (uint32_t)G;
            m2.k = (uint32_t)c->k;
            m2.B = (uint32_t)batch;
            m2.total_work = (uint32_t)(batch * G * TPR * 2);
            if (mode == M_FPN && nr == 4)
            {
                hipLaunchKernelGGL((ks_hoisted_mac2<LOGN, false, 4>), dim3(m2.total_work), dim3(256), 0, s, m2);
            }
            else if (mode == M_FPN)
            {
                hipLaunchKernelGGL((ks_hoisted_mac2<LOGN, false, 2>), dim3(m2.total_work), dim3(256), 0, s, m2);
            }
            else if (nr == 4)
            {
                hipLaunchKernelGGL((ks_hoisted_mac2<LOGN, true, 4>), dim3(m2.total_work), dim3(256), 0, s, m2);
            }
            else
            {
                hipLaunchKernelGGL((ks_hoisted_mac2<LOGN, true, 2>), dim3(m2.total_work), dim3(256), 0, s, m2);
            }
            r += nr - 1;
            continue;
        }
        HoistMacArgs m;
        m.dig = tmp;
        m.ct = in;
        m.table = tables[r];
        m.key = keys[r];
        m.corr = corrs[r];
        m.acc = acc + r * acc_stride_words;
        m.pc = c->pc;
        m.grp = grp;
        m.L = (uint32_t)L;
        m.G = (uint32_t)G;
        m.k = key_rows[r];
        m.B = (uint32_t)batch;
        m.total_work = (uint32_t)(batch * G * TPR * 2);
        if (mode == M_FPN)
        {
            hipLaunchKernelGGL((ks_hoisted_mac<LOGN, true, false>), dim3(m.total_work), dim3(256), 0, s, m);
        }
        else if (mode == M_FPR)
        {
            hipLaunchKernelGGL((ks_hoisted_mac<LOGN, true, true>), dim3(m.total_work), dim3(256), 0, s, m);
        }
        else
        {
            hipLaunchKernelGGL((ks_hoisted_mac<LOGN, false, false>), dim3(m.total_work), dim3(256), 0, s, m);
        }
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

} // namespace moai

extern "C" int moai_hoist_correction(moai_ctx *c, const uint64_t *galois_key, uint32_t galois_elt, size_t L, uint64_t *correction,
                                     void *stream)
{
    MOAI_AUDIT(stream, galois_key, correction);
    trace_op("hoist_correction", L, 1);
    int rc = check_level(c, L, 2);
    if (rc)
    {
        return rc;
    }
    if (!galois_key || !correction)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    if (c->k < 2 || L > c->k - 1)
    {
        return set_error(MOAI_EINVAL, "L exceeds the key's decomposition size");
    }
    if (!(galois_elt & 1u) || galois_elt >= 2 * c->n)
    {
        return set_error(MOAI_EINVAL, "Galois element is not valid");
    }
    uint32_t key_rows = 0;
    rc = key_rows_for(c, galois_key, L, &key_rows);
    if (rc)
    {
        return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = c->n * sizeof(uint64_t);
    std::lock_guard<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
    void *wsp;
    rc = workspace(c, align256((L + 1) * row_bytes), s, &wsp);
    if (rc)
    {
        return rc;
    }
    uint64_t *mask = static_cast<uint64_t *>(wsp);
    hipLaunchKernelGGL(galois_sign_mask_kernel, dim3((uint32_t)((c->n + 255) / 256)), dim3(256), 0, s, mask, (uint32_t)c->logn, galois_elt,
                       (uint32_t)(L + 1));
    MOAI_LAUNCH_CHECK();
    std::vector<uint32_t> pidx(L + 1);
    for (size_t i = 0; i < L; ++i)
    {
        pidx[i] = (uint32_t)i;
    }
    pidx[L] = (uint32_t)(c->k - 1);
    RowMap rm;
    rc = make_rowmap(c, L + 1, pidx.data(), &rm);
    if (rc)
    {
        return rc;
    }
    rc = ntt_launch(c, mask, 1, L + 1, rm, false, s);
    if (rc)
    {
        return rc;
    }
    HoistCorrArgs g;
    g.key = galois_key;
    g.mask = mask;
    g.out = correction;
    g.pc = c->pc;
    g.L = (uint32_t)L;
    g.k = (uint32_t)c->k;
    g.krows = key_rows;
    g.n2 = (uint32_t)(c->n >> 1);
    MOAI_CHECK_GRID_ROWS(2 * (L + 1));
    hipLaunchKernelGGL(ks_hoist_correction_kernel, rgrid(c, 2 * (L + 1)), dim3(256), 0, s, g);
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

extern "C" int moai_apply_galois_hoisted(moai_ctx *c, const uint64_t *in, uint64_t *const *outs, size_t L, const uint32_t *galois_elts,
                                         const uint64_t *const *galois_keys, const uint64_t *const *corrections, size_t R, size_t batch,
                                         int *used_fallback, void *stream)
{
    MOAI_AUDIT(stream, in);
    for (size_t r = 0; outs && galois_keys && corrections && r < R; ++r)
    {
        MOAI_AUDIT(stream, outs[r], galois_keys[r], corrections[r]);
    }
    if (R > 64 && outs && galois_elts && galois_keys && corrections)
    {
        // the accumulators of one pass are sized for 64 rotations: more are done 64 at a time, each pass with its own digit
        // decomposition (the results do not depend on the grouping)
        int any_fallback = 0;
        for (size_t r0 = 0; r0 < R; r0 += 64)
        {
            int fb = 0;
            const int rc = moai_apply_galois_hoisted(c, in, outs + r0, L, galois_elts + r0, galois_keys + r0, corrections + r0,
                                                     R - r0 < 64 ? R - r0 : 64, batch, &fb, stream);
            if (rc)
            {
                return rc;
            }
            any_fallback |= fb;
        }
        if (used_fallback)
        {
            *used_fallback = any_fallback;
        }
        return MOAI_OK;
    }
    trace_op("apply_galois_hoisted", L, batch * R);
    if (used_fallback)
    {
        *used_fallback = 0;
    }
    int rc = check_level(c, L, batch * 2);
    if (rc)
    {
        return rc;
    }
    if (batch == 0 || R == 0)
    {
        return MOAI_OK;
    }
    if (!in || !outs || !galois_elts || !galois_keys || !corrections)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    const size_t n = c->n, k = c->k;
    if (k < 2)
    {
        return set_error(MOAI_ELOGIC, "keyswitching is not supported by the context");
    }
    if (L > k - 1)
    {
        return set_error(MOAI_EINVAL, "L exceeds the key's decomposition size");
    }
    hipStream_t s = (hipStream_t)stream;
    std::vector<const uint32_t *> tables(R), itables(R);
    for (size_t r = 0; r < R; ++r)
    {
        if (!galois_keys[r] || !corrections[r] || !outs[r] || outs[r] == in)
        {
            return set_error(MOAI_EINVAL, "null key, correction or output (an output must not be the input)");
        }
        rc = galois_table(c, galois_elts[r], s, &tables[r]);
        if (rc)
        {
            return rc;
        }
        // the inverse permutation is the table of the inverse element (mod 2N, by Newton's iteration on an odd number)
        const uint32_t two_n_mask = (uint32_t)(2 * n - 1);
        uint32_t inv = galois_elts[r];
        for (int it = 0; it < 5; ++it)
        {
            inv = (inv * (2u - galois_elts[r] * inv)) & two_n_mask;
        }
        rc = galois_table(c, inv, s, &itables[r]);
        if (rc)
        {
            return rc;
        }
    }
    std::vector<uint32_t> key_rows(R);
    for (size_t r = 0; r < R; ++r)
    {
        rc = key_rows_for(c, galois_keys[r], L, &key_rows[r]);
        if (rc)
        {
            return rc;
        }
    }
    bool fallback = c->logn < 12;
    if (!fallback)
    {
        const size_t row_bytes = n * sizeof(uint64_t);
        const size_t G = ks_group_size(c, L, batch);
        const size_t sz_t = align256(batch * L * row_bytes);
        const size_t sz_tmp = align256(batch * G * L * row_bytes);
        const size_t acc_stride_words = batch * 2 * (L + 1) * n;
        const size_t sz_acc = align256(R * acc_stride_words * sizeof(uint64_t));
        const size_t sz_last = align256(batch * 2 * row_bytes);
        const size_t sz_u = align256(2 * batch * L * row_bytes);
        const size_t sz_c0 = align256(batch * L * row_bytes);
        std::unique_lock<std::mutex> op_lock(*static_cast<std::mutex *>(c->op_mutex));
        void *wsp;
        rc = workspace(c, 256 + sz_t + sz_tmp + sz_acc + sz_last + sz_u + sz_c0, s, &wsp);
        if (rc)
        {
            return rc;
        }
        char *base = static_cast<char *>(wsp);
        uint32_t *flag = reinterpret_cast<uint32_t *>(base);
        uint64_t *t = reinterpret_cast<uint64_t *>(base + 256);
        uint64_t *tmp = reinterpret_cast<uint64_t *>(base + 256 + sz_t);
        uint64_t *acc = reinterpret_cast<uint64_t *>(base + 256 + sz_t + sz_tmp);
        uint64_t *last = reinterpret_cast<uint64_t *>(base + 256 + sz_t + sz_tmp + sz_acc);
        uint64_t *u = reinterpret_cast<uint64_t *>(base + 256 + sz_t + sz_tmp + sz_acc + sz_last);
        uint64_t *pc0 = reinterpret_cast<uint64_t *>(base + 256 + sz_t + sz_tmp + sz_acc + sz_last + sz_u);
        const uint32_t n2 = (uint32_t)(n >> 1);
        // t = INTT(c1), UNPERMUTED: once for all rotations
        RowMap rm;
        rc = make_rowmap(c, L, nullptr, &rm);
        if (rc)
        {
            return rc;
        }
        rc = ntt_launch(c, t, batch, L, rm, true, s, in, 2 * L, L);
        if (rc)
        {
            return rc;
        }
        MOAI_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(any_zero_kernel, dim3(1024), dim3(256), 0, s, t, batch * L * n / 2, flag);
        MOAI_LAUNCH_CHECK();
        uint32_t host_flag = 0;
        MOAI_HIP_CHECK(hipMemcpyAsync(&host_flag, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MOAI_HIP_CHECK(hipStreamSynchronize(s));
        if (host_flag)
        {
            fallback = true; // a zero coefficient: the identity above does not hold for it
        }
        else
        {
            const bool allow_fp = (long)(batch * L) >= tuning("MOAI_KS_FP_MIN_ROWS", 16);
            std::vector<uint32_t> order;
            std::vector<int> order_mode;
            for (int mode = M_FPR; mode >= M_GUARD; --mode)
            {
                for (size_t Iidx = 0; Iidx <= L; ++Iidx)
                {
                    const uint32_t prime = (uint32_t)(Iidx == L ? k - 1 : Iidx);
                    if (ks_mode(c, prime, L, allow_fp) == mode)
                    {
                        order.push_back((uint32_t)Iidx);
                        order_mode.push_back(mode);
                    }
                }
            }
            for (size_t o0 = 0; o0 < order.size();)
            {
                const int mode = order_mode[o0];
                size_t g = 0;
                while (o0 + g < order.size() && g < G && order_mode[o0 + g] == mode)
                {
                    ++g;
                }
                KsGroup grp;
                for (size_t i = 0; i < MOAI_MAX_RNS; ++i)
                {
                    size_t Iidx = order[o0 + (i < g ? i : 0)];
                    grp.prime[i] = (uint32_t)(Iidx == L ? k - 1 : Iidx);
                    grp.slot[i] = (uint32_t)Iidx;
                }
                switch (c->logn)
                {
                case 12: rc = hoist_group<12>(c, t, tmp, in, L, batch, grp, g, mode, tables.data(), itables.data(), galois_keys, key_rows.data(), corrections, acc, acc_stride_words, R, s); break;
                case 13: rc = hoist_group<13>(c, t, tmp, in, L, batch, grp, g, mode, tables.data(), itables.data(), galois_keys, key_rows.data(), corrections, acc, acc_stride_words, R, s); break;
                case 14: rc = hoist_group<14>(c, t, tmp, in, L, batch, grp, g, mode, tables.data(), itables.data(), galois_keys, key_rows.data(), corrections, acc, acc_stride_words, R, s); break;
                case 15: rc = hoist_group<15>(c, t, tmp, in, L, batch, grp, g, mode, tables.data(), itables.data(), galois_keys, key_rows.data(), corrections, acc, acc_stride_words, R, s); break;
                default: rc = hoist_group<16>(c, t, tmp, in, L, batch, grp, g, mode, tables.data(), itables.data(), galois_keys, key_rows.data(), corrections, acc, acc_stride_words, R, s); break;
                }
                if (rc)
                {
                    return rc;
                }
                o0 += g;
            }
            // per rotation: the permuted c0 is the addend, then the shared mod-down tail (evaluator.cpp:2913-3018)
            for (size_t r = 0; r < R; ++r)
            {
                uint64_t *acc_r = acc + r * acc_stride_words;
                // pc0 [B][L][N] = perm(c0 of in): the addend of the even polynomials (add_mode 2, ciphertexts L rows apart)
                rc = galois_permute_c0(c, in, pc0, batch, L, galois_elts[r], s);
                if (rc)
                {
                    return rc;
                }
                MOAI_CHECK_GRID_ROWS(batch * 2);
                hipLaunchKernelGGL(sum_rows_kernel, rgrid(c, batch * 2), dim3(256), 0, s, acc_r, last, (uint32_t)(L + 1), (uint32_t)L, 1u,
                                   (size_t)0, c->pc, (uint32_t)(k - 1), n2);
                MOAI_LAUNCH_CHECK();
                rc = moddown(c, last, acc_r, (uint32_t)(L + 1), u, outs[r], batch * 2, L, (uint32_t)(k - 1), pc0, (uint32_t)L, 2, s);
                if (rc)
                {
                    return rc;
                }
            }
            return MOAI_OK;
        }
    }
    // fallback: the reference's sequence per rotation
    if (used_fallback)
    {
        *used_fallback = 1;
    }
    for (size_t r = 0; r < R; ++r)
    {
        rc = moai_apply_galois_to(c, in, outs[r], L, galois_elts[r], galois_keys[r], batch, stream);
        if (rc)
        {
            return rc;
        }
    }
    return MOAI_OK;
}

// ---- level-trimmed key residency ------------------------------------------------------------------------------------------------
extern "C" size_t moai_key_words(const moai_ctx *c, size_t levels)
{
    if (!c || c->k < 2)
    {
        return 0;
    }
    const size_t lv = levels > c->k - 1 ? c->k - 1 : levels;
    return lv * 2 * (lv + 1) * c->n;
}

extern "C" int moai_key_trim(moai_ctx *c, const uint64_t *full_key, size_t levels, uint64_t *trimmed, void *stream)
{
    MOAI_AUDIT(stream, full_key, trimmed);
    if (!c || !full_key || !trimmed)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    int rc = enter_device(c);
    if (rc)
    {
        return rc;
    }
    const size_t k = c->k, n = c->n;
    if (k < 2 || levels < 1 || levels > k - 1)
    {
        return set_error(MOAI_EINVAL, "levels must lie in 1 .. %zu", k < 2 ? (size_t)0 : k - 1);
    }
    {
        std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
        if (c->key_layouts.count(full_key))
        {
            return set_error(MOAI_EINVAL, "the source of moai_key_trim must be a key in the reference's layout");
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = levels + 1;
    for (size_t J = 0; J < levels; ++J)
    {
        for (size_t K = 0; K < 2; ++K)
        {
            const uint64_t *src = full_key + (J * 2 + K) * k * n;
            uint64_t *dst = trimmed + (J * 2 + K) * rows * n;
            MOAI_HIP_CHECK(hipMemcpyAsync(dst, src, levels * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
            MOAI_HIP_CHECK(hipMemcpyAsync(dst + levels * n, src + (k - 1) * n, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
        }
    }
    std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
    c->key_layouts[trimmed] = moai_ctx::KeyLayout{ (uint32_t)levels, (uint32_t)rows };
    return MOAI_OK;
}

extern "C" int moai_key_forget(moai_ctx *c, const uint64_t *key)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
    c->key_layouts.erase(key);
    return MOAI_OK;
}
