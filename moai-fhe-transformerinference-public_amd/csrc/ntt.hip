// ntt.hip -- launchers of the negacyclic NTT kernels (C ABI: moai_ntt_forward / moai_ntt_inverse).
#include <cstdlib>

#include "ntt_kernels.hip.h"
#include "launch.h"

namespace moai {

int ntt_mode(const moai_ctx *c, uint32_t prime);

static bool naive_requested()
{
    static int v = -1;
    if (v < 0)
    {
        const char *e = getenv("MOAI_NTT_NAIVE");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

static long env_long(const char *name, long dflt)
{
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
}

template <int LOGN, int MODE>
static void launch_fwd_mode(const moai_ctx *c, NttArgs a, hipStream_t s)
{
    constexpr uint32_t tpr = 1u << (LOGN - 12);
    a.total_work = a.n_poly * a.Lsel * tpr;
    if (MODE >= M_FPN)
    {
        a.tw = c->fwd_twf;
        a.twb = c->fwd_twfb;
    }
    // (the software-pipelined strided pass of the key switch, fwd_strided_tiles, does not pay here: measured in round 3 on 256 x 2
    // polynomials, 11.25 against 11.37 ms with the bench's 60-bit primes and 7.77 against 7.40 ms -- slower -- with MOAI's FP64
    // rows.  An in-place transform reads as much as it writes and has no cached operand; five resident workgroups of one tile
    // each keep more of that traffic in flight than four pipelined ones.)
    hipLaunchKernelGGL((ntt_fwd_strided<LOGN, MODE>), dim3(a.total_work), dim3(256), 0, s, a);
    hipLaunchKernelGGL((ntt_fwd_contig<LOGN, MODE>), dim3(a.total_work), dim3(256), 0, s, a);
}

// forward transform: the rows are split by the arithmetic their prime allows (ntt_mode) and every class
// gets its own pair of launches
template <int LOGN>
static void launch_fwd(const moai_ctx *c, const NttArgs &base, hipStream_t s)
{
    for (int mode = M_GUARD; mode <= M_FPR; ++mode)
    {
        NttArgs a = base;
        a.Lsel = 0;
        for (uint32_t r = 0; r < a.L; ++r)
        {
            if (ntt_mode(c, a.rows.idx[r]) == mode)
            {
                a.selp.idx[a.Lsel] = a.rows.idx[r];
                a.sel.idx[a.Lsel++] = (uint32_t)r;
            }
        }
        if (a.Lsel == 0)
        {
            continue;
        }
        switch (mode)
        {
        case M_GUARD:
        {
            // integer primes: below 2^60 the butterfly with the approximate Shoup quotient (M_LAZY8), else the exact one with a
            // guard every second stage (M_GUARD2); same residues either way (modarith.hip.h).  One launch pair per class.
            static const long lazy8 = env_long("MOAI_NTT_LAZY8", 1);
            NttArgs lo = a, hi = a;
            lo.Lsel = hi.Lsel = 0;
            for (uint32_t i = 0; i < a.Lsel; ++i)
            {
                NttArgs &dst = (lazy8 && c->primes[a.selp.idx[i]] < (1ull << 60)) ? lo : hi;
                dst.selp.idx[dst.Lsel] = a.selp.idx[i];
                dst.sel.idx[dst.Lsel++] = a.sel.idx[i];
            }
            if (lo.Lsel)
            {
                launch_fwd_mode<LOGN, M_LAZY8>(c, lo, s);
            }
            if (hi.Lsel)
            {
                launch_fwd_mode<LOGN, M_GUARD2>(c, hi, s);
            }
            break;
        }
        case M_NOGUARD: launch_fwd_mode<LOGN, M_NOGUARD>(c, a, s); break;
        case M_FPN: launch_fwd_mode<LOGN, M_FPN>(c, a, s); break;
        default: launch_fwd_mode<LOGN, M_FPR>(c, a, s); break;
        }
    }
}

// inverse transform: rows of primes below 2^51 run in exact FP64 arithmetic like the forward transform (FPN / FPR, modarith.hip.h
// gs_bfly_fp; MOAI_NTT_FP=0 keeps them on the integer units), rows of primes below 2^60 take the integer butterflies with the
// approximate Shoup quotient (M_LAZY8), the others the exact ones; one pair of launches per class, same residues
template <int LOGN, int IM>
static void launch_inv_class(NttArgs a, hipStream_t s)
{
    constexpr uint32_t tpr = 1u << (LOGN - 12);
    if (!a.Lsel)
    {
        return;
    }
    a.total_work = a.n_poly * a.Lsel * tpr;
    hipLaunchKernelGGL((ntt_inv_contig<LOGN, IM>), dim3(a.total_work), dim3(256), 0, s, a);
    hipLaunchKernelGGL((ntt_inv_strided<LOGN, IM>), dim3(a.total_work), dim3(256), 0, s, a);
}

template <int LOGN>
static void launch_inv(const moai_ctx *c, const NttArgs &base, hipStream_t s)
{
    static const long lazy8 = env_long("MOAI_NTT_LAZY8", 1);
    NttArgs cls[4] = { base, base, base, base }; // exact, lazy8, FPN, FPR
    for (NttArgs &a : cls)
    {
        a.Lsel = 0;
    }
    cls[2].tw = cls[3].tw = c->inv_twf;
    cls[2].twb = cls[3].twb = c->inv_twfb;
    for (uint32_t r = 0; r < base.L; ++r)
    {
        const uint32_t prime = base.rows.idx[r];
        const int m = ntt_mode(c, prime);
        NttArgs &dst = m == M_FPN ? cls[2] : (m == M_FPR ? cls[3] : ((lazy8 && c->primes[prime] < (1ull << 60)) ? cls[1] : cls[0]));
        dst.selp.idx[dst.Lsel] = prime;
        dst.sel.idx[dst.Lsel++] = r;
    }
    launch_inv_class<LOGN, 0>(cls[0], s);
    launch_inv_class<LOGN, 1>(cls[1], s);
    launch_inv_class<LOGN, 2>(cls[2], s);
    launch_inv_class<LOGN, 3>(cls[3], s);
}

} // namespace moai
namespace moai {
// 36 q < 2^64: forward butterflies may skip the per-stage guard (modarith.hip.h ct_bfly_noguard)
bool noguard_ok(uint64_t q)
{
    return q < (~0ull) / 36;
}

// the arithmetic mode of the forward transform under context prime `prime` (modarith.hip.h M_*):
// FP64 below 2^51 (MOAI_NTT_FP=0 keeps everything on the integer units), else integer with or without guards
int ntt_mode(const moai_ctx *c, uint32_t prime)
{
    static const long fp = env_long("MOAI_NTT_FP", 1);
    const int m = (int)c->pc_host[prime].fp_mode;
    if (fp && m)
    {
        return m;
    }
    return noguard_ok(c->primes[prime]) ? M_NOGUARD : M_GUARD;
}

// single-launch transform (ntt_coop); its queue state lives in a per-stream arena
static size_t coop_grid(const moai_ctx *c)
{
    static const long wpc = env_long("MOAI_NTT_COOP_WPC", 4);
    return (size_t)c->num_cu * (size_t)(wpc > 0 ? wpc : 4);
}

static uint32_t coop_delay()
{
    static const long d = env_long("MOAI_NTT_COOP_DELAY", 4);
    return (uint32_t)(d > 0 ? d : 1);
}

static size_t coop_steps_cap(const moai_ctx *c, size_t rows)
{
    return rows + coop_delay() + coop_grid(c) + 8;
}

static size_t coop_state_bytes(const moai_ctx *c, size_t rows)
{
    return sizeof(CoopState) + sizeof(uint32_t) * (rows + 8 * coop_steps_cap(c, rows));
}

template <int LOGN>
static int launch_coop(moai_ctx *c, const NttArgs &base, bool inverse, void *state_mem, hipStream_t s)
{
    static const long occ = env_long("MOAI_NTT_COOP_OCC", 4);
    NttArgs a = base;
    const uint32_t rows = a.n_poly * a.L;
    const uint32_t grid = (uint32_t)coop_grid(c);
    CoopArgs ca;
    ca.rows = rows;
    ca.delay = coop_delay();
    ca.steps_cap = (uint32_t)coop_steps_cap(c, rows);
    ca.st = static_cast<CoopState *>(state_mem);
    ca.done = reinterpret_cast<uint32_t *>(static_cast<char *>(state_mem) + sizeof(CoopState));
    ca.rowmap = ca.done + rows;
    MOAI_HIP_CHECK(hipMemsetAsync(state_mem, 0, coop_state_bytes(c, rows), s));
    a.total_work = grid;
    if (occ == 3)
    {
        if (inverse)
        {
            hipLaunchKernelGGL((ntt_coop<LOGN, true, 3>), dim3(grid), dim3(256), 0, s, a, ca);
        }
        else
        {
            hipLaunchKernelGGL((ntt_coop<LOGN, false, 3>), dim3(grid), dim3(256), 0, s, a, ca);
        }
    }
    else
    {
        if (inverse)
        {
            hipLaunchKernelGGL((ntt_coop<LOGN, true, 4>), dim3(grid), dim3(256), 0, s, a, ca);
        }
        else
        {
            hipLaunchKernelGGL((ntt_coop<LOGN, false, 4>), dim3(grid), dim3(256), 0, s, a, ca);
        }
    }
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

// data [n_poly][L][N]; rows maps row -> prime.  Returns a MOAI_* code.
int ntt_launch(moai_ctx *c, uint64_t *data, size_t n_poly, size_t L, const RowMap &rows, bool inverse, hipStream_t s,
               const uint64_t *src, size_t src_stride_rows, size_t src_off_rows)
{
    if (n_poly == 0 || L == 0)
    {
        return MOAI_OK;
    }
    if (src && !inverse)
    {
        return set_error(MOAI_ELOGIC, "only the inverse transform reads from another buffer");
    }
    if (src && (src_stride_rows > 0xffffffffull || src_off_rows > 0xffffffffull || src_off_rows + L > src_stride_rows))
    {
        return set_error(MOAI_EINVAL, "source slice outside its layout");
    }
    if (n_poly * L * (c->n >> (c->logn >= 12 ? 12 : 0)) > 0x7fffffffull || n_poly > 0xffffffffull)
    {
        return set_error(MOAI_EINVAL, "batch too large for one launch");
    }
    NttArgs a;
    a.data = data;
    a.tw = inverse ? c->inv_tw : c->fwd_tw;
    a.twb = inverse ? c->inv_twb : c->fwd_twb;
    a.pc = c->pc;
    a.rows = rows;
    for (size_t r = 0; r < MOAI_MAX_RNS; ++r)
    {
        a.sel.idx[r] = (uint32_t)(r < L ? r : 0);
        a.selp.idx[r] = rows.idx[r < L ? r : 0];
    }
    a.Lsel = (uint32_t)L;
    a.L = (uint32_t)L;
    a.n_poly = (uint32_t)n_poly;
    a.total_work = 0;
    a.src = nullptr;
    a.src_stride = a.src_off = 0;
    static const long ldstw = env_long("MOAI_NTT_LDSTW", 1);
    a.lds_twiddles = ldstw ? 1u : 0u;
    const int logn = c->logn;
    static const long coop_on = env_long("MOAI_NTT_COOP", 0);
    if (src)
    {
        if (naive_requested() || logn <= 11 || coop_on)
        {
            // the paths that transform in place: bring the slice over first
            const size_t row_bytes = c->n * sizeof(uint64_t);
            MOAI_HIP_CHECK(hipMemcpy2DAsync(data, L * row_bytes, src + src_off_rows * c->n, src_stride_rows * row_bytes, L * row_bytes, n_poly,
                                            hipMemcpyDeviceToDevice, s));
        }
        else
        {
            a.src = src;
            a.src_stride = (uint32_t)src_stride_rows;
            a.src_off = (uint32_t)src_off_rows;
        }
    }
    if (naive_requested() && logn >= 1)
    {
        uint32_t bx = (uint32_t)(((c->n >> 1) + 255) / 256);
        dim3 grid(bx, (uint32_t)(n_poly * L));
        if (!inverse)
        {
            for (int st = 0; st < logn; ++st)
            {
                hipLaunchKernelGGL(ntt_stage_global<false>, grid, dim3(256), 0, s, a, logn, st, st == logn - 1);
            }
        }
        else
        {
            for (int st = logn - 1; st >= 0; --st)
            {
                hipLaunchKernelGGL(ntt_stage_global<true>, grid, dim3(256), 0, s, a, logn, st, st == 0);
            }
        }
        MOAI_LAUNCH_CHECK();
        return MOAI_OK;
    }
    if (logn <= 11)
    {
        dim3 grid((uint32_t)(n_poly * L));
        if (inverse)
        {
            hipLaunchKernelGGL(ntt_small<true>, grid, dim3(256), 0, s, a, logn);
        }
        else
        {
            hipLaunchKernelGGL(ntt_small<false>, grid, dim3(256), 0, s, a, logn);
        }
        MOAI_LAUNCH_CHECK();
        return MOAI_OK;
    }
    {
        // opt-in (MOAI_NTT_COOP=1): one persistent launch per transform, the two passes meeting in L2.
        // Measured on MI355X (round 1): no faster than two launches -- L2 is write-through, so only the
        // second-pass reads could be saved, and keeping enough rows in flight to avoid dependency stalls
        // overflows the 4 MiB L2.
        static const long coop = env_long("MOAI_NTT_COOP", 0);
        if (coop)
        {
            void *st = nullptr;
            int rc = reserve_for_stream(c, (void *)((uintptr_t)s ^ 1u), coop_state_bytes(c, n_poly * L), &st, false);
            if (rc)
            {
                return rc;
            }
            switch (logn)
            {
            case 12:
                return launch_coop<12>(c, a, inverse, st, s);
            case 13:
                return launch_coop<13>(c, a, inverse, st, s);
            case 14:
                return launch_coop<14>(c, a, inverse, st, s);
            case 15:
                return launch_coop<15>(c, a, inverse, st, s);
            case 16:
                return launch_coop<16>(c, a, inverse, st, s);
            default:
                return set_error(MOAI_ELOGIC, "unsupported poly_modulus_degree 2^%d", logn);
            }
        }
    }
    // The two passes of one transform exchange the whole polynomial through memory.  Launching them
    // per chunk of polynomials that fits the 256 MiB Infinity Cache lets the second pass read what
    // the first one just wrote from the cache instead of HBM (MOAI_NTT_CHUNK_MB=0 disables).
    size_t chunk = n_poly;
    {
        static long chunk_mb = -1;
        if (chunk_mb < 0)
        {
            const char *e = getenv("MOAI_NTT_CHUNK_MB");
            chunk_mb = e ? atol(e) : 0;
        }
        if (chunk_mb > 0)
        {
            size_t per_poly = L * c->n * sizeof(uint64_t);
            chunk = ((size_t)chunk_mb << 20) / per_poly;
            if (chunk < 1)
            {
                chunk = 1;
            }
        }
    }
#define MOAI_NTT_CASE(LG)               \
    case LG:                            \
        if (inverse)                    \
        {                               \
            launch_inv<LG>(c, a, s);    \
        }                               \
        else                            \
        {                               \
            launch_fwd<LG>(c, a, s);    \
        }                               \
        break;
    for (size_t p0 = 0; p0 < n_poly; p0 += chunk)
    {
        a.data = data + p0 * L * c->n;
        if (a.src)
        {
            a.src = src + p0 * src_stride_rows * c->n;
        }
        a.n_poly = (uint32_t)(n_poly - p0 < chunk ? n_poly - p0 : chunk);
        switch (logn)
        {
            MOAI_NTT_CASE(12)
            MOAI_NTT_CASE(13)
            MOAI_NTT_CASE(14)
            MOAI_NTT_CASE(15)
            MOAI_NTT_CASE(16)
        default:
            return set_error(MOAI_ELOGIC, "unsupported poly_modulus_degree 2^%d", logn);
        }
    }
#undef MOAI_NTT_CASE
    MOAI_LAUNCH_CHECK();
    return MOAI_OK;
}

int make_rowmap(const moai_ctx *c, size_t L, const uint32_t *prime_index, RowMap *out)
{
    if (L > MOAI_MAX_RNS)
    {
        return set_error(MOAI_EINVAL, "L = %zu exceeds MOAI_MAX_RNS", L);
    }
    for (size_t r = 0; r < L; r++)
    {
        uint32_t p = prime_index ? prime_index[r] : (uint32_t)r;
        if (p >= c->k)
        {
            return set_error(MOAI_ERANGE, "prime index %u out of range (k = %zu)", p, c->k);
        }
        out->idx[r] = (uint32_t)p;
    }
    for (size_t r = L; r < MOAI_MAX_RNS; r++)
    {
        out->idx[r] = 0;
    }
    return MOAI_OK;
}

} // namespace moai

using namespace moai;

static int ntt_entry(moai_ctx *c, uint64_t *data, size_t n_poly, size_t L, const uint32_t *prime_index, void *stream,
                     bool inverse)
{
    if (!c || (!data && n_poly * L))
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    RowMap rows;
    int rc = make_rowmap(c, L, prime_index, &rows);
    if (!rc)
    {
        rc = enter_device(c);
    }
    if (rc)
    {
        return rc;
    }
    return ntt_launch(c, data, n_poly, L, rows, inverse, (hipStream_t)stream);
}

extern "C" int moai_ntt_forward(moai_ctx *c, uint64_t *data, size_t n_poly, size_t L, const uint32_t *prime_index,
                                void *stream)
{
    MOAI_AUDIT(stream, data);
    trace_op("ntt_forward", L, n_poly);
    return ntt_entry(c, data, n_poly, L, prime_index, stream, false);
}

extern "C" int moai_ntt_inverse(moai_ctx *c, uint64_t *data, size_t n_poly, size_t L, const uint32_t *prime_index,
                                void *stream)
{
    MOAI_AUDIT(stream, data);
    trace_op("ntt_inverse", L, n_poly);
    return ntt_entry(c, data, n_poly, L, prime_index, stream, true);
}
