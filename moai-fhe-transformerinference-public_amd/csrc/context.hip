// context.hip -- host-side precomputation and the device arena of a moai_ctx.
//
// Product code (not the oracle): re-provides, per prime, what the reference precomputes in
// NTTTables::initialize (SEAL/util/ntt.cpp:241-300), Modulus::set_value (SEAL/modulus.cpp:36-77)
// and RNSTool::initialize's inv_q_last_mod_q (SEAL/util/rns.cpp:769-775).  One table per prime,
// shared by every level of the modulus chain.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>

#include "common.h"

namespace moai {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error()
{
    return g_err;
}

static std::mutex g_tuning_mu;
static std::map<std::string, long> g_tuning;

static std::atomic<int> g_trace_on{ 0 };
static std::mutex g_trace_mu;
static std::map<std::pair<std::string, size_t>, unsigned long long> g_trace;

void trace_op(const char *name, size_t L, size_t units)
{
    if (!g_trace_on.load(std::memory_order_relaxed))
    {
        return;
    }
    std::lock_guard<std::mutex> g(g_trace_mu);
    g_trace[std::make_pair(std::string(name), L)] += units;
}

// ---- stream audit (common.h) ----------------------------------------------------------------------------------------
std::atomic<int> g_stream_audit{ [] {
    const char *e = getenv("MOAI_STREAM_AUDIT");
    return e && e[0] != '0' ? 1 : 0;
}() };
struct AuditBlock
{
    size_t bytes;
    const void *stream;
    bool released;
};
static std::mutex g_audit_mu;
static std::map<uintptr_t, AuditBlock> g_audit_blocks; // by base address
static std::atomic<unsigned long long> g_audit_checked{ 0 }, g_audit_violations{ 0 };

// one line at process exit, so that a run under MOAI_STREAM_AUDIT=1 shows that the audit was live and what it saw
static struct AuditSummary
{
    ~AuditSummary()
    {
        if (g_stream_audit.load() || g_audit_checked.load())
        {
            fprintf(stderr, "[stream audit] %llu labelled pointers checked at enqueue, %llu violations\n", g_audit_checked.load(),
                    g_audit_violations.load());
        }
    }
} g_audit_summary;

int audit_ptrs(const char *fn, const void *stream, std::initializer_list<const void *> ptrs)
{
    std::lock_guard<std::mutex> g(g_audit_mu);
    for (const void *p : ptrs)
    {
        if (!p)
        {
            continue;
        }
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        auto it = g_audit_blocks.upper_bound(a);
        if (it == g_audit_blocks.begin())
        {
            continue;
        }
        --it;
        if (a >= it->first + it->second.bytes)
        {
            continue; // not in a labelled block
        }
        g_audit_checked.fetch_add(1, std::memory_order_relaxed);
        if (it->second.released || it->second.stream != stream)
        {
            g_audit_violations.fetch_add(1, std::memory_order_relaxed);
            fprintf(stderr, "[stream audit] %s: pointer %p (block %p + %zu, %zu bytes) %s stream %p, work enqueued on stream %p\n", fn, p,
                    (void *)it->first, (size_t)(a - it->first), it->second.bytes,
                    it->second.released ? "was released to the cache of" : "belongs to", it->second.stream, stream);
            return set_error(MOAI_ELOGIC, "stream audit: %s was handed %p, a block %s stream %p, to enqueue on stream %p", fn, p,
                             it->second.released ? "released to the cache of" : "labelled with", it->second.stream, stream);
        }
    }
    return MOAI_OK;
}

int enter_device(const moai_ctx *c)
{
    MOAI_HIP_CHECK(hipSetDevice(c->device));
    return MOAI_OK;
}

long tuning(const char *name, long dflt)
{
    {
        std::lock_guard<std::mutex> g(g_tuning_mu);
        auto it = g_tuning.find(name);
        if (it != g_tuning.end())
        {
            return it->second;
        }
    }
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
}

typedef unsigned __int128 u128;

static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q)
{
    return (uint64_t)(((u128)a * b) % q);
}

static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q)
{
    uint64_t r = 1;
    a %= q;
    while (e)
    {
        if (e & 1)
        {
            r = mulmod(r, a, q);
        }
        a = mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}

static bool is_prime_u64(uint64_t n)
{
    static const uint64_t bases[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    if (n < 2)
    {
        return false;
    }
    for (uint64_t p : bases)
    {
        if (n == p)
        {
            return true;
        }
        if (n % p == 0)
        {
            return false;
        }
    }
    uint64_t d = n - 1;
    int r = 0;
    while (!(d & 1))
    {
        d >>= 1;
        r++;
    }
    for (uint64_t a : bases)
    {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1)
        {
            continue;
        }
        bool comp = true;
        for (int i = 1; i < r; i++)
        {
            x = mulmod(x, x, n);
            if (x == n - 1)
            {
                comp = false;
                break;
            }
        }
        if (comp)
        {
            return false;
        }
    }
    return true;
}

static inline uint32_t bitrev(uint32_t x, int bits)
{
    return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0;
}

// The smallest primitive 2N-th root of unity mod q: the value the reference's
// try_minimal_primitive_root converges to (SEAL/util/numth.cpp:386-413).
static bool minimal_primitive_root(uint64_t two_n, uint64_t q, uint64_t *out)
{
    if ((q - 1) % two_n)
    {
        return false;
    }
    uint64_t cof = (q - 1) / two_n;
    uint64_t root = 0;
    for (uint64_t c = 2; c < 4096 && c < q; c++)
    {
        uint64_t r = powmod(c, cof, q);
        if (powmod(r, two_n >> 1, q) == q - 1)
        {
            root = r;
            break;
        }
    }
    if (!root)
    {
        return false;
    }
    // all primitive 2N-th roots are the odd powers of one of them
    uint64_t sq = mulmod(root, root, q);
    uint64_t cur = root, best = root;
    for (uint64_t i = 0; i < two_n; i += 2)
    {
        if (cur < best)
        {
            best = cur;
        }
        cur = mulmod(cur, sq, q);
    }
    *out = best;
    return true;
}

static inline Tw make_tw(uint64_t w, uint64_t q)
{
    Tw t;
    t.w = w;
    t.wq = (uint64_t)((((u128)w) << 64) / q);
    return t;
}

static void build_prime(int logn, uint64_t q, uint64_t psi, Tw *fwd, Tw *inv, PrimeConst *pc)
{
    const size_t n = (size_t)1 << logn;
    uint64_t psi_inv = powmod(psi, q - 2, q);
    uint64_t p = 1, pi = 1;
    // fwd[bitrev(i)] = psi^i ; inv[bitrev(i)] = psi^-i   (i >= 1; index 0 is never used by a stage)
    for (size_t i = 0; i < n; i++)
    {
        uint32_t r = bitrev((uint32_t)i, logn);
        fwd[r] = make_tw(p, q);
        inv[r] = make_tw(pi, q);
        p = mulmod(p, psi, q);
        pi = mulmod(pi, psi_inv, q);
    }
    memset(pc, 0, sizeof(*pc));
    pc->q = q;
    pc->q2 = q << 1;
    pc->nq = 0 - q;
    pc->n4q = 0 - (q << 2);
    {
        // floor(2^128 / q)
        u128 hi = ((u128)1 << 64);
        uint64_t q1 = (uint64_t)(hi / q);
        u128 rem = hi % q;
        uint64_t q0 = (uint64_t)((rem << 64) / q);
        pc->cr0 = q0;
        pc->cr1 = q1;
    }
    // FP64 arithmetic modes (modarith.hip.h): q < 2^51; without intermediate reductions when sixteen stages
    // starting from |x| <= q/2 and growing by at most 2 q each (the w-only butterfly, ct_bfly_fp1) stay below 2^52
    if (q < (1ull << 51))
    {
        const double qd = (double)q, qinv = 1.0 / (double)q;
        memcpy(&pc->qd, &qd, 8);
        memcpy(&pc->qinv, &qinv, 8);
        pc->fp_mode = (33 * (unsigned __int128)q < ((unsigned __int128)1 << 52)) ? 2 : 3; // M_FPN : M_FPR
    }
    uint64_t ninv = powmod((uint64_t)n % q, q - 2, q);
    pc->ninv = make_tw(ninv, q);
    pc->ninv_w1 = make_tw(mulmod(ninv, n > 1 ? inv[1].w : 1, q), q);
}

} // namespace moai

using namespace moai;

extern "C" const char *moai_last_error(void)
{
    return moai::last_error();
}

extern "C" int moai_set_tuning(const char *name, long value)
{
    if (!name)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    std::lock_guard<std::mutex> g(g_tuning_mu);
    g_tuning[name] = value;
    return MOAI_OK;
}

extern "C" int moai_op_trace(int enable)
{
    std::lock_guard<std::mutex> g(g_trace_mu);
    if (enable)
    {
        g_trace.clear();
    }
    g_trace_on.store(enable ? 1 : 0);
    return MOAI_OK;
}

extern "C" size_t moai_op_trace_dump(char *buf, size_t cap)
{
    std::lock_guard<std::mutex> g(g_trace_mu);
    std::string out;
    for (const auto &kv : g_trace)
    {
        out += kv.first.first + " " + std::to_string(kv.first.second) + " " + std::to_string(kv.second) + "\n";
    }
    if (buf && cap)
    {
        const size_t nbytes = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), nbytes);
        buf[nbytes] = 0;
    }
    return out.size() + 1;
}

extern "C" int moai_debug_stream_audit(int enable)
{
    const int was = moai::g_stream_audit.exchange(enable ? 1 : 0);
    return was;
}

extern "C" void moai_debug_block_label(const void *ptr, size_t bytes, const void *stream, int state)
{
    if (!ptr || !moai::g_stream_audit.load(std::memory_order_relaxed))
    {
        return;
    }
    std::lock_guard<std::mutex> g(moai::g_audit_mu);
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    if (state == 0)
    {
        moai::g_audit_blocks.erase(a); // back to the device allocator
    }
    else
    {
        moai::g_audit_blocks[a] = moai::AuditBlock{ bytes, stream, state == 2 };
    }
}

extern "C" void moai_debug_stream_audit_counts(unsigned long long *checked, unsigned long long *violations)
{
    if (checked)
    {
        *checked = moai::g_audit_checked.load();
    }
    if (violations)
    {
        *violations = moai::g_audit_violations.load();
    }
}

extern "C" int moai_version(void)
{
    return 100;
}

extern "C" int moai_ctx_create(int logn, const uint64_t *primes, size_t k, int device, moai_ctx **out)
{
    if (!out || !primes)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    *out = nullptr;
    if (logn < 1 || logn > 16)
    {
        return set_error(MOAI_EINVAL, "coeff_count_power out of range (1..16)");
    }
    if (k < 1 || k > MOAI_MAX_RNS)
    {
        return set_error(MOAI_EINVAL, "prime count out of range (1..%d)", MOAI_MAX_RNS);
    }
    const size_t n = (size_t)1 << logn;
    for (size_t i = 0; i < k; i++)
    {
        uint64_t q = primes[i];
        if (q >> 61 || q < 3 || (q - 1) % (2 * n) != 0 || !is_prime_u64(q))
        {
            return set_error(MOAI_EINVAL, "invalid modulus %llu: need a prime = 1 mod 2N below 2^61",
                             (unsigned long long)q);
        }
        for (size_t j = 0; j < i; j++)
        {
            if (primes[j] == q)
            {
                return set_error(MOAI_EINVAL, "coeff_modulus is not pairwise distinct");
            }
        }
    }
    MOAI_HIP_CHECK(hipSetDevice(device));
    moai_ctx *c = new moai_ctx();
    c->device = device;
    c->logn = logn;
    c->n = n;
    c->k = k;
    c->primes.assign(primes, primes + k);
    c->roots.resize(k);
    c->pc_host.resize(k);
    c->mutex = new std::mutex();
    c->op_mutex = new std::mutex();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
    {
        c->num_cu = prop.multiProcessorCount;
    }

    std::vector<Tw> fwd(k * n), inv(k * n);
    bool ok = true;
    {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt ? (nt > 16 ? 16 : nt) : 4;
        std::vector<std::thread> th;
        std::mutex mu;
        size_t next = 0;
        for (unsigned t = 0; t < nt; t++)
        {
            th.emplace_back([&]() {
                for (;;)
                {
                    size_t i;
                    {
                        std::lock_guard<std::mutex> g(mu);
                        i = next++;
                    }
                    if (i >= k)
                    {
                        return;
                    }
                    uint64_t psi;
                    if (!minimal_primitive_root(2 * n, primes[i], &psi))
                    {
                        std::lock_guard<std::mutex> g(mu);
                        ok = false;
                        continue;
                    }
                    c->roots[i] = psi;
                    build_prime(logn, primes[i], psi, &fwd[i * n], &inv[i * n], &c->pc_host[i]);
                }
            });
        }
        for (auto &t : th)
        {
            t.join();
        }
    }
    if (!ok)
    {
        moai_ctx_destroy(c);
        return set_error(MOAI_EINVAL, "invalid modulus: no primitive 2N-th root");
    }
    // per-thread ordering of the contiguous pass's last four stages (ntt_kernels.hip.h fwd_contig_tile)
    std::vector<Tw> fwdb, invb;
    const size_t nb = logn >= 12 ? (n >> 12) * 15 * 256 : 0;
    if (nb)
    {
        const int r1 = logn - 8;
        fwdb.resize(k * nb);
        invb.resize(k * nb);
        for (size_t p = 0; p < k; p++)
        {
            for (size_t tile = 0; tile < (n >> 12); tile++)
            {
                for (int u = 4; u < 8; u++)
                {
                    for (uint32_t i2 = 0; i2 < (1u << (u - 4)); i2++)
                    {
                        const size_t slot = (1u << (u - 4)) - 1 + i2;
                        for (uint32_t tid = 0; tid < 256; tid++)
                        {
                            const uint32_t b = tid >> 4, tl = tid & 15u;
                            const size_t src = ((size_t)1 << (r1 + u)) + ((tile * 16 + b) << u) + ((tl << (u - 4)) | i2);
                            const size_t dst = p * nb + ((tile * 15 + slot) << 8) + tid;
                            fwdb[dst] = fwd[p * n + src];
                            invb[dst] = inv[p * n + src];
                        }
                    }
                }
            }
        }
    }
    // FP64 forward tables: {double w, double RN(w / q)} in the two words of a Tw (modarith.hip.h)
    std::vector<Tw> fwdf, fwdfb, invf, invfb;
    std::vector<double> fwdf1;
    if (logn >= 12)
    {
        fwdf1.assign(k * n, 0.0);
        fwdf.assign(k * n, Tw{ 0, 0 });
        fwdfb.assign(k * nb, Tw{ 0, 0 });
        invf.assign(k * n, Tw{ 0, 0 });
        invfb.assign(k * nb, Tw{ 0, 0 });
        auto to_fp = [](Tw t, uint64_t q) {
            double w = (double)t.w, wq = (double)t.w / (double)q;
            Tw o;
            memcpy(&o.w, &w, 8);
            memcpy(&o.wq, &wq, 8);
            return o;
        };
        for (size_t p = 0; p < k; p++)
        {
            if (!c->pc_host[p].fp_mode)
            {
                continue;
            }
            for (size_t i = 0; i < n; i++)
            {
                fwdf[p * n + i] = to_fp(fwd[p * n + i], primes[p]);
                fwdf1[p * n + i] = (double)fwd[p * n + i].w;
                invf[p * n + i] = to_fp(inv[p * n + i], primes[p]);
            }
            for (size_t i = 0; i < nb; i++)
            {
                fwdfb[p * nb + i] = to_fp(fwdb[p * nb + i], primes[p]);
                invfb[p * nb + i] = to_fp(invb[p * nb + i], primes[p]);
            }
        }
    }
    // inv_qlast[l*k + i] = q_l^-1 mod q_i  (for all l != i; the rescale uses l = L-1 > i, the
    // key-switch mod-down uses l = k-1)
    c->inv_qlast_host.assign(k * k, Tw{ 0, 0 });
    for (size_t l = 0; l < k; l++)
    {
        for (size_t i = 0; i < k; i++)
        {
            if (i != l)
            {
                uint64_t qi = primes[i];
                c->inv_qlast_host[l * k + i] = make_tw(powmod(primes[l] % qi, qi - 2, qi), qi);
            }
        }
    }
    hipError_t e;
    if ((e = hipMalloc(&c->fwd_tw, sizeof(Tw) * k * n)) != hipSuccess ||
        (e = hipMalloc(&c->inv_tw, sizeof(Tw) * k * n)) != hipSuccess ||
        (e = hipMalloc(&c->pc, sizeof(PrimeConst) * k)) != hipSuccess ||
        (e = hipMalloc(&c->inv_qlast, sizeof(Tw) * k * k)) != hipSuccess ||
        (nb && (e = hipMalloc(&c->fwd_twf1, sizeof(double) * k * n)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->fwd_twf1, fwdf1.data(), sizeof(double) * k * n, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMalloc(&c->fwd_twf, sizeof(Tw) * k * n)) != hipSuccess) ||
        (nb && (e = hipMalloc(&c->fwd_twfb, sizeof(Tw) * k * nb)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->fwd_twf, fwdf.data(), sizeof(Tw) * k * n, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->fwd_twfb, fwdfb.data(), sizeof(Tw) * k * nb, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMalloc(&c->inv_twf, sizeof(Tw) * k * n)) != hipSuccess) ||
        (nb && (e = hipMalloc(&c->inv_twfb, sizeof(Tw) * k * nb)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->inv_twf, invf.data(), sizeof(Tw) * k * n, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->inv_twfb, invfb.data(), sizeof(Tw) * k * nb, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMalloc(&c->fwd_twb, sizeof(Tw) * k * nb)) != hipSuccess) ||
        (nb && (e = hipMalloc(&c->inv_twb, sizeof(Tw) * k * nb)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->fwd_twb, fwdb.data(), sizeof(Tw) * k * nb, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMemcpy(c->inv_twb, invb.data(), sizeof(Tw) * k * nb, hipMemcpyHostToDevice)) != hipSuccess) ||
        (e = hipMemcpy(c->fwd_tw, fwd.data(), sizeof(Tw) * k * n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(c->inv_tw, inv.data(), sizeof(Tw) * k * n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(c->pc, c->pc_host.data(), sizeof(PrimeConst) * k, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(c->inv_qlast, c->inv_qlast_host.data(), sizeof(Tw) * k * k, hipMemcpyHostToDevice)) !=
            hipSuccess)
    {
        moai_ctx_destroy(c);
        return set_error(MOAI_EHIP, "context upload failed: %s", hipGetErrorString(e));
    }
    c->galois_tables.assign(n, nullptr);
    *out = c;
    return MOAI_OK;
}

extern "C" void moai_ctx_destroy(moai_ctx *c)
{
    if (!c)
    {
        return;
    }
    (void)hipFree(c->fwd_tw);
    (void)hipFree(c->inv_tw);
    (void)hipFree(c->pc);
    (void)hipFree(c->inv_qlast);
    (void)hipFree(c->fwd_twf);
    (void)hipFree(c->fwd_twf1);
    (void)hipFree(c->fwd_twfb);
    (void)hipFree(c->inv_twf);
    (void)hipFree(c->inv_twfb);
    (void)hipFree(c->fwd_twb);
    (void)hipFree(c->inv_twb);
    for (auto &kv : c->ws)
    {
        (void)hipFree(kv.second.ptr);
    }
    for (uint32_t *t : c->galois_tables)
    {
        if (t)
        {
            (void)hipFree(t);
        }
    }
    (void)hipFree(c->ckks_src_map);
    (void)hipFree(c->ckks_inv_roots);
    delete static_cast<std::mutex *>(c->mutex);
    delete static_cast<std::mutex *>(c->op_mutex);
    delete c;
}

namespace moai {
// grow the arena of `stream` to at least `bytes`; synchronises the device when it has to reallocate
static int reserve_for_stream_locked(moai_ctx *c, moai_ctx::Arena &a, size_t bytes);
int reserve_for_stream(moai_ctx *c, void *stream, size_t bytes, void **out, bool headroom)
{
    std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
    moai_ctx::Arena &a = c->ws[stream];
    if (bytes > a.bytes)
    {
        if (headroom)
        {
            // an operation outgrew the arena: over-allocate so that a slowly growing batch or level mix does not pay a
            // device synchronisation plus a multi-GiB hipFree / hipMalloc on every call (falls back to the exact size)
            const size_t want = bytes + bytes / 4 > a.bytes + a.bytes / 2 ? bytes + bytes / 4 : a.bytes + a.bytes / 2;
            int rc = reserve_for_stream_locked(c, a, want);
            if (rc == MOAI_OK)
            {
                if (out)
                {
                    *out = a.ptr;
                }
                return MOAI_OK;
            }
        }
        int rc = reserve_for_stream_locked(c, a, bytes);
        if (rc)
        {
            return rc;
        }
    }
    if (out)
    {
        *out = a.ptr;
    }
    return MOAI_OK;
}

int reserve_for_stream_locked(moai_ctx *c, moai_ctx::Arena &a, size_t bytes)
{
    (void)c;
    {
        MOAI_HIP_CHECK(hipDeviceSynchronize());
        if (a.ptr)
        {
            MOAI_HIP_CHECK(hipFree(a.ptr));
            a.ptr = nullptr;
            a.bytes = 0;
        }
        hipError_t e = hipMalloc(&a.ptr, bytes);
        if (e != hipSuccess)
        {
            a.ptr = nullptr;
            (void)hipGetLastError(); // reported through the return value, see moai_malloc
            return set_error(MOAI_ENOMEM, "workspace of %zu bytes: %s", bytes, hipGetErrorString(e));
        }
        a.bytes = bytes;
    }
    return MOAI_OK;
}
} // namespace moai

extern "C" int moai_ctx_reserve(moai_ctx *c, size_t bytes)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    // reserves the arena of the default stream; other streams grow theirs on first use
    return moai::reserve_for_stream(c, nullptr, bytes, nullptr, false);
}

extern "C" int moai_ctx_reserve_stream(moai_ctx *c, void *stream, size_t bytes)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    return moai::reserve_for_stream(c, stream, bytes, nullptr, false);
}

extern "C" size_t moai_ctx_coeff_count(const moai_ctx *c)
{
    return c ? c->n : 0;
}

extern "C" size_t moai_ctx_prime_count(const moai_ctx *c)
{
    return c ? c->k : 0;
}

extern "C" uint64_t moai_ctx_root(const moai_ctx *c, size_t prime)
{
    return (c && prime < c->k) ? c->roots[prime] : 0;
}

// ---- memory / stream plumbing ---------------------------------------------------------------------
extern "C" int moai_malloc(void **dptr, size_t bytes)
{
    if (!dptr)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e != hipSuccess)
    {
        // the failure is reported through the return value; do not leave it behind as the thread's "last error",
        // where the next launch check would find it after the caller has freed memory and retried successfully
        (void)hipGetLastError();
        return set_error(MOAI_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return MOAI_OK;
}

extern "C" int moai_free(void *dptr)
{
    MOAI_HIP_CHECK(hipFree(dptr));
    return MOAI_OK;
}

extern "C" int moai_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    MOAI_AUDIT(stream, dst);
    MOAI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    MOAI_AUDIT(stream, src);
    MOAI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    MOAI_AUDIT(stream, dst, src);
    MOAI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_memset_zero(void *dst, size_t bytes, void *stream)
{
    MOAI_AUDIT(stream, dst);
    MOAI_HIP_CHECK(hipMemsetAsync(dst, 0, bytes, (hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_stream_create(void **stream)
{
    hipStream_t s;
    MOAI_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return MOAI_OK;
}

extern "C" int moai_stream_destroy(void *stream)
{
    MOAI_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_stream_sync(void *stream)
{
    MOAI_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_device_info(int device, char *name, size_t cap, int *cus, size_t *hbm)
{
    hipDeviceProp_t prop;
    MOAI_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && cap)
    {
        // some ROCm builds leave prop.name empty for this part: describe the device by what the runtime does report
        if (prop.name[0])
        {
            snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
        }
        else
        {
            snprintf(name, cap, "%s, %d CUs, %zu GiB", prop.gcnArchName, prop.multiProcessorCount, prop.totalGlobalMem >> 30);
        }
    }
    if (cus)
    {
        *cus = prop.multiProcessorCount;
    }
    if (hbm)
    {
        *hbm = prop.totalGlobalMem;
    }
    return MOAI_OK;
}

extern "C" int moai_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    size_t f = 0, t = 0;
    MOAI_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes)
    {
        *free_bytes = f;
    }
    if (total_bytes)
    {
        *total_bytes = t;
    }
    return MOAI_OK;
}

extern "C" int moai_event_create(void **ev)
{
    hipEvent_t e;
    MOAI_HIP_CHECK(hipEventCreate(&e));
    *ev = (void *)e;
    return MOAI_OK;
}

extern "C" int moai_event_destroy(void *ev)
{
    MOAI_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
    return MOAI_OK;
}

extern "C" int moai_event_synchronize(void *ev)
{
    MOAI_HIP_CHECK(hipEventSynchronize((hipEvent_t)ev));
    return MOAI_OK;
}

extern "C" int moai_host_malloc(void **hptr, size_t bytes)
{
    if (!hptr)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault);
    if (e != hipSuccess)
    {
        return set_error(MOAI_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return MOAI_OK;
}

extern "C" int moai_host_free(void *hptr)
{
    if (hptr)
    {
        MOAI_HIP_CHECK(hipHostFree(hptr));
    }
    return MOAI_OK;
}

extern "C" int moai_event_record(void *ev, void *stream)
{
    MOAI_HIP_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return MOAI_OK;
}

extern "C" int moai_event_elapsed_ms(void *start, void *stop, float *ms)
{
    MOAI_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    MOAI_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MOAI_OK;
}
