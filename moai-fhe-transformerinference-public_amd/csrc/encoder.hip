// encoder.hip -- CKKSEncoder::encode_internal on the device (SEAL/ckks.h:457-637).
//
// The floating-point part is an FP64 inverse DWT over std::complex<double>.  It is kept bit-identical to
// the reference's scalar x86-64 build by performing exactly its operations per butterfly -- sums,
// differences and the four products of a complex multiply each rounded on their own (this file is
// compiled with -ffp-contract=off, FP64 denormals are always honoured on CDNA) -- in any order ACROSS
// butterflies, which is all a parallel schedule changes.  Root tables are computed on the host with the
// same libm calls the reference makes (util/croots.cpp:18-42) and uploaded.
//
// Two launches per batch of value vectors, both HBM/latency bound and tiny next to the NTTs that follow:
//   ckks_fft_contig   : gather the slot vector, first min(logn, 12) stages inside 4096-element LDS tiles
//   ckks_fft_finish   : remaining logn-12 stages in registers (elements 4096 apart), std::round, exact
//                       residues for a group of primes, max |coefficient| for the caller's range check
// followed by the library's forward NTT on the residues.
#include <cmath>
#include <complex>
#include <mutex>

#include "launch.h"
#include "modarith.hip.h"

namespace moai {

constexpr int ENC_TILE_LOG = 12;
constexpr uint32_t ENC_TILE = 1u << ENC_TILE_LOG;
constexpr int ENC_ROWS_PER_BLOCK = 4;

struct EncArgs
{
    const double *values;
    const uint32_t *src_map; // [N]: which slot lands on position p (inverse of matrix_reps_index_map_)
    const double2 *roots;    // inv_root_powers_ [N]
    double2 *scratch;        // [n_batch][N]
    uint64_t *dst;           // [n_batch][L][N]
    unsigned long long *max_bits; // [n_batch] or null: bit pattern of the largest |coefficient|
    const PrimeConst *pc;
    RowMap rows;
    uint32_t L;
    uint32_t logn;
    uint32_t values_size;
    uint32_t is_complex;
    const int32_t *mask; // masked-constant form: values[b][s] = mask[s] == 1 ? values[b] : 0 (values holds one double per vector)
    double fix; // scale / N
};

// (x, y) <- (x + y, (x - y) * r): DWTHandler::transform_from_rev's butterfly with
// Arithmetic<complex<double>, complex<double>, double> (SEAL/ckks.h:46-81)
__device__ __forceinline__ void gs_cbfly(double &xr, double &xi, double &yr, double &yi, double rr, double ri)
{
    double ur = xr, ui = xi, vr = yr, vi = yi;
    xr = ur + vr;
    xi = ui + vi;
    double a = ur - vr, b = ui - vi;
    double ac = a * rr, bd = b * ri, ad = a * ri, bc = b * rr;
    yr = ac - bd;
    yi = ad + bc;
}

// the last stage with the scalar folded in (dwthandler.h:273-314): x' = (u + v) * s, y' = (u - v) * (r * s)
__device__ __forceinline__ void gs_cbfly_last(double &xr, double &xi, double &yr, double &yi, double rr, double ri,
                                              double s)
{
    double ur = xr, ui = xi, vr = yr, vi = yi;
    double sr = rr * s, si = ri * s;
    xr = (ur + vr) * s;
    xi = (ui + vi) * s;
    double a = ur - vr, b = ui - vi;
    double ac = a * sr, bd = b * si, ad = a * si, bc = b * sr;
    yr = ac - bd;
    yi = ad + bc;
}

__global__ __launch_bounds__(256) void ckks_fft_contig(EncArgs g)
{
    __shared__ double re[ENC_TILE];
    __shared__ double im[ENC_TILE];
    const uint32_t n = 1u << g.logn;
    const uint32_t slots = n >> 1;
    const uint32_t tile = n < ENC_TILE ? n : ENC_TILE;
    const uint32_t tiles = n / tile;
    const uint32_t b = blockIdx.x / tiles;
    const uint32_t base = (blockIdx.x % tiles) * tile;
    const uint32_t tid = threadIdx.x;

    // conj_values[matrix_reps_index_map_[i]] = values[i], [.. i + slots] = conj(values[i]) (ckks.h:505-510),
    // read through the inverse map so that the writes are the contiguous side
    const double *vals = g.mask ? g.values + b : g.values + (size_t)b * g.values_size * (g.is_complex ? 2 : 1);
    for (uint32_t k = tid; k < tile; k += 256)
    {
        uint32_t s = g.src_map[base + k];
        uint32_t slot = s & (slots - 1);
        double r = 0.0, i = 0.0;
        if (g.mask)
        {
            // a real constant on the masked slots: its conjugate has imaginary part -0.0 like std::conj(double)
            if (slot < g.values_size)
            {
                r = g.mask[slot] == 1 ? vals[0] : 0.0;
                i = s >= slots ? -0.0 : 0.0;
            }
        }
        else if (slot < g.values_size)
        {
            r = g.is_complex ? vals[2 * slot] : vals[slot];
            i = g.is_complex ? vals[2 * slot + 1] : 0.0;
            if (s >= slots)
            {
                i = -i;
            }
        }
        re[k] = r;
        im[k] = i;
    }
    __syncthreads();

    const uint32_t stages = g.logn < (uint32_t)ENC_TILE_LOG ? g.logn : (uint32_t)ENC_TILE_LOG;
    for (uint32_t st = 0; st < stages; st++)
    {
        // stage st: gap = 2^st, m = n / 2^(st+1) groups, roots at [n - 2m + 1 + group]
        const uint32_t gap = 1u << st;
        const uint32_t m = n >> (st + 1);
        const uint32_t root0 = n - 2 * m + 1;
        const bool last = (st + 1 == g.logn);
        for (uint32_t t = tid; t < (tile >> 1); t += 256)
        {
            uint32_t grp = t >> st;
            uint32_t j = t & (gap - 1);
            uint32_t x = (grp << (st + 1)) + j;
            uint32_t y = x + gap;
            double2 r = g.roots[root0 + (base >> (st + 1)) + grp];
            double xr = re[x], xi = im[x], yr = re[y], yi = im[y];
            if (last)
            {
                gs_cbfly_last(xr, xi, yr, yi, r.x, r.y, g.fix);
            }
            else
            {
                gs_cbfly(xr, xi, yr, yi, r.x, r.y);
            }
            re[x] = xr;
            im[x] = xi;
            re[y] = yr;
            im[y] = yi;
        }
        __syncthreads();
    }
    double2 *out = g.scratch + (size_t)b * n + base;
    for (uint32_t k = tid; k < tile; k += 256)
    {
        out[k] = make_double2(re[k], im[k]);
    }
}

// The same twelve stages for n >= 4096 with sixteen elements per thread: four stages at a time in registers (the element
// index bits 0-3, then 4-7, then 8-11 vary inside a thread), two exchanges through one 32 KiB LDS buffer (real parts, then
// imaginary parts).  Every butterfly is the same gs_cbfly on the same operands as in ckks_fft_contig -- only which thread
// performs it changes -- so the results are the same bits.  LDS slot of element e: e ^ ((e >> 4) & 15) (at most two-way bank
// conflicts in every access pattern below).
__device__ __forceinline__ uint32_t enc_slot(uint32_t e)
{
    return e ^ ((e >> 4) & 15u);
}

// stages S0..S0+3 on the sixteen elements of a thread; element k of the thread is tile element e0 + k * estep
template <int S0>
__device__ __forceinline__ void enc_round(double (&re)[16], double (&im)[16], const EncArgs &g, uint32_t n, uint32_t base, uint32_t e0,
                                          uint32_t estep)
{
#pragma unroll
    for (int s = 0; s < 4; s++)
    {
        const uint32_t st = S0 + s;
        const uint32_t m = n >> (st + 1);
        const double2 *roots = g.roots + (n - 2 * m + 1) + (base >> (st + 1));
        const bool last = (st + 1 == g.logn);
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            if (k & (1 << s))
            {
                continue;
            }
            const uint32_t x = e0 + (uint32_t)k * estep; // the pair's lower element inside the tile
            const double2 r = roots[x >> (st + 1)];
            if (last)
            {
                gs_cbfly_last(re[k], im[k], re[k + (1 << s)], im[k + (1 << s)], r.x, r.y, g.fix);
            }
            else
            {
                gs_cbfly(re[k], im[k], re[k + (1 << s)], im[k + (1 << s)], r.x, r.y);
            }
        }
    }
}

__global__ __launch_bounds__(256) void ckks_fft_contig16(EncArgs g)
{
    __shared__ double buf[ENC_TILE];
    const uint32_t n = 1u << g.logn;
    const uint32_t slots = n >> 1;
    const uint32_t tiles = n >> ENC_TILE_LOG;
    const uint32_t b = blockIdx.x / tiles;
    const uint32_t base = (blockIdx.x % tiles) << ENC_TILE_LOG;
    const uint32_t tid = threadIdx.x;
    double re[16], im[16];

    // gather (ckks.h:505-510 through the inverse map), elements 16 tid .. 16 tid + 15
    const double *vals = g.mask ? g.values + b : g.values + (size_t)b * g.values_size * (g.is_complex ? 2 : 1);
    const uint4 *sm = reinterpret_cast<const uint4 *>(g.src_map + base + 16 * tid);
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++)
    {
        const uint4 s4 = sm[k4];
        const uint32_t ss[4] = { s4.x, s4.y, s4.z, s4.w };
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t s = ss[u];
            const uint32_t slot = s & (slots - 1);
            double r = 0.0, i = 0.0;
            if (g.mask)
            {
                if (slot < g.values_size)
                {
                    r = g.mask[slot] == 1 ? vals[0] : 0.0;
                    i = s >= slots ? -0.0 : 0.0;
                }
            }
            else if (slot < g.values_size)
            {
                r = g.is_complex ? vals[2 * slot] : vals[slot];
                i = g.is_complex ? vals[2 * slot + 1] : 0.0;
                if (s >= slots)
                {
                    i = -i;
                }
            }
            re[4 * k4 + u] = r;
            im[4 * k4 + u] = i;
        }
    }
    enc_round<0>(re, im, g, n, base, 16 * tid, 1);
    // exchange 1: element 16 tid + k  ->  thread (hi4, lo4) = (bits 8-11, bits 0-3) holds hi4 * 256 + k * 16 + lo4
    const uint32_t hi4 = tid >> 4, lo4 = tid & 15u;
#pragma unroll
    for (int part = 0; part < 2; part++)
    {
        double(&v)[16] = part ? im : re;
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            buf[enc_slot(16 * tid + k)] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            v[k] = buf[enc_slot(hi4 * 256 + k * 16 + lo4)];
        }
        __syncthreads();
    }
    enc_round<4>(re, im, g, n, base, hi4 * 256 + lo4, 16);
    // exchange 2: ->  thread tid holds k * 256 + tid
#pragma unroll
    for (int part = 0; part < 2; part++)
    {
        double(&v)[16] = part ? im : re;
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            buf[enc_slot(hi4 * 256 + k * 16 + lo4)] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            v[k] = buf[enc_slot(k * 256 + tid)];
        }
        __syncthreads();
    }
    enc_round<8>(re, im, g, n, base, tid, 256);
    double2 *out = g.scratch + (size_t)b * n + base;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        out[k * 256 + tid] = make_double2(re[k], im[k]);
    }
}

// |c| of an integer-valued double as mant * 2^sh with mant < 2^53 (sh < 0 only shifts out zero bits)
__device__ __forceinline__ void split_rounded(double c, uint64_t &mag, int &sh)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(c);
    const int e = (int)((bits >> 52) & 0x7ff);
    mag = 0;
    sh = 0;
    if (e != 0)
    {
        const uint64_t mant = (bits & ((1ull << 52) - 1)) | (1ull << 52);
        sh = e - 1075;
        if (sh <= 0)
        {
            mag = sh > -64 ? (mant >> (-sh)) : 0;
            sh = 0;
        }
        else if (sh <= 11)
        {
            mag = mant << sh;
            sh = 0;
        }
        else
        {
            mag = mant;
        }
    }
}

// (mag * 2^sh) mod q for sh > 0: coefficients of 64 bits and more (the reference's barrett_reduce_128 and
// RNSBase::decompose branches, ckks.h:575-629); rare, kept out of line
__device__ __noinline__ uint64_t residue_shifted(uint64_t mag, int sh, uint64_t q, uint64_t cr0, uint64_t cr1)
{
    uint64_t r = barrett64(mag, q, cr1);
    while (sh > 0)
    {
        int s = sh < 63 ? sh : 63;
        r = barrett128(r << s, r >> (64 - s), q, cr0, cr1);
        sh -= s;
    }
    return r;
}

template <int R>
__global__ __launch_bounds__(256) void ckks_fft_finish(EncArgs g)
{
    constexpr uint32_t K = 1u << R;
    const uint32_t n = 1u << g.logn;
    const uint32_t tile = n >> R; // 4096, or n when R == 0
    const uint32_t p_raw = blockIdx.x * 256 + threadIdx.x;
    const bool active = p_raw < tile; // N < 256 only; inactive lanes still take part in the wave reduction
    const uint32_t p = active ? p_raw : tile - 1;
    const uint32_t b = blockIdx.z;
    const double2 *in = g.scratch + (size_t)b * n;
    double xr[K], xi[K];
#pragma unroll
    for (uint32_t k = 0; k < K; k++)
    {
        double2 v = in[p + k * tile];
        xr[k] = v.x;
        xi[k] = v.y;
    }
    // element p + k*tile: stage t pairs k with k + 2^t; group = k >> (t+1); m = 2^(R-t-1) groups
#pragma unroll
    for (int t = 0; t < R; t++)
    {
        const uint32_t m = 1u << (R - t - 1);
        const uint32_t root0 = n - 2 * m + 1;
#pragma unroll
        for (uint32_t k = 0; k < K; k++)
        {
            if (!(k & (1u << t)))
            {
                double2 r = g.roots[root0 + (k >> (t + 1))];
                if (t == R - 1)
                {
                    gs_cbfly_last(xr[k], xi[k], xr[k + (1u << t)], xi[k + (1u << t)], r.x, r.y, g.fix);
                }
                else
                {
                    gs_cbfly(xr[k], xi[k], xr[k + (1u << t)], xi[k + (1u << t)], r.x, r.y);
                }
            }
        }
    }
    if (blockIdx.y == 0 && g.max_bits)
    {
        // max over fabs(real part) (ckks.h:527-531); non-negative doubles order like their bit patterns
        unsigned long long mx = 0;
#pragma unroll
        for (uint32_t k = 0; k < K; k++)
        {
            unsigned long long v = active ? (unsigned long long)__double_as_longlong(fabs(xr[k])) : 0ull;
            mx = v > mx ? v : mx;
        }
        for (int off = 32; off > 0; off >>= 1)
        {
            unsigned long long o = __shfl_down(mx, off);
            mx = o > mx ? o : mx;
        }
        if ((threadIdx.x & 63) == 0)
        {
            atomicMax(g.max_bits + b, mx);
        }
    }
    // std::round (halfway cases away from zero), then the exact integer modulo each prime with the sign
    // applied by negate_uint_mod: what all three branches of ckks.h:549-629 store
    uint64_t mag[K];
    int sh[K];
    uint32_t negmask = 0;
    int any_shift = 0;
#pragma unroll
    for (uint32_t k = 0; k < K; k++)
    {
        double c = round(xr[k]);
        negmask |= (uint32_t)(((uint64_t)__double_as_longlong(c)) >> 63) << k;
        split_rounded(c, mag[k], sh[k]);
        any_shift |= sh[k];
    }
    if (!active)
    {
        return;
    }
    const uint32_t row0 = blockIdx.y * ENC_ROWS_PER_BLOCK;
    for (uint32_t j = row0; j < row0 + ENC_ROWS_PER_BLOCK && j < g.L; j++)
    {
        const PrimeConst pc = g.pc[g.rows.idx[j]];
        uint64_t *out = g.dst + ((size_t)b * g.L + j) * n + p;
#pragma unroll
        for (uint32_t k = 0; k < K; k++)
        {
            uint64_t r = any_shift ? residue_shifted(mag[k], sh[k], pc.q, pc.cr0, pc.cr1) : barrett64(mag[k], pc.q, pc.cr1);
            out[k * tile] = (((negmask >> k) & 1u) && r) ? pc.q - r : r;
        }
    }
}

// ---- host side: tables exactly as CKKSEncoder's constructor builds them (SEAL/ckks.cpp:13-76) ------------
static uint32_t reverse_bits(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++)
    {
        r = (r << 1) | ((x >> i) & 1u);
    }
    return r;
}

namespace {
// util::ComplexRoots (SEAL/util/croots.cpp:18-75): an eighth of the circle from std::polar, the rest by
// exact symmetries
struct ComplexRoots
{
    size_t degree;
    std::vector<std::complex<double>> roots;
    explicit ComplexRoots(size_t degree_of_roots) : degree(degree_of_roots), roots(degree_of_roots / 8 + 1)
    {
        const double PI_ = 3.1415926535897932384626433832795028842;
        for (size_t i = 0; i <= degree / 8; i++)
        {
            // std::polar(1.0, theta) = (cos(theta), sin(theta)), which GCC at -O2 and above (the reference's
            // Release build) turns into ONE sincos call; glibc's sincos does not return sin()'s bits for
            // every argument, so ask for sincos explicitly instead of leaving it to this compiler
            double sn, cs;
            ::sincos(2 * PI_ * static_cast<double>(i) / static_cast<double>(degree), &sn, &cs);
            roots[i] = std::complex<double>(1.0 * cs, 1.0 * sn);
        }
    }
    std::complex<double> get_root(size_t index) const
    {
        index &= degree - 1;
        if (index <= degree / 8)
        {
            return roots[index];
        }
        else if (index <= degree / 4)
        {
            std::complex<double> a = roots[degree / 4 - index];
            return { a.imag(), a.real() };
        }
        else if (index <= degree / 2)
        {
            return -std::conj(get_root(degree / 2 - index));
        }
        else if (index <= 3 * degree / 4)
        {
            return -get_root(index - degree / 2);
        }
        return std::conj(get_root(degree - index));
    }
};
} // namespace

static int ensure_tables(moai_ctx *c)
{
    std::lock_guard<std::mutex> g(*static_cast<std::mutex *>(c->mutex));
    if (c->ckks_inv_roots)
    {
        return MOAI_OK;
    }
    const size_t n = c->n;
    const size_t slots = n >> 1;
    const int logn = c->logn;
    const uint64_t m = (uint64_t)n << 1;
    c->ckks_index_map.assign(n, 0);
    uint64_t gen = 5, pos = 1;
    for (size_t i = 0; i < slots; i++)
    {
        uint64_t index1 = (pos - 1) >> 1;
        uint64_t index2 = (m - pos - 1) >> 1;
        c->ckks_index_map[i] = reverse_bits((uint32_t)index1, logn);
        c->ckks_index_map[slots | i] = reverse_bits((uint32_t)index2, logn);
        pos *= gen;
        pos &= (m - 1);
    }
    c->ckks_inv_roots_host.assign(2 * n, 0.0);
    if (m >= 8)
    {
        ComplexRoots cr((size_t)m);
        for (size_t i = 1; i < n; i++)
        {
            std::complex<double> z = std::conj(cr.get_root((size_t)reverse_bits((uint32_t)(i - 1), logn) + 1));
            c->ckks_inv_roots_host[2 * i] = z.real();
            c->ckks_inv_roots_host[2 * i + 1] = z.imag();
        }
    }
    else
    {
        c->ckks_inv_roots_host[2] = 0;
        c->ckks_inv_roots_host[3] = -1;
    }
    std::vector<uint32_t> src(n);
    for (size_t i = 0; i < n; i++)
    {
        src[c->ckks_index_map[i]] = (uint32_t)i;
    }
    uint32_t *d_src = nullptr;
    double *d_roots = nullptr;
    MOAI_HIP_CHECK(hipMalloc(&d_src, sizeof(uint32_t) * n));
    MOAI_HIP_CHECK(hipMalloc(&d_roots, sizeof(double) * 2 * n));
    MOAI_HIP_CHECK(hipMemcpy(d_src, src.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    MOAI_HIP_CHECK(hipMemcpy(d_roots, c->ckks_inv_roots_host.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    c->ckks_src_map = d_src;
    c->ckks_inv_roots = d_roots;
    return MOAI_OK;
}

// significant bits of the product of the selected primes (schoolbook multi-word product)
static int product_bit_count(const moai_ctx *c, size_t L, const uint32_t *prime_index)
{
    std::vector<uint64_t> acc(1, 1);
    for (size_t j = 0; j < L; j++)
    {
        size_t idx = prime_index ? prime_index[j] : j;
        if (idx >= c->k)
        {
            return 0;
        }
        const uint64_t q = c->primes[idx];
        unsigned __int128 carry = 0;
        for (size_t w = 0; w < acc.size(); w++)
        {
            unsigned __int128 t = (unsigned __int128)acc[w] * q + carry;
            acc[w] = (uint64_t)t;
            carry = t >> 64;
        }
        if (carry)
        {
            acc.push_back((uint64_t)carry);
        }
    }
    int bits = 64 * (int)(acc.size() - 1);
    for (uint64_t top = acc.back(); top; top >>= 1)
    {
        bits++;
    }
    return bits;
}

template <int R>
static void launch_finish(const EncArgs &g, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL(ckks_fft_finish<R>, grid, dim3(256), 0, s, g);
}

} // namespace moai

using namespace moai;

extern "C" int moai_total_coeff_modulus_bit_count(const moai_ctx *c, size_t L, const uint32_t *prime_index)
{
    if (!c || L == 0 || L > c->k)
    {
        set_error(MOAI_EINVAL, "invalid level");
        return 0;
    }
    return product_bit_count(c, L, prime_index);
}

extern "C" int moai_ckks_tables(moai_ctx *c, uint32_t *index_map, double *inv_root_powers)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    int rc = ensure_tables(c);
    if (rc)
    {
        return rc;
    }
    if (index_map)
    {
        std::copy(c->ckks_index_map.begin(), c->ckks_index_map.end(), index_map);
    }
    if (inv_root_powers)
    {
        std::copy(c->ckks_inv_roots_host.begin(), c->ckks_inv_roots_host.end(), inv_root_powers);
    }
    return MOAI_OK;
}

static int encode_impl(moai_ctx *c, const double *values, const int32_t *mask, int is_complex, size_t values_size,
                       size_t n_batch, uint64_t *dst, size_t L, const uint32_t *prime_index, double scale,
                       double *max_coeff, void *stream)
{
    if (!c)
    {
        return set_error(MOAI_EINVAL, "null context");
    }
    if (L == 0 || L > c->k || L > MOAI_MAX_RNS)
    {
        return set_error(MOAI_EINVAL, "invalid level");
    }
    if (c->logn < 3 || c->logn > ENC_TILE_LOG + 4)
    {
        return set_error(MOAI_ELOGIC, "encoder supports 8 <= N <= 2^16");
    }
    if (values_size > (c->n >> 1))
    {
        return set_error(MOAI_EINVAL, "values_size is too large");
    }
    if (!values && values_size > 0)
    {
        return set_error(MOAI_EINVAL, "values cannot be null");
    }
    RowMap rows;
    int rc = make_rowmap(c, L, prime_index, &rows);
    if (rc)
    {
        return rc;
    }
    // ckks.h:493-497
    const int total_bits = product_bit_count(c, L, prime_index);
    if (scale <= 0 || (static_cast<int>(std::log2(scale)) + 1 >= total_bits))
    {
        return set_error(MOAI_EINVAL, "scale out of bounds");
    }
    if (n_batch == 0)
    {
        return MOAI_OK;
    }
    if (n_batch > 65535)
    {
        return set_error(MOAI_EINVAL, "at most 65535 vectors per call");
    }
    if (!dst)
    {
        return set_error(MOAI_EINVAL, "null argument");
    }
    rc = enter_device(c);
    if (!rc)
    {
        rc = ensure_tables(c);
    }
    if (rc)
    {
        return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> op(*static_cast<std::mutex *>(c->op_mutex));
    void *scratch = nullptr;
    rc = workspace(c, n_batch * c->n * sizeof(double2), s, &scratch);
    if (rc)
    {
        return rc;
    }
    if (max_coeff)
    {
        MOAI_HIP_CHECK(hipMemsetAsync(max_coeff, 0, sizeof(double) * n_batch, s));
    }
    EncArgs g;
    g.values = values;
    g.mask = mask;
    g.src_map = c->ckks_src_map;
    g.roots = reinterpret_cast<const double2 *>(c->ckks_inv_roots);
    g.scratch = static_cast<double2 *>(scratch);
    g.dst = dst;
    g.max_bits = reinterpret_cast<unsigned long long *>(max_coeff);
    g.pc = c->pc;
    g.rows = rows;
    g.L = (uint32_t)L;
    g.logn = (uint32_t)c->logn;
    g.values_size = (uint32_t)values_size;
    g.is_complex = is_complex ? 1u : 0u;
    g.fix = scale / static_cast<double>(c->n);

    const uint32_t tile = c->n < ENC_TILE ? (uint32_t)c->n : ENC_TILE;
    const uint32_t tiles = (uint32_t)(c->n / tile);
    if (c->logn >= ENC_TILE_LOG)
    {
        hipLaunchKernelGGL(ckks_fft_contig16, dim3((uint32_t)(n_batch * tiles)), dim3(256), 0, s, g);
    }
    else
    {
        hipLaunchKernelGGL(ckks_fft_contig, dim3((uint32_t)(n_batch * tiles)), dim3(256), 0, s, g);
    }
    MOAI_LAUNCH_CHECK();
    const int R = c->logn > ENC_TILE_LOG ? c->logn - ENC_TILE_LOG : 0;
    dim3 grid((tile + 255) / 256, (uint32_t)((L + ENC_ROWS_PER_BLOCK - 1) / ENC_ROWS_PER_BLOCK), (uint32_t)n_batch);
    switch (R)
    {
    case 0: launch_finish<0>(g, grid, s); break;
    case 1: launch_finish<1>(g, grid, s); break;
    case 2: launch_finish<2>(g, grid, s); break;
    case 3: launch_finish<3>(g, grid, s); break;
    default: launch_finish<4>(g, grid, s); break;
    }
    MOAI_LAUNCH_CHECK();
    // ckks.h:631-634: ntt_negacyclic_harvey on every row
    return ntt_launch(c, dst, n_batch, L, rows, false, s);
}

extern "C" int moai_ckks_encode(moai_ctx *c, const double *values, int is_complex, size_t values_size,
                                size_t n_batch, uint64_t *dst, size_t L, const uint32_t *prime_index, double scale,
                                double *max_coeff, void *stream)
{
    MOAI_AUDIT(stream, dst, values, max_coeff);
    trace_op("ckks_encode", L, n_batch);
    return encode_impl(c, values, nullptr, is_complex, values_size, n_batch, dst, L, prime_index, scale, max_coeff, stream);
}

extern "C" int moai_ckks_encode_masked(moai_ctx *c, const double *constants, const int32_t *mask, size_t mask_size,
                                       size_t n_batch, uint64_t *dst, size_t L, const uint32_t *prime_index,
                                       double scale, double *max_coeff, void *stream)
{
    MOAI_AUDIT(stream, dst, constants, mask, max_coeff);
    trace_op("ckks_encode_masked", L, n_batch);
    if (!mask && mask_size > 0)
    {
        return set_error(MOAI_EINVAL, "mask cannot be null");
    }
    return encode_impl(c, constants, mask, 0, mask_size, n_batch, dst, L, prime_index, scale, max_coeff, stream);
}
