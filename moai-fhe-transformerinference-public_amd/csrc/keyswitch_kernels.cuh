// keyswitch_kernels.cuh -- the key-switch inner product with the NTT passes fused around it.
//
// Reference loop (SEAL/evaluator.cpp:2817-2911), for every output modulus I and digit J:
//     operand = NTT_{q_I}( t_J mod q_I ) ;  acc_{K,I} += operand (*) key[J][K][I]
// On the device the "mod q_I" is applied while the strided pass loads the digit (ks_fwd_strided), and
// the contiguous pass never writes the transformed digit back: one workgroup owns a 4096-coefficient
// tile of the output (modulus I, ciphertext b), loops over the L digits, finishes each digit's last 8
// stages in registers and multiplies it straight into 128-bit accumulators for both key components
// (ks_contig_mac).  Per (I, J) coefficient this moves 8 B (read t) + 16 B (intermediate) + 16 B (key,
// shared by the batch through L2) instead of 56 B + key for the unfused sequence
// reduce -> NTT -> NTT -> MAC.
#pragma once
#include "ntt_kernels.cuh"

namespace moai {

struct KsGroup
{
    uint16_t prime[MOAI_MAX_RNS]; // context prime of group member g
    uint16_t slot[MOAI_MAX_RNS];  // row of acc it produces (I, or L for the special prime)
};

struct KsP1Args
{
    const uint64_t *t;  // [B][L][N] digits in coefficient form
    uint64_t *tmp;      // [B][G][L][N] after the strided pass, lazy [0,4q)
    const Tw *tw;
    const PrimeConst *pc;
    KsGroup grp;
    uint32_t L;
    uint32_t G;
    uint32_t total_work;
};

// work id -> (tile fastest, then group member, digit, ciphertext): neighbours read the same digit tile
// MODE 0: reference discipline (guard per butterfly, digits normalised with two conditional subtracts)
// MODE 1: every prime below 2^64/36 and 36 q^2 L < 2^128 (MOAI's chain): no guards, and the
//         unreduced digit (< 33q) goes straight into the 128-bit MAC
template <int LOGN, int MODE>
__global__ __launch_bounds__(256, 4) void ks_fwd_strided(KsP1Args a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ uint64_t lds[4096];
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    w /= TPR;
    const uint32_t g = w % a.G;
    w /= a.G;
    const uint32_t J = w % a.L;
    const uint32_t b = w / a.L;
    const uint32_t prime = a.grp.prime[g];
    const PrimeConst *pc = a.pc + prime;
    LoadBarrett op;
    op.q = pc->q;
    op.cr1 = pc->cr1;
    const uint64_t *in = a.t + (((size_t)b * a.L + J) << LOGN);
    uint64_t *out = a.tmp + ((((size_t)b * a.G + g) * a.L + J) << LOGN);
    fwd_strided_tile<LOGN, LoadBarrett, (MODE != 0)>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->q, pc->q2, lds,
                                                     threadIdx.x, op);
}

struct KsP2Args
{
    const uint64_t *tmp; // [B][G][L][N]
    const uint64_t *key; // [k-1][2][k][N]
    uint64_t *acc;       // [B][2][L+1][N]
    const Tw *tw;
    const PrimeConst *pc;
    KsGroup grp;
    uint32_t L;
    uint32_t G;
    uint32_t k;
    uint32_t B;          // ciphertexts in the batch
    uint32_t S;          // digit range split: split s sums digits [s*jchunk, (s+1)*jchunk) into its own acc copy
    uint32_t jchunk;
    size_t split_stride; // words between the acc copies of consecutive splits
    uint32_t total_work;
};

__device__ __forceinline__ void mac128r(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b)
{
    uint64_t pl = a * b;
    uint64_t ph = mulhi64(a, b);
    lo += pl;
    hi += ph + (lo < pl ? 1 : 0);
}

template <int LOGN, int MODE>
__global__ __launch_bounds__(256, 2) void ks_contig_mac(KsP2Args a)
{
    constexpr int R1 = LOGN - 8;
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048];
    uint64_t *lds = reinterpret_cast<uint64_t *>(lds2);

    // ciphertext index fastest: the workgroups that run side by side on one XCD walk the same key tiles
    // (modulus I, tile, J = 0..L-1), so each key line is fetched from HBM once per XCD, not once per ciphertext
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t bq = w % a.B;
    w /= a.B;
    const uint32_t tile = w % TPR;
    w /= TPR;
    const uint32_t g = w % a.G;
    const uint32_t split = w / a.G;
    const uint32_t j0 = split * a.jchunk;
    const uint32_t j1 = (j0 + a.jchunk < a.L) ? j0 + a.jchunk : a.L;
    const uint32_t prime = a.grp.prime[g];
    const uint32_t slot = a.grp.slot[g];
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q, q2 = pc->q2;
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << LOGN);
    const uint32_t tid = threadIdx.x;
    const uint32_t b = tid >> 4;
    const uint32_t tl = tid & 15u;
    const uint32_t blk = (tile << 4) + b;
    const uint32_t myrow = tid;

    uint64_t lo0[16], hi0[16], lo1[16], hi1[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
    {
        lo0[e] = hi0[e] = lo1[e] = hi1[e] = 0;
    }

    // software pipeline: digit J+1 is loaded into the (then dead) coefficient registers while digit J is
    // being multiplied into the accumulators
    const uint64_t *__restrict__ dig = a.tmp + ((((size_t)bq * a.G + g) * a.L) << LOGN) + ((size_t)tile << 12);
    uint64_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        x[j] = dig[((size_t)j0 << LOGN) + ((b << 8) | ((uint32_t)j << 4) | tl)];
    }
    for (uint32_t J = j0; J < j1; ++J)
    {
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            const int half = 8 >> u;
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                if (!(j & half))
                {
                    Tw t = tw[(1u << (R1 + u)) + (blk << u) + (uint32_t)(j >> (4 - u))];
                    ct_bfly_t<(MODE != 0)>(x[j], x[j + half], t.w, t.wq, q, q2);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            lds[phys_contig((b << 8) | ((uint32_t)j << 4) | tl)] = x[j];
        }
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 8; ++c)
        {
            ulonglong2 v = lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))];
            x[2 * c] = v.x;
            x[2 * c + 1] = v.y;
        }
#pragma unroll
        for (int u = 4; u < 8; ++u)
        {
            const int half = 8 >> (u - 4);
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                if (!(j & half))
                {
                    // the plain table here: the same 15 entries per thread are re-read for every digit J, and a
                    // thread's 8 last-stage entries share one 128-byte line (measured faster than the
                    // per-thread-ordered copy the NTT kernels use)
                    uint32_t t_ = (tl << 4) | (uint32_t)j;
                    Tw t = tw[(1u << (R1 + u)) + (blk << u) + (t_ >> (8 - u))];
                    ct_bfly_t<(MODE != 0)>(x[j], x[j + half], t.w, t.wq, q, q2);
                }
            }
        }
        // rows are private to their thread: write the canonical values back into the same slots
#pragma unroll
        for (int c = 0; c < 8; ++c)
        {
            ulonglong2 v;
            if (MODE == 0)
            {
                v.x = csub(csub(x[2 * c], q2), q);
                v.y = csub(csub(x[2 * c + 1], q2), q);
            }
            else
            {
                v.x = x[2 * c];
                v.y = x[2 * c + 1];
            }
            lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))] = v;
        }
        if (J + 1 < j1)
        {
            const uint64_t *__restrict__ nxt = dig + ((size_t)(J + 1) << LOGN);
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                x[j] = nxt[(b << 8) | ((uint32_t)j << 4) | tl];
            }
        }
        lds_barrier();
        // coalesced view of the tile: chunk ch = it * 256 + tid; multiply into both key components
        const ulonglong2 *__restrict__ k0 =
            reinterpret_cast<const ulonglong2 *>(a.key + (((size_t)(J * 2 + 0) * a.k + prime) << LOGN)) + ((size_t)tile << 11);
        const ulonglong2 *__restrict__ k1 =
            reinterpret_cast<const ulonglong2 *>(a.key + (((size_t)(J * 2 + 1) * a.k + prime) << LOGN)) + ((size_t)tile << 11);
#pragma unroll
        for (int it = 0; it < 8; ++it)
        {
            uint32_t ch = (uint32_t)it * 256u + tid;
            uint32_t rr = ch >> 3;
            ulonglong2 v = lds2[(rr << 3) | ((ch & 7u) ^ (rr & 7u))];
            ulonglong2 ka = k0[ch];
            ulonglong2 kb = k1[ch];
            mac128r(lo0[2 * it], hi0[2 * it], v.x, ka.x);
            mac128r(lo0[2 * it + 1], hi0[2 * it + 1], v.y, ka.y);
            mac128r(lo1[2 * it], hi1[2 * it], v.x, kb.x);
            mac128r(lo1[2 * it + 1], hi1[2 * it + 1], v.y, kb.y);
        }
        lds_barrier(); // the tile is dead: the next digit may overwrite it
    }
    const uint64_t cr0 = pc->cr0, cr1 = pc->cr1;
    ulonglong2 *__restrict__ o0 =
        reinterpret_cast<ulonglong2 *>(a.acc + split * a.split_stride + ((((size_t)bq * 2 + 0) * (a.L + 1) + slot) << LOGN)) +
        ((size_t)tile << 11);
    ulonglong2 *__restrict__ o1 =
        reinterpret_cast<ulonglong2 *>(a.acc + split * a.split_stride + ((((size_t)bq * 2 + 1) * (a.L + 1) + slot) << LOGN)) +
        ((size_t)tile << 11);
#pragma unroll
    for (int it = 0; it < 8; ++it)
    {
        uint32_t ch = (uint32_t)it * 256u + tid;
        ulonglong2 r0, r1;
        r0.x = barrett128(lo0[2 * it], hi0[2 * it], q, cr0, cr1);
        r0.y = barrett128(lo0[2 * it + 1], hi0[2 * it + 1], q, cr0, cr1);
        r1.x = barrett128(lo1[2 * it], hi1[2 * it], q, cr0, cr1);
        r1.y = barrett128(lo1[2 * it + 1], hi1[2 * it + 1], q, cr0, cr1);
        o0[ch] = r0;
        o1[ch] = r1;
    }
}

// ---- "divide by the last modulus and round" with its element-wise halves fused into the NTT ---------
// (RNSTool::divide_and_round_q_last_ntt_inplace SEAL/util/rns.cpp:830-901; key-switch mod-down
// SEAL/evaluator.cpp:2913-3018).  `last` [P][N] is the dropped row in coefficient form.
//   strided pass load:   u = ([last + q_last/2] mod q_last) mod q_i + (q_i - (q_last/2 mod q_i))
//   contiguous pass end: out_i = (acc_i - NTT(u)_i) * q_last^-1 mod q_i   (+ out_i when accumulating)
struct LoadExpandLast
{
    uint64_t ql, half, q, cr1, fix;
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return barrett64(csub(v + half, ql), q, cr1) + fix;
    }
};

struct StoreModDown
{
    const ulonglong2 *acc; // tile base of the row being divided (first of `splits` partial copies)
    ulonglong2 *out;       // tile base of the result row
    uint64_t q;
    Tw inv;
    int accumulate;
    uint32_t splits;       // partial sums to add up (key switch on few ciphertexts), 1 otherwise
    size_t split_stride;   // 16-byte chunks between consecutive partial copies
    __device__ __forceinline__ void operator()(uint32_t ch, ulonglong2 u) const
    {
        ulonglong2 x = acc[ch], r;
        for (uint32_t sp = 1; sp < splits; ++sp)
        {
            ulonglong2 y = acc[ch + sp * split_stride];
            x.x = csub(x.x + y.x, q);
            x.y = csub(x.y + y.y, q);
        }
        r.x = csub(mul_shoup_lazy(x.x + q - u.x, inv.w, inv.wq, q), q);
        r.y = csub(mul_shoup_lazy(x.y + q - u.y, inv.w, inv.wq, q), q);
        if (accumulate)
        {
            ulonglong2 c = out[ch];
            r.x = csub(r.x + c.x, q);
            r.y = csub(r.y + c.y, q);
        }
        out[ch] = r;
    }
};

struct ModDownArgs
{
    const uint64_t *last; // [P][N] coefficient form, canonical under prime_last
    uint64_t *u;          // [P][Lout][N] scratch between the two passes
    const uint64_t *acc;  // row (p, i) at acc + (p * acc_stride + i) * N
    uint64_t *out;        // [P][Lout][N]
    const Tw *tw;
    const Tw *twb;
    const PrimeConst *pc;
    const Tw *inv_last;   // q_last^-1 mod q_i
    uint32_t prime_last;
    uint32_t acc_stride;
    uint32_t Lout;
    uint32_t P;
    int accumulate;
    uint32_t acc_splits;     // >= 1
    size_t acc_split_stride; // words between partial copies of acc
    uint32_t total_work;
};

template <int LOGN, bool NOGUARD>
__global__ __launch_bounds__(256, 4) void moddown_strided(ModDownArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ uint64_t lds[4096];
    // tile fastest, then prime, then polynomial: neighbours read the same `last` row
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    w /= TPR;
    const uint32_t i = w % a.Lout;
    const uint32_t p = w / a.Lout;
    const PrimeConst *pc = a.pc + i;
    LoadExpandLast op;
    op.ql = a.pc[a.prime_last].q;
    op.half = op.ql >> 1;
    op.q = pc->q;
    op.cr1 = pc->cr1;
    op.fix = pc->q - barrett64(op.half, pc->q, pc->cr1);
    fwd_strided_tile<LOGN, LoadExpandLast, NOGUARD>(a.last + ((size_t)p << LOGN), a.u + (((size_t)p * a.Lout + i) << LOGN), tile,
                                                    a.tw + ((size_t)i << LOGN), pc->q, pc->q2, lds, threadIdx.x, op);
}

template <int LOGN, bool NOGUARD>
__global__ __launch_bounds__(256) void moddown_contig(ModDownArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048];
    // polynomial fastest: the workgroups that share a twiddle slice run together
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t p = w % a.P;
    w /= a.P;
    const uint32_t tile = w % TPR;
    const uint32_t i = w / TPR;
    const PrimeConst *pc = a.pc + i;
    StoreModDown st;
    st.acc = reinterpret_cast<const ulonglong2 *>(a.acc + (((size_t)p * a.acc_stride + i) << LOGN)) + ((size_t)tile << 11);
    st.out = reinterpret_cast<ulonglong2 *>(a.out + (((size_t)p * a.Lout + i) << LOGN)) + ((size_t)tile << 11);
    st.q = pc->q;
    st.inv = a.inv_last[i];
    st.accumulate = a.accumulate;
    st.splits = a.acc_splits;
    st.split_stride = a.acc_split_stride >> 1;
    fwd_contig_tile<LOGN, NOGUARD, StoreModDown>(a.u + (((size_t)p * a.Lout + i) << LOGN), tile, a.tw + ((size_t)i << LOGN), pc->q,
                                                 pc->q2, lds2, threadIdx.x, a.twb + (size_t)i * ((size_t)TPR * 15 * 256),
                                                 pc->cr1, st);
}

} // namespace moai
