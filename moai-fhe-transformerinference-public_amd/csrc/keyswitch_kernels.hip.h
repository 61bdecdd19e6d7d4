// keyswitch_kernels.hip.h -- the key-switch inner product with the NTT passes fused around it.
//
// Reference loop (SEAL/evaluator.cpp:2817-2911), for every output modulus I and digit J:
//     operand = NTT_{q_I}( t_J mod q_I ) ;  acc_{K,I} += operand (*) key[J][K][I]
// On the device the "mod q_I" is applied while the strided pass loads the digit (ks_fwd_strided), and
// the contiguous pass never writes the transformed digit back: one workgroup owns a 4096-coefficient
// tile of the output (modulus I, ciphertext b), loops over the L digits, finishes each digit's last 8
// stages in registers and multiplies it straight into 128-bit accumulators for both key components
// (ks_contig_mac8).  Per (I, J) coefficient this moves 8 B (read t) + 16 B (intermediate) + 16 B (key,
// shared by the batch through L2) instead of 56 B + key for the unfused sequence
// reduce -> NTT -> NTT -> MAC.
#pragma once
#include "ntt_kernels.hip.h"

namespace moai {

struct KsGroup
{
    uint32_t prime[MOAI_MAX_RNS]; // context prime of group member g (32-bit entries: scalar loads, see RowMap)
    uint32_t slot[MOAI_MAX_RNS];  // row of acc it produces (I, or L for the special prime)
};

struct KsP1Args
{
    const uint64_t *t;  // [B][L][N] digits in coefficient form
    uint64_t *tmp;      // [B][G][L][N] after the strided pass, lazy [0,4q)
    const Tw *tw;
    const double *tw1;  // FP64 modes: the forward powers as plain doubles (PRE variant), or nullptr
    const PrimeConst *pc;
    KsGroup grp;
    uint32_t L;
    uint32_t G;
    uint32_t total_work;
};

// work id -> (tile fastest, then group member, digit, ciphertext): neighbours read the same digit tile
// MODE (modarith.hip.h M_*), one per launch: the host groups the output moduli by the arithmetic they allow
//   M_GUARD    reference discipline (guard per butterfly, digits normalised with two conditional subtracts)
//   M_NOGUARD  prime below 2^64/36 and 36 q^2 L < 2^128: no guards, and the unreduced digit (< 33q) goes
//              straight into the 128-bit MAC
//   M_FPN/FPR  prime below 2^51: FP64 butterflies (tw = the FP64 table), canonical integer into the MAC
// ITEMS > 1 (FP64 modes): a workgroup takes ITEMS consecutive tiles of its row, software-pipelined (fwd_strided_tiles); the grid
// is total_work / ITEMS
template <int LOGN, int MODE, bool PRE = false, int ITEMS = 1>
__global__ __launch_bounds__(256, 4) void ks_fwd_strided(KsP1Args a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    static_assert(ITEMS >= 1 && TPR % ITEMS == 0, "a workgroup's tiles belong to one row");
    __shared__ uint64_t lds[4096 + (ITEMS > 1 ? 512 : 0)]; // the exchange buffer; pipelined: + phase B's twiddles
    uint32_t w = xcd_remap(blockIdx.x, a.total_work / ITEMS);
    const uint32_t tile = (w % (TPR / ITEMS)) * ITEMS;
    w /= TPR / ITEMS;
    const uint32_t g = w % a.G;
    w /= a.G;
    const uint32_t J = w % a.L;
    const uint32_t b = w / a.L;
    // 16-bit kernarg entries arrive through a vector load: pin them back to scalars, or the twiddle pointer lives in VGPRs and
    // the uniform twiddles of phase A come through vector loads (whose waits would also cover the prefetch of the next tile)
    const uint32_t prime = __builtin_amdgcn_readfirstlane(a.grp.prime[g]);
    if ((uint32_t)__builtin_amdgcn_readfirstlane(a.grp.slot[g]) == J)
    {
        // output modulus = the digit's own prime: the transform of the reduced digit is the target's NTT-form row
        // itself (the reference copies it, SEAL/evaluator.cpp:2836-2839); ks_contig_mac8 reads it from there
        return;
    }
    const PrimeConst *pc = a.pc + prime;
    const uint64_t *in = a.t + (((size_t)b * a.L + J) << LOGN);
    uint64_t *out = a.tmp + ((((size_t)b * a.G + g) * a.L + J) << LOGN);
    if (MODE >= M_FPN)
    {
        // FP64 modes: the conversion to a double reduces modulo q_I on the way; a digit of 53 bits and more
        // (its own prime is an integer-mode one) takes an integer Barrett step first.  Workgroup-uniform.
        if (a.pc[J].q >> 52)
        {
            LoadBarrettFp op;
            op.q = pc->q;
            op.cr1 = pc->cr1;
            op.qd = pc->qd;
            op.qinv = pc->qinv;
            if constexpr (ITEMS > 1)
            {
                fwd_strided_tiles<LOGN, LoadBarrettFp, MODE, ITEMS>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->qd, pc->qinv, lds, threadIdx.x, op);
            }
            else
            {
                fwd_strided_tile<LOGN, LoadBarrettFp, MODE, PRE>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->qd, pc->qinv, lds, threadIdx.x, op,
                                                                 PRE ? a.tw1 + ((size_t)prime << LOGN) : nullptr);
            }
        }
        else
        {
            LoadFp52 op;
            op.qd = pc->qd;
            op.qinv = pc->qinv;
            if constexpr (ITEMS > 1)
            {
                fwd_strided_tiles<LOGN, LoadFp52, MODE, ITEMS>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->qd, pc->qinv, lds, threadIdx.x, op);
            }
            else
            {
                fwd_strided_tile<LOGN, LoadFp52, MODE, PRE>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->qd, pc->qinv, lds, threadIdx.x, op,
                                                            PRE ? a.tw1 + ((size_t)prime << LOGN) : nullptr);
            }
        }
    }
    // digit J is canonical under prime J: it needs reducing only when that prime is the larger one
    // (SEAL/evaluator.cpp:2846-2854); the branch is workgroup-uniform
    else if (a.pc[J].q > pc->q)
    {
        LoadBarrett op;
        op.q = pc->q;
        op.cr1 = pc->cr1;
        fwd_strided_tile<LOGN, LoadBarrett, MODE>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->q, pc->q2, lds, threadIdx.x, op);
    }
    else
    {
        fwd_strided_tile<LOGN, LoadIdentity, MODE>(in, out, tile, a.tw + ((size_t)prime << LOGN), pc->q, pc->q2, lds, threadIdx.x);
    }
}

struct KsP2Args
{
    const uint64_t *tmp; // [B][G][L][N]
    const uint64_t *tgt; // the target in NTT form: row (b, J) at tgt + ((b * tgt_stride + tgt_off + J) << LOGN)
    uint32_t tgt_stride, tgt_off;
    const uint64_t *key; // [digits][2][k][N]: k rows per key polynomial, the special prime's row LAST (the reference's
                         // layout has k = all primes, SEAL/kswitchkeys.h:340; a level-trimmed key fewer, moai_key_trim)
    uint64_t *acc;       // [B][2][L+1][N]
    const Tw *tw;
    const double *tw1;   // FP64 modes: the forward powers as plain doubles
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

// ---- contiguous pass + key MAC, 8 coefficients per thread --------------------------------------------------------
// With the NTT kernels' 16 coefficients per thread the 2 x 16 128-bit accumulators need 250 VGPRs (two waves
// per SIMD; measured 6-12 % slower).  Here a workgroup owns a 2048-coefficient tile (8 blocks of 256): radix-8 /
// radix-8 / radix-4 with two LDS exchanges, and the MAC runs straight from the registers that finish the
// transform; 128 VGPRs, four waves per SIMD.
// layouts of the 8 registers of thread (b = tid >> 5, r = tid & 31), t = coefficient index inside block b:
//   P1  t = (j << 5) | r                         stages 0..2 (t bits 7..5)
//   P2  t = (r >> 2) << 5 | j << 2 | (r & 3)      stages 3..5 (t bits 4..2)
//   P3  t = r << 3 | j                            stages 6..7 (t bits 1..0); 8 contiguous coefficients
// LDS swizzles keep every exchange conflict free: exchange 1 stores t with bits 4..2 ^= bits 7..5, exchange 2
// stores t with bits 2..1 ^= bits 6..5.
__device__ __forceinline__ uint32_t phys8_a(uint32_t e)
{
    return e ^ (((e >> 5) & 7u) << 2);
}

__device__ __forceinline__ uint32_t phys8_b(uint32_t e)
{
    return e ^ (((e >> 5) & 3u) << 1);
}

// PF (FP64 modes): where the digit's sixteen key residues per thread (eight 16-byte loads) are issued.  0: next to the
// products that use them, two loads per quarter of the MAC with a wait behind each pair -- what the compiler makes of the
// straightforward loop (the periodic reduction's branch keeps it from moving them): four exposed L2 round trips per digit.
// 1: all eight at the head of the MAC, one wait.  2: all eight at the head of the iteration, next to the digit's own loads,
// so that they are in flight during the whole transform (32 more registers live across it).
// (Measured in round 3 and not kept: the digit's VALUES prefetched one digit ahead without registers -- each wave copying the next
// digit's 4 KiB into its own part of exchange buffer A with global_load_lds_dwordx4 as soon as it has read that buffer back, the
// twiddles of stages 0..2 in LDS, no global load at the head of an iteration: bit-identical, 0.496 against 0.502 ms per key switch
// at l = 35 and 0.120 against 0.118 at l = 15 -- noise.  The wait for the key residues at the head of the MAC also waits for the
// copy issued before them (vector memory operations complete in order), and the kernel's idle fifth is not this round trip.)
template <int LOGN, int MODE, int PF = 0>
__global__ __launch_bounds__(256, 4) void ks_contig_mac8(KsP2Args a)
{
    constexpr int R1 = LOGN - 8;
    constexpr uint32_t TPR8 = 1u << (LOGN - 11);
    __shared__ ulonglong2 lds_a2[1024];
    __shared__ ulonglong2 lds_b2[1024];
    // FP64 modes: the seven twiddles of stages 0..2 per 256-block ([block][8] doubles) -- the same for every digit of the loop
    // and for the block's 32 threads: fetched once per workgroup, read from LDS in every iteration instead of seven global loads
    constexpr bool TW012 = MODE >= M_FPN;
    __shared__ double lds_tw[TW012 ? 64 : 1];
    uint64_t *lds_a = reinterpret_cast<uint64_t *>(lds_a2);
    uint64_t *lds_b = reinterpret_cast<uint64_t *>(lds_b2);

    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t bq = w % a.B;
    w /= a.B;
    const uint32_t tile = w % TPR8;
    w /= TPR8;
    const uint32_t g = w % a.G;
    const uint32_t split = w / a.G;
    const uint32_t j0 = split * a.jchunk;
    const uint32_t j1 = (j0 + a.jchunk < a.L) ? j0 + a.jchunk : a.L;
    const uint32_t prime = a.grp.prime[g];
    const uint32_t slot = a.grp.slot[g];
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q;
    const uint64_t bq1 = mode_q<MODE>(*pc), bq2 = mode_q2<MODE>(*pc); // the butterflies' (q, q2) under MODE
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << LOGN);
    const double *__restrict__ tw1 = a.tw1 + ((size_t)prime << LOGN);
    const uint32_t tid0 = threadIdx.x;

    uint64_t lo0[8], hi0[8], lo1[8], hi1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
        lo0[e] = hi0[e] = lo1[e] = hi1[e] = 0;
    }
    const uint64_t *__restrict__ dig = a.tmp + ((((size_t)bq * a.G + g) * a.L) << LOGN) + ((size_t)tile << 11);

    // FP64 modes: the thirteen per-thread twiddles of stages 3..7 are the same for every digit of the loop (they
    // depend on the prime, the tile and the thread only) -- 26 registers instead of 13 loads behind the two LDS
    // barriers of every iteration
    double twr[13];
    if (MODE >= M_FPN)
    {
        const uint32_t b = tid0 >> 5, r = tid0 & 31u, hi3 = r >> 2;
        const uint32_t blk = (tile << 3) + b;
        twr[0] = tw1[(1u << (R1 + 3)) + (blk << 3) + hi3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            twr[1 + i] = tw1[(1u << (R1 + 4)) + (blk << 4) + ((hi3 << 1) | (uint32_t)i)];
            twr[7 + i] = tw1[(1u << (R1 + 6)) + (blk << 6) + ((r << 1) | (uint32_t)i)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            twr[3 + i] = tw1[(1u << (R1 + 5)) + (blk << 5) + ((hi3 << 2) | (uint32_t)i)];
            twr[9 + i] = tw1[(1u << (R1 + 7)) + (blk << 7) + ((r << 2) | (uint32_t)i)];
        }
    }
    if (TW012)
    {
        if (tid0 < 64u && (tid0 & 7u) < 7u)
        {
            const uint32_t b_ = tid0 >> 3, i = tid0 & 7u;
            const uint32_t u = i == 0 ? 0u : (i < 3 ? 1u : 2u);
            lds_tw[tid0] = tw1[(1u << (R1 + u)) + ((((uint32_t)tile << 3) + b_) << u) + (i - ((1u << u) - 1u))];
        }
        lds_barrier(); // written by wave 0, read by all four
    }
    for (uint32_t J = j0; J < j1; ++J)
    {
        // opaque copy of the thread index: keeps the per-thread address arithmetic inside the loop instead of
        // hoisted into (and spilled from) registers the accumulators need
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));
        const uint32_t b = tid >> 5;
        const uint32_t r = tid & 31u;
        const uint32_t hi3 = r >> 2, lo2 = r & 3u;
        const uint32_t blk = (tile << 3) + b;
        // chunk (16 bytes = 2 coefficients) of this thread's 8 contiguous coefficients inside the tile
        const uint32_t ch0 = (b << 7) | (r << 2);
        const uint64_t *__restrict__ base = dig + ((size_t)J << LOGN);
        uint64_t x[8];
        const bool direct = (slot == J); // workgroup-uniform
        const uint32_t krow = slot == a.L ? a.k - 1 : prime; // the special prime's row is the last one of the key's layout
        const ulonglong2 *__restrict__ k0 =
            reinterpret_cast<const ulonglong2 *>(a.key + (((size_t)(J * 2 + 0) * a.k + krow) << LOGN)) + ((size_t)tile << 10) + ch0;
        const ulonglong2 *__restrict__ k1 =
            reinterpret_cast<const ulonglong2 *>(a.key + (((size_t)(J * 2 + 1) * a.k + krow) << LOGN)) + ((size_t)tile << 10) + ch0;
        ulonglong2 kpa[4], kpb[4];
        if (MODE >= M_FPN && PF == 2)
        {
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                kpa[c] = k0[c];
                kpb[c] = k1[c];
            }
        }
        if (direct)
        {
            // digit under its own prime: canonical NTT-form values straight from the target row
            const ulonglong2 *__restrict__ trow =
                reinterpret_cast<const ulonglong2 *>(a.tgt + (((size_t)bq * a.tgt_stride + a.tgt_off + J) << LOGN)) + ((size_t)tile << 10) + ch0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                ulonglong2 v = trow[c];
                // FP64 modes: as doubles, like the values the butterflies leave (canonical below q: nothing to fold)
                x[2 * c] = MODE >= M_FPN ? d2u(fp_from_u52(v.x)) : v.x;
                x[2 * c + 1] = MODE >= M_FPN ? d2u(fp_from_u52(v.y)) : v.y;
            }
        }
        else
        {
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            x[j] = base[(b << 8) | ((uint32_t)j << 5) | r];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
        {
            const int half = 4 >> u;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if (!(j & half))
                {
                    if (MODE >= M_FPN)
                    {
                        ct_bfly_fp1_sel(x[j], x[j + half], lds_tw[(b << 3) + ((1u << u) - 1u) + (uint32_t)(j >> (3 - u))], u2d(bq1), u2d(bq2),
                                        MODE == M_FPR && !(u & 1));
                    }
                    else
                    {
                        Tw t = tw[(1u << (R1 + u)) + (blk << u) + (uint32_t)(j >> (3 - u))];
                        ct_bfly_t<MODE>(x[j], x[j + half], t.w, t.wq, bq1, bq2);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            lds_a[phys8_a((b << 8) | ((uint32_t)j << 5) | r)] = x[j];
        }
        lds_wave_sync(); // block b's 32 threads are half a wave: both exchanges stay inside one wave
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            x[j] = lds_a[phys8_a((b << 8) | (hi3 << 5) | ((uint32_t)j << 2) | lo2)];
        }
#pragma unroll
        for (int u = 3; u < 6; ++u)
        {
            const int half = 4 >> (u - 3);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if (!(j & half))
                {
                    uint32_t t_ = (hi3 << 5) | ((uint32_t)j << 2) | lo2;
                    if (MODE >= M_FPN)
                    {
                        ct_bfly_fp1_sel(x[j], x[j + half], twr[(u == 3) ? 0 : (u == 4) ? 1 + (j >> 2) : 3 + (j >> 1)], u2d(bq1), u2d(bq2), MODE == M_FPR && !(u & 1));
                    }
                    else
                    {
                        Tw t = tw[(1u << (R1 + u)) + (blk << u) + (t_ >> (8 - u))];
                        ct_bfly_t<MODE>(x[j], x[j + half], t.w, t.wq, bq1, bq2);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            lds_b[phys8_b((b << 8) | (hi3 << 5) | ((uint32_t)j << 2) | lo2)] = x[j];
        }
        lds_wave_sync(); // block b's 32 threads are half a wave: both exchanges stay inside one wave
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            // coefficients (r << 3) | 2c, 2c+1: chunk c of the row, stored at chunk c ^ ((r >> 2) & 3)
            ulonglong2 v = lds_b2[((b << 8) | (r << 3)) / 2 + ((uint32_t)c ^ ((r >> 2) & 3u))];
            x[2 * c] = v.x;
            x[2 * c + 1] = v.y;
        }
#pragma unroll
        for (int u = 6; u < 8; ++u)
        {
            const int half = 2 >> (u - 6);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if (!(j & half))
                {
                    uint32_t t_ = (r << 3) | (uint32_t)j;
                    if (MODE >= M_FPN)
                    {
                        ct_bfly_fp1_sel(x[j], x[j + half], twr[(u == 6) ? 7 + (j >> 2) : 9 + (j >> 1)], u2d(bq1), u2d(bq2), MODE == M_FPR && !(u & 1));
                    }
                    else
                    {
                        Tw t = tw[(1u << (R1 + u)) + (blk << u) + (t_ >> (8 - u))];
                        ct_bfly_t<MODE>(x[j], x[j + half], t.w, t.wq, bq1, bq2);
                    }
                }
            }
        }
        } // !direct
        // FP64 modes: when the sixteen running sums are folded back to |.| <= q/2.  Every product is below 0.88 q (fp_mulmod_q with
        // a balanced digit), so M_FPR (2^53 / q > 4) may add three of them to a folded sum, M_FPN (q < 2^52 / 25) sixteen
        const bool fold = MODE == M_FPR ? ((J - j0) % 3u == 2u) : (((J - j0) & 15u) == 15u);
        if (MODE >= M_FPN && PF != 0)
        {
            // the same products and sums as below, in the same order per accumulator: the same bits.
            // The digit value enters the products as the butterflies left it in M_FPN (below 33q < 2^52: the quotient estimate of
            // fp_mulmod_q is then off by at most 2, the product below 2.5 q, h - c q = r - l still an integer below 2^50, and
            // sixteen of them on a folded sum stay below 40.5 q < 2^53 = 50 q at least) and folded to |v| <= q/2 in M_FPR
            // (2^53 is only 4q there).  A copied digit (canonical, below q) needs no folding in either mode.
            const double qd = u2d(bq1), qinv = u2d(bq2);
            if (PF == 1)
            {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    kpa[c] = k0[c];
                    kpb[c] = k1[c];
                }
            }
            double sm[16];
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const double vx = MODE == M_FPN ? u2d(x[2 * c]) : fp_red(u2d(x[2 * c]), qd, qinv);
                const double vy = MODE == M_FPN ? u2d(x[2 * c + 1]) : fp_red(u2d(x[2 * c + 1]), qd, qinv);
                sm[4 * c + 0] = u2d(lo0[2 * c]) + fp_mulmod_q(vx, fp_from_u52(kpa[c].x), qd, qinv);
                sm[4 * c + 1] = u2d(lo0[2 * c + 1]) + fp_mulmod_q(vy, fp_from_u52(kpa[c].y), qd, qinv);
                sm[4 * c + 2] = u2d(lo1[2 * c]) + fp_mulmod_q(vx, fp_from_u52(kpb[c].x), qd, qinv);
                sm[4 * c + 3] = u2d(lo1[2 * c + 1]) + fp_mulmod_q(vy, fp_from_u52(kpb[c].y), qd, qinv);
            }
            if (fold) // wave-uniform: one branch around all sixteen reductions
            {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                {
                    sm[i] = fp_red(sm[i], qd, qinv);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                lo0[2 * c] = d2u(sm[4 * c + 0]);
                lo0[2 * c + 1] = d2u(sm[4 * c + 1]);
                lo1[2 * c] = d2u(sm[4 * c + 2]);
                lo1[2 * c + 1] = d2u(sm[4 * c + 3]);
            }
        }
        else if (MODE >= M_FPN)
        {
            // FP64 modes: the MAC stays on the FP64 pipe as well (fp_mulmod_q with the key residue canonical, below 2^51; the
            // digit folded to |v| <= q/2 first in M_FPR, as it comes in M_FPN -- bounds above), each product is reduced
            // exactly like a butterfly's, and the running sums stay below 2^53.
            const double qd = u2d(bq1), qinv = u2d(bq2);
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const double vx = MODE == M_FPN ? u2d(x[2 * c]) : fp_red(u2d(x[2 * c]), qd, qinv);
                const double vy = MODE == M_FPN ? u2d(x[2 * c + 1]) : fp_red(u2d(x[2 * c + 1]), qd, qinv);
                const ulonglong2 ka = k0[c];
                const ulonglong2 kb = k1[c];
                double s0 = u2d(lo0[2 * c]) + fp_mulmod_q(vx, fp_from_u52(ka.x), qd, qinv);
                double s1 = u2d(lo0[2 * c + 1]) + fp_mulmod_q(vy, fp_from_u52(ka.y), qd, qinv);
                double s2 = u2d(lo1[2 * c]) + fp_mulmod_q(vx, fp_from_u52(kb.x), qd, qinv);
                double s3 = u2d(lo1[2 * c + 1]) + fp_mulmod_q(vy, fp_from_u52(kb.y), qd, qinv);
                if (fold)
                {
                    s0 = fp_red(s0, qd, qinv);
                    s1 = fp_red(s1, qd, qinv);
                    s2 = fp_red(s2, qd, qinv);
                    s3 = fp_red(s3, qd, qinv);
                }
                lo0[2 * c] = d2u(s0);
                lo0[2 * c + 1] = d2u(s1);
                lo1[2 * c] = d2u(s2);
                lo1[2 * c + 1] = d2u(s3);
            }
        }
        else
        {
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                uint64_t vx = x[2 * c], vy = x[2 * c + 1];
                if (MODE == M_GUARD && !direct)
                {
                    vx = csub(csub(vx, bq2), q);
                    vy = csub(csub(vy, bq2), q);
                }
                ulonglong2 ka = k0[c];
                ulonglong2 kb = k1[c];
                mac128r(lo0[2 * c], hi0[2 * c], vx, ka.x);
                mac128r(lo0[2 * c + 1], hi0[2 * c + 1], vy, ka.y);
                mac128r(lo1[2 * c], hi1[2 * c], vx, kb.x);
                mac128r(lo1[2 * c + 1], hi1[2 * c + 1], vy, kb.y);
            }
        }
        // no barrier here: the next digit writes buffer A, whose readers all passed the second barrier above,
        // and buffer B is written again only after the next first barrier
    }
    const uint32_t ch0 = ((tid0 >> 5) << 7) | ((tid0 & 31u) << 2);
    const uint64_t cr0 = pc->cr0, cr1 = pc->cr1;
    ulonglong2 *__restrict__ o0 =
        reinterpret_cast<ulonglong2 *>(a.acc + split * a.split_stride + ((((size_t)bq * 2 + 0) * (a.L + 1) + slot) << LOGN)) +
        ((size_t)tile << 10) + ch0;
    ulonglong2 *__restrict__ o1 =
        reinterpret_cast<ulonglong2 *>(a.acc + split * a.split_stride + ((((size_t)bq * 2 + 1) * (a.L + 1) + slot) << LOGN)) +
        ((size_t)tile << 10) + ch0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        ulonglong2 r0, r1;
        if (MODE >= M_FPN)
        {
            const double qd = u2d(bq1), qinv = u2d(bq2);
            r0.x = fp_to_canonical(u2d(lo0[2 * c]), qd, qinv);
            r0.y = fp_to_canonical(u2d(lo0[2 * c + 1]), qd, qinv);
            r1.x = fp_to_canonical(u2d(lo1[2 * c]), qd, qinv);
            r1.y = fp_to_canonical(u2d(lo1[2 * c + 1]), qd, qinv);
        }
        else
        {
            r0.x = barrett128(lo0[2 * c], hi0[2 * c], q, cr0, cr1);
            r0.y = barrett128(lo0[2 * c + 1], hi0[2 * c + 1], q, cr0, cr1);
            r1.x = barrett128(lo1[2 * c], hi1[2 * c], q, cr0, cr1);
            r1.y = barrett128(lo1[2 * c + 1], hi1[2 * c + 1], q, cr0, cr1);
        }
        o0[c] = r0;
        o1[c] = r1;
    }
}

// ---- hoisted rotations: several Galois automorphisms of ONE ciphertext share its digit decomposition --------------
// rotate_vector(ct, step_r) for r = 1..R (the baby steps of MOAI's bootstrapping transforms, include/source/
// bootstrapping/Bootstrapper.cpp:2017-2022, 2082-2088) is, per rotation, apply_galois_ntt on both polynomials and
// switch_key_inplace on the permuted c1 (SEAL/evaluator.cpp:2631-2654, 2724-3020): l (l + 1) transforms each.  With
// sigma the automorphism, t = INTT(c1) and s(j) its sign at output coefficient j, the digit SEAL transforms is
//     [ sigma(t_J) ]_{q_I}(j) = [ t_J(pi j) ]_{q_I}                        where s(j) = +1
//                             = [ q_J - t_J(pi j) ]_{q_I} = -[t_J(pi j)]_{q_I} + (q_J mod q_I)   where s(j) = -1, t_J(pi j) != 0
// (also when q_J < q_I, where SEAL does not reduce: q_J - t = -t + q_J), i.e.
//     [sigma(t_J)]_{q_I} = sigma([t_J]_{q_I}) + delta_{J,I} m      (mod q_I),   m(j) = 1 where s(j) = -1,
// as long as no coefficient of t is zero.  NTT_I is linear and turns sigma into the index permutation
// apply_galois_ntt uses, so with D_{J,I} = NTT_I([t_J]_{q_I}) computed ONCE
//     acc_r[K][I] = sum_J perm_r(D_{J,I}) (*) key_r[J][K][I]  +  NTT_I(m_r) (*) sum_J delta_{J,I} key_r[J][K][I]
// equals the reference's sum modulo q_I, and the canonical results are the same bits.  The second term depends on the
// key and the level only ("correction", ks_hoist_correction_kernel).  A zero coefficient of t (probability 2^-46 each;
// certain for transparent or test ciphertexts) falls back to the per-rotation path (host side).
struct HoistCorrArgs
{
    const uint64_t *key;   // [k-1][2][k][N]
    const uint64_t *mask;  // [L+1][N]: NTT of the sign mask under slot i's prime
    uint64_t *out;         // [2][L+1][N]
    const PrimeConst *pc;
    uint32_t L, k, n2;
    uint32_t krows;        // rows per key polynomial in the key's layout (k for the reference's, fewer for a trimmed key)
};

// out[K][slot] = mask[slot] (*) sum_{J < L} (q_J mod q_slot) key[J][K][slot]      blockIdx.y = K * (L + 1) + slot
__global__ __launch_bounds__(256) void ks_hoist_correction_kernel(HoistCorrArgs g)
{
    const uint32_t K = blockIdx.y / (g.L + 1), slot = blockIdx.y % (g.L + 1);
    const uint32_t prime = slot == g.L ? g.k - 1 : slot;
    const PrimeConst *pc = g.pc + prime;
    const uint64_t q = pc->q, cr0 = pc->cr0, cr1 = pc->cr1;
    const ulonglong2 *mk = reinterpret_cast<const ulonglong2 *>(g.mask) + (size_t)slot * g.n2;
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + ((size_t)K * (g.L + 1) + slot) * g.n2;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < g.n2; j += gridDim.x * 256u)
    {
        uint64_t lx = 0, hx = 0, ly = 0, hy = 0;
        for (uint32_t J = 0; J < g.L; ++J)
        {
            if (J == slot)
            {
                continue; // q_J mod q_J = 0: the digit under its own prime is the target row itself
            }
            const uint64_t delta = barrett64(g.pc[J].q, q, cr1);
            const ulonglong2 kv =
                (reinterpret_cast<const ulonglong2 *>(g.key) + ((size_t)(J * 2 + K) * g.krows + (slot == g.L ? g.krows - 1 : slot)) * g.n2)[j];
            mac128r(lx, hx, kv.x, delta);
            mac128r(ly, hy, kv.y, delta);
        }
        const ulonglong2 m = mk[j];
        ulonglong2 r;
        r.x = mulmod_barrett(barrett128(lx, hx, q, cr0, cr1), m.x, q, cr0, cr1);
        r.y = mulmod_barrett(barrett128(ly, hy, q, cr0, cr1), m.y, q, cr0, cr1);
        o[j] = r;
    }
}

// m[j] = 1 where x -> x^elt negates the coefficient that lands on j (GaloisTool::apply_galois, SEAL/util/galois.cpp:133-190)
__global__ __launch_bounds__(256) void galois_sign_mask_kernel(uint64_t *m, uint32_t logn, uint32_t elt, uint32_t copies)
{
    const uint32_t n = 1u << logn;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n)
    {
        const uint32_t raw = (uint32_t)(((uint64_t)i * elt) & (2u * n - 1u));
        const uint32_t j = raw & (n - 1u);
        const uint64_t v = raw >> logn;
        for (uint32_t c = 0; c < copies; ++c)
        {
            m[(size_t)c * n + j] = v;
        }
    }
}

__global__ __launch_bounds__(256) void any_zero_kernel(const uint64_t *v, size_t words2, uint32_t *flag)
{
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(v);
    bool z = false;
    for (size_t j = (size_t)blockIdx.x * 256u + threadIdx.x; j < words2; j += (size_t)gridDim.x * 256u)
    {
        const ulonglong2 x = p[j];
        z = z || x.x == 0 || x.y == 0;
    }
    if (z)
    {
        atomicOr(flag, 1u);
    }
}

struct HoistMacArgs
{
    const uint64_t *dig;   // [B][G][L][N]: D_{J,I}, canonical, for the G output moduli of this launch
    const uint64_t *ct;    // the UNPERMUTED input [B][2][L][N]: its c1 row J is the digit under prime J
    const uint32_t *table; // NTT-domain index permutation of this rotation (galois_table)
    const uint64_t *key;   // this rotation's key [k-1][2][k][N]
    const uint64_t *corr;  // this rotation's correction [2][L+1][N]
    uint64_t *acc;         // this rotation's [B][2][L+1][N]
    const PrimeConst *pc;
    KsGroup grp;
    uint32_t L, G, k, B;
    uint32_t total_work;
};

// acc[b][K][slot] = sum_J perm(D[b][g][J]) (*) key[J][K][prime] + corr[K][slot]; one workgroup = 2048 outputs of one (b, g)
template <int LOGN, bool FP, bool FPR>
__global__ __launch_bounds__(256, 4) void ks_hoisted_mac(HoistMacArgs a)
{
    constexpr uint32_t TPR8 = 1u << (LOGN - 11);
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t bq = w % a.B;
    w /= a.B;
    const uint32_t tile = w % TPR8;
    const uint32_t g = w / TPR8;
    const uint32_t prime = a.grp.prime[g];
    const uint32_t slot = a.grp.slot[g];
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q;
    uint32_t src[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
        src[e] = a.table[(tile << 11) + ((uint32_t)e << 8) + threadIdx.x];
    }
    const size_t obase = ((size_t)tile << 11) + threadIdx.x;
    uint64_t lo0[8], hi0[8], lo1[8], hi1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
        lo0[e] = hi0[e] = lo1[e] = hi1[e] = 0;
    }
    const double qd = u2d(pc->qd), qinv = u2d(pc->qinv);
    for (uint32_t J = 0; J < a.L; ++J)
    {
        const uint64_t *__restrict__ row = (slot == J) ? a.ct + (((size_t)(bq * 2 + 1) * a.L + J) << LOGN)
                                                       : a.dig + ((((size_t)bq * a.G + g) * a.L + J) << LOGN);
        const uint32_t krow = slot == a.L ? a.k - 1 : prime; // a.k = rows per key polynomial in this key's layout
        const uint64_t *__restrict__ k0 = a.key + (((size_t)(J * 2 + 0) * a.k + krow) << LOGN) + obase;
        const uint64_t *__restrict__ k1 = a.key + (((size_t)(J * 2 + 1) * a.k + krow) << LOGN) + obase;
        // all twenty-four loads of the digit first (the compiler otherwise issues them three at a time with a full wait behind
        // each triple -- the periodic reduction's branch pins them: eight exposed round trips per digit), then the arithmetic,
        // then ONE wave-uniform branch around the reductions.  Same operations per accumulator in the same order: same bits.
        uint64_t xs[8], kas[8], kbs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
        {
            xs[e] = row[src[e]];
            kas[e] = k0[(size_t)e << 8];
            kbs[e] = k1[(size_t)e << 8];
        }
        if (FP)
        {
            double s0[8], s1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
            {
                const double v = fp_red(fp_from_u52(xs[e]), qd, qinv);
                s0[e] = u2d(lo0[e]) + fp_mulmod_q(v, fp_from_u52(kas[e]), qd, qinv);
                s1[e] = u2d(lo1[e]) + fp_mulmod_q(v, fp_from_u52(kbs[e]), qd, qinv);
            }
            if (FPR || (J & 15u) == 15u)
            {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                {
                    s0[e] = fp_red(s0[e], qd, qinv);
                    s1[e] = fp_red(s1[e], qd, qinv);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
            {
                lo0[e] = d2u(s0[e]);
                lo1[e] = d2u(s1[e]);
            }
        }
        else
        {
#pragma unroll
            for (int e = 0; e < 8; ++e)
            {
                mac128r(lo0[e], hi0[e], xs[e], kas[e]);
                mac128r(lo1[e], hi1[e], xs[e], kbs[e]);
            }
        }
    }
    const uint64_t cr0 = pc->cr0, cr1 = pc->cr1;
    uint64_t *__restrict__ o0 = a.acc + ((((size_t)bq * 2 + 0) * (a.L + 1) + slot) << LOGN) + obase;
    uint64_t *__restrict__ o1 = a.acc + ((((size_t)bq * 2 + 1) * (a.L + 1) + slot) << LOGN) + obase;
    const uint64_t *__restrict__ c0 = a.corr + (((size_t)0 * (a.L + 1) + slot) << LOGN) + obase;
    const uint64_t *__restrict__ c1 = a.corr + (((size_t)1 * (a.L + 1) + slot) << LOGN) + obase;
    uint64_t cv0[8], cv1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
        cv0[e] = c0[(size_t)e << 8];
        cv1[e] = c1[(size_t)e << 8];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
        uint64_t r0, r1;
        if (FP)
        {
            r0 = fp_to_canonical(u2d(lo0[e]), qd, qinv);
            r1 = fp_to_canonical(u2d(lo1[e]), qd, qinv);
        }
        else
        {
            r0 = barrett128(lo0[e], hi0[e], q, cr0, cr1);
            r1 = barrett128(lo1[e], hi1[e], q, cr0, cr1);
        }
        o0[(size_t)e << 8] = csub(r0 + cv0[e], q);
        o1[(size_t)e << 8] = csub(r1 + cv1[e], q);
    }
}

// The same sums for NR = 2 or 4 rotations in one pass over the digits (FP64 arithmetic only): a workgroup walks 2048 INPUT
// positions, every digit value is loaded once and multiplied into every rotation's keys at the positions it is sent to
// (itable = the inverse permutation: itable[pos] is where position pos lands), and the sums are stored at those positions.
// Each output still adds its J terms in the order J = 0 .. L-1 with the same operations as ks_hoisted_mac: the same bits.
struct HoistMac2Args
{
    const uint64_t *dig;
    const uint64_t *ct;
    const uint32_t *itable[4];
    const uint64_t *key[4];
    const uint64_t *corr[4];
    uint64_t *acc[4];
    const PrimeConst *pc;
    KsGroup grp;
    uint32_t L, G, k, B;
    uint32_t krows[4];     // rows per key polynomial of each rotation's key
    uint32_t total_work;
};

template <int LOGN, bool FPR, int NR>
__global__ __launch_bounds__(256, 4) void ks_hoisted_mac2(HoistMac2Args a)
{
    constexpr uint32_t TPR8 = 1u << (LOGN - 11);
    constexpr int E = 8 / NR; // positions per thread and pass: NR * E accumulator pairs, NR passes of 256 * E positions cover the tile
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t bq = w % a.B;
    w /= a.B;
    const uint32_t tile = w % TPR8;
    const uint32_t g = w / TPR8;
    const uint32_t prime = a.grp.prime[g];
    const uint32_t slot = a.grp.slot[g];
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q;
    const double qd = u2d(pc->qd), qinv = u2d(pc->qinv);
    for (uint32_t pass = 0; pass < (uint32_t)NR; ++pass)
    {
        const uint32_t pbase = (tile << 11) + pass * (256u * E) + threadIdx.x;
        uint32_t dst[NR][E];
        double s0[NR][E], s1[NR][E];
#pragma unroll
        for (int r = 0; r < NR; ++r)
        {
#pragma unroll
            for (int e = 0; e < E; ++e)
            {
                dst[r][e] = a.itable[r][pbase + ((uint32_t)e << 8)];
                s0[r][e] = 0.0;
                s1[r][e] = 0.0;
            }
        }
        for (uint32_t J = 0; J < a.L; ++J)
        {
            const uint64_t *__restrict__ row = (slot == J) ? a.ct + (((size_t)(bq * 2 + 1) * a.L + J) << LOGN)
                                                           : a.dig + ((((size_t)bq * a.G + g) * a.L + J) << LOGN);
            double v[E];
#pragma unroll
            for (int e = 0; e < E; ++e)
            {
                v[e] = fp_red(fp_from_u52(row[pbase + ((uint32_t)e << 8)]), qd, qinv);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
            {
                const uint32_t kr = a.krows[r], krow = slot == a.L ? kr - 1 : prime;
                const uint64_t *__restrict__ k0 = a.key[r] + (((size_t)(J * 2 + 0) * kr + krow) << LOGN);
                const uint64_t *__restrict__ k1 = a.key[r] + (((size_t)(J * 2 + 1) * kr + krow) << LOGN);
                // this rotation's key residues first, then the products, then one wave-uniform branch around the reductions
                // (see ks_hoisted_mac): same operations per accumulator in the same order
                uint64_t ka[E], kb[E];
#pragma unroll
                for (int e = 0; e < E; ++e)
                {
                    ka[e] = k0[dst[r][e]];
                    kb[e] = k1[dst[r][e]];
                }
#pragma unroll
                for (int e = 0; e < E; ++e)
                {
                    s0[r][e] += fp_mulmod_q(v[e], fp_from_u52(ka[e]), qd, qinv);
                    s1[r][e] += fp_mulmod_q(v[e], fp_from_u52(kb[e]), qd, qinv);
                }
            }
            if (FPR || (J & 15u) == 15u)
            {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                    {
                        s0[r][e] = fp_red(s0[r][e], qd, qinv);
                        s1[r][e] = fp_red(s1[r][e], qd, qinv);
                    }
                }
            }
        }
        // the sixteen correction values first, in one batch: written as load-add-store per accumulator, the compiled epilogue made
        // sixteen dependent round trips (it cannot see that the stores never touch the correction rows)
        uint64_t c0v[NR][E], c1v[NR][E];
#pragma unroll
        for (int r = 0; r < NR; ++r)
        {
            const uint64_t *__restrict__ c0 = a.corr[r] + (((size_t)0 * (a.L + 1) + slot) << LOGN);
            const uint64_t *__restrict__ c1 = a.corr[r] + (((size_t)1 * (a.L + 1) + slot) << LOGN);
#pragma unroll
            for (int e = 0; e < E; ++e)
            {
                c0v[r][e] = c0[dst[r][e]];
                c1v[r][e] = c1[dst[r][e]];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
        {
            uint64_t *__restrict__ o0 = a.acc[r] + ((((size_t)bq * 2 + 0) * (a.L + 1) + slot) << LOGN);
            uint64_t *__restrict__ o1 = a.acc[r] + ((((size_t)bq * 2 + 1) * (a.L + 1) + slot) << LOGN);
#pragma unroll
            for (int e = 0; e < E; ++e)
            {
                const uint32_t d = dst[r][e];
                o0[d] = csub(fp_to_canonical(s0[r][e], qd, qinv) + c0v[r][e], q);
                o1[d] = csub(fp_to_canonical(s1[r][e], qd, qinv) + c1v[r][e], q);
            }
        }
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

// the same expansion for the FP64 modes: the integer result is below 2 q_i < 2^52, so the conversion is exact.
// (Measured in round 3 and not kept: the expansion entirely in doubles -- v = hi 2^32 + lo, hi * (2^32 mod q_i) reduced with
// fp_mulmod_q, minus q_last mod q_i when v + half wraps -- is 22 % SLOWER per launch than this integer Barrett step, 372 against
// 304 us at pack 48: FP64 operations issue at the rate of the 32-bit multiplies they replace, and there are more of them.  The same
// with the two halves made doubles without v_cvt_f64_u32 (a register pair {x, 0x43300000} minus 2^52): 364 against 296 us.)
struct LoadExpandLastFp
{
    uint64_t ql, half, q, cr1, fix, qd, qinv;
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return d2u(fp_red(fp_from_u52(barrett64(csub(v + half, ql), q, cr1) + fix), u2d(qd), u2d(qinv)));
    }
};

struct StoreModDown
{
    const ulonglong2 *acc; // tile base of the row being divided (first of `splits` partial copies)
    ulonglong2 *out;       // tile base of the result row
    const ulonglong2 *add; // tile base of the row to add to the result (key switch), or nullptr (rescale)
    const ulonglong2 *add2; // tile base of a second row to add (a sum that the result is accumulated into), or nullptr
    uint64_t q;
    Tw inv;
    Tw sc;                 // use_sc: the divided row is acc * sc mod q (a scalar plaintext product fused into the rescale)
    int use_sc;
    uint32_t splits;       // partial sums to add up (key switch on few ciphertexts), 1 otherwise
    size_t split_stride;   // 16-byte chunks between consecutive partial copies
    // the operands of four chunks, fetched together (fwd_contig_tile calls fetch before it finishes them): the compiled loop
    // used to make up to three dependent round trips per chunk -- accumulator, addend, second addend, each behind its own
    // null-pointer branch and wait -- twenty-four per tile
    ulonglong2 av[4], cv[4], dv[4];
    __device__ __forceinline__ void fetch(const uint32_t (&chs)[4])
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            av[i] = acc[chs[i]];
        }
        if (add)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                cv[i] = add[chs[i]];
            }
        }
        if (add2)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                dv[i] = add2[chs[i]];
            }
        }
    }
    __device__ __forceinline__ void operator()(int i, uint32_t ch, ulonglong2 u) const
    {
        ulonglong2 x = av[i], r;
        for (uint32_t sp = 1; sp < splits; ++sp)
        {
            ulonglong2 y = acc[ch + sp * split_stride];
            x.x = csub(x.x + y.x, q);
            x.y = csub(x.y + y.y, q);
        }
        if (use_sc)
        {
            // multiply_poly_scalar_coeffmod (polyarithsmallmod.cpp:226-278): canonical product, as the separate call gives
            x.x = csub(mul_shoup_lazy(x.x, sc.w, sc.wq, q), q);
            x.y = csub(mul_shoup_lazy(x.y, sc.w, sc.wq, q), q);
        }
        r.x = csub(mul_shoup_lazy(x.x + q - u.x, inv.w, inv.wq, q), q);
        r.y = csub(mul_shoup_lazy(x.y + q - u.y, inv.w, inv.wq, q), q);
        if (add)
        {
            r.x = csub(r.x + cv[i].x, q);
            r.y = csub(r.y + cv[i].y, q);
        }
        if (add2)
        {
            r.x = csub(r.x + dv[i].x, q);
            r.y = csub(r.y + dv[i].y, q);
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
    // what is added to the result (the key switch adds its output to a ciphertext, evaluator.cpp:3012-3017):
    // row (p, i) of the addend is at addend + ((p / 2) * addend_bstride + (p % 2) * Lout + i) * N, i.e. the first two
    // polynomials of ciphertexts stored addend_bstride rows apart; add_mode 0: nothing, 1: every polynomial,
    // 2: even polynomials only (apply_galois: c0 gets the permuted c0, c1 starts from zero)
    const uint64_t *addend;
    uint32_t addend_bstride;
    int add_mode;
    const uint64_t *addend2; // or null: [P][Lout][N] like out (may be out itself), added to every polynomial after `addend`
    uint32_t acc_splits;     // >= 1
    size_t acc_split_stride; // words between partial copies of acc
    uint32_t total_work;
    RowMap sel;              // the output rows (= primes) of this launch: one arithmetic mode per launch
    uint32_t Lsel;
    int has_scal;            // rows of acc are multiplied by scal[i] (reduced scalar + Shoup quotient) before the division
    Tw scal[MOAI_MAX_RNS];
};

// (one tile per workgroup: the software-pipelined form of the key switch's strided pass, fwd_strided_tiles, measured the same here --
// 313 against 308 us per launch at pack 48 -- this pass is bound by its load operation's arithmetic)
template <int LOGN, int MODE>
__global__ __launch_bounds__(256, 4) void moddown_strided(ModDownArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ uint64_t lds[4096];
    // tile fastest, then prime, then polynomial: neighbours read the same `last` row
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    w /= TPR;
    const uint32_t i = __builtin_amdgcn_readfirstlane(a.sel.idx[w % a.Lsel]);
    const uint32_t p = w / a.Lsel;
    const PrimeConst *pc = a.pc + i;
    const uint64_t ql = a.pc[a.prime_last].q, half = ql >> 1;
    const uint64_t fix = pc->q - barrett64(half, pc->q, pc->cr1);
    const uint64_t *in = a.last + ((size_t)p << LOGN);
    uint64_t *outp = a.u + (((size_t)p * a.Lout + i) << LOGN);
    if (MODE >= M_FPN)
    {
        LoadExpandLastFp op;
        op.ql = ql;
        op.half = half;
        op.q = pc->q;
        op.cr1 = pc->cr1;
        op.fix = fix;
        op.qd = pc->qd;
        op.qinv = pc->qinv;
        fwd_strided_tile<LOGN, LoadExpandLastFp, MODE>(in, outp, tile, a.tw + ((size_t)i << LOGN), pc->qd, pc->qinv, lds, threadIdx.x, op);
    }
    else
    {
        LoadExpandLast op;
        op.ql = ql;
        op.half = half;
        op.q = pc->q;
        op.cr1 = pc->cr1;
        op.fix = fix;
        fwd_strided_tile<LOGN, LoadExpandLast, MODE>(in, outp, tile, a.tw + ((size_t)i << LOGN), pc->q, pc->q2, lds, threadIdx.x, op);
    }
}

template <int LOGN, int MODE>
__global__ __launch_bounds__(256) void moddown_contig(ModDownArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    // (the first four stages' twiddles through LDS, as in ntt_fwd_contig, measured slower here -- 756 against 716-739 us per launch:
    // 36 KiB of LDS per workgroup leave four of them on a CU where this kernel's 93 registers allow five)
    __shared__ ulonglong2 lds2[2048];
    // polynomial fastest: the workgroups that share a twiddle slice run together
    uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t p = w % a.P;
    w /= a.P;
    const uint32_t tile = w % TPR;
    const uint32_t i = __builtin_amdgcn_readfirstlane(a.sel.idx[w / TPR]);
    const PrimeConst &pc = a.pc[i];
    StoreModDown st;
    st.acc = reinterpret_cast<const ulonglong2 *>(a.acc + (((size_t)p * a.acc_stride + i) << LOGN)) + ((size_t)tile << 11);
    st.out = reinterpret_cast<ulonglong2 *>(a.out + (((size_t)p * a.Lout + i) << LOGN)) + ((size_t)tile << 11);
    st.q = pc.q;
    st.inv = a.inv_last[i];
    st.add = nullptr;
    if (a.add_mode == 1 || (a.add_mode == 2 && !(p & 1u)))
    {
        st.add = reinterpret_cast<const ulonglong2 *>(a.addend + (((size_t)(p >> 1) * a.addend_bstride + (size_t)(p & 1u) * a.Lout + i) << LOGN)) +
                 ((size_t)tile << 11);
    }
    st.add2 = a.addend2 ? reinterpret_cast<const ulonglong2 *>(a.addend2 + (((size_t)p * a.Lout + i) << LOGN)) + ((size_t)tile << 11) : nullptr;
    st.splits = a.acc_splits;
    st.split_stride = a.acc_split_stride >> 1;
    st.use_sc = a.has_scal;
    st.sc = a.scal[i];
    // the tile hands the store canonical integers in every mode
    fwd_contig_tile<LOGN, MODE, StoreModDown>(a.u + (((size_t)p * a.Lout + i) << LOGN), tile, a.tw + ((size_t)i << LOGN), mode_q<MODE>(pc),
                                              mode_q2<MODE>(pc), lds2, threadIdx.x, a.twb + (size_t)i * ((size_t)TPR * 15 * 256), pc.cr1,
                                              st);
}

} // namespace moai
