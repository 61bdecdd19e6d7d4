// ntt_kernels.hip.h -- negacyclic NTT / INTT for gfx950.
//
// What it computes is the reference's ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey
// (SEAL/util/ntt.cpp:394-475, butterflies SEAL/util/dwthandler.h:94-356): forward = Cooley-Tukey,
// natural order in, bit-reversed order out; inverse = Gentleman-Sande, bit-reversed in, natural
// out, N^-1 folded into the last stage.  How it computes it is CDNA4's: for N = 2^LOGN with
// 12 <= LOGN <= 16 the LOGN stages are split into a STRIDED pass (the first LOGN-8 stages: 256
// interleaved sub-transforms whose elements are 256 apart) and a CONTIGUOUS pass (the last 8
// stages: independent 256-point transforms on consecutive coefficients).  A 256-thread workgroup
// owns a 4096-coefficient tile (32 KiB); every thread keeps 16 coefficients in VGPRs and runs
// radix-16 (4 stages) on them, the tile is transposed once through LDS (XOR-swizzled so both
// sides are bank-conflict free), and the thread runs the remaining <= 4 stages.  Global accesses
// are 128-byte runs (strided pass) or whole-wave 1 KiB rows (contiguous pass, staged through LDS).
// One RNS prime per block column: a workgroup works under a single prime, its twiddles come from
// that prime's table, and the blockIdx -> tile map keeps the workgroups that share twiddles on
// one XCD so the table lines stay in that XCD's L2.
//
// Lazy ranges: forward keeps values in [0,4q) between stages, inverse in [0,2q), like the
// reference; the pass that finishes a transform writes canonical residues.
#pragma once
#include "modarith.hip.h"

namespace moai {

// Observed dispatch is round-robin over the 8 XCDs (blocks b and b+8 share an XCD).  Give each XCD
// a contiguous range of work ids so that neighbours in work order share an L2.  Speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t total)
{
    if (total & 7u)
    {
        return b;
    }
    return (b & 7u) * (total >> 3) + (b >> 3);
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not for its
// outstanding global loads (a __syncthreads() would drain vmcnt as well and serialise the prefetch below).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exchange between lanes of ONE wave through LDS: a wave's LDS instructions execute in issue order, so data written by
// one lane is there for another lane of the same wave once the wave's LDS counter has drained; no s_barrier, the
// other waves of the workgroup run on.  (The asm also keeps the compiler from moving LDS accesses across it.)
__device__ __forceinline__ void lds_wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- LDS tile layouts ------------------------------------------------------------------------------
// contiguous pass: element e of the 4096-tile lives in 16-byte chunk (e>>1); chunks are XOR-swizzled
// within each 128-byte row so that "one row per lane" (ds_*_b128) and "one column per lane"
// (ds_*_b64) accesses are both conflict free.
__device__ __forceinline__ uint32_t phys_contig(uint32_t e)
{
    uint32_t row = e >> 4;
    uint32_t c = (e >> 1) & 7u;
    return (row << 4) | (((c ^ (row & 7u)) << 1) | (e & 1u));
}

template <int GB>
__device__ __forceinline__ uint32_t phys_strided(uint32_t e)
{
    // only the 16-column tile (LOGN = 16) needs it: odd 256-blocks are shifted by half a bank row
    if (GB == 4)
    {
        return e ^ (((e >> 8) & 1u) << 4);
    }
    return e;
}

struct NttArgs
{
    uint64_t *data;            // [n_poly][L][N]
    const Tw *tw;              // fwd or inv table, [k][N]
    const Tw *twb;             // same direction, per-thread order for the contiguous pass's last 4 stages
    const PrimeConst *pc;      // [k]
    RowMap rows;               // row r -> prime
    RowMap sel;                // the rows this launch transforms: sel.idx[0..Lsel) (rows of one arithmetic mode)
    RowMap selp;               // their primes: selp.idx[i] = rows.idx[sel.idx[i]]
    uint32_t L;
    uint32_t Lsel;
    uint32_t n_poly;
    uint32_t total_work;       // grid size
    // inverse transform only: when set, the first pass reads polynomial p's row r from src row p * src_stride + src_off + r
    // instead of from `data` (the transform of a slice of a larger layout, written into `data`, without a copy first)
    const uint64_t *src;
    uint32_t src_stride;
    uint32_t src_off;
    uint32_t lds_twiddles; // forward contiguous pass: the first four stages' twiddles through LDS (MOAI_NTT_LDSTW=0: global loads)
};

// (q, q2) arguments of a tile function under MODE: the integer pair, or the bit patterns of (double q, 1/q)
template <int MODE>
__device__ __forceinline__ uint64_t mode_q(const PrimeConst &pc)
{
    return MODE >= M_FPN ? pc.qd : (MODE == M_LAZY8 ? pc.nq : pc.q);
}
template <int MODE>
__device__ __forceinline__ uint64_t mode_q2(const PrimeConst &pc)
{
    return MODE >= M_FPN ? pc.qinv : (MODE == M_LAZY8 ? pc.n4q : pc.q2);
}

// =====================================================================================================
// forward, strided pass: stages 0 .. LOGN-9
// =====================================================================================================
struct LoadIdentity
{
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return v;
    }
};

// v mod q on load (modulo_poly_coeffs, SEAL/util/polyarithsmallmod.cpp:18-41): the key switch feeds
// digits that are canonical under another prime (SEAL/evaluator.cpp:2844-2854)
struct LoadBarrett
{
    uint64_t q, cr1;
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return barrett64(v, q, cr1);
    }
};

// FP64 modes: any integer below 2^53 -> the double holding its balanced residue (|.| <= q/2); serves both as
// the plain load and as the "mod q_I" of the key switch
struct LoadFp
{
    uint64_t qd, qinv;
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return d2u(fp_red(fp_from_u64(v), u2d(qd), u2d(qinv)));
    }
};

// the same for an integer known to be below 2^52 (a key-switch digit canonical under a prime below 2^52)
struct LoadFp52
{
    uint64_t qd, qinv;
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return d2u(fp_red(fp_from_u52(v), u2d(qd), u2d(qinv)));
    }
};

// the same for integers of any size (a key-switch digit under a 52..61-bit prime): integer Barrett step first
struct LoadBarrettFp
{
    uint64_t q, cr1, qd, qinv;
    __device__ __forceinline__ uint64_t operator()(uint64_t v) const
    {
        return d2u(fp_red(fp_from_u52(barrett64(v, q, cr1)), u2d(qd), u2d(qinv))); // reduced below q < 2^51
    }
};


// the unrolled stage loops of the tiles carry the stage number in a loop variable; M_GUARD2 needs it as a constant
template <int MODE>
__device__ __forceinline__ void ct_bfly_stage(uint64_t &x, uint64_t &y, uint64_t w, uint64_t wq, uint64_t q, uint64_t q2, int stages_left)
{
    if (MODE == M_GUARD2)
    {
        if (stages_left & 1)
        {
            ct_bfly_guard2<false>(x, y, w, wq, q, q2);
        }
        else
        {
            ct_bfly_guard2<true>(x, y, w, wq, q, q2);
        }
    }
    else
    {
        ct_bfly_t<MODE>(x, y, w, wq, q, q2);
    }
}

// one tile of the strided pass: reads row `inp`, writes row `outp` (may be the same row)
// PRE (FP64 modes, with tw1 = the forward powers as plain doubles): the per-thread twiddles of phase B -- fifteen at N = 2^16 --
// are fetched as 8-byte doubles BEFORE the exchange instead of as 16-byte {w, w/q} pairs one or two at a time between the
// butterflies that use them (ten exposed round trips to L2 per tile in the compiled loop); the butterflies then take their
// quotient estimate from RN(y w) RN(1/q) like the contiguous key-switch pass (ct_bfly_fp1: growth below 2q per stage instead of
// 0.75q; 46-bit primes still end all sixteen stages below 28q < 2^52).  Same residues.
// the 16 values a thread takes from its tile of the input row (tile base = row + tile * G), not yet converted
template <int LOGN>
__device__ __forceinline__ void strided_load_raw(const uint64_t *__restrict__ irow, const uint32_t tid, uint64_t (&raw)[16])
{
    constexpr int GB = 12 - (LOGN - 8);
    constexpr uint32_t G = 1u << GB;
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        uint32_t e = (uint32_t)j * 256u + tid;
        raw[j] = irow[((e >> GB) << 8) + (e & (G - 1))];
    }
}

// everything after the load: the butterflies of both phases, the exchange and the stores of one tile (row = tile base of the
// output row).  SYNC_FIRST: the workgroup has used `lds` for an earlier tile -- wait until everybody has read it.  The barriers
// wait for LDS traffic only (lds_barrier), so global loads and stores of neighbouring tiles stay in flight across them.
// LDSTW: phase B takes its twiddles (entries 16..255 of the table) from the workgroup's copy in LDS, `ldstw`
template <int LOGN, int MODE, bool PRE, bool SYNC_FIRST, bool LDSTW = false>
__device__ __forceinline__ void strided_core(uint64_t (&x)[16], uint64_t *__restrict__ row, const Tw *__restrict__ tw, uint64_t q,
                                             uint64_t q2, uint64_t *lds, const uint32_t tid, const double *__restrict__ tw1,
                                             const Tw *ldstw = nullptr)
{
    // phase A's fifteen twiddles (entries 1..15 of the table) are the same for every thread: read through the constant address
    // space, so that they stay SCALAR loads also inside a loop over tiles -- behind the stores of an earlier iteration the compiler
    // turns a uniform global load into a vector load (it cannot see that no kernel ever writes the tables), and the wait for a
    // vector load also waits for the prefetch of the next tile issued before it
    typedef const Tw __attribute__((address_space(4))) *TwConst;
    const TwConst twa = (TwConst)(uintptr_t)tw;
    constexpr int R1 = LOGN - 8;
    constexpr int RB = R1 - 4;
    constexpr int GB = 12 - R1;
    constexpr uint32_t G = 1u << GB;
    // phase A: top four bits of t live in the register index
#ifdef MOAI_DIAG_NOCOMPUTE
    const int diag_stages = (q == 0x7ff8dead0000beefull) ? 4 : 0; // diagnostic build: loads, exchange and stores only
#define MOAI_DIAG_STAGE_OK(u) && ((u) < diag_stages)
#else
#define MOAI_DIAG_STAGE_OK(u)
#endif
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (!(j & half) MOAI_DIAG_STAGE_OK(u))
            {
                const uint32_t ti = (1u << u) + (uint32_t)(j >> (4 - u));
                ct_bfly_stage<MODE>(x[j], x[j + half], twa[ti].w, twa[ti].wq, q, q2, LOGN - 1 - u);
            }
        }
    }
    if (RB > 0)
    {
        double twb[16];
        if (PRE)
        {
            const uint32_t thp = tid >> GB;
#pragma unroll
            for (int s = 4; s < R1; ++s)
            {
                const int cnt = 16 >> (R1 - s); // distinct twiddles of this stage per thread
#pragma unroll
                for (int i = 0; i < cnt; ++i)
                {
                    twb[cnt - 1 + i] = tw1[(1u << s) + ((thp << (4 - (R1 - s))) | (uint32_t)i)];
                }
            }
        }
        if (SYNC_FIRST)
        {
            lds_barrier();
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            lds[phys_strided<GB>((uint32_t)j * 256u + tid)] = x[j];
        }
        lds_barrier();
        const uint32_t g = tid & (G - 1);
        const uint32_t th = tid >> GB;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            x[j] = lds[phys_strided<GB>((th << (GB + 4)) | ((uint32_t)j << GB) | g)];
        }
        // phase B: low RB bits of t
#pragma unroll
        for (int s = 4; s < R1; ++s)
        {
            const int half = 1 << (R1 - 1 - s);
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                if (!(j & half) MOAI_DIAG_STAGE_OK(s - 4))
                {
                    if (PRE)
                    {
                        ct_bfly_fp1<MODE == M_FPR>(x[j], x[j + half], twb[(16 >> (R1 - s)) - 1 + (j >> (R1 - s))], u2d(q), u2d(q2));
                    }
                    else
                    {
                        uint32_t t_ = (th << 4) | (uint32_t)j;
                        Tw t = LDSTW ? ldstw[(1u << s) + (t_ >> (R1 - s))] : tw[(1u << s) + (t_ >> (R1 - s))];
                        ct_bfly_stage<MODE>(x[j], x[j + half], t.w, t.wq, q, q2, LOGN - 1 - s);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            uint32_t t_ = (th << 4) | (uint32_t)j;
#ifdef MOAI_DIAG_NOSTORE
            if (x[j] == 0x7ff8dead0000beefull) // diagnostic build: the arithmetic stays, the stores (practically) never happen
#endif
            row[(t_ << 8) + g] = x[j]; // (non-temporal stores measured the same: 2080 against 2092 us in the key switch's pass)
        }
    }
    else
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            uint32_t e = (uint32_t)j * 256u + tid;
            row[((e >> GB) << 8) + (e & (G - 1))] = x[j];
        }
    }
}

// LDSTW (N = 2^16, `lds` with 512 more words): phase B's twiddles -- entries 16..255 of the table, the same for every tile -- are
// copied into LDS, one 16-byte load per thread issued with the tile's own loads, and read from there behind the exchange
template <int LOGN, class LoadOp = LoadIdentity, int MODE = M_GUARD, bool PRE = false, bool LDSTW = false>
__device__ __forceinline__ void fwd_strided_tile(const uint64_t *__restrict__ inp, uint64_t *__restrict__ rowp, uint32_t tile,
                                                 const Tw *__restrict__ tw, uint64_t q, uint64_t q2, uint64_t *lds,
                                                 const uint32_t tid, LoadOp op = LoadOp(), const double *__restrict__ tw1 = nullptr)
{
    constexpr uint32_t G = 1u << (12 - (LOGN - 8));
    static_assert(!LDSTW || LOGN == 16, "the LDS copy holds the 256 entries phase B of N = 2^16 indexes");
    Tw *ldstw = reinterpret_cast<Tw *>(lds + 4096);
    uint64_t x[16];
    strided_load_raw<LOGN>(inp + tile * G, tid, x);
    if (LDSTW)
    {
        ldstw[tid] = tw[tid]; // visible to the workgroup behind the exchange's barrier
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        x[j] = op(x[j]);
    }
    strided_core<LOGN, MODE, PRE, false, LDSTW>(x, rowp + tile * G, tw, q, q2, lds, tid, tw1, LDSTW ? ldstw : nullptr);
}

// ITEMS consecutive tiles of one row by one workgroup, software-pipelined: the loads of tile i + 1 are issued before the
// butterflies of tile i, and the stores of tile i drain under the butterflies of tile i + 1.  A workgroup that loads, computes and
// stores ONE tile keeps the memory system busy only through its neighbours on the CU, and 32 KiB of LDS plus ~96 VGPRs allow four
// or five of them: measured on the key switch's strided pass (round 3, tools/diag), 2.37 ms per launch against 1.96 ms for its
// memory traffic alone and 1.62 ms for its arithmetic alone -- the two did not overlap.
template <int LOGN, class LoadOp, int MODE, int ITEMS>
__device__ __forceinline__ void fwd_strided_tiles(const uint64_t *inp, uint64_t *rowp, uint32_t tile0,
                                                  const Tw *__restrict__ tw, uint64_t q, uint64_t q2, uint64_t *lds, const uint32_t tid,
                                                  LoadOp op)
{
    constexpr uint32_t G = 1u << (12 - (LOGN - 8));
    static_assert(LOGN == 16, "the LDS copy holds the 256 entries phase B of N = 2^16 indexes");
    uint64_t raw[16], x[16];
    strided_load_raw<LOGN>(inp + tile0 * G, tid, raw);
    // phase B's twiddles into LDS (`lds` + 4096 words: 256 entries of 16 bytes, one per thread; entries 16..255 are used).
    // Fetched from memory inside the loop, every wait for one of them would also wait for the stores of the previous tile
    // and for the prefetch of the next one: vector memory operations complete in order.
    Tw *ldstw = reinterpret_cast<Tw *>(lds + 4096);
    ldstw[tid] = tw[tid];
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        x[j] = op(raw[j]);
    }
    // one loop body (not unrolled: four copies of it would not fit the instruction cache).  The first tile's values are converted
    // before the loop, so the body starts from arithmetic results and not from pending loads: entered that way, its first wait
    // would also cover the loads it has just issued.
#pragma unroll 1
    for (uint32_t it = 0; it < (uint32_t)ITEMS; ++it)
    {
        // opaque copy of the thread index: the per-thread addresses (16 loads, 16 stores, 32 LDS slots, the twiddles) are
        // recomputed in every iteration instead of hoisted out of the loop into registers the prefetch needs
        uint32_t t = tid;
        asm volatile("" : "+v"(t));
        const bool more = it + 1u < (uint32_t)ITEMS;
        if (more)
        {
            strided_load_raw<LOGN>(inp + (tile0 + it + 1u) * G, t, raw);
        }
        strided_core<LOGN, MODE, false, true, true>(x, rowp + (tile0 + it) * G, tw, q, q2, lds, t, nullptr, ldstw);
        if (more)
        {
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                x[j] = op(raw[j]);
            }
        }
    }
}

// five waves per SIMD (96 VGPRs) where the body fits; the modes whose body does not (12, 8 and 14 spilled registers
// under that cap) run faster at four: M_GUARD2 12.17 -> 11.90 ms per batch transform, FP64 / unguarded rows 2-3 %
template <int LOGN, int MODE = M_GUARD>
__global__ __launch_bounds__(256, (MODE == M_GUARD2 || MODE == M_LAZY8 || MODE == M_FPR || MODE == M_NOGUARD) ? 4 : 5) void ntt_fwd_strided(NttArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    // the modes that run four workgroups per CU anyway take phase B's twiddles through LDS at N = 2^16 (4 KiB more)
    constexpr bool LDSTW = LOGN == 16 && (MODE == M_GUARD2 || MODE == M_LAZY8 || MODE == M_FPR || MODE == M_NOGUARD);
    __shared__ uint64_t lds[4096 + (LDSTW ? 512 : 0)];
    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    const uint32_t srow = w / TPR;                 // over n_poly * Lsel selected rows
    const uint32_t si = srow % a.Lsel;
    // 16-bit kernarg entries arrive through a vector load: pin them back to scalars, or every address and
    // modulus derived from them lives in VGPRs
    const uint32_t r = __builtin_amdgcn_readfirstlane(a.sel.idx[si]);
    const uint32_t prime = __builtin_amdgcn_readfirstlane(a.selp.idx[si]);
    const PrimeConst &pc = a.pc[prime];
    uint64_t *rowp = a.data + ((size_t)((srow / a.Lsel) * a.L + r) << LOGN);
    if (MODE >= M_FPN)
    {
        LoadFp op;
        op.qd = pc.qd;
        op.qinv = pc.qinv;
        fwd_strided_tile<LOGN, LoadFp, MODE, false, LDSTW>(rowp, rowp, tile, a.tw + ((size_t)prime << LOGN), pc.qd, pc.qinv, lds, threadIdx.x, op);
    }
    else
    {
        fwd_strided_tile<LOGN, LoadIdentity, MODE, false, LDSTW>(rowp, rowp, tile, a.tw + ((size_t)prime << LOGN), mode_q<MODE>(pc), mode_q2<MODE>(pc),
                                                                 lds, threadIdx.x);
    }
}

// =====================================================================================================
// forward, contiguous pass: stages LOGN-8 .. LOGN-1 on 16 consecutive 256-blocks; writes canonical
// =====================================================================================================
// what the contiguous pass does with a finished 16-byte chunk (index ch within the 4096-coefficient tile)
// fetch(chs) is called with four chunk indices before the four calls that finish them: a store operation that needs operands
// from memory issues all its loads there, in one batch, instead of one round trip per chunk and operand
struct StoreTile
{
    ulonglong2 *out; // tile base
    __device__ __forceinline__ void fetch(const uint32_t (&)[4])
    {
    }
    __device__ __forceinline__ void operator()(int, uint32_t ch, ulonglong2 v) const
    {
        out[ch] = v;
    }
};

// M_NOGUARD: input below 20q (strided pass without guards), stages without guards, one Barrett step at
// the end (cr1 = high word of floor(2^128/q)); M_GUARD: the reference's [0,4q) discipline; FP64 modes: doubles
// in, canonical integers out (q, q2 carry the bit patterns of (double q, 1/q))
// ldstw (or null): 240 entries of LDS beside lds2 for the twiddles of the first four stages -- fifteen per 256-block, shared by the
// block's sixteen threads.  Every wave fetches the sixty of its own four blocks with ONE load per lane and reads them back from LDS
// (broadcast reads, no s_barrier: the blocks of a wave are its own), instead of fifteen 16-byte global loads per thread issued one
// or two ahead of the butterflies that use them.
template <int LOGN, int MODE = M_GUARD, class StoreOp = StoreTile>
__device__ __forceinline__ void fwd_contig_tile(uint64_t *__restrict__ rowp, uint32_t tile, const Tw *__restrict__ tw,
                                                uint64_t q, uint64_t q2, ulonglong2 *lds2, const uint32_t tid,
                                                const Tw *__restrict__ twb, uint64_t cr1, StoreOp store, Tw *ldstw = nullptr)
{
    constexpr int R1 = LOGN - 8;
    uint64_t *lds = reinterpret_cast<uint64_t *>(lds2);
    uint64_t *__restrict__ base = rowp + ((size_t)tile << 12);
    const uint32_t b = tid >> 4;
    const uint32_t tl = tid & 15u;
    const uint32_t blk = (tile << 4) + b;
    const Tw *__restrict__ twbt = twb + (size_t)tile * (15 * 256);

    if (ldstw)
    {
        const uint32_t lane = tid & 63u;
        if (lane < 60u)
        {
            const uint32_t bb = ((tid >> 6) << 2) + lane / 15u, i = lane % 15u;       // block of this wave, slot 2^u - 1 + k
            const uint32_t u = i == 0 ? 0u : (i < 3 ? 1u : (i < 7 ? 2u : 3u)), k = i - ((1u << u) - 1u);
            ldstw[bb * 15u + i] = tw[(1u << (R1 + u)) + (((tile << 4) + bb) << u) + k];
        }
    }
    uint64_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        x[j] = base[(b << 8) | ((uint32_t)j << 4) | tl];
    }
    if (ldstw)
    {
        lds_wave_sync();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (!(j & half))
            {
                Tw t = ldstw ? ldstw[b * 15u + ((1u << u) - 1u) + (uint32_t)(j >> (4 - u))]
                             : tw[(1u << (R1 + u)) + (blk << u) + (uint32_t)(j >> (4 - u))];
                ct_bfly_stage<MODE>(x[j], x[j + half], t.w, t.wq, q, q2, 7 - u);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        lds[phys_contig((b << 8) | ((uint32_t)j << 4) | tl)] = x[j];
    }
    // a 256-block belongs to 16 consecutive threads, i.e. to one wave: every exchange of this pass is wave-local
    lds_wave_sync();
    const uint32_t myrow = tid; // (b << 4) | th
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
                // slot = 2^(u-4) - 1 + (j >> (8-u)); consecutive threads read consecutive entries
                Tw t = twbt[(((1u << (u - 4)) - 1u + (uint32_t)(j >> (8 - u))) << 8) + tid];
                ct_bfly_stage<MODE>(x[j], x[j + half], t.w, t.wq, q, q2, 7 - u);
            }
        }
    }
    // a thread overwrites the row it alone has read
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        ulonglong2 v;
        if (MODE >= M_FPN)
        {
            v.x = fp_to_canonical(u2d(x[2 * c]), u2d(q), u2d(q2));
            v.y = fp_to_canonical(u2d(x[2 * c + 1]), u2d(q), u2d(q2));
        }
        else if (MODE == M_NOGUARD)
        {
            v.x = barrett64(x[2 * c], q, cr1);
            v.y = barrett64(x[2 * c + 1], q, cr1);
        }
        else if (MODE == M_LAZY8)
        {
            // values below 8q; (q, q2) = (2^64 - q, 2^64 - 4q), and 2^64 - 2q = (2^64 - 4q) / 2 + 2^63
            const uint64_t n2q = (q2 >> 1) | 0x8000000000000000ull;
            v.x = csub_sign(csub_sign(csub_sign(x[2 * c], q2), n2q), q);
            v.y = csub_sign(csub_sign(csub_sign(x[2 * c + 1], q2), n2q), q);
        }
        else if (MODE == M_GUARD2)
        {
            // the last stage is a guarded one: values below 6q
            v.x = csub(csub(csub(x[2 * c], q2 << 1), q2), q);
            v.y = csub(csub(csub(x[2 * c + 1], q2 << 1), q2), q);
        }
        else
        {
            v.x = csub(csub(x[2 * c], q2), q);
            v.y = csub(csub(x[2 * c + 1], q2), q);
        }
        lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))] = v;
    }
    lds_wave_sync();
    // every wave stores the 64 rows (8 KiB, contiguous in memory) its own lanes finished: 1 KiB per wave instruction
    const uint32_t wbase = (tid >> 6) << 9, lane = tid & 63u;
#pragma unroll
    for (int h = 0; h < 2; ++h)
    {
        uint32_t chs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            chs[i] = wbase + (uint32_t)(4 * h + i) * 64u + lane;
        }
        store.fetch(chs);
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const uint32_t rr = chs[i] >> 3;
            store(i, chs[i], lds2[(rr << 3) | ((chs[i] & 7u) ^ (rr & 7u))]);
        }
    }
}

template <int LOGN, int MODE = M_GUARD>
__device__ __forceinline__ void fwd_contig_tile(uint64_t *__restrict__ rowp, uint32_t tile, const Tw *__restrict__ tw,
                                                uint64_t q, uint64_t q2, ulonglong2 *lds2, const uint32_t tid,
                                                const Tw *__restrict__ twb, uint64_t cr1 = 0, Tw *ldstw = nullptr)
{
    StoreTile st;
    st.out = reinterpret_cast<ulonglong2 *>(rowp + ((size_t)tile << 12));
    fwd_contig_tile<LOGN, MODE, StoreTile>(rowp, tile, tw, q, q2, lds2, tid, twb, cr1, st, ldstw);
}

template <int LOGN, int MODE = M_GUARD>
__global__ __launch_bounds__(256) void ntt_fwd_contig(NttArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048 + 240]; // the tile, and the first four stages' twiddles (fwd_contig_tile, ldstw)
    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t pol = w % a.n_poly;
    const uint32_t rest = w / a.n_poly;
    const uint32_t tile = rest % TPR;
    const uint32_t r = __builtin_amdgcn_readfirstlane(a.sel.idx[rest / TPR]);
    const uint32_t prime = __builtin_amdgcn_readfirstlane(a.selp.idx[rest / TPR]);
    const PrimeConst &pc = a.pc[prime];
    fwd_contig_tile<LOGN, MODE>(a.data + (((size_t)pol * a.L + r) << LOGN), tile, a.tw + ((size_t)prime << LOGN),
                                mode_q<MODE>(pc), mode_q2<MODE>(pc), lds2, threadIdx.x,
                                a.twb + (size_t)prime * ((size_t)TPR * 15 * 256), pc.cr1, a.lds_twiddles ? reinterpret_cast<Tw *>(lds2 + 2048) : nullptr);
}

// =====================================================================================================
// inverse, contiguous pass: stages LOGN-1 .. LOGN-8 (gap 1 .. 128); lazy [0,2q) out
// =====================================================================================================
// LZ: the M_LAZY8 butterflies (modarith.hip.h), with (q, q2) = (2^64 - q, 2^64 - 4q); values below 4q instead of 2q
// IM: 0 exact integer butterflies, 1 M_LAZY8, 2 / 3 exact FP64 (FPN / FPR, modarith.hip.h gs_bfly_fp: canonical integers in,
// doubles out to the strided pass; tw / twb = the FP64 inverse tables, (q, q2) = the bit patterns of (double q, 1/q))
template <int LOGN, int IM = 0>
__device__ __forceinline__ void inv_contig_tile(uint64_t *rowp, uint32_t tile, const Tw *__restrict__ tw,
                                                uint64_t q, uint64_t q2, ulonglong2 *lds2, const uint32_t tid,
                                                const Tw *__restrict__ twb, const uint64_t *srcp = nullptr, Tw *ldstw = nullptr)
{
    constexpr int R1 = LOGN - 8;
    uint64_t *lds = reinterpret_cast<uint64_t *>(lds2);
    uint64_t *base = rowp + ((size_t)tile << 12);
    const uint32_t b = tid >> 4;
    const uint32_t tl = tid & 15u;
    const uint32_t blk = (tile << 4) + b;
    const Tw *__restrict__ twbt = twb + (size_t)tile * (15 * 256);

    // the per-thread twiddles of the first four stages do not depend on the data: fetch them while the tile is
    // staged through LDS instead of one by one behind the barrier (-3.7 % on the inverse transform; the same
    // prefetch in the forward passes costs registers they do not have and measured slower)
    Tw tb[15];
#pragma unroll
    for (int i = 0; i < 15; ++i)
    {
        tb[i] = twbt[((uint32_t)i << 8) + tid];
    }
    // ldstw (optional, unused by the kernels): the last four stages' twiddles through LDS like the forward pass's first four
    // (fwd_contig_tile).  Measured on the bench's batch, same box: inverse 11.05 ms with it against 10.99 without -- this pass
    // already has its first four stages' twiddles in registers before the tile arrives, and the extra 4 KiB of LDS buy nothing.
    if (ldstw)
    {
        const uint32_t ln = tid & 63u;
        if (ln < 60u)
        {
            const uint32_t bb = ((tid >> 6) << 2) + ln / 15u, i = ln % 15u;
            const uint32_t u = i == 0 ? 0u : (i < 3 ? 1u : (i < 7 ? 2u : 3u)), k = i - ((1u << u) - 1u);
            ldstw[bb * 15u + i] = tw[(1u << (R1 + u)) + (((tile << 4) + bb) << u) + k];
        }
    }
    // the row read may be another buffer's (NttArgs::src) or the one written: the tile is whole in LDS before any of it is stored
    const ulonglong2 *in2 = reinterpret_cast<const ulonglong2 *>(srcp ? srcp + ((size_t)tile << 12) : base);
    // every wave stages the 64 rows its own lanes transform (8 KiB, contiguous): the exchanges of this pass are wave-local
    const uint32_t wbase = (tid >> 6) << 9, lane = tid & 63u;
#pragma unroll
    for (int it = 0; it < 8; ++it)
    {
        uint32_t ch = wbase + (uint32_t)it * 64u + lane;
        uint32_t rr = ch >> 3;
        lds2[(rr << 3) | ((ch & 7u) ^ (rr & 7u))] = in2[ch];
    }
    lds_wave_sync();
    const uint32_t myrow = tid;
    uint64_t x[16];
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        ulonglong2 v = lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))];
        // FP64 modes: any input below 2^52 (canonical or lazy, like the integer butterflies accept); FPN folds it to |.| <= q/2
        // here -- its schedule of sum reductions counts from there --, FPR folds in every butterfly anyway
        x[2 * c] = IM == 2 ? d2u(fp_red(fp_from_u52(v.x), u2d(q), u2d(q2))) : (IM == 3 ? d2u(fp_from_u52(v.x)) : v.x);
        x[2 * c + 1] = IM == 2 ? d2u(fp_red(fp_from_u52(v.y), u2d(q), u2d(q2))) : (IM == 3 ? d2u(fp_from_u52(v.y)) : v.y);
    }
#pragma unroll
    for (int u = 7; u >= 4; --u)
    {
        const int half = 8 >> (u - 4);
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (!(j & half))
            {
                Tw t = tb[(1 << (u - 4)) - 1 + (j >> (8 - u))];
                gs_bfly_im<IM>(x[j], x[j + half], t.w, t.wq, q, q2, (u & 3) == 0); // FPN: sums folded in every fourth stage
            }
        }
    }
    // rows are private to their thread: no barrier needed before writing them back
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        ulonglong2 v;
        v.x = x[2 * c];
        v.y = x[2 * c + 1];
        lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))] = v;
    }
    lds_wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        x[j] = lds[phys_contig((b << 8) | ((uint32_t)j << 4) | tl)];
    }
#pragma unroll
    for (int u = 3; u >= 0; --u)
    {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (!(j & half))
            {
                Tw t = ldstw ? ldstw[b * 15u + ((1u << u) - 1u) + (uint32_t)(j >> (4 - u))]
                             : tw[(1u << (R1 + u)) + (blk << u) + (uint32_t)(j >> (4 - u))];
                gs_bfly_im<IM>(x[j], x[j + half], t.w, t.wq, q, q2, (u & 3) == 0); // FPN: sums folded in every fourth stage
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        base[(b << 8) | ((uint32_t)j << 4) | tl] = x[j];
    }
}

template <int LOGN, int IM = 0>
__global__ __launch_bounds__(256) void ntt_inv_contig(NttArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048];
    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t pol = w % a.n_poly;
    const uint32_t rest = w / a.n_poly;
    const uint32_t tile = rest % TPR;
    const uint32_t r = __builtin_amdgcn_readfirstlane(a.sel.idx[rest / TPR]);
    const uint32_t prime = __builtin_amdgcn_readfirstlane(a.selp.idx[rest / TPR]);
    const PrimeConst &pc = a.pc[prime];
    const uint64_t *srcp = a.src ? a.src + (((size_t)pol * a.src_stride + a.src_off + r) << LOGN) : nullptr;
    inv_contig_tile<LOGN, IM>(a.data + (((size_t)pol * a.L + r) << LOGN), tile, a.tw + ((size_t)prime << LOGN),
                              IM >= 2 ? pc.qd : (IM == 1 ? pc.nq : pc.q), IM >= 2 ? pc.qinv : (IM == 1 ? pc.n4q : pc.q2), lds2, threadIdx.x,
                              a.twb + (size_t)prime * ((size_t)TPR * 15 * 256), srcp);
}

// =====================================================================================================
// inverse, strided pass: stages LOGN-9 .. 0, N^-1 folded into stage 0; writes canonical
// =====================================================================================================
template <int LOGN, int IM = 0>
__device__ __forceinline__ void inv_strided_tile(uint64_t *__restrict__ rowp, uint32_t tile, const Tw *__restrict__ tw,
                                                 const PrimeConst *pc, uint64_t *lds, const uint32_t tid)
{
    constexpr bool LZ = (IM == 1);
    constexpr int R1 = LOGN - 8;
    constexpr int RB = R1 - 4;
    constexpr int GB = 12 - R1;
    constexpr uint32_t G = 1u << GB;
    const uint64_t q = IM >= 2 ? pc->qd : (LZ ? pc->nq : pc->q);
    const uint64_t q2 = IM >= 2 ? pc->qinv : (LZ ? pc->n4q : pc->q2);
    uint64_t *__restrict__ row = rowp + tile * G;

    // LDSTW (M_LAZY8 at N = 2^16: four workgroups per CU either way): phase B's twiddles -- entries 16..255 of the table, shared by
    // the sixteen threads of a group -- through a copy in LDS (`lds` + 4096 words), as in the forward strided pass
    constexpr bool LDSTW = LOGN == 16 && IM == 1;
    Tw *ldstw = reinterpret_cast<Tw *>(lds + 4096);
    uint64_t x[16];
    if (RB > 0)
    {
        const uint32_t g = tid & (G - 1);
        const uint32_t th = tid >> GB;
        if (LDSTW)
        {
            ldstw[tid] = tw[tid];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            uint32_t t_ = (th << 4) | (uint32_t)j;
            x[j] = row[(t_ << 8) + g];
        }
        if (LDSTW)
        {
            lds_barrier();
        }
#pragma unroll
        for (int s = R1 - 1; s >= 4; --s)
        {
            const int half = 1 << (R1 - 1 - s);
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                if (!(j & half))
                {
                    uint32_t t_ = (th << 4) | (uint32_t)j;
                    Tw t = LDSTW ? ldstw[(1u << s) + (t_ >> (R1 - s))] : tw[(1u << s) + (t_ >> (R1 - s))];
                    gs_bfly_im<IM>(x[j], x[j + half], t.w, t.wq, q, q2, ((R1 - 1 - s) & 3) == 3); // stage number in this pass
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            lds[phys_strided<GB>((th << (GB + 4)) | ((uint32_t)j << GB) | g)] = x[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            x[j] = lds[phys_strided<GB>((uint32_t)j * 256u + tid)];
        }
    }
    else
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            uint32_t e = (uint32_t)j * 256u + tid;
            x[j] = row[((e >> GB) << 8) + (e & (G - 1))];
        }
    }
#pragma unroll
    for (int u = 3; u >= 1; --u)
    {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (!(j & half))
            {
                Tw t = tw[(1u << u) + (uint32_t)(j >> (4 - u))];
                gs_bfly_im<IM>(x[j], x[j + half], t.w, t.wq, q, q2, ((RB + 3 - u) & 3) == 3);
            }
        }
    }
    {
        const Tw ninv = pc->ninv;
        const Tw ninv_w1 = pc->ninv_w1;
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            if (IM >= 2)
            {
                gs_bfly_last_fp<IM == 3>(x[j], x[j + 8], fp_from_u64(ninv.w), fp_from_u64(ninv_w1.w), u2d(q), u2d(q2));
            }
            else if (LZ)
            {
                gs_bfly_last_lazy8(x[j], x[j + 8], ninv, ninv_w1, q, q2);
            }
            else
            {
                gs_bfly_last(x[j], x[j + 8], ninv, ninv_w1, q, q2);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        uint32_t e = (uint32_t)j * 256u + tid;
        // exact: below 2q; LZ: below 4q, with 2^64 - 2q = (2^64 - 4q) / 2 + 2^63
        row[((e >> GB) << 8) + (e & (G - 1))] = IM >= 2 ? fp_to_canonical(u2d(x[j]), u2d(q), u2d(q2))
                                                        : (LZ ? csub_sign(csub_sign(x[j], (q2 >> 1) | 0x8000000000000000ull), q) : csub(x[j], q));
    }
}

template <int LOGN, int IM = 0>
__global__ __launch_bounds__(256, IM == 1 ? 4 : 5) void ntt_inv_strided(NttArgs a)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ uint64_t lds[4096 + ((LOGN == 16 && IM == 1) ? 512 : 0)]; // the exchange buffer (+ phase B's twiddles, inv_strided_tile)
    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    const uint32_t srow = w / TPR; // over n_poly * Lsel selected rows
    const uint32_t si = srow % a.Lsel;
    const uint32_t r = __builtin_amdgcn_readfirstlane(a.sel.idx[si]);
    const uint32_t prime = __builtin_amdgcn_readfirstlane(a.selp.idx[si]);
    inv_strided_tile<LOGN, IM>(a.data + ((size_t)((srow / a.Lsel) * a.L + r) << LOGN), tile, a.tw + ((size_t)prime << LOGN), a.pc + prime, lds,
                               threadIdx.x);
}

// =====================================================================================================
// single-launch transform: both passes of a row meet in one XCD's L2
// =====================================================================================================
// The two passes exchange the whole row (N x 8 B) and no CU can hold a 512 KiB row, but an XCD's
// 4 MiB L2 can.  This persistent kernel therefore keeps all work on one row inside one XCD: every
// workgroup reads the id of the XCD it runs on (HW_REG_XCC_ID) and pulls (row, pass, tile) items from
// that XCD's own queue with one fetch-add per item; rows are claimed from a global ticket by the
// workgroup that draws tile 0 of a step.  Queue order per XCD, in steps of 2*TPR items:
//     [first-pass tiles of step s][second-pass tiles of step s - DELAY]
// so a row's second pass follows its first pass by DELAY rows: long enough that the first pass has
// normally finished, short enough that the row is still in L2.  A first-pass tile publishes with
// "s_waitcnt vmcnt(0); barrier; one lane bumps done[row]" (its stores have then reached L2); a
// second-pass tile waits for done[row] == TPR, drops its CU's L1 (agent-scope acquire) and reads the
// row from L2.  HBM then sees one read and one write per coefficient instead of two of each.
//
// Correctness does not depend on dispatch order or placement:
//  * a row's items are only ever handed out by the queue of the XCD that claimed the row, so producer
//    and consumer share an L2 by construction (not by an assumption about the dispatcher);
//  * a workgroup blocks only on items that sit EARLIER in its own queue, i.e. on workgroups that are
//    already running and never block themselves (first-pass tiles, the row claim), so there is no
//    co-residency requirement and no deadlock for any grid size >= 1;
//  * every spin is bounded; a timeout sets CoopState::error (checked by the host) instead of hanging.
struct CoopState
{
    uint32_t next_row; // global row ticket
    uint32_t error;    // non-zero after a spin timeout
    uint32_t pad0[30];
    struct
    {
        uint32_t head;
        uint32_t pad[31];
    } q[8];
    // followed by: uint32_t done[rows]; uint32_t rowmap[8][steps_cap]
};

struct CoopArgs
{
    CoopState *st;
    uint32_t *done;     // [rows] first-pass tiles completed
    uint32_t *rowmap;   // [8][steps_cap]: 0 = unset, 1 = end of work, r + 2 = row r
    uint32_t rows;      // n_poly * L
    uint32_t steps_cap;
    uint32_t delay;     // steps between a row's first and second pass in the queue
};

#define MOAI_SPIN_LIMIT (1u << 22)

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ uint32_t ld_relaxed(uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int LOGN, bool INV, int WPS>
__global__ __launch_bounds__(256, WPS) void ntt_coop(NttArgs a, CoopArgs c)
{
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048];
    __shared__ uint32_t sh[2];
    uint64_t *lds = reinterpret_cast<uint64_t *>(lds2);
    const uint32_t tid0 = threadIdx.x;
    const uint32_t xcc = xcc_id();
    uint32_t *rowmap = c.rowmap + (size_t)xcc * c.steps_cap;

    for (;;)
    {
        __syncthreads(); // LDS tile and sh[] of the previous item are dead
        // opaque copy of the thread index: stops the compiler from hoisting the ~70 per-thread address
        // computations of both tile bodies out of this loop and spilling them
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));
        if (tid == 0)
        {
            const uint32_t it = __hip_atomic_fetch_add(&c.st->q[xcc].head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t step = it / (2 * TPR);
            const uint32_t rem = it % (2 * TPR);
            const uint32_t pass = rem / TPR;
            const uint32_t tile = rem % TPR;
            uint32_t v = 0; // 0 = empty slot, 1 = end of work, r + 2 = row r
            if (!(pass == 1 && step < c.delay))
            {
                const uint32_t sr = pass ? step - c.delay : step;
                if (sr >= c.steps_cap)
                {
                    v = 1;
                }
                else if (pass == 0 && tile == 0)
                {
                    uint32_t r = __hip_atomic_fetch_add(&c.st->next_row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = r < c.rows ? r + 2 : 1;
                    __hip_atomic_store(&rowmap[sr], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                else
                {
                    uint32_t i = 0;
                    while ((v = ld_relaxed(&rowmap[sr])) == 0 && ++i < MOAI_SPIN_LIMIT)
                    {
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (v == 0)
                    {
                        __hip_atomic_store(&c.st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = 1;
                    }
                }
                if (v >= 2 && pass == 1)
                {
                    // wait for the TPR first-pass tiles of this row, then drop this CU's L1
                    uint32_t *d = &c.done[v - 2];
                    uint32_t i = 0;
                    while (ld_relaxed(d) < TPR && ++i < MOAI_SPIN_LIMIT)
                    {
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (i >= MOAI_SPIN_LIMIT)
                    {
                        __hip_atomic_store(&c.st->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = 1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            sh[0] = v;
            sh[1] = (pass << 16) | tile;
        }
        __syncthreads();
        // wave-uniform by construction: keep them in SGPRs so that row / twiddle bases stay scalar
        const uint32_t v = __builtin_amdgcn_readfirstlane(sh[0]);
        const uint32_t pt = __builtin_amdgcn_readfirstlane(sh[1]);
        const uint32_t pass = pt >> 16;
        const uint32_t tile = pt & 0xffffu;
        if (v == 0)
        {
            continue;
        }
        if (v == 1)
        {
            if (pass == 1)
            {
                return; // the rows behind this point of the queue do not exist
            }
            continue;
        }
        // tickets run prime-major (polynomial index fastest) so that all XCDs work under the same prime
        // at the same time and its twiddle tables stay resident in every L2
        const uint32_t ticket = v - 2;
        const uint32_t prow = (ticket % a.n_poly) * a.L + ticket / a.n_poly;
        const uint32_t prime = a.rows.idx[prow % a.L];
        const Tw *tw = a.tw + ((size_t)prime << LOGN);
        const PrimeConst *pc = a.pc + prime;
        uint64_t *rowp = a.data + ((size_t)prow << LOGN);
        if (pass == 0)
        {
            if (INV)
            {
                inv_contig_tile<LOGN>(rowp, tile, tw, pc->q, pc->q2, lds2, tid, a.twb + (size_t)prime * ((size_t)TPR * 15 * 256));
            }
            else
            {
                fwd_strided_tile<LOGN>(rowp, rowp, tile, tw, pc->q, pc->q2, lds, tid);
            }
            // publish: every wave's stores have reached L2, then one lane bumps the row counter
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
            {
                __hip_atomic_fetch_add(&c.done[ticket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        else
        {
            if (INV)
            {
                inv_strided_tile<LOGN>(rowp, tile, tw, pc, lds, tid);
            }
            else
            {
                fwd_contig_tile<LOGN>(rowp, tile, tw, pc->q, pc->q2, lds2, tid, a.twb + (size_t)prime * ((size_t)TPR * 15 * 256));
            }
        }
    }
}

// =====================================================================================================
// small transforms (N <= 2048): one workgroup per RNS row, whole row in LDS
// =====================================================================================================
template <bool INV>
__global__ __launch_bounds__(256) void ntt_small(NttArgs a, int logn)
{
    __shared__ uint64_t lds[2048];
    const uint32_t n = 1u << logn;
    const uint32_t prow = blockIdx.x;
    const uint32_t prime = a.rows.idx[prow % a.L];
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << logn);
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q;
    const uint64_t q2 = pc->q2;
    uint64_t *__restrict__ row = a.data + ((size_t)prow << logn);
    const uint32_t tid = threadIdx.x;

    for (uint32_t i = tid; i < n; i += 256)
    {
        lds[i] = row[i];
    }
    __syncthreads();
    if (!INV)
    {
        for (int s = 0; s < logn; ++s)
        {
            const int lg = logn - 1 - s; // log2(gap)
            for (uint32_t bf = tid; bf < (n >> 1); bf += 256)
            {
                uint32_t blk = bf >> lg;
                uint32_t off = bf & ((1u << lg) - 1);
                uint32_t i0 = (blk << (lg + 1)) | off;
                Tw t = tw[(1u << s) + blk];
                uint64_t x = lds[i0], y = lds[i0 + (1u << lg)];
                ct_bfly(x, y, t.w, t.wq, q, q2);
                lds[i0] = x;
                lds[i0 + (1u << lg)] = y;
            }
            __syncthreads();
        }
        for (uint32_t i = tid; i < n; i += 256)
        {
            row[i] = csub(csub(lds[i], q2), q);
        }
    }
    else
    {
        for (int s = logn - 1; s >= 1; --s)
        {
            const int lg = logn - 1 - s;
            for (uint32_t bf = tid; bf < (n >> 1); bf += 256)
            {
                uint32_t blk = bf >> lg;
                uint32_t off = bf & ((1u << lg) - 1);
                uint32_t i0 = (blk << (lg + 1)) | off;
                Tw t = tw[(1u << s) + blk];
                uint64_t x = lds[i0], y = lds[i0 + (1u << lg)];
                gs_bfly(x, y, t.w, t.wq, q, q2);
                lds[i0] = x;
                lds[i0 + (1u << lg)] = y;
            }
            __syncthreads();
        }
        const Tw ninv = pc->ninv;
        const Tw ninv_w1 = pc->ninv_w1;
        for (uint32_t bf = tid; bf < (n >> 1); bf += 256)
        {
            uint64_t x = lds[bf], y = lds[bf + (n >> 1)];
            gs_bfly_last(x, y, ninv, ninv_w1, q, q2);
            lds[bf] = x;
            lds[bf + (n >> 1)] = y;
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256)
        {
            row[i] = csub(lds[i], q);
        }
    }
}

// =====================================================================================================
// one radix-2 stage over global memory (debug / cross-check path: MOAI_NTT_NAIVE=1)
// =====================================================================================================
template <bool INV>
__global__ __launch_bounds__(256) void ntt_stage_global(NttArgs a, int logn, int s, int last)
{
    const uint32_t half_n = 1u << (logn - 1);
    const uint32_t bf = blockIdx.x * 256u + threadIdx.x;
    const uint32_t prow = blockIdx.y;
    if (bf >= half_n)
    {
        return;
    }
    const uint32_t prime = a.rows.idx[prow % a.L];
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << logn);
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q;
    const uint64_t q2 = pc->q2;
    uint64_t *__restrict__ row = a.data + ((size_t)prow << logn);
    const int lg = logn - 1 - s;
    uint32_t blk = bf >> lg;
    uint32_t off = bf & ((1u << lg) - 1);
    uint32_t i0 = (blk << (lg + 1)) | off;
    uint32_t i1 = i0 + (1u << lg);
    uint64_t x = row[i0], y = row[i1];
    if (!INV)
    {
        Tw t = tw[(1u << s) + blk];
        ct_bfly(x, y, t.w, t.wq, q, q2);
        if (last)
        {
            x = csub(csub(x, q2), q);
            y = csub(csub(y, q2), q);
        }
    }
    else if (s > 0)
    {
        Tw t = tw[(1u << s) + blk];
        gs_bfly(x, y, t.w, t.wq, q, q2);
    }
    else
    {
        gs_bfly_last(x, y, pc->ninv, pc->ninv_w1, q, q2);
        x = csub(x, q);
        y = csub(y, q);
    }
    row[i0] = x;
    row[i1] = y;
}

} // namespace moai
