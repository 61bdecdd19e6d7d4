// ntt_kernels.cuh -- negacyclic NTT / INTT for gfx950.
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
#include "modarith.cuh"

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
    const PrimeConst *pc;      // [k]
    RowMap rows;               // row r -> prime
    uint32_t L;
    uint32_t n_poly;
    uint32_t total_work;       // grid size
};

// =====================================================================================================
// forward, strided pass: stages 0 .. LOGN-9
// =====================================================================================================
template <int LOGN>
__global__ __launch_bounds__(256) void ntt_fwd_strided(NttArgs a)
{
    constexpr int R1 = LOGN - 8;
    constexpr int RB = R1 - 4;
    constexpr int GB = 12 - R1;
    constexpr uint32_t G = 1u << GB;
    constexpr uint32_t TPR = 256u / G;
    __shared__ uint64_t lds[RB > 0 ? 4096 : 1];

    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    const uint32_t prow = w / TPR;
    const uint32_t prime = a.rows.idx[prow % a.L];
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << LOGN);
    const uint64_t q = a.pc[prime].q;
    const uint64_t q2 = a.pc[prime].q2;
    uint64_t *__restrict__ row = a.data + ((size_t)prow << LOGN) + tile * G;
    const uint32_t tid = threadIdx.x;

    uint64_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        uint32_t e = (uint32_t)j * 256u + tid;
        x[j] = row[((e >> GB) << 8) + (e & (G - 1))];
    }
    // phase A: top four bits of t live in the register index
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (!(j & half))
            {
                Tw t = tw[(1u << u) + (uint32_t)(j >> (4 - u))];
                ct_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
            }
        }
    }
    if (RB > 0)
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            lds[phys_strided<GB>((uint32_t)j * 256u + tid)] = x[j];
        }
        __syncthreads();
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
                if (!(j & half))
                {
                    uint32_t t_ = (th << 4) | (uint32_t)j;
                    Tw t = tw[(1u << s) + (t_ >> (R1 - s))];
                    ct_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            uint32_t t_ = (th << 4) | (uint32_t)j;
            row[(t_ << 8) + g] = x[j];
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

// =====================================================================================================
// forward, contiguous pass: stages LOGN-8 .. LOGN-1 on 16 consecutive 256-blocks; writes canonical
// =====================================================================================================
template <int LOGN>
__global__ __launch_bounds__(256) void ntt_fwd_contig(NttArgs a)
{
    constexpr int R1 = LOGN - 8;
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048];
    uint64_t *lds = reinterpret_cast<uint64_t *>(lds2);

    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t pol = w % a.n_poly;
    const uint32_t rest = w / a.n_poly;
    const uint32_t tile = rest % TPR;
    const uint32_t r = rest / TPR;
    const uint32_t prime = a.rows.idx[r];
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << LOGN);
    const uint64_t q = a.pc[prime].q;
    const uint64_t q2 = a.pc[prime].q2;
    uint64_t *__restrict__ base = a.data + (((size_t)pol * a.L + r) << LOGN) + ((size_t)tile << 12);
    const uint32_t tid = threadIdx.x;
    const uint32_t b = tid >> 4;
    const uint32_t tl = tid & 15u;
    const uint32_t blk = (tile << 4) + b;

    uint64_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        x[j] = base[(b << 8) | ((uint32_t)j << 4) | tl];
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
                Tw t = tw[(1u << (R1 + u)) + (blk << u) + (uint32_t)(j >> (4 - u))];
                ct_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        lds[phys_contig((b << 8) | ((uint32_t)j << 4) | tl)] = x[j];
    }
    __syncthreads();
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
                uint32_t t_ = (tl << 4) | (uint32_t)j;
                Tw t = tw[(1u << (R1 + u)) + (blk << u) + (t_ >> (8 - u))];
                ct_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
            }
        }
    }
    __syncthreads(); // every thread has read its row before anyone overwrites the tile
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        ulonglong2 v;
        v.x = csub(csub(x[2 * c], q2), q);
        v.y = csub(csub(x[2 * c + 1], q2), q);
        lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))] = v;
    }
    __syncthreads();
    ulonglong2 *__restrict__ out2 = reinterpret_cast<ulonglong2 *>(base);
#pragma unroll
    for (int it = 0; it < 8; ++it)
    {
        uint32_t ch = (uint32_t)it * 256u + tid;
        uint32_t rr = ch >> 3;
        out2[ch] = lds2[(rr << 3) | ((ch & 7u) ^ (rr & 7u))];
    }
}

// =====================================================================================================
// inverse, contiguous pass: stages LOGN-1 .. LOGN-8 (gap 1 .. 128); lazy [0,2q) out
// =====================================================================================================
template <int LOGN>
__global__ __launch_bounds__(256) void ntt_inv_contig(NttArgs a)
{
    constexpr int R1 = LOGN - 8;
    constexpr uint32_t TPR = 1u << (LOGN - 12);
    __shared__ ulonglong2 lds2[2048];
    uint64_t *lds = reinterpret_cast<uint64_t *>(lds2);

    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t pol = w % a.n_poly;
    const uint32_t rest = w / a.n_poly;
    const uint32_t tile = rest % TPR;
    const uint32_t r = rest / TPR;
    const uint32_t prime = a.rows.idx[r];
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << LOGN);
    const uint64_t q = a.pc[prime].q;
    const uint64_t q2 = a.pc[prime].q2;
    uint64_t *__restrict__ base = a.data + (((size_t)pol * a.L + r) << LOGN) + ((size_t)tile << 12);
    const uint32_t tid = threadIdx.x;
    const uint32_t b = tid >> 4;
    const uint32_t tl = tid & 15u;
    const uint32_t blk = (tile << 4) + b;

    const ulonglong2 *__restrict__ in2 = reinterpret_cast<const ulonglong2 *>(base);
#pragma unroll
    for (int it = 0; it < 8; ++it)
    {
        uint32_t ch = (uint32_t)it * 256u + tid;
        uint32_t rr = ch >> 3;
        lds2[(rr << 3) | ((ch & 7u) ^ (rr & 7u))] = in2[ch];
    }
    __syncthreads();
    const uint32_t myrow = tid;
    uint64_t x[16];
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        ulonglong2 v = lds2[(myrow << 3) | ((uint32_t)c ^ (myrow & 7u))];
        x[2 * c] = v.x;
        x[2 * c + 1] = v.y;
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
                uint32_t t_ = (tl << 4) | (uint32_t)j;
                Tw t = tw[(1u << (R1 + u)) + (blk << u) + (t_ >> (8 - u))];
                gs_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
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
    __syncthreads();
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
                Tw t = tw[(1u << (R1 + u)) + (blk << u) + (uint32_t)(j >> (4 - u))];
                gs_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        base[(b << 8) | ((uint32_t)j << 4) | tl] = x[j];
    }
}

// =====================================================================================================
// inverse, strided pass: stages LOGN-9 .. 0, N^-1 folded into stage 0; writes canonical
// =====================================================================================================
template <int LOGN>
__global__ __launch_bounds__(256) void ntt_inv_strided(NttArgs a)
{
    constexpr int R1 = LOGN - 8;
    constexpr int RB = R1 - 4;
    constexpr int GB = 12 - R1;
    constexpr uint32_t G = 1u << GB;
    constexpr uint32_t TPR = 256u / G;
    __shared__ uint64_t lds[RB > 0 ? 4096 : 1];

    const uint32_t w = xcd_remap(blockIdx.x, a.total_work);
    const uint32_t tile = w % TPR;
    const uint32_t prow = w / TPR;
    const uint32_t prime = a.rows.idx[prow % a.L];
    const Tw *__restrict__ tw = a.tw + ((size_t)prime << LOGN);
    const PrimeConst *pc = a.pc + prime;
    const uint64_t q = pc->q;
    const uint64_t q2 = pc->q2;
    uint64_t *__restrict__ row = a.data + ((size_t)prow << LOGN) + tile * G;
    const uint32_t tid = threadIdx.x;

    uint64_t x[16];
    if (RB > 0)
    {
        const uint32_t g = tid & (G - 1);
        const uint32_t th = tid >> GB;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            uint32_t t_ = (th << 4) | (uint32_t)j;
            x[j] = row[(t_ << 8) + g];
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
                    Tw t = tw[(1u << s) + (t_ >> (R1 - s))];
                    gs_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
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
                gs_bfly(x[j], x[j + half], t.w, t.wq, q, q2);
            }
        }
    }
    {
        const Tw ninv = pc->ninv;
        const Tw ninv_w1 = pc->ninv_w1;
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            gs_bfly_last(x[j], x[j + 8], ninv, ninv_w1, q, q2);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        uint32_t e = (uint32_t)j * 256u + tid;
        row[((e >> GB) << 8) + (e & (G - 1))] = csub(x[j], q);
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
