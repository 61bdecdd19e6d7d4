// common.h -- shared declarations of the MI355X evaluator library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>
#include <map>
#include <string>
#include <vector>

#include "../../include/moai_hip.h"

namespace moai {

// One precomputed power of psi with its Shoup quotient floor(w * 2^64 / q)
// (the role of SEAL's MultiplyUIntModOperand, SEAL/util/uintarithsmallmod.h:255-286).
struct alignas(16) Tw
{
    uint64_t w;
    uint64_t wq;
};

// Per-prime constants, one 128-byte record per prime in device memory.
struct alignas(16) PrimeConst
{
    uint64_t q;        // modulus
    uint64_t q2;       // 2q
    uint64_t cr0;      // floor(2^128 / q), low word   (Modulus::const_ratio, SEAL/modulus.cpp:36-77)
    uint64_t cr1;      //                   high word
    Tw ninv;           // N^-1 mod q                     (NTTTables::inv_degree_modulo, ntt.cpp:290-296)
    Tw ninv_w1;        // N^-1 * inv_tw[1] mod q: last inverse stage with the scaling folded in
    uint64_t qd;       // bit pattern of (double) q          } FP64 arithmetic modes (modarith.hip.h),
    uint64_t qinv;     // bit pattern of RN(1.0 / q)         } meaningful when fp_mode != 0
    uint64_t fp_mode;  // 0: integer only; M_FPN or M_FPR
    uint64_t nq;       // 2^64 - q     } M_LAZY8 (modarith.hip.h): stored, not derived on the device, so that the compiler
    uint64_t n4q;      // 2^64 - 4 q   } keeps  t * (-q)  a multiply-add chain instead of a 64-bit subtract
    uint64_t pad[3];
};

// Row -> context-prime map passed by value to kernels (rows of one RNS polynomial).
struct RowMap
{
    // 32-bit entries: a kernel indexes these kernarg arrays with a value it computes (uniform over the workgroup), and gfx950 has
    // scalar loads for dwords only -- 16-bit entries came through a VECTOR load and a full s_waitcnt vmcnt(0) at the head of every
    // workgroup (and of every term of the plaintext dot products)
    uint32_t idx[MOAI_MAX_RNS];
};

int set_error(int code, const char *fmt, ...);

#define MOAI_HIP_CHECK(expr)                                                                   \
    do                                                                                         \
    {                                                                                          \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
        {                                                                                      \
            return ::moai::set_error(MOAI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                     __FILE__, __LINE__);                                      \
        }                                                                                      \
    } while (0)

#define MOAI_LAUNCH_CHECK() MOAI_HIP_CHECK(hipGetLastError())

// Stream audit (debug; MOAI_STREAM_AUDIT=1 or moai_debug_stream_audit(1)).  A caller that keeps device blocks in a
// stream-ordered cache (the seal:: shim's util::DevicePool) labels every block with the stream it belongs to
// (moai_debug_block_label); every entry point that enqueues work then checks that each device pointer it was handed
// lies in a block labelled with the stream the work is enqueued on, and is not a block that has been released.
// An unlabelled pointer (plain moai_malloc memory, host arrays) is not checked.
extern std::atomic<int> g_stream_audit;
int audit_ptrs(const char *fn, const void *stream, std::initializer_list<const void *> ptrs);
#define MOAI_AUDIT(stream, ...)                                                             \
    do                                                                                      \
    {                                                                                       \
        if (::moai::g_stream_audit.load(std::memory_order_relaxed))                         \
        {                                                                                   \
            int _arc = ::moai::audit_ptrs(__func__, (const void *)(stream), { __VA_ARGS__ }); \
            if (_arc)                                                                       \
            {                                                                               \
                return _arc;                                                                \
            }                                                                               \
        }                                                                                   \
    } while (0)

// row-per-block launches put the row count in gridDim.y, which HIP limits to 65535
#define MOAI_CHECK_GRID_ROWS(rows)                                                                                   \
    do                                                                                                               \
    {                                                                                                                \
        if ((size_t)(rows) > 65535u)                                                                                 \
        {                                                                                                            \
            return ::moai::set_error(MOAI_EINVAL, "batch too large for one launch: %zu rows (limit 65535)", (size_t)(rows)); \
        }                                                                                                            \
    } while (0)

} // namespace moai

// The opaque context of the C ABI.
struct moai_ctx
{
    int device = 0;
    int logn = 0;
    size_t n = 0;
    size_t k = 0;
    int num_cu = 256;
    std::vector<uint64_t> primes;      // host copies
    std::vector<uint64_t> roots;       // psi per prime
    std::vector<moai::PrimeConst> pc_host;
    moai::Tw *fwd_tw = nullptr;        // [k][N]: index m+i = psi^bitrev(m+i)        (ntt.cpp:269-278)
    moai::Tw *inv_tw = nullptr;        // [k][N]: index m+i = psi^-bitrev(m+i) (same indexing as fwd)
    // the twiddles of the last four stages of the contiguous pass, re-ordered so that the 256 threads of a
    // workgroup read consecutive entries: [k][N/4096 tiles][15 slots][256 threads] (N >= 4096 only)
    moai::Tw *fwd_twb = nullptr;
    moai::Tw *inv_twb = nullptr;
    // forward tables for the FP64 modes: {double w, double RN(w/q)} per entry, same indexing as fwd_tw /
    // fwd_twb; rows of primes that have no FP64 mode are zero
    moai::Tw *fwd_twf = nullptr;
    moai::Tw *fwd_twfb = nullptr;
    // the inverse tables in the same form (inv_tw / inv_twb indexing): the inverse transform of rows below 2^51
    moai::Tw *inv_twf = nullptr;
    moai::Tw *inv_twfb = nullptr;
    // the same powers as plain doubles, 8 bytes per entry (fwd_tw indexing): the contiguous key-switch kernel is
    // bound by twiddle fetches and takes its quotient estimate from RN(1/q) instead (modarith.hip.h ct_bfly_fp1)
    double *fwd_twf1 = nullptr;
    moai::PrimeConst *pc = nullptr;    // [k]
    // inv_q_last_mod_q[l][i] = q_l^-1 mod q_i as Shoup operands, l in [1,k), i < l  (rns.cpp:769-775)
    moai::Tw *inv_qlast = nullptr;     // [k][k]
    std::vector<moai::Tw> inv_qlast_host;
    // workspace arenas, one per stream so that concurrent callers (MOAI's OpenMP threads, each on its
    // own stream) never share scratch memory
    struct Arena
    {
        void *ptr = nullptr;
        size_t bytes = 0;
    };
    std::map<void *, Arena> ws;
    // Galois permutation tables, built lazily per element (galois.cpp:18-51)
    std::vector<uint32_t *> galois_tables; // [N] entries index (elt-1)>>1, device pointers
    // CKKSEncoder tables (SEAL/ckks.cpp:13-76), built on the first moai_ckks_encode
    std::vector<uint32_t> ckks_index_map;     // matrix_reps_index_map_ [N]
    std::vector<double> ckks_inv_roots_host;  // inv_root_powers_ [N] (re, im)
    uint32_t *ckks_src_map = nullptr;         // device, inverse of ckks_index_map
    double *ckks_inv_roots = nullptr;         // device copy
    // Key layouts (moai_key_trim): a key-switch key registered here is [digits][2][rows][N] with the special prime's row last
    // instead of the reference's [k-1][2][k][N]; unregistered pointers have the reference's layout.
    struct KeyLayout
    {
        uint32_t digits, rows;
    };
    std::map<const void *, KeyLayout> key_layouts;
    void *mutex = nullptr;
    // serialises the enqueue of multi-kernel operations that share a stream's workspace arena: callers
    // on different host threads may target the same stream (MOAI's OpenMP loops do)
    void *op_mutex = nullptr;
};
