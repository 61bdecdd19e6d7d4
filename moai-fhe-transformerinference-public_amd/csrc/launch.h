// launch.h -- host-side entry points shared between the translation units of the library.
#pragma once
#include "common.h"

namespace moai {

// a tuning knob: the value set through moai_set_tuning, else the environment variable of that name, else dflt
long tuning(const char *name, long dflt);
// operation census for the end-to-end bench (moai_op_trace): counts `units` (polynomials, ciphertexts or products, as the
// entry point's own batch argument counts them) per (entry point, level); a relaxed atomic load when it is off
void trace_op(const char *name, size_t L, size_t units);
bool noguard_ok(uint64_t q);
// makes the context's device current for the calling thread (contexts of several devices may live in one process);
// every operation entry point calls it before it allocates or launches
int enter_device(const moai_ctx *c);
// arithmetic mode (modarith.hip.h M_*) of the forward transform under a context prime
int ntt_mode(const moai_ctx *c, uint32_t prime);
int make_rowmap(const moai_ctx *c, size_t L, const uint32_t *prime_index, RowMap *out);
// src (inverse only): polynomial p's row r is read from src row p * src_stride_rows + src_off_rows + r, the result lands in
// `data` [n_poly][L][N] -- the inverse transform of a slice of a larger layout without copying the slice first
int ntt_launch(moai_ctx *c, uint64_t *data, size_t n_poly, size_t L, const RowMap &rows, bool inverse,
               hipStream_t s, const uint64_t *src = nullptr, size_t src_stride_rows = 0, size_t src_off_rows = 0);
// returns the context workspace grown to at least `bytes` (grows only outside stream capture)
int workspace(moai_ctx *c, size_t bytes, hipStream_t s, void **out);
// headroom: allocate max(1.25 x bytes, 1.5 x the current size) when the arena has to grow
int reserve_for_stream(moai_ctx *c, void *stream, size_t bytes, void **out, bool headroom);
// device pointer to the Galois permutation table of `elt` (built on first use)
int galois_table(moai_ctx *c, uint32_t elt, hipStream_t s, const uint32_t **out);
// out [batch][L][N] = the Galois permutation of polynomial 0 of every ciphertext of in [batch][2][L][N]
int galois_permute_c0(moai_ctx *c, const uint64_t *in, uint64_t *out, size_t batch, size_t L, uint32_t galois_elt, hipStream_t s);

} // namespace moai
