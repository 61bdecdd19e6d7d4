/*
 * moai_hip.h -- C ABI of the MI355X-native RNS-CKKS evaluator hot path.
 *
 * This is the drop-in boundary: everything MOAI's seal::Evaluator needs for ciphertext arithmetic,
 * as plain C entry points over device pointers.  The host-side seal:: shim (and any other binding:
 * ctypes, cgo, JNI) sits above this header; nothing below it is visible to callers.
 *
 * Each entry point names the reference interface it replaces.  Paths are relative to the
 * reference checkout; SEAL/ = thirdparty/SEAL-4.1-bs/native/src/seal/.
 *
 * Conventions
 *   - All residue data is uint64_t in the reference's own layout [poly][rns prime][coefficient]
 *     (SEAL/ciphertext.h:337-349), with an optional leading batch dimension.  Pointers are DEVICE
 *     pointers unless a parameter is documented "host".
 *   - Inputs are canonical residues in [0, q_i); outputs are canonical residues, bit-identical to
 *     the reference CPU path on the same inputs.
 *   - A data level with L primes uses context primes [0, L); the key level is all k primes and
 *     prime k-1 is the special prime (SEAL/context.cpp:455-522).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue work and
 *     return; they never synchronise the device.  Workspace is drawn from the context's arena,
 *     which grows (hipMalloc) only outside stream capture -- call moai_ctx_reserve first when
 *     capturing into a hipGraph.
 *   - Return value: 0 on success, a negative MOAI_E* code otherwise; moai_last_error() gives the
 *     message (thread local).  No exceptions cross this boundary: the C++ shim re-creates the
 *     reference's exceptions (std::invalid_argument / std::logic_error / std::out_of_range)
 *     BEFORE enqueueing, as the reference raises them synchronously.
 */
#ifndef MOAI_HIP_H
#define MOAI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOAI_OK 0
#define MOAI_EINVAL (-1)   /* bad argument (the shim maps it to std::invalid_argument) */
#define MOAI_ELOGIC (-2)   /* unsupported parameters (std::logic_error)                */
#define MOAI_ERANGE (-3)   /* index out of range (std::out_of_range)                   */
#define MOAI_EHIP (-4)     /* HIP runtime failure                                      */
#define MOAI_ENOMEM (-5)

#define MOAI_MAX_RNS 64    /* largest number of RNS rows one call may address */

typedef struct moai_ctx moai_ctx;

const char *moai_last_error(void);
int moai_version(void);

/* ---- context: per-prime tables on the device -------------------------------------------------
 * Replaces SEALContext's per-level ContextData precomputation for the hot path
 * (SEAL/context.cpp:422-522, NTTTables::initialize SEAL/util/ntt.cpp:241-300, RNSTool
 * inv_q_last_mod_q SEAL/util/rns.cpp:769-775, Modulus::const_ratio SEAL/modulus.cpp:36-77).
 * Tables are shared per prime, not duplicated per level.  primes: host, k entries, each an NTT
 * prime (= 1 mod 2N) below 2^61.  coeff_count_power in [1, 16].
 */
int moai_ctx_create(int coeff_count_power, const uint64_t *primes, size_t k, int device, moai_ctx **out);
void moai_ctx_destroy(moai_ctx *ctx);
int moai_ctx_reserve(moai_ctx *ctx, size_t workspace_bytes); /* arena of the default (NULL) stream */
/* arena of `stream`: call before capturing that stream into a hipGraph (arenas never grow under capture) */
int moai_ctx_reserve_stream(moai_ctx *ctx, void *stream, size_t workspace_bytes);
size_t moai_ctx_coeff_count(const moai_ctx *ctx);
size_t moai_ctx_prime_count(const moai_ctx *ctx);
/* psi = minimal primitive 2N-th root of prime i (NTTTables::get_root) */
uint64_t moai_ctx_root(const moai_ctx *ctx, size_t prime);

/* ---- memory / streams (the device arena behind seal::DynArray / MemoryPool, SEAL/dynarray.h) -- */
int moai_malloc(void **dptr, size_t bytes);
/* page-locked host memory: copies from it are asynchronous for real (staging buffers of host-side callers) */
int moai_host_malloc(void **hptr, size_t bytes);
int moai_host_free(void *hptr);
int moai_free(void *dptr);
int moai_memcpy_h2d(void *dst, const void *src_host, size_t bytes, void *stream);
int moai_memcpy_d2h(void *dst_host, const void *src, size_t bytes, void *stream);
int moai_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
/* n <= 64 separate device blocks of `words` 64-bit words (even) each <-> one packed array [n][words], in ONE launch: what the shim's
 * call combiner does around a batched operation (it used to enqueue one copy per caller and direction).  src / dst: HOST arrays of
 * device pointers.  No counterpart in the reference (plumbing of SURVEY 8 row f4). */
int moai_gather_blocks(moai_ctx *ctx, const uint64_t *const *src, uint64_t *packed, size_t n, size_t words, void *stream);
int moai_scatter_blocks(moai_ctx *ctx, const uint64_t *packed, uint64_t *const *dst, size_t n, size_t words, void *stream);
int moai_memset_zero(void *dst, size_t bytes, void *stream);
int moai_stream_create(void **stream);
int moai_stream_destroy(void *stream);
int moai_stream_sync(void *stream);

/* ---- negacyclic NTT ----------------------------------------------------------------------------
 * data: uint64[n_poly][L][N] in place.  Row (p, r) is transformed under context prime
 * prime_index[r] (host array of L entries) or prime r when prime_index is NULL.
 * forward:  natural order in (values in [0, 4q), as the reference's lazy transform accepts) -> bit-reversed
 *           order out, canonical [0,q)
 *           (ntt_negacyclic_harvey, SEAL/util/ntt.cpp:408-437; Evaluator::transform_to_ntt_inplace
 *           SEAL/evaluator.cpp:2468-2514)
 * inverse:  bit-reversed in -> natural out, scaled by N^-1, canonical
 *           (inverse_ntt_negacyclic_harvey, SEAL/util/ntt.cpp:453-475;
 *           Evaluator::transform_from_ntt_inplace SEAL/evaluator.cpp:2516-2561)
 */
int moai_ntt_forward(moai_ctx *ctx, uint64_t *data, size_t n_poly, size_t L, const uint32_t *prime_index,
                     void *stream);
int moai_ntt_inverse(moai_ctx *ctx, uint64_t *data, size_t n_poly, size_t L, const uint32_t *prime_index,
                     void *stream);

/* ---- element-wise RNS polynomial arithmetic (SEAL/util/polyarithsmallmod.cpp) -------------------
 * All operate on uint64[n_poly][L][N] with row r under prime r; out may alias an input.
 */
/* add_poly_coeffmod :43-86 / Evaluator::add_inplace SEAL/evaluator.cpp:155-240 */
int moai_add(moai_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly, size_t L,
             void *stream);
/* sub_poly_coeffmod :88-133 / Evaluator::sub_inplace SEAL/evaluator.cpp:263-350 */
int moai_sub(moai_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly, size_t L,
             void *stream);
/* negate_poly_coeffmod polyarithsmallmod.h:77-106 / Evaluator::negate_inplace SEAL/evaluator.cpp:130-153 */
int moai_negate(moai_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n_poly, size_t L, void *stream);
/*
 * dyadic_product_coeffmod :226-278.  a: [n_poly][L][N]; b: [n_poly_b][L][N] with n_poly_b == n_poly
 * or n_poly_b == 1 (broadcast: Evaluator::multiply_plain_ntt SEAL/evaluator.cpp:2336-2373 multiplies
 * every ciphertext polynomial by the one plaintext).
 */
int moai_dyadic_mul(moai_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_poly,
                    size_t n_poly_b, size_t L, void *stream);
/*
 * multiply_poly_scalar_coeffmod :197-224 with one scalar per RNS row (host array scalars[L], any
 * uint64; reduced mod q_r first like polyarithsmallmod.h:209-217).  This is multiply_plain by a
 * scalar-encoded plaintext, whose rows are constant (SEAL/ckks.cpp:131-150), without materialising
 * N*L words.
 */
int moai_mul_scalar_rows(moai_ctx *ctx, const uint64_t *a, const uint64_t *scalars, uint64_t *out, size_t n_poly,
                         size_t L, void *stream);
/* add_poly_scalar_coeffmod :135-164, one scalar per row (add_plain of a scalar-encoded plaintext
 * touches polynomial 0 only: Evaluator::add_plain_inplace SEAL/evaluator.cpp:2014-2018). */
int moai_add_scalar_rows(moai_ctx *ctx, const uint64_t *a, const uint64_t *scalars, uint64_t *out, size_t n_poly,
                         size_t L, void *stream);

/* out = base + sum_{t < terms} x[t] (*) scalars[t]   -- the accumulation chain of MOAI's column-packed ct x pt product
 * (include/source/matrix_mul/Ct_pt_matrix_mul.hpp:19-42: Evaluator::multiply_plain by a scalar-encoded plaintext,
 * SEAL/evaluator.cpp:2336-2373 with SEAL/ckks.cpp:131-150, then Evaluator::add_inplace, :155-240, once per row of W) as one
 * pass per sixteen terms.  x: HOST array of `terms` device pointers, each [size][L][N]; scalars: HOST [terms][L], the plaintexts'
 * constant rows, reduced; base: device [size][L][N] or NULL (may be `out`); out: device [size][L][N], not one of the terms.
 * Canonical residues out: the same bits as the reference's `terms` products and sums. */
int moai_scalar_dot(moai_ctx *ctx, const uint64_t *const *x, const uint64_t *scalars, size_t terms, const uint64_t *base,
                    uint64_t *out, size_t size, size_t L, void *stream);
/* The same chain with full plaintexts (MOAI's masked products, Ct_pt_matrix_mul.hpp:103-170: multiply_plain_ntt's dyadic
 * product, SEAL/evaluator.cpp:2336-2373, then add_inplace): out = base + sum_t x[t] (*) p[t].  x: HOST array of device
 * pointers as above; p: DEVICE [terms][L][N], the plaintexts back to back in NTT form (moai_ckks_encode_masked writes them so). */
int moai_vector_dot(moai_ctx *ctx, const uint64_t *const *x, const uint64_t *p, size_t terms, const uint64_t *base, uint64_t *out,
                    size_t size, size_t L, void *stream);
/* ... and for ciphertext operands: out[3][L][N] = base + sum_t multiply(x[t], y[t]) with size-2 operands in separate blocks
 * (Evaluator::multiply's ckks_multiply, SEAL/evaluator.cpp:770-909, then add_inplace: the inner loops of MOAI's
 * Ct_ct_matrix_mul.hpp:32-41 and :121-134).  x, y: HOST arrays of `terms` device pointers; base: device [3][L][N] or NULL. */
int moai_ct_dot_ptrs(moai_ctx *ctx, const uint64_t *const *x, const uint64_t *const *y, size_t terms, const uint64_t *base, uint64_t *out,
                     size_t L, void *stream);

/* ---- ciphertext products -------------------------------------------------------------------------
 * Evaluator::ckks_multiply SEAL/evaluator.cpp:770-909, size 2 x size 2 -> size 3:
 * out[b] = (x0*y0, x0*y1 + x1*y0, x1*y1).  x, y: [batch][2][L][N]; out: [batch][3][L][N]
 * (out must not alias x or y).
 */
int moai_ct_multiply(moai_ctx *ctx, const uint64_t *x, const uint64_t *y, uint64_t *out, size_t L, size_t batch,
                     void *stream);
/* Evaluator::ckks_square SEAL/evaluator.cpp:1223-1282: (x0^2, 2 x0 x1, x1^2) */
int moai_ct_square(moai_ctx *ctx, const uint64_t *x, uint64_t *out, size_t L, size_t batch, void *stream);
/* Evaluator::multiply for operands of any size, SEAL/evaluator.cpp:862-900 (the dest_size != 3 branch of ckks_multiply):
 * out[b][k] = sum over i + j = k of x[b][i] (*) y[b][j], k < size_x + size_y - 1.
 * x: [batch][size_x][L][N]; y: [batch][size_y][L][N]; out: [batch][size_x + size_y - 1][L][N], not an operand.
 * Sizes 2..16, product at most 16 polynomials (SEAL_CIPHERTEXT_SIZE_MAX). For 2 x 2 moai_ct_multiply is the same result. */
int moai_ct_multiply_general(moai_ctx *ctx, const uint64_t *x, size_t size_x, const uint64_t *y, size_t size_y,
                             uint64_t *out, size_t L, size_t batch, void *stream);
/* sum over j < count of ckks_multiply(x[j], y[j]) (evaluator.cpp:805-860) accumulated with add_inplace
 * (:155-240): the inner loop of include/source/matrix_mul/Ct_ct_matrix_mul.hpp:33-42 and :117-131 in one pass.
 * x, y: [count][2][L][N]; out: [3][L][N].  Same canonical residues as the reference's multiply-reduce-add
 * sequence.  Primes of at most 61 bits (SEAL's own bound, util/defines.h:40). */
int moai_ct_dot(moai_ctx *ctx, const uint64_t *x, const uint64_t *y, uint64_t *out, size_t count, size_t L,
                void *stream);
/* out[poly] = sum over t < terms of multiply_plain(x[x_index[t]][poly], p[p_index[t]]) (evaluator.cpp:2336-2373)
 * accumulated with add_inplace: the inner loop of the baby-step / giant-step linear transforms of MOAI's
 * bootstrapping (include/source/bootstrapping/Bootstrapper.cpp:2028-2046, 2095-2113) for a whole batch.
 * x: operands, operand k = x + k * n_poly * L * N, each [n_poly][L][N] (n_poly = batch * ciphertext size);
 * p: plaintexts in NTT form, [n_pt][L][N], shared by the batch; out: [n_poly][L][N]; x_index / p_index: host
 * arrays.  terms <= 64.  Same canonical residues as the reference's multiply-reduce-add sequence. */
int moai_ct_pt_dot(moai_ctx *ctx, const uint64_t *x, const uint64_t *p, uint64_t *out, const uint32_t *x_index,
                   const uint32_t *p_index, size_t terms, size_t n_poly, size_t L, void *stream);
/* Two such sums over the same operands in one pass -- two giant steps of one transform (Bootstrapper.cpp:2024-2062): every
 * baby-step ciphertext is read once for both.  out = sum over t < terms of x[x_index[t]] (.) p[p_index[t]];
 * out2 = sum over t < terms2 <= terms of x[x_index[t]] (.) p[p_index2[t]] (the last giant step of a transform is shorter).
 * The same residues as two moai_ct_pt_dot calls. */
int moai_ct_pt_dot2(moai_ctx *ctx, const uint64_t *x, const uint64_t *p, uint64_t *out, uint64_t *out2,
                    const uint32_t *x_index, const uint32_t *p_index, const uint32_t *p_index2, size_t terms, size_t terms2,
                    size_t n_poly, size_t L, void *stream);
/* out = sum over ALL r < rows of multiply_plain(x[r], p[r]) accumulated with add_inplace: one output column of
 * ct_pt_matrix_mul_wo_pre_w_mask (include/source/matrix_mul/Ct_pt_matrix_mul.hpp:120-150, every weight a full plaintext),
 * any number of rows in one call; with p2 / out2 (both or neither) a second column over the same ciphertexts in the same
 * pass.  x: [rows][n_poly][L][N]; p, p2: [rows][L][N] in NTT form; out, out2: [n_poly][L][N].  The sum is split over
 * workgroups and folded by a second small kernel (exact: every partial sum is a canonical residue).  Same residues as the
 * reference's multiply-reduce-add sequence. */
int moai_ct_pt_dot_rows(moai_ctx *ctx, const uint64_t *x, const uint64_t *p, const uint64_t *p2, uint64_t *out, uint64_t *out2,
                        size_t rows, size_t n_poly, size_t L, void *stream);

/*
 * Column-packed ciphertext x plaintext matrix product with scalar-encoded weights: the body of
 * ct_pt_matrix_mul_wo_pre (include/source/matrix_mul/Ct_pt_matrix_mul.hpp:4-49, :51-101) before its
 * rescale, i.e. out[c] = sum_j multiply_plain(X[j], encode(W[j][c])) for every output column c.
 * x: [rows][size][L][N]; w: DEVICE array uint64[L][rows][cols], w[r][j][c] = the residue under prime r
 * of the scalar plaintext CKKSEncoder::encode(W[j][c], scale) (SEAL/ckks.cpp:101-150), canonical;
 * out: [cols][size][L][N] (must not alias x).  Follow with moai_rescale(out, ..., batch = cols).
 */
int moai_ct_pt_matmul(moai_ctx *ctx, const uint64_t *x, const uint64_t *w, uint64_t *out, size_t rows, size_t cols,
                      size_t size, size_t L, void *stream);

/* ---- level changes ---------------------------------------------------------------------------------
 * Evaluator::rescale_to_next SEAL/evaluator.cpp:1682-1720 -> mod_switch_scale_to_next :1402-1481 ->
 * RNSTool::divide_and_round_q_last_ntt_inplace SEAL/util/rns.cpp:830-901.
 * in: [batch][size][L][N] -> out: [batch][size][L-1][N]; in is preserved; out may not alias in.
 */
int moai_rescale(moai_ctx *ctx, const uint64_t *in, uint64_t *out, size_t size, size_t L, size_t batch,
                 void *stream);
/* multiply_plain by a scalar plaintext (moai_mul_scalar_rows) followed by moai_rescale, in one pass over the
 * ciphertext: Evaluator::multiply_const + rescale_to_next_inplace of the fork (SEAL/evaluator.cpp:395-418,
 * :1682-1720), the pair the *_reduced_error compositions and MOAI's polynomial evaluations issue for every
 * coefficient.  Same residues as the two calls.  in: [batch][size][L][N], scalars[L] (host), out: [batch][size][L-1][N]. */
int moai_mul_scalar_rescale(moai_ctx *ctx, const uint64_t *in, const uint64_t *scalars, uint64_t *out, size_t size,
                            size_t L, size_t batch, void *stream);
/* moai_rescale / moai_mul_scalar_rescale followed by Evaluator::add_inplace (SEAL/evaluator.cpp:155-240) of `addend`, the
 * addition done by the rescale's last kernel: what add_[inplace_]reduced_error of the fork issues after its level adjustment
 * (SEAL/evaluator.cpp:447-480) and MOAI's polynomial evaluations issue per coefficient (rescale, then add to the running sum).
 * Same residues as the separate calls.  addend: [batch][size][L-1][N], may be `out` itself (accumulate in place). */
int moai_rescale_add(moai_ctx *ctx, const uint64_t *in, const uint64_t *addend, uint64_t *out, size_t size, size_t L,
                     size_t batch, void *stream);
int moai_mul_scalar_rescale_add(moai_ctx *ctx, const uint64_t *in, const uint64_t *scalars, const uint64_t *addend,
                                uint64_t *out, size_t size, size_t L, size_t batch, void *stream);
/* Evaluator::mod_switch_drop_to_next SEAL/evaluator.cpp:1483-1546 applied `drop` times:
 * in: [batch][size][L][N] -> out: [batch][size][L-drop][N].  out must not alias in (except batch*size == 1,
 * where the kept rows already are in place). */
int moai_mod_drop(moai_ctx *ctx, const uint64_t *in, uint64_t *out, size_t size, size_t L, size_t drop,
                  size_t batch, void *stream);

/* ---- Galois automorphism and key switching -----------------------------------------------------------
 * GaloisTool::apply_galois_ntt SEAL/util/galois.cpp:192-218 with the table of :18-51:
 * out[p][r][i] = in[p][r][table[i]].  galois_elt odd, < 2N.  out must not alias in.
 */
int moai_galois_permute(moai_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n_poly, size_t L,
                        uint32_t galois_elt, void *stream);
/* GaloisTool::get_elt_from_step SEAL/util/galois.cpp:53-95 (generator 5 in this fork,
 * SEAL/util/galois.h:169).  Returns 0 and sets the error for |step| >= N/2. */
uint32_t moai_galois_elt_from_step(const moai_ctx *ctx, int step);
/*
 * Evaluator::switch_key_inplace SEAL/evaluator.cpp:2724-3020 (CKKS branch):
 *   ct[b] (2 polys, NTT form, L primes) += ModDown_p( sum_J NTT_{q_I}([INTT(target[b]_J)]_{q_I}) (*) key[J][.][I] )
 * ct: [batch][2][L][N] in place; target: [batch][L][N] (NTT form, not modified);
 * key: uint64[k-1][2][k][N] -- the reference's KSwitchKeys entry, vector<PublicKey> of k-1 size-2
 * ciphertexts at the key level (SEAL/kswitchkeys.h:340, keygenerator.cpp:303-336) flattened.
 * Every ciphertext of the batch is switched with the same key.
 */
int moai_switch_key(moai_ctx *ctx, uint64_t *ct, const uint64_t *target, const uint64_t *key, size_t L,
                    size_t batch, void *stream);
/* Evaluator::relinearize_internal SEAL/evaluator.cpp:1345-1400 for size 3 -> 2:
 * ct3: [batch][3][L][N]; out: [batch][2][L][N] (may not alias ct3). */
int moai_relinearize(moai_ctx *ctx, const uint64_t *ct3, const uint64_t *relin_key, uint64_t *out, size_t L,
                     size_t batch, void *stream);
/* Evaluator::apply_galois_inplace SEAL/evaluator.cpp:2563-2665 (rotate_vector / complex_conjugate
 * with the key present, rotate_internal :2667-2697): ct: [batch][2][L][N] in place. */
int moai_apply_galois(moai_ctx *ctx, uint64_t *ct, size_t L, uint32_t galois_elt, const uint64_t *galois_key,
                      size_t batch, void *stream);
/* The same with a separate destination (Evaluator::apply_galois / rotate_vector / complex_conjugate with a
 * `destination`, SEAL/evaluator.h:1093-1101, 1191-1227, 1262-1270: "destination = encrypted; ..._inplace(destination)"
 * without the deep copy): in, out: [batch][2][L][N]; out may be in. */
int moai_apply_galois_to(moai_ctx *ctx, const uint64_t *in, uint64_t *out, size_t L, uint32_t galois_elt,
                         const uint64_t *galois_key, size_t batch, void *stream);
/* acc = add_inplace(acc, apply_galois(in)): a rotation whose result is added to a running sum, the pair
 * rotate_vector + add_inplace_reduced_error (equal levels) of the giant steps of Bootstrapper::bsgs_linear_transform
 * (include/source/bootstrapping/Bootstrapper.cpp:2049-2059); the addition rides on the key switch's last kernel.
 * in, acc: [batch][2][L][N], distinct.  Same residues as moai_apply_galois_to followed by moai_add. */
int moai_apply_galois_acc(moai_ctx *ctx, const uint64_t *in, uint64_t *acc, size_t L, uint32_t galois_elt,
                          const uint64_t *galois_key, size_t batch, void *stream);

/* ---- MOAI-owned integer kernel ---------------------------------------------------------------------------
 * Bootstrapper::modraise_inplace include/source/bootstrapping/Bootstrapper.cpp:2938-2992:
 * in: [batch][2][1][N] (NTT form under prime 0) -> out: [batch][2][L_out][N] (NTT form). */
int moai_modraise(moai_ctx *ctx, const uint64_t *in, uint64_t *out, size_t L_out, size_t batch, void *stream);

/* ---- CKKS encoder ---------------------------------------------------------
 * CKKSEncoder::encode_internal, vector form, SEAL/ckks.h:457-637, for n_batch value vectors in one call:
 * scatter into the conjugate-symmetric slot vector (matrix_reps_index_map_, ckks.cpp:34-52, generator 5),
 * FP64 inverse DWT with scale / N folded into the last stage (util/dwthandler.h:202-356 on
 * std::complex<double>, every product and sum rounded separately as the reference's x86-64 build does),
 * std::round, exact residues of the rounded integers, forward NTT.
 *   values     device, [n_batch][values_size] doubles (is_complex = 0) or (re, im) pairs (is_complex = 1);
 *              values_size <= N/2, shorter vectors are zero-padded like the reference
 *   dst        device, [n_batch][L][N], rows over context primes prime_index[0..L) (NULL = 0..L-1), NTT form
 *   max_coeff  device, [n_batch] doubles, or NULL: max |coefficient| before rounding, i.e. the quantity
 *              ckks.h:527-538 turns into max_coeff_bit_count; the caller compares it with
 *              moai_total_coeff_modulus_bit_count to raise "encoded values are too large"
 * The three decomposition branches of the reference (<= 64 bits, <= 128 bits, multi-word) all produce the
 * exact integer modulo each prime; so does the single device routine.
 * Errors: MOAI_EINVAL "values_size is too large" / "scale out of bounds" (ckks.h:469-497). */
int moai_ckks_encode(moai_ctx *ctx, const double *values, int is_complex, size_t values_size, size_t n_batch,
                     uint64_t *dst, size_t L, const uint32_t *prime_index, double scale, double *max_coeff,
                     void *stream);
/* The same for vectors of the form MOAI's masked matrix products encode
 * (include/source/matrix_mul/Ct_pt_matrix_mul.hpp:124-146): values[b][s] = constants[b] where mask[s] == 1, else
 * 0, for s < mask_size (the rest zero-padded).  constants: device, [n_batch] doubles; mask: device, [mask_size]
 * int32 (MOAI's bias_vec).  Saves building and uploading n_batch * N/2 doubles; same residues as
 * moai_ckks_encode on the expanded vectors. */
int moai_ckks_encode_masked(moai_ctx *ctx, const double *constants, const int32_t *mask, size_t mask_size,
                            size_t n_batch, uint64_t *dst, size_t L, const uint32_t *prime_index, double scale,
                            double *max_coeff, void *stream);
/* ContextData::total_coeff_modulus_bit_count (SEAL/context.cpp:169-173): significant bits of the product of
 * the L primes; 0 on error. */
int moai_total_coeff_modulus_bit_count(const moai_ctx *ctx, size_t L, const uint32_t *prime_index);
/* The tables the encoder uses, for inspection by tests: matrix_reps_index_map_ (host copy, N entries) and
 * inv_root_powers_ (host copy, N (re, im) pairs; ckks.cpp:54-71 via util/croots.cpp). */
int moai_ckks_tables(moai_ctx *ctx, uint32_t *index_map, double *inv_root_powers);

/* ---- tuning -------------------------------------------------------------------------------------------------------
 * Overrides a performance knob for the whole process (same names as the environment variables read by the
 * library, which it takes precedence over).  Results never depend on these.  Currently:
 *   MOAI_KS_FP_MIN_ROWS  batch * L from which the key switch uses the FP64 arithmetic modes (default 16) */
int moai_set_tuning(const char *name, long value);

/* ---- hoisted rotations -------------------------------------------------------------------------------------------
 * out[r] = apply_galois(in, galois_elts[r], galois_keys[r]) for r < R -- R calls of Evaluator::rotate_vector /
 * apply_galois on the SAME ciphertext (SEAL/evaluator.cpp:2563-2665 over :2724-3020; the baby steps of MOAI's
 * bootstrapping transforms, include/source/bootstrapping/Bootstrapper.cpp:2017-2022, 2082-2088) -- with ONE digit
 * decomposition (the l (l + 1) transforms of switch_key_inplace) shared by all of them.  Bit-identical to the R separate
 * calls: the digit of a permuted polynomial is the permuted digit plus (q_J mod q_I) times the rotation's sign mask, and
 * that second term is the per-(key, level) constant moai_hoist_correction computes once (csrc/keyswitch_kernels.hip.h
 * has the derivation).  The identity needs INTT(c1) free of zero coefficients; the call checks that on the device and
 * otherwise makes the R separate calls itself (*used_fallback = 1).
 *   in, outs[r]  [batch][2][L][N] (no output may alias the input)   correction [2][L+1][N] (rows 0..L-1 under primes
 *   0..L-1, row L under the special prime), computed for the same (key, galois_elt, L).  outs / galois_keys /
 *   corrections: host arrays of R device pointers.
 * Any R: the accumulators of one pass hold 64 rotations, more are done 64 at a time (one decomposition per pass).
 * The call SYNCHRONISES the stream once (it reads the zero-coefficient flag back): unlike the other key-switch entry
 * points it cannot run under HIP-graph stream capture. */
int moai_hoist_correction(moai_ctx *ctx, const uint64_t *galois_key, uint32_t galois_elt, size_t L, uint64_t *correction,
                          void *stream);
int moai_apply_galois_hoisted(moai_ctx *ctx, const uint64_t *in, uint64_t *const *outs, size_t L, const uint32_t *galois_elts,
                              const uint64_t *const *galois_keys, const uint64_t *const *corrections, size_t R, size_t batch,
                              int *used_fallback, void *stream);

/* ---- level-trimmed key residency ------------------------------------------------------------------------------------
 * switch_key_inplace at l data primes reads only the digits J < l and the rows {0 .. l-1, special prime} of a key
 * (SEAL/evaluator.cpp:2818, 2831; the reference keeps every key whole, SEAL/kswitchkeys.h:340: 1.32 GB each at MOAI's
 * parameters).  moai_key_trim copies exactly that part of a key in the reference's layout [k-1][2][k][N] into
 * `trimmed`, [levels][2][levels+1][N] with the special prime's row last (moai_key_words(ctx, levels) words), and records
 * the layout of that pointer in the context: every key-switch entry point above accepts it in place of the full key for
 * l <= levels, with the same results bit for bit (the same words are read), and fails with MOAI_ERANGE for l > levels.
 * MOAI's 31 default rotation keys are only used at chain index <= 14 (Ct_ct_matrix_mul.hpp:29,95,112,147): 15 levels =
 * 19 % of the full size.  moai_key_forget drops the record (before the block is freed or reused). */
size_t moai_key_words(const moai_ctx *ctx, size_t levels);
int moai_key_trim(moai_ctx *ctx, const uint64_t *full_key, size_t levels, uint64_t *trimmed, void *stream);
int moai_key_forget(moai_ctx *ctx, const uint64_t *key);

/* ---- stream audit (debug) ----------------------------------------------------------------------------------------
 * A caller that recycles device blocks in a stream-ordered cache (the seal:: shim's util::DevicePool: a released block may be
 * handed out again on the SAME stream without synchronising, which is only safe when everything that touches the block is
 * enqueued on that stream) can have that invariant checked.  With MOAI_STREAM_AUDIT=1 in the environment, or after
 * moai_debug_stream_audit(1), moai_debug_block_label records the stream a block belongs to (state 1 = in use, 2 = released to
 * the cache, 0 = forget: back to the device allocator), and every entry point above that enqueues work fails with MOAI_ELOGIC
 * (and a line on stderr naming the function, the pointer and both streams) when handed a device pointer inside a block
 * labelled with another stream or inside a released block.  Unlabelled memory is not checked.  The reference's counterpart is
 * the single-threaded ownership of its MemoryPoolMT blocks (SEAL/util/mempool.h:228); it has no device streams.
 * moai_debug_stream_audit returns the previous setting; _counts reports pointers checked / violations found so far. */
int moai_debug_stream_audit(int enable);
void moai_debug_block_label(const void *ptr, size_t bytes, const void *stream, int state);
void moai_debug_stream_audit_counts(unsigned long long *checked, unsigned long long *violations);

/* ---- operation census ----------------------------------------------------------------------------------------------
 * moai_op_trace(1) clears and starts, moai_op_trace(0) stops counting what the entry points above were asked to do:
 * per (entry point, level L) the sum of the call's own batch argument (polynomials for the element-wise and NTT calls,
 * ciphertexts for the scheme-level ones, terms x polynomials for the fused sums).  bench.py uses it to price the same
 * operations on the CPU oracle.  moai_op_trace_dump writes "name L count" lines (NUL-terminated, truncated to cap)
 * and returns the size the whole text needs.  Off by default; costs one relaxed atomic load per call. */
int moai_op_trace(int enable);
size_t moai_op_trace_dump(char *buf, size_t cap);

/* ---- measurement support -----------------------------------------------------------------------------------
 * Average duration in milliseconds of the NTT kernels of the last moai_ntt_* call recorded with
 * HIP events on the caller's stream is not provided here; callers time with their own events
 * around the calls (bench.py does).  moai_device_info fills name (<= 255 chars) and CU count. */
int moai_device_info(int device, char *name, size_t name_cap, int *compute_units, size_t *hbm_bytes);
/* free and total device memory of the current device (hipMemGetInfo) */
int moai_mem_info(size_t *free_bytes, size_t *total_bytes);
/* hipEvent helpers so that pure-C / ctypes callers can time the stream the kernels run on */
int moai_event_create(void **event);
int moai_event_destroy(void *event);
int moai_event_record(void *event, void *stream);
int moai_event_synchronize(void *event);
int moai_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */

#ifdef __cplusplus
}
#endif
#endif
