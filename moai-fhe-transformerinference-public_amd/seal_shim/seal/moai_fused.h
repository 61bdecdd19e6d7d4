// seal/moai_fused.h -- drop-in replacements for MOAI routines that the device can execute as ONE kernel
// instead of rows*cols evaluator calls.  Same signature, same results (bit-identical ciphertexts).
//
// moai_fused::ct_pt_matrix_mul_wo_pre replaces
// include/source/matrix_mul/Ct_pt_matrix_mul.hpp:4-49 (and the "_large" variant :51-101, which differs
// only in its OpenMP blocking): out[i] = rescale( sum_j multiply_plain(enc_X[j], encode(W[j][i])) ).
//
// moai_fused::ct_ct_matrix_mul_colpacking replaces include/source/matrix_mul/Ct_ct_matrix_mul.hpp:6-56
// (Q K^T): out[i] = rescale(relinearize( sum_j multiply(X[j], rotate(W[j], i * num_batch)) )).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <memory>

#include "seal/seal.h"

namespace moai_fused
{
    // ---- packed ciphertexts -----------------------------------------------------------------------------------
    // pack(): one seal::Ciphertext holding all of `cts` (same size, level, scale, NTT form) back to back.  Every
    // Evaluator method applied to it performs the operation on each member with the device's batched kernels,
    // so a per-ciphertext MOAI routine called on the pack (gelu_v2, exp, invert_sqrt, eval_odd_deg9_poly, ...)
    // does the work of the OpenMP loop MOAI wraps around it, with the same result for every member bit for
    // bit.  unpack() splits it again.
    inline seal::Ciphertext pack(const std::vector<seal::Ciphertext> &cts, const seal::SEALContext &context)
    {
        using namespace seal;
        if (cts.empty())
        {
            throw std::invalid_argument("nothing to pack");
        }
        const Ciphertext &f = cts[0];
        for (auto &c : cts)
        {
            if (c.parms_id() != f.parms_id() || c.size() != f.size() || c.is_ntt_form() != f.is_ntt_form() ||
                c.scale() != f.scale() || c.batch() != 1)
            {
                throw std::invalid_argument("pack: ciphertexts differ in level, size, form or scale");
            }
        }
        Ciphertext out;
        out.resize_batch(context, f.parms_id(), f.size(), cts.size());
        out.is_ntt_form() = f.is_ntt_form();
        out.scale() = f.scale();
        const std::size_t words = f.size() * f.coeff_modulus_size() * f.poly_modulus_degree();
        for (std::size_t b = 0; b < cts.size(); b++)
        {
            util::hip_check(moai_memcpy_d2d(out.device_data() + b * words, cts[b].device_data(), words * 8, context.stream()));
        }
        return out;
    }
    inline void unpack(const seal::Ciphertext &packed, const seal::SEALContext &context, std::vector<seal::Ciphertext> &cts)
    {
        using namespace seal;
        const std::size_t B = packed.batch();
        std::vector<Ciphertext> out(B);
        const std::size_t words = packed.size() * packed.coeff_modulus_size() * packed.poly_modulus_degree();
        for (std::size_t b = 0; b < B; b++)
        {
            out[b].resize(context, packed.parms_id(), packed.size());
            out[b].is_ntt_form() = packed.is_ntt_form();
            out[b].scale() = packed.scale();
            util::hip_check(moai_memcpy_d2d(out[b].device_data(), packed.device_data() + b * words, words * 8, context.stream()));
        }
        cts = std::move(out);
    }

    inline std::vector<seal::Ciphertext> ct_pt_matrix_mul_wo_pre(const std::vector<seal::Ciphertext> &enc_X,
                                                                 const std::vector<std::vector<double>> &W, int col_X,
                                                                 int col_W, int row_W, const seal::SEALContext &seal_context,
                                                                 int computed_cols = -1)
    {
        using namespace seal;
        std::vector<Ciphertext> output(static_cast<std::size_t>(col_W));
        // the "_large" variant of the reference fills only the first 128 * (col_W / 128) columns
        col_W = computed_cols >= 0 ? computed_cols : col_W;
        if (col_W == 0)
        {
            return output;
        }
        if (col_X != row_W)
        {
            std::cout << "ERROR: bad dimensions of X or W. " << std::endl;
            return output;
        }
        const double scale = enc_X[0].scale();
        const parms_id_type pid = enc_X[0].parms_id();
        auto cd = seal_context.get_context_data(pid);
        if (!cd || !cd->next_context_data())
        {
            throw std::invalid_argument("end of modulus switching chain reached");
        }
        const auto &cm = cd->parms().coeff_modulus();
        const std::size_t L = cm.size(), n = seal_context.n();
        const std::size_t rows = static_cast<std::size_t>(row_W), cols = static_cast<std::size_t>(col_W);
        for (std::size_t j = 0; j < rows; j++)
        {
            if (enc_X[j].parms_id() != pid || enc_X[j].size() != 2 || !enc_X[j].is_ntt_form())
            {
                throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            }
        }
        // the scalar plaintexts' residues, exactly as CKKSEncoder::encode(double, parms_id, scale) makes them;
        // output columns in chunks, so that the staging buffers stay within a few GiB however wide W is
        CKKSEncoder encoder(seal_context);
        void *st = seal_context.stream();
        const std::size_t ct_words = 2 * L * n;
        const std::size_t chunk = std::max<std::size_t>(16, std::min<std::size_t>(cols, (std::size_t(4) << 30) / (ct_words * 8)));
        util::DeviceArray dx(rows * ct_words, st), dw(L * rows * chunk, st), dout(chunk * ct_words, st),
            dres(chunk * 2 * (L - 1) * n, st);
        for (std::size_t j = 0; j < rows; j++)
        {
            util::hip_check(moai_memcpy_d2d(dx.get() + j * ct_words, enc_X[j].device_data(), ct_words * 8, st));
        }
        const parms_id_type next_id = cd->next_context_data()->parms_id();
        std::vector<std::uint64_t> w;
        for (std::size_t c0 = 0; c0 < cols; c0 += chunk)
        {
            const std::size_t cc = std::min(chunk, cols - c0);
            w.assign(L * rows * cc, 0);
            for (std::size_t j = 0; j < rows; j++)
            {
                for (std::size_t c = 0; c < cc; c++)
                {
                    Plaintext p;
                    encoder.encode(W[j][c0 + c], pid, enc_X[j].scale(), p);
                    for (std::size_t r = 0; r < L; r++)
                    {
                        w[(r * rows + j) * cc + c] = p.scalar_rows()[r];
                    }
                }
            }
            util::hip_check(moai_memcpy_h2d(dw.get(), w.data(), w.size() * 8, st));
            util::hip_check(moai_ct_pt_matmul(seal_context.device(), dx.get(), dw.get(), dout.get(), rows, cc, 2, L, st));
            util::hip_check(moai_rescale(seal_context.device(), dout.get(), dres.get(), 2, L, cc, st));
            for (std::size_t c = 0; c < cc; c++)
            {
                Ciphertext &o = output[c0 + c];
                o.resize(seal_context, next_id, 2);
                util::hip_check(moai_memcpy_d2d(o.device_data(), dres.get() + c * 2 * (L - 1) * n, 2 * (L - 1) * n * 8, st));
                o.is_ntt_form() = true;
                o.scale() = scale; // Ct_pt_matrix_mul.hpp:41
            }
            seal_context.sync(); // w is refilled for the next chunk
        }
        seal_context.sync(); // w and the staging buffers go out of scope
        return output;
    }
    // include/source/matrix_mul/Ct_pt_matrix_mul.hpp:51-101: the same product with another OpenMP blocking; like
    // the reference it fills the first 128 * (col_W / 128) output columns
    inline std::vector<seal::Ciphertext> ct_pt_matrix_mul_wo_pre_large(const std::vector<seal::Ciphertext> &enc_X,
                                                                       const std::vector<std::vector<double>> &W, int col_X,
                                                                       int col_W, int row_W,
                                                                       const seal::SEALContext &seal_context)
    {
        return ct_pt_matrix_mul_wo_pre(enc_X, W, col_X, col_W, row_W, seal_context, 128 * (col_W / 128));
    }

    // moai_fused::ct_pt_matrix_mul_wo_pre_w_mask replaces include/source/matrix_mul/Ct_pt_matrix_mul.hpp:103-170
    // (self-output and final feed-forward products of the 12-layer run, test_full_scheme.hpp:601,928): every
    // weight is encoded as the VECTOR mask * w -- rows*cols FP64 transforms, which cannot be shared because
    // rounding does not commute with the scaling by w.  Here a column's `rows` plaintexts are produced by one
    // moai_ckks_encode_masked call straight from the weights (nothing but `rows` doubles crosses PCIe), the
    // products and sums of the column are moai_ct_pt_dot passes, and all columns are rescaled in one call.
    // Same ciphertexts bit for bit; like the reference only the first 128 * (col_W / 128) columns are computed.
    inline std::vector<seal::Ciphertext> ct_pt_matrix_mul_wo_pre_w_mask(const std::vector<seal::Ciphertext> &enc_X,
                                                                        const std::vector<std::vector<double>> &W,
                                                                        const std::vector<int> &bias_vec, int col_X,
                                                                        int col_W, int row_W,
                                                                        const seal::SEALContext &seal_context)
    {
        using namespace seal;
        std::vector<Ciphertext> output(static_cast<std::size_t>(col_W));
        if (col_X != row_W)
        {
            std::cout << "ERROR: bad dimensions of X or W. " << std::endl;
            return output;
        }
        const double scale = enc_X[0].scale();
        const parms_id_type pid = enc_X[0].parms_id();
        auto cd = seal_context.get_context_data(pid);
        if (!cd || !cd->next_context_data())
        {
            throw std::invalid_argument("end of modulus switching chain reached");
        }
        const auto &cm = cd->parms().coeff_modulus();
        const std::size_t L = cm.size(), n = seal_context.n(), slots = n >> 1;
        const std::size_t rows = static_cast<std::size_t>(row_W);
        const std::size_t cols = static_cast<std::size_t>(128 * (col_W / 128));
        for (std::size_t j = 0; j < rows; j++)
        {
            if (enc_X[j].parms_id() != pid || enc_X[j].size() != 2 || !enc_X[j].is_ntt_form())
            {
                throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            }
            if (enc_X[j].scale() != scale)
            {
                throw std::invalid_argument("scale mismatch");
            }
        }
        // CKKSEncoder::encode and multiply_plain checks (ckks.h:493-497, evaluator.cpp:2351-2357)
        if (scale <= 0 || (static_cast<int>(std::log2(scale)) + 1 >= cd->total_coeff_modulus_bit_count()))
        {
            throw std::invalid_argument("scale out of bounds");
        }
        if (static_cast<int>(std::log2(scale * scale)) >= cd->total_coeff_modulus_bit_count())
        {
            throw std::invalid_argument("scale out of bounds");
        }
        if (bias_vec.size() < slots)
        {
            throw std::invalid_argument("bias_vec is shorter than the slot count");
        }
        if (cols == 0)
        {
            return output;
        }
        void *st = seal_context.stream();
        moai_ctx *dev = seal_context.device();
        const std::size_t ct_words = 2 * L * n;
        static_assert(sizeof(int) == 4, "bias_vec is uploaded as int32");
        // two columns per round: their 2 * rows plaintexts come from one moai_ckks_encode_masked call, and one pass over the
        // input ciphertexts (moai_ct_pt_dot_rows) forms both columns' sums
        const std::size_t group = cols >= 2 ? 2 : 1;
        util::DeviceArray dmask((slots + 1) / 2, st), dconst(2 * group * rows, st); // constants, then max |coeff| per vector
        util::hip_check(moai_memcpy_h2d(dmask.get(), bias_vec.data(), slots * 4, st));
        util::DeviceArray dX(rows * ct_words, st), dP(group * rows * L * n, st), dacc(cols * ct_words, st);
        for (std::size_t j = 0; j < rows; j++)
        {
            util::hip_check(moai_memcpy_d2d(dX.get() + j * ct_words, enc_X[j].device_data(), ct_words * 8, st));
        }
        std::vector<double> wcol(group * rows), mx(group * rows);
        const int total_bits = cd->total_coeff_modulus_bit_count();
        for (std::size_t c = 0; c < cols; c += group)
        {
            const std::size_t g = std::min(group, cols - c);
            for (std::size_t k = 0; k < g; k++)
            {
                for (std::size_t j = 0; j < rows; j++)
                {
                    wcol[k * rows + j] = W[j][c + k];
                }
            }
            util::hip_check(moai_memcpy_h2d(dconst.get(), wcol.data(), g * rows * 8, st));
            double *max_dev = reinterpret_cast<double *>(dconst.get() + group * rows);
            util::hip_check(moai_ckks_encode_masked(dev, reinterpret_cast<const double *>(dconst.get()),
                                                    reinterpret_cast<const std::int32_t *>(dmask.get()), slots, g * rows, dP.get(), L,
                                                    nullptr, scale, max_dev, st));
            std::uint64_t *acc = dacc.get() + c * ct_words;
            util::hip_check(moai_ct_pt_dot_rows(dev, dX.get(), dP.get(), g == 2 ? dP.get() + rows * L * n : nullptr, acc,
                                                g == 2 ? acc + ct_words : nullptr, rows, 2, L, st));
            util::hip_check(moai_memcpy_d2h(mx.data(), max_dev, g * rows * 8, st));
            seal_context.sync(); // wcol / mx are reused by the next round
            for (std::size_t k = 0; k < g * rows; k++)
            {
                int bits = static_cast<int>(std::ceil(std::log2(std::max<>(mx[k], 1.0)))) + 1;
                if (!(bits < total_bits))
                {
                    throw std::invalid_argument("encoded values are too large");
                }
            }
        }
        util::DeviceArray dres(cols * 2 * (L - 1) * n, st);
        util::hip_check(moai_rescale(dev, dacc.get(), dres.get(), 2, L, cols, st));
        const parms_id_type next_id = cd->next_context_data()->parms_id();
        for (std::size_t c = 0; c < cols; c++)
        {
            output[c].resize(seal_context, next_id, 2);
            util::hip_check(moai_memcpy_d2d(output[c].device_data(), dres.get() + c * 2 * (L - 1) * n, 2 * (L - 1) * n * 8, st));
            output[c].is_ntt_form() = true;
            output[c].scale() = scale; // Ct_pt_matrix_mul.hpp:160
        }
        seal_context.sync();
        return output;
    }

    namespace detail
    {
        // The key switches Evaluator::rotate_internal (SEAL/evaluator.cpp:2667-2722) performs for `steps`, as
        // the list of Galois elements in the order it applies them: one element when the key exists,
        // otherwise the non-adjacent form, each term through rotate_internal again, a term of N/2 skipped.
        inline void rotation_sequence(const seal::SEALContext &context, const seal::GaloisKeys &keys, int steps,
                                      std::vector<std::uint32_t> &out)
        {
            if (steps == 0)
            {
                return;
            }
            std::uint32_t elt = moai_galois_elt_from_step(context.device(), steps);
            if (!elt)
            {
                throw std::invalid_argument("step count too large");
            }
            if (keys.has_key(elt))
            {
                out.push_back(elt);
                return;
            }
            std::vector<int> naf_steps = seal::util::naf(steps);
            if (naf_steps.size() == 1)
            {
                throw std::invalid_argument("Galois key not present");
            }
            for (int st : naf_steps)
            {
                if (static_cast<std::size_t>(std::abs(st)) != (context.n() >> 1))
                {
                    rotation_sequence(context, keys, st, out);
                }
            }
        }
    } // namespace detail

    // Same arguments and the same ciphertexts, bit for bit, as MOAI's ct_ct_matrix_mul_colpacking.  What
    // changes is the schedule: (1) the col_X columns travel as one batch, so every rotation is one batched
    // key switch instead of col_X single ones; (2) rows whose rotation sequences share a prefix (3*256 =
    // [-256, +1024] and 7*256 = [-256, +2048] both start with -256) share the prefix's result -- the same
    // operations on the same operands give the same bits, so nothing but the count of key switches changes
    // (355 -> 169 per column for 128 rows); (3) the col_X multiply + add_inplace pairs of a row are one
    // moai_ct_dot pass; (4) relinearize and rescale run once over all rows.
    inline std::vector<seal::Ciphertext> ct_ct_matrix_mul_colpacking(const std::vector<seal::Ciphertext> &enc_X,
                                                                     const std::vector<seal::Ciphertext> &enc_W,
                                                                     const seal::GaloisKeys &RotK,
                                                                     const seal::RelinKeys &relin_keys,
                                                                     const seal::SEALContext &seal_context, int col_X,
                                                                     int row_X, int col_W, int row_W, int num_batch)
    {
        using namespace seal;
        std::vector<Ciphertext> output(static_cast<std::size_t>(row_X));
        if (col_X != col_W || row_X != row_W)
        {
            std::cout << "ERROR: bad dimensions of X or W. " << std::endl;
            return output;
        }
        const double scale = enc_X[0].scale();
        const parms_id_type pid = enc_X[0].parms_id();
        auto cd = seal_context.get_context_data(pid);
        if (!cd)
        {
            throw std::invalid_argument("encrypted1 is not valid for encryption parameters");
        }
        const std::size_t cols = static_cast<std::size_t>(col_X), rows = static_cast<std::size_t>(row_X);
        for (std::size_t j = 0; j < cols; j++)
        {
            // the checks of Evaluator::multiply_inplace / add_inplace (evaluator.cpp:596-640, 155-180)
            if (enc_X[j].parms_id() != pid || enc_W[j].parms_id() != pid)
            {
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            }
            if (enc_X[j].size() != 2 || enc_W[j].size() != 2)
            {
                throw std::logic_error("only size-2 ciphertexts are multiplied on the device");
            }
            if (!enc_X[j].is_ntt_form() || !enc_W[j].is_ntt_form())
            {
                throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
            }
            if (enc_X[j].scale() != enc_X[0].scale() || enc_W[j].scale() != enc_W[0].scale())
            {
                throw std::invalid_argument("scale mismatch");
            }
        }
        if (RotK.parms_id() != seal_context.key_parms_id())
        {
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        }
        if (relin_keys.parms_id() != seal_context.key_parms_id())
        {
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        }
        if (relin_keys.size() < 1)
        {
            throw std::invalid_argument("not enough relinearization keys");
        }
        if (!cd->next_context_data())
        {
            throw std::invalid_argument("end of modulus switching chain reached");
        }
        // scale of the products (evaluator.cpp:789-795, 904-908)
        const double new_scale = enc_X[0].scale() * enc_W[0].scale();
        if (new_scale <= 0 || (static_cast<int>(std::log2(new_scale)) >= cd->total_coeff_modulus_bit_count()))
        {
            throw std::invalid_argument("scale out of bounds");
        }
        // rotation sequences per row, rows ordered so that shared prefixes are adjacent
        std::vector<std::vector<std::uint32_t>> seq(rows);
        for (std::size_t i = 1; i < rows; i++)
        {
            detail::rotation_sequence(seal_context, RotK, static_cast<int>(i) * num_batch, seq[i]);
        }
        std::vector<std::size_t> order(rows);
        for (std::size_t i = 0; i < rows; i++)
        {
            order[i] = i;
        }
        std::sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return seq[a] < seq[b]; });

        const std::size_t L = cd->parms().coeff_modulus().size(), n = seal_context.n();
        const std::size_t ct_words = 2 * L * n;
        void *st = seal_context.stream();
        moai_ctx *dev = seal_context.device();
        util::DeviceArray dX(cols * ct_words, st), dW(cols * ct_words, st), d3(rows * 3 * L * n, st);
        for (std::size_t j = 0; j < cols; j++)
        {
            util::hip_check(moai_memcpy_d2d(dX.get() + j * ct_words, enc_X[j].device_data(), ct_words * 8, st));
            util::hip_check(moai_memcpy_d2d(dW.get() + j * ct_words, enc_W[j].device_data(), ct_words * 8, st));
        }
        // The rotation sequences form a prefix tree: a node = all columns of W after a prefix of key switches; the rows
        // whose sequence ends at a node take their products there.  The children of a node rotate the SAME ciphertexts by
        // different steps: with two or more of them that is one hoisted call (one digit decomposition for all of them,
        // moai_apply_galois_hoisted; same bits), otherwise one batched key switch.
        struct Node
        {
            std::map<std::uint32_t, std::unique_ptr<Node>> child;
            std::vector<std::size_t> ends;
        };
        Node root;
        for (std::size_t i = 0; i < rows; i++)
        {
            Node *at = &root;
            for (std::uint32_t e : seq[i])
            {
                auto &c = at->child[e];
                if (!c)
                {
                    c.reset(new Node());
                }
                at = c.get();
            }
            at->ends.push_back(i);
        }
        (void)order;
        static const bool hoist = [] {
            const char *e = std::getenv("MOAI_SHIM_HOIST");
            return !(e && e[0] == '0');
        }();
        std::function<void(const Node &, const std::uint64_t *)> walk = [&](const Node &node, const std::uint64_t *buf) {
            for (std::size_t i : node.ends)
            {
                util::hip_check(moai_ct_dot(dev, dX.get(), buf, d3.get() + i * 3 * L * n, cols, L, st));
            }
            if (node.child.empty())
            {
                return;
            }
            std::vector<util::DeviceArray> kids;
            kids.reserve(node.child.size());
            std::vector<std::uint32_t> elts;
            std::vector<const std::uint64_t *> kptr, cptr;
            std::vector<std::uint64_t *> optr;
            for (auto &kv : node.child)
            {
                kids.emplace_back(cols * ct_words, st);
                elts.push_back(kv.first);
                const std::size_t index = GaloisKeys::get_index(kv.first);
                kptr.push_back(RotK.device_key(index, L));
                optr.push_back(kids.back().get());
            }
            if (hoist && elts.size() >= 2 && seal_context.logn() >= 12)
            {
                for (std::uint32_t e : elts)
                {
                    cptr.push_back(RotK.hoist_correction(seal_context, GaloisKeys::get_index(e), e, L));
                }
                int fell_back = 0;
                util::hip_check(moai_apply_galois_hoisted(dev, buf, optr.data(), L, elts.data(), kptr.data(), cptr.data(), elts.size(), cols,
                                                          &fell_back, st));
            }
            else
            {
                for (std::size_t c = 0; c < elts.size(); c++)
                {
                    util::hip_check(moai_apply_galois_to(dev, buf, optr[c], L, elts[c], kptr[c], cols, st));
                }
            }
            std::size_t c = 0;
            for (auto &kv : node.child)
            {
                walk(*kv.second, kids[c].get());
                kids[c] = util::DeviceArray(); // release as soon as the subtree is done
                c++;
            }
        };
        walk(root, dW.get());
        util::DeviceArray d2(rows * ct_words, st), dres(rows * 2 * (L - 1) * n, st);
        util::hip_check(moai_relinearize(dev, d3.get(), relin_keys.device_key(0, L), d2.get(), L, rows, st));
        util::hip_check(moai_rescale(dev, d2.get(), dres.get(), 2, L, rows, st));
        const parms_id_type next_id = cd->next_context_data()->parms_id();
        for (std::size_t i = 0; i < rows; i++)
        {
            output[i].resize(seal_context, next_id, 2);
            util::hip_check(moai_memcpy_d2d(output[i].device_data(), dres.get() + i * 2 * (L - 1) * n, 2 * (L - 1) * n * 8, st));
            output[i].is_ntt_form() = true;
            output[i].scale() = scale; // Ct_ct_matrix_mul.hpp:48
        }
        seal_context.sync(); // the staging buffers go out of scope
        return output;
    }
    // moai_fused::ct_ct_matrix_mul_diagpacking replaces include/source/matrix_mul/Ct_ct_matrix_mul.hpp:58-153
    // (softmax(QK^T) V): same arguments, same ciphertexts bit for bit.  The rotations of X that share a step, the
    // baby-step rotations of all columns of W, the relinearize / rescale of all partial products and the giant-step
    // rotations each run as batched calls; the products of a (column, giant step) pair are one moai_ct_dot.
    inline std::vector<seal::Ciphertext> ct_ct_matrix_mul_diagpacking(const std::vector<seal::Ciphertext> &enc_X,
                                                                      const std::vector<seal::Ciphertext> &enc_W,
                                                                      const seal::GaloisKeys &RotK,
                                                                      const seal::RelinKeys &relin_keys,
                                                                      const seal::SEALContext &seal_context, int col_X,
                                                                      int row_X, int col_W, int row_W, int num_batch)
    {
        using namespace seal;
        (void)row_W;
        const double scale = enc_X[0].scale();
        std::vector<Ciphertext> output(static_cast<std::size_t>(col_W));
        // Ct_ct_matrix_mul.hpp:71-79
        int g = static_cast<int>(std::sqrt(static_cast<double>(col_X)));
        if (g * g < col_X)
        {
            g++;
        }
        int b = col_X / g;
        if (b * g < col_X)
        {
            b++;
        }
        const parms_id_type pid = enc_X[0].parms_id();
        auto cd = seal_context.get_context_data(pid);
        if (!cd)
        {
            throw std::invalid_argument("encrypted1 is not valid for encryption parameters");
        }
        if (row_X < col_X)
        {
            throw std::logic_error("the replacement expects every diagonal of X to be present (row_X >= col_X)");
        }
        for (int i = 0; i < col_X; i++)
        {
            const Ciphertext &c = enc_X[static_cast<std::size_t>(i)];
            if (c.parms_id() != pid || c.size() != 2 || !c.is_ntt_form() || c.scale() != scale)
            {
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            }
        }
        for (int i = 0; i < col_W; i++)
        {
            const Ciphertext &c = enc_W[static_cast<std::size_t>(i)];
            if (c.parms_id() != pid || c.size() != 2 || !c.is_ntt_form() || c.scale() != enc_W[0].scale())
            {
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            }
        }
        if (RotK.parms_id() != seal_context.key_parms_id())
        {
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        }
        if (relin_keys.parms_id() != seal_context.key_parms_id() || relin_keys.size() < 1)
        {
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        }
        if (!cd->next_context_data())
        {
            throw std::invalid_argument("end of modulus switching chain reached");
        }
        const double new_scale = scale * enc_W[0].scale();
        if (new_scale <= 0 || (static_cast<int>(std::log2(new_scale)) >= cd->total_coeff_modulus_bit_count()))
        {
            throw std::invalid_argument("scale out of bounds");
        }
        const std::size_t L = cd->parms().coeff_modulus().size(), n = seal_context.n();
        const std::size_t ctw = 2 * L * n, G = static_cast<std::size_t>(g), Bg = static_cast<std::size_t>(b);
        const std::size_t cols = static_cast<std::size_t>(col_W), nx = static_cast<std::size_t>(col_X);
        void *st = seal_context.stream();
        moai_ctx *dev = seal_context.device();
        std::vector<std::uint32_t> seq;
        auto rotate = [&](std::uint64_t *data, int step, std::size_t count) {
            // Evaluator::rotate_vector on `count` contiguous ciphertexts
            seq.clear();
            detail::rotation_sequence(seal_context, RotK, step, seq);
            for (std::uint32_t elt : seq)
            {
                util::hip_check(moai_apply_galois(dev, data, L, elt, RotK.device_key(GaloisKeys::get_index(elt), L), count, st));
            }
        };
        // rot_enc_X[index], index = i*g + j: enc_X[index] rotated by (col_X - i*g) * num_batch (:86-101)
        util::DeviceArray rx(Bg * G * ctw, st);
        for (std::size_t idx = 0; idx < nx; idx++)
        {
            util::hip_check(moai_memcpy_d2d(rx.get() + idx * ctw, enc_X[idx].device_data(), ctw * 8, st));
        }
        if (Bg * G > nx)
        {
            util::hip_check(moai_memset_zero(rx.get() + nx * ctw, (Bg * G - nx) * ctw * 8, st));
        }
        for (std::size_t i = 0; i < Bg; i++)
        {
            const int rot_ind = (col_X - static_cast<int>(i * G)) * num_batch;
            const std::size_t first = i * G, count = std::min(G, nx > first ? nx - first : 0);
            if (count && rot_ind != col_X * num_batch)
            {
                rotate(rx.get() + first * ctw, rot_ind, count);
            }
        }
        // baby steps: c_g[i][k] = enc_W[i] rotated by k * num_batch (:107-113), stored [i][k]
        util::DeviceArray cg(cols * G * ctw, st), stage(cols * ctw, st);
        for (std::size_t i = 0; i < cols; i++)
        {
            util::hip_check(moai_memcpy_d2d(stage.get() + i * ctw, enc_W[i].device_data(), ctw * 8, st));
        }
        for (std::size_t k = 0; k < G; k++)
        {
            util::DeviceArray rot(cols * ctw, st);
            util::hip_check(moai_memcpy_d2d(rot.get(), stage.get(), cols * ctw * 8, st));
            if (k)
            {
                rotate(rot.get(), static_cast<int>(k) * num_batch, cols);
            }
            for (std::size_t i = 0; i < cols; i++)
            {
                util::hip_check(moai_memcpy_d2d(cg.get() + (i * G + k) * ctw, rot.get() + i * ctw, ctw * 8, st));
            }
        }
        // giant steps: out[i][j] = sum_k c_g[i][k] * rot_enc_X[j*g + k] (:116-141), all relinearized and rescaled together
        util::DeviceArray d3(cols * Bg * 3 * L * n, st);
        for (std::size_t i = 0; i < cols; i++)
        {
            for (std::size_t j = 0; j < Bg; j++)
            {
                const std::size_t first = j * G, count = std::min(G, nx > first ? nx - first : 0);
                if (!count)
                {
                    throw std::logic_error("empty giant step");
                }
                util::hip_check(moai_ct_dot(dev, cg.get() + i * G * ctw, rx.get() + first * ctw, d3.get() + (j * cols + i) * 3 * L * n,
                                            count, L, st));
            }
        }
        util::DeviceArray d2(cols * Bg * ctw, st), dres(cols * Bg * 2 * (L - 1) * n, st);
        util::hip_check(moai_relinearize(dev, d3.get(), relin_keys.device_key(0, L), d2.get(), L, cols * Bg, st));
        util::hip_check(moai_rescale(dev, d2.get(), dres.get(), 2, L, cols * Bg, st));
        // output[i] = out[i][0] + sum_{j>=1} rotate(out[i][j], j*g*num_batch) (:142-150); layout [j][i]
        const std::size_t Lr = L - 1, rw = 2 * Lr * n;
        for (std::size_t j = 1; j < Bg; j++)
        {
            std::uint64_t *blk = dres.get() + j * cols * rw;
            seq.clear();
            detail::rotation_sequence(seal_context, RotK, static_cast<int>(j * G) * num_batch, seq);
            for (std::uint32_t elt : seq)
            {
                util::hip_check(moai_apply_galois(dev, blk, Lr, elt, RotK.device_key(GaloisKeys::get_index(elt), Lr), cols, st));
            }
            util::hip_check(moai_add(dev, dres.get(), blk, dres.get(), cols * 2, Lr, st));
        }
        const parms_id_type next_id = cd->next_context_data()->parms_id();
        for (std::size_t i = 0; i < cols; i++)
        {
            output[i].resize(seal_context, next_id, 2);
            util::hip_check(moai_memcpy_d2d(output[i].device_data(), dres.get() + i * rw, rw * 8, st));
            output[i].is_ntt_form() = true;
            output[i].scale() = scale; // :139
        }
        seal_context.sync();
        return output;
    }
} // namespace moai_fused
