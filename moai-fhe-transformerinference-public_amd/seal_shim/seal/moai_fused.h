// seal/moai_fused.h -- drop-in replacements for MOAI routines that the device can execute as ONE kernel
// instead of rows*cols evaluator calls.  Same signature, same results (bit-identical ciphertexts).
//
// moai_fused::ct_pt_matrix_mul_wo_pre replaces
// include/source/matrix_mul/Ct_pt_matrix_mul.hpp:4-49 (and the "_large" variant :51-101, which differs
// only in its OpenMP blocking): out[i] = rescale( sum_j multiply_plain(enc_X[j], encode(W[j][i])) ).
#pragma once
#include "seal/seal.h"

namespace moai_fused
{
    inline std::vector<seal::Ciphertext> ct_pt_matrix_mul_wo_pre(const std::vector<seal::Ciphertext> &enc_X,
                                                                 const std::vector<std::vector<double>> &W, int col_X,
                                                                 int col_W, int row_W, const seal::SEALContext &seal_context)
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
        const std::size_t L = cm.size(), n = seal_context.n();
        const std::size_t rows = static_cast<std::size_t>(row_W), cols = static_cast<std::size_t>(col_W);
        for (std::size_t j = 0; j < rows; j++)
        {
            if (enc_X[j].parms_id() != pid || enc_X[j].size() != 2 || !enc_X[j].is_ntt_form())
            {
                throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            }
        }
        // the scalar plaintexts' residues, exactly as CKKSEncoder::encode(double, parms_id, scale) makes them
        CKKSEncoder encoder(seal_context);
        std::vector<std::uint64_t> w(L * rows * cols);
        for (std::size_t j = 0; j < rows; j++)
        {
            for (std::size_t c = 0; c < cols; c++)
            {
                Plaintext p;
                encoder.encode(W[j][c], pid, enc_X[j].scale(), p);
                for (std::size_t r = 0; r < L; r++)
                {
                    w[(r * rows + j) * cols + c] = p.scalar_rows()[r];
                }
            }
        }
        void *st = seal_context.stream();
        util::DeviceArray dw(w.size(), st), dx(rows * 2 * L * n, st), dout(cols * 2 * L * n, st), dres(cols * 2 * (L - 1) * n, st);
        util::hip_check(moai_memcpy_h2d(dw.get(), w.data(), w.size() * 8, st));
        for (std::size_t j = 0; j < rows; j++)
        {
            util::hip_check(moai_memcpy_d2d(dx.get() + j * 2 * L * n, enc_X[j].device_data(), 2 * L * n * 8, st));
        }
        util::hip_check(moai_ct_pt_matmul(seal_context.device(), dx.get(), dw.get(), dout.get(), rows, cols, 2, L, st));
        util::hip_check(moai_rescale(seal_context.device(), dout.get(), dres.get(), 2, L, cols, st));
        const parms_id_type next_id = cd->next_context_data()->parms_id();
        for (std::size_t c = 0; c < cols; c++)
        {
            output[c].resize(seal_context, next_id, 2);
            util::hip_check(moai_memcpy_d2d(output[c].device_data(), dres.get() + c * 2 * (L - 1) * n, 2 * (L - 1) * n * 8, st));
            output[c].is_ntt_form() = true;
            output[c].scale() = scale; // Ct_pt_matrix_mul.hpp:41
        }
        seal_context.sync(); // w and the staging buffers go out of scope
        return output;
    }
} // namespace moai_fused
