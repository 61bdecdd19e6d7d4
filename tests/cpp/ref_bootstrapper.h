// tests/cpp/ref_bootstrapper.h -- test-side comparator: the reference's OWN source text for the evaluation half of
// Bootstrapper::bootstrap_3, cut out of the reference checkout at build time (tests/cpp/gen_ref_slices.py ->
// tests/cpp/generated/ref_bootstrapper_slices.inc, git-ignored, never committed, never shipped to the GPU box as text) and
// compiled here against the seal:: shim inside namespace refslice.  This header only declares the class those
// definitions belong to -- the members they read (Bootstrapper.h:16-48) -- and supplies the one routine that is not
// sliced: modraise_inplace writes raw residues through seal::util::iter(Ciphertext) (Bootstrapper.cpp:2973-2988),
// which a device-resident ciphertext does not offer, so it goes through moai_modraise (checked against the oracle in
// tests/test_gpu_parity.py::test_modraise).
//
// Use: a binary built where the reference checkout exists runs the reference's call sequence, one ciphertext at a time,
// through the shim's evaluator, and the packed device pipeline is compared with it bit for bit.
#pragma once
#include <cmath>
#include <complex>
#include <vector>

#include "ModularReducer.h"
#include "seal/moai_bootstrap_eval.h"
#include "seal/seal.h"

namespace refslice
{
    using namespace std;
    using namespace seal;
    using namespace seal::util;

    int giantstep(int M);
    void rotation(int logslot, int Nh, int shiftcount, const vector<complex<double>> &vec, vector<complex<double>> &rtnvec);

    class Bootstrapper
    {
    public:
        long logn, n, logNh, Nh;
        double initial_scale = 1.0, final_scale;
        SEALContext &context;
        CKKSEncoder &encoder;
        Evaluator &evaluator;
        GaloisKeys &gal_keys;
        long slot_index = 0;
        vector<vector<vector<complex<double>>>> fftcoeff1, fftcoeff2, fftcoeff3, invfftcoeff1, invfftcoeff2, invfftcoeff3;
        ::ModularReducer *mod_reducer = nullptr;

        Bootstrapper(long _logn, long _logNh, double _final_scale, SEALContext &_context, CKKSEncoder &_encoder, Evaluator &_evaluator,
                     GaloisKeys &_gal_keys)
            : logn(_logn), n(1L << _logn), logNh(_logNh), Nh(1L << _logNh), final_scale(_final_scale), context(_context),
              encoder(_encoder), evaluator(_evaluator), gal_keys(_gal_keys)
        {
        }
        // slot_index 0 of the six sets
        void set_diagonals(const moai_fused::BootDiagonals3 &d)
        {
            fftcoeff1.assign(1, d.fftcoeff1);
            fftcoeff2.assign(1, d.fftcoeff2);
            fftcoeff3.assign(1, d.fftcoeff3);
            invfftcoeff1.assign(1, d.invfftcoeff1);
            invfftcoeff2.assign(1, d.invfftcoeff2);
            invfftcoeff3.assign(1, d.invfftcoeff3);
        }

        // defined by the generated slices
        void bsgs_linear_transform(Ciphertext &rtncipher, Ciphertext &cipher, int totlen, int basicstep, int coeff_logn,
                                   const vector<vector<complex<double>>> &fftcoeff);
        void rotated_bsgs_linear_transform(Ciphertext &rtncipher, Ciphertext &cipher, int totlen, int basicstep, int coeff_logn,
                                           const vector<vector<complex<double>>> &fftcoeff);
        void sflinv_full_3(Ciphertext &rtncipher, Ciphertext &cipher);
        void sfl_full_3(Ciphertext &rtncipher, Ciphertext &cipher);
        void coefftoslot_full_3(Ciphertext &rtncipher1, Ciphertext &rtncipher2, Ciphertext &cipher);
        void slottocoeff_full_3(Ciphertext &rtncipher, Ciphertext &cipher1, Ciphertext &cipher2);
        void bootstrap_full_3(Ciphertext &rtncipher, Ciphertext &cipher);

        // not sliced (see the header comment)
        void modraise_inplace(Ciphertext &cipher)
        {
            if (cipher.size() != 2)
            {
                throw invalid_argument("Ciphertexts of size 2 are supported only!");
            }
            if (cipher.coeff_modulus_size() != 1)
            {
                throw invalid_argument("Ciphertexts in the lowest level are supported only!");
            }
            if (!cipher.is_ntt_form())
            {
                evaluator.transform_to_ntt_inplace(cipher);
            }
            Ciphertext raised;
            raised.resize_batch(context, context.first_parms_id(), 2, cipher.batch());
            hip_check(moai_modraise(context.device(), cipher.device_data(), raised.device_data(), raised.coeff_modulus_size(), cipher.batch(),
                                    context.stream()));
            raised.is_ntt_form() = true;
            raised.scale() = cipher.scale();
            cipher = std::move(raised);
        }
        // Bootstrapper.cpp:3496-3502, full-slot branch
        void bootstrap_3(Ciphertext &rtncipher, Ciphertext &cipher)
        {
            initial_scale = cipher.scale();
            bootstrap_full_3(rtncipher, cipher);
        }
    };

#include "generated/ref_bootstrapper_slices.inc"
} // namespace refslice
