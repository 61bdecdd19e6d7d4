// bootstrapping/Bootstrapper.h -- GPU-backed drop-in for MOAI's include/source/bootstrapping/Bootstrapper.h, so that
// softmax.hpp, single_att_block.hpp and the drivers under include/test compile UNCHANGED against seal_shim/
// (add -I<package>/seal_shim/bootstrapping in place of the reference's include/source/bootstrapping).
//
// Interface kept (Bootstrapper.h:14-221, Bootstrapper.cpp): the constructor of :52-69, the public data members, and the
// "level-3" full-slot family MOAI's drivers use --
//   prepare_mod_polynomial (:1973-1977), addLeftRotKeys_Linear_to_vector_3 (:89-184), addBootKeys_3 (:374-392),
//   change_logn (:508-520), genorigcoeff/genfftcoeff_3/geninvfftcoeff_3 through generate_LT_coefficient_3 (:1967-1971),
//   bsgs_linear_transform / rotated_bsgs_linear_transform (:1997-2129), sflinv_full_3 / sfl_full_3 (:2602-2623, :2460-2497),
//   coefftoslot_full_3 / slottocoeff_full_3 (:2742-2777), modraise_inplace (:2938-2992), bootstrap_full_3 (:3231-3251),
//   bootstrap_3 / bootstrap_inplace_3 (:3496-3508), set_final_scale.
// Not provided: the two-level, one-depth, hoisting, "real" and sparse-slot (logn < logNh) variants, which no MOAI
// driver calls; they throw std::logic_error.
//
// What is different inside:
//  * no NTL: the modular-reduction polynomial comes from bootstrapping/moai_remez.h, the transform diagonals from
//    bootstrapping/moai_fft_diagonals.h (both re-derived; see those files for what pins them);
//  * the evaluation runs on the device through seal/moai_bootstrap_eval.h (PackedBootstrapper3): the evaluator calls of the
//    reference's routines in their order, with the plaintext diagonals encoded once per (level, scale) and cached;
//  * bootstrap_3 is what MOAI calls from its OpenMP loops, one ciphertext per call (include/test/test_full_scheme.hpp:
//    654-660).  Concurrent callers are gathered into ONE packed run (leader / followers, bounded waits): every ciphertext
//    gets exactly the result of its own call -- the packed kernels compute each member independently, bit-identical to
//    a single-ciphertext run (tests/cpp/test_bootstrap_real.cpp) -- but the device sees batches of the caller count.
//    MOAI_BOOT_COMBINE_US sets the gathering window (default 8000 us -- a caller without company waits a quarter of it, 0.1 % of
//    a packed run and 1.5 % of a single bootstrap; 0 = never gather), MOAI_BOOT_MAX_PACK the
//    largest pack (default 48).
#pragma once

#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <tuple>

#include "ModularReducer.h"
#include "moai_fft_diagonals.h"
#include "seal/moai_bootstrap_eval.h"

// the reference's header opens these namespaces for everything that includes it (Bootstrapper.h:10-12); MOAI's
// headers rely on that
using namespace std;
using namespace seal;
using namespace seal::util;

class Bootstrapper
{
public:
    long loge;
    long logn;
    long n;
    long logNh;
    long Nh;
    long L;

    double initial_scale = 1.0;
    double final_scale;

    long boundary_K;
    long sin_cos_deg;
    long scale_factor;
    long inverse_deg;

    SEALContext &context;
    KeyGenerator &keygen;
    CKKSEncoder &encoder;
    Encryptor &encryptor;
    Decryptor &decryptor;
    Evaluator &evaluator;
    RelinKeys &relin_keys;
    GaloisKeys &gal_keys;

    vector<long> slot_vec;
    long slot_index = 0;
    // [slot index][diagonal][slot], the reference's layout (Bootstrapper.h:43-46)
    vector<vector<vector<complex<double>>>> fftcoeff1, fftcoeff2, fftcoeff3;
    vector<vector<vector<complex<double>>>> invfftcoeff1, invfftcoeff2, invfftcoeff3;

    ModularReducer *mod_reducer;

    Bootstrapper(long _loge, long _logn, long _logNh, long _L, double _final_scale, long _boundary_K, long _sin_cos_deg,
                 long _scale_factor, long _inverse_deg, SEALContext &_context, KeyGenerator &_keygen, CKKSEncoder &_encoder,
                 Encryptor &_encryptor, Decryptor &_decryptor, Evaluator &_evaluator, RelinKeys &_relin_keys, GaloisKeys &_gal_keys)
        : loge(_loge), logn(_logn), logNh(_logNh), L(_L), final_scale(_final_scale), boundary_K(_boundary_K),
          sin_cos_deg(_sin_cos_deg), scale_factor(_scale_factor), inverse_deg(_inverse_deg), context(_context), keygen(_keygen),
          encoder(_encoder), encryptor(_encryptor), decryptor(_decryptor), evaluator(_evaluator), relin_keys(_relin_keys),
          gal_keys(_gal_keys)
    {
        n = 1 << logn;
        Nh = 1 << logNh;
        mod_reducer = new ModularReducer(boundary_K, static_cast<double>(loge), sin_cos_deg, scale_factor, inverse_deg, context, encoder,
                                         encryptor, evaluator, relin_keys, decryptor);
        const char *e = std::getenv("MOAI_BOOT_COMBINE_US");
        combine_us_ = e ? std::atol(e) : 8000;
        e = std::getenv("MOAI_BOOT_MAX_PACK");
        max_pack_ = e ? static_cast<std::size_t>(std::atol(e)) : 48;
        if (max_pack_ < 1)
        {
            max_pack_ = 1;
        }
    }
    Bootstrapper(const Bootstrapper &) = delete;
    Bootstrapper &operator=(const Bootstrapper &) = delete;
    ~Bootstrapper()
    {
        delete mod_reducer;
    }

    inline void set_final_scale(double _final_scale)
    {
        std::lock_guard<std::mutex> run(run_mu_); // not while a bootstrap holds a reference to the engine
        final_scale = _final_scale;
        std::lock_guard<std::mutex> g(engine_mu_);
        engine_.reset();
    }

    // ---- keys ------------------------------------------------------------------------------------------------------
    void addLeftRotKeys_Linear_to_vector_3(vector<int> &gal_steps_vector)
    {
        moai_fused::boot_rotation_steps_3(static_cast<int>(logn), static_cast<int>(logNh), gal_steps_vector);
    }
    void addBootKeys_3(GaloisKeys &keys)
    {
        vector<int> gal_steps_vector;
        gal_steps_vector.push_back(0);
        for (int i = 0; i < logNh; i++)
        {
            gal_steps_vector.push_back((1 << i));
        }
        addLeftRotKeys_Linear_to_vector_3(gal_steps_vector);
        keygen.create_galois_keys(gal_steps_vector, keys);
        slot_vec.push_back(logn);
        select_slot_index();
    }
    void change_logn(long new_logn)
    {
        std::lock_guard<std::mutex> run(run_mu_); // not while a bootstrap holds a reference to the engine
        logn = new_logn;
        n = (1 << logn);
        select_slot_index();
        std::lock_guard<std::mutex> g(engine_mu_);
        engine_.reset();
    }

    // ---- constants -------------------------------------------------------------------------------------------------
    void prepare_mod_polynomial()
    {
        mod_reducer->generate_sin_cos_polynomial();
        mod_reducer->generate_inverse_sine_polynomial();
    }
    // the per-stage matrices live inside moai_fft_diagonals.h; kept as an entry point for callers of the reference's name
    void genorigcoeff()
    {
    }
    void genfftcoeff_3()
    {
        generate_sets(true, false);
    }
    void geninvfftcoeff_3()
    {
        generate_sets(false, true);
    }
    void generate_LT_coefficient_3()
    {
        genorigcoeff();
        generate_sets(true, true);
    }

    // ---- linear transforms (one ciphertext or a pack) ---------------------------------------------------------------
    void bsgs_linear_transform(Ciphertext &rtncipher, Ciphertext &cipher, int totlen, int basicstep, int coeff_logn,
                               const vector<vector<complex<double>>> &fftcoeff)
    {
        transform(false, totlen, basicstep, coeff_logn, fftcoeff).apply(cipher, rtncipher, gal_keys);
    }
    void rotated_bsgs_linear_transform(Ciphertext &rtncipher, Ciphertext &cipher, int totlen, int basicstep, int coeff_logn,
                                       const vector<vector<complex<double>>> &fftcoeff)
    {
        transform(true, totlen, basicstep, coeff_logn, fftcoeff).apply(cipher, rtncipher, gal_keys);
    }
    void sflinv_full_3(Ciphertext &rtncipher, Ciphertext &cipher)
    {
        engine().sflinv_full_3(rtncipher, cipher);
    }
    void sfl_full_3(Ciphertext &rtncipher, Ciphertext &cipher)
    {
        auto &e = engine();
        e.initial_scale() = initial_scale;
        e.sfl_full_3(rtncipher, cipher);
    }
    void coefftoslot_full_3(Ciphertext &rtncipher1, Ciphertext &rtncipher2, Ciphertext &cipher)
    {
        engine().coefftoslot_full_3(rtncipher1, rtncipher2, cipher);
    }
    void slottocoeff_full_3(Ciphertext &rtncipher, Ciphertext &cipher1, Ciphertext &cipher2)
    {
        auto &e = engine();
        e.initial_scale() = initial_scale;
        e.slottocoeff_full_3(rtncipher, cipher1, cipher2);
    }
    void modraise_inplace(Ciphertext &cipher)
    {
        engine().modraise_inplace(cipher);
    }

    // ---- the bootstrap ----------------------------------------------------------------------------------------------
    // one packed (or single) ciphertext straight through the pipeline; `cipher` is consumed like the reference's
    void bootstrap_full_3(Ciphertext &rtncipher, Ciphertext &cipher)
    {
        std::lock_guard<std::mutex> run(run_mu_);
        engine().bootstrap_3(rtncipher, cipher);
    }
    void bootstrap_3(Ciphertext &rtncipher, Ciphertext &cipher)
    {
        initial_scale = cipher.scale(); // the reference writes this member from every calling thread as well (:3497)
        if (logn != logNh)
        {
            throw std::logic_error("bootstrap_sparse_3 (logn < logNh) is not provided");
        }
        if (cipher.batch() != 1 || combine_us_ <= 0 || max_pack_ == 1)
        {
            bootstrap_full_3(rtncipher, cipher);
            return;
        }
        gather_and_run(rtncipher, cipher);
    }
    void bootstrap_inplace_3(Ciphertext &cipher)
    {
        Ciphertext rtncipher;
        bootstrap_3(rtncipher, cipher);
        cipher = rtncipher;
    }
    // how the calls of this object were grouped so far: {packed runs, ciphertexts}
    std::pair<std::size_t, std::size_t> gather_statistics() const
    {
        std::lock_guard<std::mutex> run(run_mu_);
        return { runs_, members_ };
    }

    // the variants no MOAI driver calls
    void bootstrap(Ciphertext &, Ciphertext &)
    {
        unsupported("bootstrap");
    }
    void bootstrap_inplace(Ciphertext &)
    {
        unsupported("bootstrap_inplace");
    }
    void bootstrap_sparse_3(Ciphertext &, Ciphertext &)
    {
        unsupported("bootstrap_sparse_3");
    }
    void bootstrap_real_3(Ciphertext &, Ciphertext &)
    {
        unsupported("bootstrap_real_3");
    }
    void bootstrap_hoisting(Ciphertext &, Ciphertext &)
    {
        unsupported("bootstrap_hoisting");
    }
    void generate_LT_coefficient()
    {
        unsupported("generate_LT_coefficient");
    }
    void addBootKeys(GaloisKeys &)
    {
        unsupported("addBootKeys");
    }

private:
    [[noreturn]] static void unsupported(const char *what)
    {
        throw std::logic_error(std::string("Bootstrapper::") + what + " is not provided: only the level-3 full-slot family is");
    }
    void select_slot_index()
    {
        slot_index = -1;
        for (std::size_t i = 0; i < slot_vec.size(); i++)
        {
            if (slot_vec[i] == logn)
            {
                slot_index = static_cast<long>(i);
                break;
            }
        }
        if (slot_index == -1)
        {
            throw std::invalid_argument("LT coefficients were not generated for this logn");
        }
    }
    void generate_sets(bool forward, bool inverse)
    {
        auto size_to = [&](vector<vector<vector<complex<double>>>> &v) { v.resize(slot_vec.size()); };
        size_to(fftcoeff1);
        size_to(fftcoeff2);
        size_to(fftcoeff3);
        size_to(invfftcoeff1);
        size_to(invfftcoeff2);
        size_to(invfftcoeff3);
        for (std::size_t u = 0; u < slot_vec.size(); u++)
        {
            if (slot_vec[u] != logNh)
            {
                throw std::logic_error("sparse-slot transform coefficients (logn < logNh) are not provided");
            }
            moai_boot::LevelThreeDiagonals d = moai_boot::level_three_diagonals(static_cast<int>(slot_vec[u]), boundary_K);
            if (forward)
            {
                fftcoeff1[u] = std::move(d.fftcoeff1);
                fftcoeff2[u] = std::move(d.fftcoeff2);
                fftcoeff3[u] = std::move(d.fftcoeff3);
            }
            if (inverse)
            {
                invfftcoeff1[u] = std::move(d.invfftcoeff1);
                invfftcoeff2[u] = std::move(d.invfftcoeff2);
                invfftcoeff3[u] = std::move(d.invfftcoeff3);
            }
        }
        std::lock_guard<std::mutex> g(engine_mu_);
        engine_.reset();
    }

    // the device pipeline, built on first use from the members above
    moai_fused::PackedBootstrapper3 &engine()
    {
        std::lock_guard<std::mutex> g(engine_mu_);
        if (!engine_)
        {
            if (logn != logNh)
            {
                throw std::logic_error("only logn == logNh is provided");
            }
            select_slot_index();
            const std::size_t u = static_cast<std::size_t>(slot_index);
            if (fftcoeff1.size() <= u || invfftcoeff1.size() <= u || fftcoeff1[u].empty() || invfftcoeff1[u].empty())
            {
                throw std::logic_error("generate_LT_coefficient_3() has not run");
            }
            moai_fused::BootDiagonals3 d;
            d.fftcoeff1 = fftcoeff1[u];
            d.fftcoeff2 = fftcoeff2[u];
            d.fftcoeff3 = fftcoeff3[u];
            d.invfftcoeff1 = invfftcoeff1[u];
            d.invfftcoeff2 = invfftcoeff2[u];
            d.invfftcoeff3 = invfftcoeff3[u];
            engine_.reset(new moai_fused::PackedBootstrapper3(context, encoder, evaluator, relin_keys, gal_keys, static_cast<int>(logn),
                                                              static_cast<int>(logNh), final_scale, d, mod_reducer->packed_reducer()));
        }
        return *engine_;
    }
    moai_fused::BsgsLinearTransform &transform(bool rotated, int totlen, int basicstep, int coeff_logn,
                                               const vector<vector<complex<double>>> &fftcoeff)
    {
        std::lock_guard<std::mutex> g(engine_mu_);
        auto key = std::make_tuple(static_cast<const void *>(&fftcoeff), rotated, totlen, basicstep, coeff_logn);
        auto it = transforms_.find(key);
        if (it == transforms_.end())
        {
            it = transforms_
                     .emplace(key, std::unique_ptr<moai_fused::BsgsLinearTransform>(new moai_fused::BsgsLinearTransform(
                                       context, static_cast<int>(Nh), totlen, basicstep, coeff_logn, fftcoeff, rotated)))
                     .first;
        }
        return *it->second;
    }

    // ---- gathering of concurrent bootstrap_3 calls ------------------------------------------------------------------
    struct Request
    {
        Ciphertext *out;
        Ciphertext *in;
        bool done = false;
        std::exception_ptr error;
    };
    // Every caller queues its request; the caller at the head of the queue leads ONE packed run (its own request is part
    // of it), the others sleep until their request is done or they reach the head.  All waits end: a leader's wait for
    // company is bounded by the window, and a finished run always wakes the queue.
    void gather_and_run(Ciphertext &rtncipher, Ciphertext &cipher)
    {
        Request me{ &rtncipher, &cipher };
        std::unique_lock<std::mutex> lk(gather_mu_);
        pending_.push_back(&me);
        gather_cv_.notify_all();
        while (!me.done)
        {
            if (leader_active_ || pending_.front() != &me)
            {
                gather_cv_.wait(lk);
                continue;
            }
            leader_active_ = true;
            // wait for company: until the pack is full, nobody new arrived for a quarter of the window, or the window ends
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(combine_us_);
            while (pending_.size() < max_pack_)
            {
                const std::size_t seen = pending_.size();
                auto quiet = std::chrono::steady_clock::now() + std::chrono::microseconds(combine_us_ / 4 + 1);
                if (deadline < quiet)
                {
                    quiet = deadline;
                }
                gather_cv_.wait_until(lk, quiet, [&] { return pending_.size() > seen; });
                if (pending_.size() == seen || std::chrono::steady_clock::now() >= deadline)
                {
                    break;
                }
            }
            const std::size_t take = pending_.size() < max_pack_ ? pending_.size() : max_pack_;
            std::vector<Request *> batch(pending_.begin(), pending_.begin() + static_cast<std::ptrdiff_t>(take));
            pending_.erase(pending_.begin(), pending_.begin() + static_cast<std::ptrdiff_t>(take));
            lk.unlock();
            // whatever happens in the run (an allocation that fails outside the groups' own try blocks included), the batch is
            // marked done and the queue is woken: no caller may sleep for ever behind a leader that left with an exception
            std::exception_ptr run_error;
            try
            {
                run_batch(batch);
            }
            catch (...)
            {
                run_error = std::current_exception();
            }
            lk.lock();
            for (Request *r : batch)
            {
                if (run_error && !r->error)
                {
                    r->error = run_error;
                }
                r->done = true;
            }
            leader_active_ = false;
            gather_cv_.notify_all();
        }
        lk.unlock();
        if (me.error)
        {
            std::rethrow_exception(me.error);
        }
    }
    // members are grouped by (level, scale); each group is one packed run
    void run_batch(const std::vector<Request *> &batch)
    {
        std::vector<bool> used(batch.size(), false);
        for (std::size_t i = 0; i < batch.size(); i++)
        {
            if (used[i])
            {
                continue;
            }
            std::vector<std::size_t> group;
            for (std::size_t j = i; j < batch.size(); j++)
            {
                if (!used[j] && batch[j]->in->parms_id() == batch[i]->in->parms_id() && batch[j]->in->scale() == batch[i]->in->scale() &&
                    batch[j]->in->size() == batch[i]->in->size() && batch[j]->in->is_ntt_form() == batch[i]->in->is_ntt_form())
                {
                    group.push_back(j);
                    used[j] = true;
                }
            }
            std::exception_ptr err;
            try
            {
                std::lock_guard<std::mutex> run(run_mu_);
                auto &e = engine();
                if (group.size() == 1)
                {
                    e.bootstrap_3(*batch[group[0]]->out, *batch[group[0]]->in);
                }
                else
                {
                    std::vector<Ciphertext> members;
                    members.reserve(group.size());
                    for (std::size_t j : group)
                    {
                        members.push_back(*batch[j]->in);
                    }
                    Ciphertext packed = moai_fused::pack(members, context), packed_out;
                    members.clear();
                    e.bootstrap_3(packed_out, packed);
                    std::vector<Ciphertext> outs;
                    moai_fused::unpack(packed_out, context, outs);
                    for (std::size_t g = 0; g < group.size(); g++)
                    {
                        *batch[group[g]]->out = std::move(outs[g]);
                    }
                }
                runs_++;
                members_ += group.size();
            }
            catch (...)
            {
                err = std::current_exception();
            }
            for (std::size_t j : group)
            {
                batch[j]->error = err;
            }
        }
    }

    mutable std::mutex engine_mu_, run_mu_, gather_mu_;
    std::condition_variable gather_cv_;
    std::vector<Request *> pending_;
    bool leader_active_ = false;
    long combine_us_ = 2000;
    std::size_t max_pack_ = 48;
    std::size_t runs_ = 0, members_ = 0;
    std::unique_ptr<moai_fused::PackedBootstrapper3> engine_;
    std::map<std::tuple<const void *, bool, int, int, int>, std::unique_ptr<moai_fused::BsgsLinearTransform>> transforms_;
};
