// seal/moai_bootstrap_lt.h -- the baby-step / giant-step linear transforms of MOAI's bootstrapping for a
// BATCH of ciphertexts with the plaintext diagonals encoded once and kept on the device.
//
// What it replaces: Bootstrapper::bsgs_linear_transform and ::rotated_bsgs_linear_transform
// (include/source/bootstrapping/Bootstrapper.cpp:1997-2062, 2064-2129), which MOAI calls once per
// ciphertext from an OpenMP loop; per call they rotate the input gs times, and for every diagonal run
// rotation() (common/func.cpp:216-225), a full-level CKKSEncoder::encode, mod_switch_to, multiply_plain and
// add_inplace_reduced_error.  The ciphertexts this class returns are the ones that sequence produces, bit for
// bit (tests/cpp/test_bootstrap_lt.cpp transcribes the sequence through the evaluator and compares):
//   * a rotation of the batch is one batched key switch (moai_apply_galois, batch = number of inputs);
//   * the diagonals of a (level, scale) pair are encoded in ONE moai_ckks_encode call on first use and cached
//     -- encoding at the ciphertext's level gives the rows the reference gets by encoding at the top level
//     and dropping rows (the coefficients do not depend on the level);
//   * the multiply_plain + add chain of a giant step is one moai_ct_pt_dot pass over the batch.
// Modular sums are associative, so the order of the additions does not matter for the residues; scale and
// level bookkeeping follow the reference (product scale = input scale squared, level unchanged).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <complex>
#include <map>
#include <memory>

#include "seal/moai_fused.h"
#include "seal/seal.h"

namespace moai_fused
{
    // include/source/bootstrapping/common/func.cpp:204-214
    inline int giantstep(int M)
    {
        int minval = M, mink = 1, currval;
        for (int k = 1; k <= 3 * std::sqrt(M); k++)
        {
            currval = static_cast<int>(std::ceil((M + 0.0) / (k + 0.0))) + k - 1;
            if (currval < minval)
            {
                minval = currval;
                mink = k;
            }
        }
        return mink;
    }

    // common/func.cpp:216-225; a negative index is undefined behaviour there and refused here
    inline void rotation(int logslot, int Nh, int shiftcount, const std::vector<std::complex<double>> &vec,
                         std::vector<std::complex<double>> &rtnvec)
    {
        int slotlen = (1 << logslot);
        int repeatcount = Nh / slotlen;
        rtnvec.clear();
        for (int j = 0; j < repeatcount; j++)
        {
            for (int i = 0; i < slotlen; i++)
            {
                int idx = (slotlen + i + shiftcount) % slotlen;
                if (idx < 0)
                {
                    throw std::invalid_argument("rotation: shift below -slotlen");
                }
                rtnvec.push_back(vec[static_cast<std::size_t>(idx)]);
            }
        }
    }

    class BsgsLinearTransform
    {
    public:
        // rotated = false: Bootstrapper::bsgs_linear_transform (fftcoeff has 2*totlen+1 diagonals, indexed
        // from -totlen); rotated = true: ::rotated_bsgs_linear_transform (totlen+1 diagonals from 0)
        BsgsLinearTransform(const seal::SEALContext &context, int Nh, int totlen, int basicstep, int coeff_logn,
                            const std::vector<std::vector<std::complex<double>>> &fftcoeff, bool rotated)
            : context_(context), Nh_(Nh)
        {
            if (!rotated)
            {
                // Bootstrapper.cpp:1999-2003
                int gs1 = giantstep(2 * totlen + 1);
                int basicstart1 = -totlen + gs1 * static_cast<int>(std::floor((totlen + 0.0) / (gs1 + 0.0)));
                int giantfirst1 = -static_cast<int>(std::floor((totlen + 0.0) / (gs1 + 0.0)));
                int giantlast1 = static_cast<int>(std::floor((2 * totlen + 0.0) / (gs1 + 0.0))) + giantfirst1;
                for (int i = basicstart1; i < basicstart1 + gs1; i++)
                {
                    baby_steps_.push_back(i == 0 ? 0 : (Nh + i * basicstep) % Nh); // :2017-2022
                }
                for (int i = giantfirst1; i <= giantlast1; i++)
                {
                    Giant g;
                    g.step = i != 0 ? (Nh + i * gs1 * basicstep) % Nh : 0; // :2049-2050
                    int jlast = i != giantlast1 ? basicstart1 + gs1 - 1 : totlen - i * gs1; // :2027, :2038
                    for (int j = basicstart1; j <= jlast; j++)
                    {
                        g.baby.push_back(static_cast<std::uint32_t>(j - basicstart1));
                        g.diag.push_back(add_diagonal(coeff_logn, (-i) * gs1 * basicstep, fftcoeff.at(static_cast<std::size_t>((i * gs1 + j) + totlen))));
                    }
                    giants_.push_back(std::move(g));
                }
            }
            else
            {
                // Bootstrapper.cpp:2066-2067
                int gs2 = giantstep(totlen + 1);
                int giantlast2 = static_cast<int>(std::floor((totlen + 0.0) / (gs2 + 0.0)));
                for (int i = 0; i < gs2; i++)
                {
                    baby_steps_.push_back(i == 0 ? 0 : (Nh + i * basicstep) % Nh); // :2082-2088
                }
                for (int i = 0; i <= giantlast2; i++)
                {
                    Giant g;
                    g.step = i != 0 ? (Nh + i * gs2 * basicstep) % Nh : 0; // :2115-2116
                    int jlast = i != giantlast2 ? gs2 - 1 : totlen - i * gs2; // :2093, :2104
                    for (int j = 0; j <= jlast; j++)
                    {
                        g.baby.push_back(static_cast<std::uint32_t>(j));
                        g.diag.push_back(add_diagonal(coeff_logn, (-i) * gs2 * basicstep, fftcoeff.at(static_cast<std::size_t>(i * gs2 + j))));
                    }
                    giants_.push_back(std::move(g));
                }
            }
            for (auto &g : giants_)
            {
                if (g.baby.size() > 64)
                {
                    throw std::logic_error("more than 64 diagonals in one giant step");
                }
            }
        }

        std::size_t diagonal_count() const
        {
            return diagonals_.size() / static_cast<std::size_t>(Nh_);
        }
        std::size_t key_switches_per_ciphertext(const seal::GaloisKeys &keys) const
        {
            std::size_t c = 0;
            std::vector<std::uint32_t> seq;
            for (int s : baby_steps_)
            {
                seq.clear();
                detail::rotation_sequence(context_, keys, s, seq);
                c += seq.size();
            }
            for (auto &g : giants_)
            {
                seq.clear();
                detail::rotation_sequence(context_, keys, g.step, seq);
                c += seq.size();
            }
            return c;
        }

        // out[b] = [rotated_]bsgs_linear_transform(in[b]); every input at the same level and scale
        void apply(const std::vector<seal::Ciphertext> &in, std::vector<seal::Ciphertext> &out,
                   const seal::GaloisKeys &gal_keys)
        {
            using namespace seal;
            const std::size_t B = in.size();
            std::vector<Ciphertext> result(B);
            if (B == 0)
            {
                out.clear();
                return;
            }
            const parms_id_type pid = in[0].parms_id();
            const double scale = in[0].scale();
            for (auto &c : in)
            {
                if (c.parms_id() != pid || c.scale() != scale || c.batch() != 1)
                {
                    throw std::invalid_argument("the batch must share one level and one scale");
                }
                if (c.size() != 2 || !c.is_ntt_form())
                {
                    throw std::invalid_argument("encrypted must be a size-2 ciphertext in NTT form");
                }
            }
            const std::size_t ct_words = 2 * in[0].coeff_modulus_size() * context_.n();
            void *st = context_.stream();
            run(
                pid, scale, B, gal_keys,
                [&](std::uint64_t *src) {
                    for (std::size_t b = 0; b < B; b++)
                    {
                        util::hip_check(moai_memcpy_d2d(src + b * ct_words, in[b].device_data(), ct_words * 8, st));
                    }
                },
                [&](const std::uint64_t *acc, double new_scale) {
                    for (std::size_t b = 0; b < B; b++)
                    {
                        result[b].resize(context_, pid, 2);
                        util::hip_check(moai_memcpy_d2d(result[b].device_data(), acc + b * ct_words, ct_words * 8, st));
                        result[b].is_ntt_form() = true;
                        result[b].scale() = new_scale;
                    }
                });
            out = std::move(result);
        }

        // the same on a packed ciphertext (moai_fused::pack): out may be the input object
        void apply(const seal::Ciphertext &in, seal::Ciphertext &out, const seal::GaloisKeys &gal_keys)
        {
            using namespace seal;
            if (in.size() != 2 || !in.is_ntt_form())
            {
                throw std::invalid_argument("encrypted must be a size-2 ciphertext in NTT form");
            }
            const parms_id_type pid = in.parms_id();
            const std::size_t B = in.batch();
            const std::size_t batch_words = B * 2 * in.coeff_modulus_size() * context_.n();
            void *st = context_.stream();
            Ciphertext result;
            run(
                pid, in.scale(), B, gal_keys,
                [&](std::uint64_t *src) { util::hip_check(moai_memcpy_d2d(src, in.device_data(), batch_words * 8, st)); },
                [&](const std::uint64_t *acc, double new_scale) {
                    result.resize_batch(context_, pid, 2, B);
                    util::hip_check(moai_memcpy_d2d(result.device_data(), acc, batch_words * 8, st));
                    result.is_ntt_form() = true;
                    result.scale() = new_scale;
                });
            out = std::move(result);
        }

    private:
        template <typename Load, typename Store>
        void run(const seal::parms_id_type &pid, double scale, std::size_t B, const seal::GaloisKeys &gal_keys, Load load, Store store)
        {
            using namespace seal;
            auto cd = context_.get_context_data(pid);
            if (!cd)
            {
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            }
            if (gal_keys.parms_id() != context_.key_parms_id())
            {
                throw std::invalid_argument("galois_keys is not valid for encryption parameters");
            }
            // multiply_plain's scale check (SEAL/evaluator.cpp:2351-2357)
            const double new_scale = scale * scale;
            if (new_scale <= 0 || (static_cast<int>(std::log2(new_scale)) >= cd->total_coeff_modulus_bit_count()))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            const std::size_t L = cd->parms().coeff_modulus().size(), n = context_.n();
            const std::size_t ct_words = 2 * L * n, batch_words = B * ct_words;
            void *st = context_.stream();
            moai_ctx *dev = context_.device();
            const std::uint64_t *plain = encoded(pid, scale);

            // babies: [n_baby][B][2][L][N]
            const std::size_t nb = baby_steps_.size();
            util::DeviceArray babies(nb * batch_words, st);
            std::size_t zero_baby = nb;
            for (std::size_t k = 0; k < nb; k++)
            {
                if (baby_steps_[k] == 0)
                {
                    zero_baby = k;
                }
            }
            std::uint64_t *src = babies.get() + (zero_baby < nb ? zero_baby : 0) * batch_words;
            util::DeviceArray src_own;
            if (zero_baby == nb)
            {
                src_own.resize(batch_words, st);
                src = src_own.get();
            }
            load(src);
            std::vector<std::uint32_t> seq;
            // The baby steps rotate ONE source: when every step has its own key (the bootstrapping key list has them all,
            // Bootstrapper.cpp:89-184) and the steps sit back to back in `babies`, they are one hoisted call -- one digit
            // decomposition instead of one per step, same bits (include/moai_hip.h, moai_apply_galois_hoisted).
            // MOAI_SHIM_HOIST=0 keeps the separate calls.
            if (!hoisted_babies(src, babies.get(), batch_words, zero_baby, L, B, gal_keys))
            {
                for (std::size_t k = 0; k < nb; k++)
                {
                    if (k == zero_baby)
                    {
                        continue;
                    }
                    std::uint64_t *dst = babies.get() + k * batch_words;
                    rotate_batch_from(src, dst, baby_steps_[k], L, B, gal_keys, seq);
                }
            }
            // Giant steps two at a time where their baby-step lists nest (all of a transform's giant steps use the same babies,
            // the last one a leading part of them): one pass over the babies feeds both inner sums (moai_ct_pt_dot2).
            // MOAI_SHIM_DOT2=0 keeps one moai_ct_pt_dot per giant step.
            static const bool pairs = [] {
                const char *e = std::getenv("MOAI_SHIM_DOT2");
                return !(e && e[0] == '0');
            }();
            util::DeviceArray acc(batch_words, st), giant(batch_words, st), giant2;
            bool first = true;
            std::vector<std::uint32_t> xi, pi, pi2;
            auto fold = [&](std::uint64_t *dst, const Giant &g) {
                if (g.step != 0 && !first)
                {
                    // the rotation's last key switch adds its result to the running sum itself (moai_apply_galois_acc)
                    seq.clear();
                    detail::rotation_sequence(context_, gal_keys, g.step, seq);
                    if (!seq.empty())
                    {
                        for (std::size_t h = 0; h + 1 < seq.size(); h++)
                        {
                            util::hip_check(moai_apply_galois(dev, dst, L, seq[h], gal_keys.device_key(seal::GaloisKeys::get_index(seq[h]), L), B, st));
                        }
                        util::hip_check(moai_apply_galois_acc(dev, dst, acc.get(), L, seq.back(),
                                                              gal_keys.device_key(seal::GaloisKeys::get_index(seq.back()), L), B, st));
                        return;
                    }
                }
                if (g.step != 0)
                {
                    rotate_batch(dst, g.step, L, B, gal_keys, seq);
                }
                if (first)
                {
                    if (dst != acc.get())
                    {
                        util::hip_check(moai_memcpy_d2d(acc.get(), dst, batch_words * 8, st));
                    }
                    first = false;
                }
                else
                {
                    util::hip_check(moai_add(dev, acc.get(), dst, acc.get(), B * 2, L, st));
                }
            };
            auto leads = [](const Giant &shorter, const Giant &longer) {
                return shorter.baby.size() <= longer.baby.size() && std::equal(shorter.baby.begin(), shorter.baby.end(), longer.baby.begin());
            };
            for (std::size_t k = 0; k < giants_.size();)
            {
                const Giant &ga = giants_[k];
                std::uint64_t *dst = first && ga.step == 0 ? acc.get() : giant.get();
                const Giant *gb = pairs && k + 1 < giants_.size() ? &giants_[k + 1] : nullptr;
                if (gb && (leads(*gb, ga) || leads(ga, *gb)))
                {
                    if (!giant2.get())
                    {
                        giant2.resize(batch_words, st);
                    }
                    const bool a_long = leads(*gb, ga);
                    const Giant &lng = a_long ? ga : *gb, &sht = a_long ? *gb : ga;
                    xi.assign(lng.baby.begin(), lng.baby.end());
                    pi.assign(lng.diag.begin(), lng.diag.end());
                    pi2.assign(sht.diag.begin(), sht.diag.end());
                    util::hip_check(moai_ct_pt_dot2(dev, babies.get(), plain, a_long ? dst : giant2.get(), a_long ? giant2.get() : dst, xi.data(),
                                                    pi.data(), pi2.data(), xi.size(), pi2.size(), B * 2, L, st));
                    fold(dst, ga);
                    fold(giant2.get(), *gb);
                    k += 2;
                    continue;
                }
                xi.assign(ga.baby.begin(), ga.baby.end());
                pi.assign(ga.diag.begin(), ga.diag.end());
                util::hip_check(moai_ct_pt_dot(dev, babies.get(), plain, dst, xi.data(), pi.data(), xi.size(), B * 2, L, st));
                fold(dst, ga);
                k += 1;
            }
            store(acc.get(), new_scale);
            context_.sync(); // staging buffers go out of scope
        }

        struct Giant
        {
            int step = 0;
            std::vector<std::uint32_t> baby, diag;
        };

        std::uint32_t add_diagonal(int coeff_logn, int shift, const std::vector<std::complex<double>> &coeff)
        {
            std::vector<std::complex<double>> rotated;
            rotation(coeff_logn, Nh_, shift, coeff, rotated);
            if (rotated.size() != static_cast<std::size_t>(Nh_))
            {
                throw std::invalid_argument("diagonal length does not divide the slot count");
            }
            diagonals_.insert(diagonals_.end(), rotated.begin(), rotated.end());
            return static_cast<std::uint32_t>(diagonals_.size() / static_cast<std::size_t>(Nh_) - 1);
        }

        // the first key switch of every non-zero baby step in one moai_apply_galois_hoisted call (for a step with its own key that
        // is the whole rotation); false when that form does not apply
        bool hoisted_babies(const std::uint64_t *src, std::uint64_t *babies, std::size_t batch_words, std::size_t zero_baby, std::size_t L,
                            std::size_t B, const seal::GaloisKeys &keys) const
        {
            static const bool enabled = [] {
                const char *e = std::getenv("MOAI_SHIM_HOIST");
                return !(e && e[0] == '0');
            }();
            const std::size_t nb = baby_steps_.size();
            if (std::getenv("MOAI_SHIM_HOIST_DEBUG"))
            {
                std::fprintf(stderr, "[hoist] L=%zu: baby steps", L);
                for (int s : baby_steps_)
                {
                    std::fprintf(stderr, " %d", s);
                }
                std::fprintf(stderr, "; giant steps");
                for (auto &g : giants_)
                {
                    std::fprintf(stderr, " %d", g.step);
                }
                std::fprintf(stderr, "\n");
            }
            if (!enabled || nb < 3 || context_.logn() < 12)
            {
                return false;
            }
            const std::size_t count = zero_baby < nb ? nb - 1 : nb;
            std::vector<std::uint32_t> elts(count);
            std::vector<const std::uint64_t *> kptr(count), cptr(count);
            std::vector<std::uint64_t *> optr(count);
            // a step without its own key takes the NAF path of rotate_internal (SEAL/evaluator.cpp:2699-2721): several key
            // switches in a row.  Its FIRST one still starts from the shared source and joins the hoisted call; the others
            // follow in place.
            std::vector<std::vector<std::uint32_t>> rest(count);
            std::vector<std::uint32_t> seq;
            for (std::size_t k = 0, i = 0; k < nb; k++)
            {
                if (k == zero_baby)
                {
                    continue;
                }
                seq.clear();
                detail::rotation_sequence(context_, keys, baby_steps_[k], seq);
                if (seq.empty())
                {
                    return false;
                }
                elts[i] = seq[0];
                rest[i].assign(seq.begin() + 1, seq.end());
                const std::size_t index = seal::GaloisKeys::get_index(seq[0]);
                kptr[i] = keys.device_key(index, L);
                cptr[i] = keys.hoist_correction(context_, index, seq[0], L);
                optr[i] = babies + k * batch_words;
                i++;
            }
            int fell_back = 0;
            seal::util::hip_check(moai_apply_galois_hoisted(context_.device(), src, optr.data(), L, elts.data(), kptr.data(), cptr.data(), count, B,
                                                            &fell_back, context_.stream()));
            for (std::size_t i = 0; i < count; i++)
            {
                for (std::uint32_t elt : rest[i])
                {
                    seal::util::hip_check(moai_apply_galois(context_.device(), optr[i], L, elt, keys.device_key(seal::GaloisKeys::get_index(elt), L), B,
                                                            context_.stream()));
                }
            }
            return true;
        }

        // the same from `src` into `dst`: the first key switch reads the source, the rest work in place
        void rotate_batch_from(const std::uint64_t *src, std::uint64_t *dst, int step, std::size_t L, std::size_t B,
                               const seal::GaloisKeys &keys, std::vector<std::uint32_t> &seq) const
        {
            seq.clear();
            detail::rotation_sequence(context_, keys, step, seq);
            if (seq.empty())
            {
                seal::util::hip_check(moai_memcpy_d2d(dst, src, B * 2 * L * context_.n() * 8, context_.stream()));
                return;
            }
            for (std::size_t h = 0; h < seq.size(); h++)
            {
                const std::uint32_t elt = seq[h];
                seal::util::hip_check(moai_apply_galois_to(context_.device(), h == 0 ? src : dst, dst, L, elt,
                                                           keys.device_key(seal::GaloisKeys::get_index(elt), L), B, context_.stream()));
            }
        }

        // rotate_vector's key switches (Evaluator::rotate_internal) on a whole batch in place
        void rotate_batch(std::uint64_t *data, int step, std::size_t L, std::size_t B, const seal::GaloisKeys &keys,
                          std::vector<std::uint32_t> &seq) const
        {
            seq.clear();
            detail::rotation_sequence(context_, keys, step, seq);
            for (std::uint32_t elt : seq)
            {
                seal::util::hip_check(moai_apply_galois(context_.device(), data, L, elt,
                                                        keys.device_key(seal::GaloisKeys::get_index(elt), L), B, context_.stream()));
            }
        }

        // all diagonals encoded at (level, scale): CKKSEncoder::encode(value, scale) followed by
        // mod_switch_to (SEAL/evaluator.h:1371-1378), i.e. the first L rows of the same coefficients
        const std::uint64_t *encoded(const seal::parms_id_type &pid, double scale)
        {
            auto key = std::make_pair(pid, scale);
            auto it = cache_.find(key);
            if (it != cache_.end())
            {
                return it->second->get();
            }
            using namespace seal;
            auto cd = context_.get_context_data(pid);
            auto top = context_.first_context_data();
            // the reference encodes at the first level: its checks run against that level's modulus
            if (scale <= 0 || (static_cast<int>(std::log2(scale)) + 1 >= top->total_coeff_modulus_bit_count()))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            const std::size_t L = cd->parms().coeff_modulus().size(), n = context_.n();
            const std::size_t nd = diagonal_count();
            void *st = context_.stream();
            auto buf = std::make_shared<util::DeviceArray>(nd * L * n, st);
            const std::size_t words = nd * static_cast<std::size_t>(Nh_) * 2;
            util::DeviceArray staging(words + nd, st);
            util::hip_check(moai_memcpy_h2d(staging.get(), reinterpret_cast<const double *>(diagonals_.data()), words * 8, st));
            double *max_dev = reinterpret_cast<double *>(staging.get() + words);
            util::hip_check(moai_ckks_encode(context_.device(), reinterpret_cast<const double *>(staging.get()), 1,
                                             static_cast<std::size_t>(Nh_), nd, buf->get(), L, nullptr, scale, max_dev, st));
            std::vector<double> mx(nd);
            util::hip_check(moai_memcpy_d2h(mx.data(), max_dev, nd * 8, st));
            context_.sync();
            for (double m : mx)
            {
                int bits = static_cast<int>(std::ceil(std::log2(std::max<>(m, 1.0)))) + 1;
                if (!(bits < top->total_coeff_modulus_bit_count()))
                {
                    throw std::invalid_argument("encoded values are too large");
                }
            }
            cache_[key] = buf;
            return buf->get();
        }

        seal::SEALContext context_;
        int Nh_;
        std::vector<int> baby_steps_;
        std::vector<Giant> giants_;
        std::vector<std::complex<double>> diagonals_; // [n_diag][Nh], already run through rotation()
        std::map<std::pair<seal::parms_id_type, double>, std::shared_ptr<seal::util::DeviceArray>> cache_;
    };
} // namespace moai_fused
