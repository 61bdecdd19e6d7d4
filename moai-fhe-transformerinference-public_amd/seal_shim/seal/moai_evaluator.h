// seal/moai_evaluator.h -- seal::Evaluator over the C ABI.  Same public methods, argument meaning and
// exceptions as SEAL/evaluator.{h,cpp} (CKKS paths) including the fork's additions
// (SEAL/evaluator.cpp:395-594, SEAL/evaluator.h:1296-1386).  Every method validates on the host first,
// exactly where the reference throws, and only then enqueues device work.
#pragma once

namespace seal
{
    namespace util
    {
        // SEAL/util/common.h are_close
        inline bool are_close(double a, double b)
        {
            double scale_factor = std::max({ std::fabs(a), std::fabs(b), 1.0 });
            return std::fabs(a - b) < std::numeric_limits<double>::epsilon() * scale_factor;
        }
    } // namespace util

    class Evaluator
    {
    public:
        // the fork's constructor (SEAL/evaluator.h:84)
        Evaluator(const SEALContext &context, CKKSEncoder &encoder) : context_(context), encoder_(encoder)
        {}
        Evaluator(const Evaluator &) = delete;
        Evaluator &operator=(const Evaluator &) = delete;

        // ---- negate / add / sub ------------------------------------------------------------------------
        void negate_inplace(Ciphertext &encrypted) const
        {
            check_ct(encrypted, "encrypted");
            hip(moai_negate(dev(), encrypted.device_data(), encrypted.device_data(), encrypted.size() * encrypted.batch(),
                            encrypted.coeff_modulus_size(), st()));
        }
        void negate(const Ciphertext &encrypted, Ciphertext &destination) const
        {
            if (&encrypted == &destination)
            {
                negate_inplace(destination);
                return;
            }
            check_ct(encrypted, "encrypted");
            like(destination, encrypted);
            hip(moai_negate(dev(), encrypted.device_data(), destination.device_data(), encrypted.size() * encrypted.batch(),
                            encrypted.coeff_modulus_size(), st()));
        }
        void add_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            addsub(encrypted1, encrypted2, false);
        }
        void add(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const
        {
            if (&encrypted2 == &destination)
            {
                add_inplace(destination, encrypted1);
            }
            else if (&encrypted1 == &destination)
            {
                add_inplace(destination, encrypted2);
            }
            else if (encrypted1.size() == encrypted2.size() && encrypted1.size() >= 2)
            {
                // equal sizes: one kernel straight into the destination, no deep copy first
                check_pair(encrypted1, encrypted2);
                like(destination, encrypted1);
                hip(moai_add(dev(), encrypted1.device_data(), encrypted2.device_data(), destination.device_data(),
                             encrypted1.size() * encrypted1.batch(), encrypted1.coeff_modulus_size(), st()));
            }
            else
            {
                destination = encrypted1;
                add_inplace(destination, encrypted2);
            }
        }
        void add_many(const std::vector<Ciphertext> &encrypteds, Ciphertext &destination) const
        {
            if (encrypteds.empty())
            {
                throw std::invalid_argument("encrypteds cannot be empty");
            }
            for (auto &e : encrypteds)
            {
                if (&e == &destination)
                {
                    throw std::invalid_argument("encrypteds must be different from destination");
                }
            }
            destination = encrypteds[0];
            for (std::size_t i = 1; i < encrypteds.size(); i++)
            {
                add_inplace(destination, encrypteds[i]);
            }
        }
        void sub_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            addsub(encrypted1, encrypted2, true);
        }
        void sub(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const
        {
            if (&encrypted2 == &destination)
            {
                sub_inplace(destination, encrypted1);
                negate_inplace(destination);
            }
            else if (&encrypted1 == &destination)
            {
                sub_inplace(destination, encrypted2);
            }
            else if (encrypted1.size() == encrypted2.size() && encrypted1.size() >= 2)
            {
                check_pair(encrypted1, encrypted2);
                like(destination, encrypted1);
                hip(moai_sub(dev(), encrypted1.device_data(), encrypted2.device_data(), destination.device_data(),
                             encrypted1.size() * encrypted1.batch(), encrypted1.coeff_modulus_size(), st()));
            }
            else
            {
                destination = encrypted1;
                sub_inplace(destination, encrypted2);
            }
        }

        // ---- multiply / square / relinearize -----------------------------------------------------------
        // SEAL/evaluator.cpp:770-909 (ckks_multiply) / :1223-1282 (ckks_square); `out` must not be an operand
        void multiply_into(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &out) const
        {
            check_ct(encrypted1, "encrypted1");
            check_ct(encrypted2, "encrypted2");
            if (encrypted1.parms_id() != encrypted2.parms_id())
            {
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            }
            if (!(encrypted1.is_ntt_form() && encrypted2.is_ntt_form()))
            {
                throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
            }
            const std::size_t size1 = encrypted1.size(), size2 = encrypted2.size();
            if (size1 + size2 - 1 > 16) // product_fits_in: SEAL_CIPHERTEXT_SIZE_MAX (SEAL/evaluator.cpp:790-793)
            {
                throw std::logic_error("invalid parameters");
            }
            auto cd = context_.get_context_data(encrypted1.parms_id());
            double new_scale = encrypted1.scale() * encrypted2.scale();
            if (!scale_ok(new_scale, *cd))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            if (encrypted1.batch() != encrypted2.batch())
            {
                throw std::invalid_argument("encrypted1 and encrypted2 pack different numbers of ciphertexts");
            }
            const std::size_t L = encrypted1.coeff_modulus_size(), B = encrypted1.batch();
            if (size1 == 2 && size2 == 2 && B == 1 && &encrypted1 != &encrypted2 && lazy_products())
            {
                // not computed now: the product is recorded as (block, block) and summed with the products add_inplace brings
                // (Ciphertext::LazyTerm::src2; MOAI's ct x ct loops, Ct_ct_matrix_mul.hpp:32-41, 121-134); a square is always eager
                Ciphertext::LazyTerm term;
                term.src = (encrypted1.materialize(), encrypted1.buf_);
                term.src2 = (encrypted2.materialize(), encrypted2.buf_);
                if (term.src != term.src2)
                {
                    out.release();
                    out.stream_ = st();
                    out.dev_ = dev();
                    out.parms_id_ = encrypted1.parms_id_;
                    out.is_ntt_form_ = true;
                    out.size_ = 3;
                    out.batch_ = 1;
                    out.n_ = encrypted1.n_;
                    out.L_ = encrypted1.L_;
                    out.scale_ = new_scale;
                    out.lazy_ = std::make_shared<std::vector<Ciphertext::LazyTerm>>();
                    out.lazy_->push_back(std::move(term));
                    out.deferred_.v.store(true, std::memory_order_release);
                    return;
                }
            }
            out.resize_batch(context_, encrypted1.parms_id(), size1 + size2 - 1, B);
            if (size1 != 2 || size2 != 2)
            {
                // the dest_size != 3 branch (:862-900); ckks_square of a larger ciphertext is this product with itself (:1237-1241)
                hip(moai_ct_multiply_general(dev(), encrypted1.device_data(), size1, encrypted2.device_data(), size2,
                                             out.device_data(), L, B, st()));
            }
            else if (&encrypted1 == &encrypted2 || encrypted1.device_data() == encrypted2.device_data())
            {
                hip(moai_ct_square(dev(), encrypted1.device_data(), out.device_data(), L, B, st()));
            }
            else
            {
                hip(moai_ct_multiply(dev(), encrypted1.device_data(), encrypted2.device_data(), out.device_data(), L, B,
                                     st()));
            }
            out.is_ntt_form() = true;
            out.scale() = new_scale;
        }
        void multiply_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2,
                              MemoryPoolHandle = MemoryPoolHandle()) const
        {
            Ciphertext out;
            multiply_into(encrypted1, encrypted2, out);
            encrypted1 = std::move(out);
        }
        void multiply(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination,
                      MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // the product is built in a fresh ciphertext either way: no deep copy of an operand first
            Ciphertext out;
            multiply_into(encrypted1, encrypted2, out);
            destination = std::move(out);
        }
        void square_inplace(Ciphertext &encrypted, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            multiply_inplace(encrypted, encrypted);
        }
        void square(const Ciphertext &encrypted, Ciphertext &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            Ciphertext out;
            multiply_into(encrypted, encrypted, out);
            destination = std::move(out);
        }
        void relinearize_inplace(Ciphertext &encrypted, const RelinKeys &relin_keys,
                                 MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // SEAL/evaluator.cpp:1345-1400
            auto cd = context_.get_context_data(encrypted.parms_id());
            if (!cd)
            {
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            }
            if (relin_keys.parms_id() != context_.key_parms_id())
            {
                throw std::invalid_argument("relin_keys is not valid for encryption parameters");
            }
            if (encrypted.size() == 2)
            {
                return;
            }
            if (relin_keys.size() < encrypted.size() - 2)
            {
                throw std::invalid_argument("not enough relinearization keys");
            }
            if (!encrypted.is_ntt_form())
            {
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            }
            const std::size_t L = encrypted.coeff_modulus_size();
            if (encrypted.size() > 3)
            {
                // :1385-1393: the last polynomial is switched with relin_keys[get_index(size - 1)] into (c0, c1) and dropped,
                // until two are left.  (KeyGenerator::create_relin_keys makes the key of s^2 only, so this needs keys from
                // elsewhere -- as in the reference.)
                const std::size_t B = encrypted.batch(), rn = L * encrypted.poly_modulus_degree();
                std::size_t size = encrypted.size();
                const std::size_t stride = size * rn;
                while (size > 2)
                {
                    const std::uint64_t *key = relin_keys.device_key(RelinKeys::get_index(size - 1), L);
                    for (std::size_t b = 0; b < B; b++)
                    {
                        std::uint64_t *ct = encrypted.device_data() + b * stride;
                        hip(moai_switch_key(dev(), ct, ct + (size - 1) * rn, key, L, 1, st()));
                    }
                    size--;
                }
                Ciphertext two;
                two.resize_batch(context_, encrypted.parms_id(), 2, B);
                for (std::size_t b = 0; b < B; b++)
                {
                    hip(moai_memcpy_d2d(two.device_data() + b * 2 * rn, encrypted.device_data() + b * stride, 2 * rn * 8, st()));
                }
                two.is_ntt_form() = true;
                two.scale() = encrypted.scale();
                encrypted = std::move(two);
                return;
            }
            Ciphertext out;
            out.resize_batch(context_, encrypted.parms_id(), 2, encrypted.batch());
            util::OpCombiner &comb = util::OpCombiner::instance();
            if (encrypted.batch() == 1 && comb.enabled())
            {
                const std::uint64_t *key = relin_keys.device_key(0, L);
                const std::size_t rn = L * encrypted.poly_modulus_degree();
                comb.submit(util::OpCombiner::Key(1, L, 0, key, dev()), { encrypted.device_data(), out.device_data() },
                            [&](const std::vector<util::OpCombiner::Request> &reqs) {
                                if (reqs.size() == 1)
                                {
                                    hip(moai_relinearize(dev(), reqs[0].in, key, reqs[0].out, L, 1, st()));
                                    return;
                                }
                                util::DeviceArray tin(reqs.size() * 3 * rn, st()), tout(reqs.size() * 2 * rn, st());
                                gather_requests(dev(), reqs, tin.get(), 3 * rn, st());
                                hip(moai_relinearize(dev(), tin.get(), key, tout.get(), L, reqs.size(), st()));
                                scatter_requests(dev(), reqs, tout.get(), 2 * rn, st());
                            });
            }
            else
            {
                hip(moai_relinearize(dev(), encrypted.device_data(), relin_keys.device_key(0, L), out.device_data(), L,
                                     encrypted.batch(), st()));
            }
            out.is_ntt_form() = true;
            out.scale() = encrypted.scale();
            encrypted = std::move(out);
        }
        void relinearize(const Ciphertext &encrypted, const RelinKeys &relin_keys, Ciphertext &destination,
                         MemoryPoolHandle = MemoryPoolHandle()) const
        {
            destination = encrypted;
            relinearize_inplace(destination, relin_keys);
        }

        // ---- modulus switching / rescaling ---------------------------------------------------------------
        void mod_switch_to_next(const Ciphertext &encrypted, Ciphertext &destination,
                                MemoryPoolHandle = MemoryPoolHandle()) const
        {
            drop_levels(encrypted, destination, 1);
        }
        void mod_switch_to_next_inplace(Ciphertext &encrypted, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            Ciphertext out;
            drop_levels(encrypted, out, 1);
            encrypted = std::move(out);
        }
        void mod_switch_to_next_inplace(Plaintext &plain) const
        {
            auto cd = context_.get_context_data(plain.parms_id());
            if (!cd || !plain.is_ntt_form())
            {
                throw std::invalid_argument("plain is not valid for encryption parameters");
            }
            auto next = cd->next_context_data();
            if (!next)
            {
                throw std::invalid_argument("end of modulus switching chain reached");
            }
            drop_plain(plain, *next);
        }
        void mod_switch_to_next(const Plaintext &plain, Plaintext &destination) const
        {
            destination = plain;
            mod_switch_to_next_inplace(destination);
        }
        void mod_switch_to_inplace(Ciphertext &encrypted, parms_id_type parms_id,
                                   MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // SEAL/evaluator.cpp:1630-1658; the level-by-level loop becomes one k-level drop
            auto cd = context_.get_context_data(encrypted.parms_id());
            auto target = context_.get_context_data(parms_id);
            if (!cd)
            {
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            }
            if (!target)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            if (cd->chain_index() < target->chain_index())
            {
                throw std::invalid_argument("cannot switch to higher level modulus");
            }
            std::size_t drop = cd->chain_index() - target->chain_index();
            if (drop == 0)
            {
                return;
            }
            Ciphertext out;
            drop_levels(encrypted, out, drop);
            encrypted = std::move(out);
        }
        void mod_switch_to(const Ciphertext &encrypted, parms_id_type parms_id, Ciphertext &destination,
                           MemoryPoolHandle = MemoryPoolHandle()) const
        {
            destination = encrypted;
            mod_switch_to_inplace(destination, parms_id);
        }
        void mod_switch_to_inplace(Plaintext &plain, parms_id_type parms_id) const
        {
            auto cd = context_.get_context_data(plain.parms_id());
            auto target = context_.get_context_data(parms_id);
            if (!cd || !plain.is_ntt_form())
            {
                throw std::invalid_argument("plain is not valid for encryption parameters");
            }
            if (!target)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            if (cd->chain_index() < target->chain_index())
            {
                throw std::invalid_argument("cannot switch to higher level modulus");
            }
            if (cd->chain_index() != target->chain_index())
            {
                drop_plain(plain, *target);
            }
        }
        void mod_switch_to(const Plaintext &plain, parms_id_type parms_id, Plaintext &destination) const
        {
            destination = plain;
            mod_switch_to_inplace(destination, parms_id);
        }
        void rescale_to_next(const Ciphertext &encrypted, Ciphertext &destination,
                             MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // SEAL/evaluator.cpp:1682-1720 / :1402-1481
            check_ct(encrypted, "encrypted");
            auto cd = context_.get_context_data(encrypted.parms_id());
            if (context_.last_parms_id() == encrypted.parms_id())
            {
                throw std::invalid_argument("end of modulus switching chain reached");
            }
            if (!encrypted.is_ntt_form())
            {
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            }
            auto next = cd->next_context_data();
            const std::size_t L = encrypted.coeff_modulus_size();
            Ciphertext out;
            out.resize_batch(context_, next->parms_id(), encrypted.size(), encrypted.batch());
            // (not routed through the call combiner: measured in round 3, gathering concurrent callers' rescales into one batched
            // call made MOAI's gelu_v2 loop SLOWER -- 0.84 against 0.55 s per 128 ciphertexts -- the callers are not in step there and
            // the gathering window costs more than five small launches do)
            hip(moai_rescale(dev(), encrypted.device_data(), out.device_data(), encrypted.size(), L, encrypted.batch(), st()));
            out.is_ntt_form() = true;
            out.scale() = encrypted.scale() / static_cast<double>(cd->parms().coeff_modulus().back().value());
            destination = std::move(out);
        }
        // true when `addend` can ride on the last kernel of a rescale of `encrypted` (moai_rescale_add / moai_mul_scalar_rescale_add):
        // it sits one level below, with the same shape -- the case in which the add_inplace that would follow adds
        // polynomial to polynomial (SEAL/evaluator.cpp:198-214)
        bool rides_on_rescale_of(const Ciphertext &addend, const Ciphertext &encrypted) const
        {
            auto cd = context_.get_context_data(encrypted.parms_id());
            return cd && cd->next_context_data() && addend.parms_id() == cd->next_context_data()->parms_id() && addend.is_ntt_form() &&
                   encrypted.is_ntt_form() && addend.size() == encrypted.size() && addend.batch() == encrypted.batch() &&
                   addend.device_data() != encrypted.device_data();
        }
        // acc = rescale_to_next(encrypted) + acc in one pass, acc one level below `encrypted` (rides_on_rescale_of): the
        // pair rescale_to_next_inplace + add_inplace; acc takes the rescaled scale like add_*_reduced_error gives it
        void rescale_to_next_add_inplace(const Ciphertext &encrypted, Ciphertext &acc) const
        {
            check_ct(encrypted, "encrypted");
            if (!rides_on_rescale_of(acc, encrypted))
            {
                throw std::invalid_argument("acc does not sit one level below encrypted with its shape");
            }
            auto cd = context_.get_context_data(encrypted.parms_id());
            hip(moai_rescale_add(dev(), encrypted.device_data(), acc.device_data(), acc.device_data(), encrypted.size(),
                                 encrypted.coeff_modulus_size(), encrypted.batch(), st()));
            acc.scale() = encrypted.scale() / static_cast<double>(cd->parms().coeff_modulus().back().value());
        }
        void rescale_to_next_inplace(Ciphertext &encrypted, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            Ciphertext out;
            rescale_to_next(encrypted, out);
            encrypted = std::move(out);
        }
        void rescale_to_inplace(Ciphertext &encrypted, parms_id_type parms_id,
                                MemoryPoolHandle = MemoryPoolHandle()) const
        {
            auto cd = context_.get_context_data(encrypted.parms_id());
            auto target = context_.get_context_data(parms_id);
            if (!cd)
            {
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            }
            if (!target)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            if (cd->chain_index() < target->chain_index())
            {
                throw std::invalid_argument("cannot switch to higher level modulus");
            }
            while (encrypted.parms_id() != parms_id)
            {
                rescale_to_next_inplace(encrypted);
            }
        }
        void rescale_to(const Ciphertext &encrypted, parms_id_type parms_id, Ciphertext &destination,
                        MemoryPoolHandle = MemoryPoolHandle()) const
        {
            destination = encrypted;
            rescale_to_inplace(destination, parms_id);
        }

        // ---- plaintext operations ----------------------------------------------------------------------------
        void add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const
        {
            plain_addsub(encrypted, plain, false);
        }
        void add_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const
        {
            destination = encrypted;
            add_plain_inplace(destination, plain);
        }
        void sub_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const
        {
            plain_addsub(encrypted, plain, true);
        }
        void sub_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const
        {
            destination = encrypted;
            sub_plain_inplace(destination, plain);
        }
        void multiply_plain_inplace(Ciphertext &encrypted, const Plaintext &plain,
                                    MemoryPoolHandle = MemoryPoolHandle()) const
        {
            multiply_plain(encrypted, plain, encrypted);
        }
        void multiply_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination,
                            MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // SEAL/evaluator.cpp:2154-2198 -> multiply_plain_ntt :2336-2373; the product is written straight
            // into `destination` (no deep copy of the input first)
            check_ct(encrypted, "encrypted");
            if (!plain.is_ntt_form() || !encrypted.is_ntt_form())
            {
                throw std::invalid_argument("NTT form mismatch");
            }
            if (encrypted.parms_id() != plain.parms_id())
            {
                throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            }
            auto cd = context_.get_context_data(encrypted.parms_id());
            double new_scale = encrypted.scale() * plain.scale();
            if (!scale_ok(new_scale, *cd))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            const std::size_t L = encrypted.coeff_modulus_size();
            if (&destination != &encrypted && plain.is_scalar() && encrypted.batch() == 1 && encrypted.size() == 2 && lazy_products())
            {
                // not computed now: the destination records (block of the operand, the plaintext's constant rows); see
                // Ciphertext::LazyTerm.  Everything that can throw has been checked above, as in the eager path.
                std::shared_ptr<util::DeviceArray> src = (encrypted.materialize(), encrypted.buf_);
                destination.release();
                destination.stream_ = st();
                destination.dev_ = dev();
                destination.parms_id_ = encrypted.parms_id_;
                destination.is_ntt_form_ = encrypted.is_ntt_form_;
                destination.size_ = encrypted.size_;
                destination.batch_ = 1;
                destination.n_ = encrypted.n_;
                destination.L_ = encrypted.L_;
                destination.scale_ = new_scale;
                destination.lazy_ = std::make_shared<std::vector<Ciphertext::LazyTerm>>();
                destination.lazy_->push_back(Ciphertext::LazyTerm{ src, plain.scalar_rows() });
                destination.deferred_.v.store(true, std::memory_order_release);
                return;
            }
            if (&destination != &encrypted && !plain.is_scalar() && plain.is_masked_constant() && encrypted.batch() == 1 && encrypted.size() == 2 &&
                lazy_products())
            {
                // a masked-constant vector plaintext whose transform has not been made: the product is recorded with the
                // plaintext's (mask, constant, scale); the transforms of a whole chain are made together (Ciphertext::materialize)
                Ciphertext::LazyTerm term;
                {
                    std::lock_guard<std::mutex> g(util::lazy_mutex());
                    term.mask = plain.mask_;
                    term.c = plain.mask_c_;
                    term.pscale = plain.mask_scale_;
                }
                if (term.mask)
                {
                    term.src = (encrypted.materialize(), encrypted.buf_);
                    destination.release();
                    destination.stream_ = st();
                    destination.dev_ = dev();
                    destination.parms_id_ = encrypted.parms_id_;
                    destination.is_ntt_form_ = encrypted.is_ntt_form_;
                    destination.size_ = encrypted.size_;
                    destination.batch_ = 1;
                    destination.n_ = encrypted.n_;
                    destination.L_ = encrypted.L_;
                    destination.scale_ = new_scale;
                    destination.lazy_ = std::make_shared<std::vector<Ciphertext::LazyTerm>>();
                    destination.lazy_->push_back(std::move(term));
                    destination.deferred_.v.store(true, std::memory_order_release);
                    return;
                }
            }
            if (&destination != &encrypted)
            {
                like(destination, encrypted);
            }
            if (plain.is_scalar())
            {
                hip(moai_mul_scalar_rows(dev(), encrypted.device_data(), plain.scalar_rows().data(),
                                         destination.device_data(), encrypted.size() * encrypted.batch(), L, st()));
            }
            else
            {
                hip(moai_dyadic_mul(dev(), encrypted.device_data(), plain.device_data(), destination.device_data(),
                                    encrypted.size() * encrypted.batch(), 1, L, st()));
            }
            destination.scale() = new_scale;
        }

        // ---- NTT form ---------------------------------------------------------------------------------------------
        void transform_to_ntt_inplace(Ciphertext &encrypted) const
        {
            check_ct(encrypted, "encrypted");
            if (encrypted.is_ntt_form())
            {
                throw std::invalid_argument("encrypted is already in NTT form");
            }
            hip(moai_ntt_forward(dev(), encrypted.device_data(), encrypted.size() * encrypted.batch(),
                                 encrypted.coeff_modulus_size(), nullptr, st()));
            encrypted.is_ntt_form() = true;
        }
        void transform_to_ntt(const Ciphertext &encrypted, Ciphertext &destination) const
        {
            destination = encrypted;
            transform_to_ntt_inplace(destination);
        }
        void transform_from_ntt_inplace(Ciphertext &encrypted) const
        {
            check_ct(encrypted, "encrypted");
            if (!encrypted.is_ntt_form())
            {
                throw std::invalid_argument("encrypted_ntt is not in NTT form");
            }
            hip(moai_ntt_inverse(dev(), encrypted.device_data(), encrypted.size() * encrypted.batch(),
                                 encrypted.coeff_modulus_size(), nullptr, st()));
            encrypted.is_ntt_form() = false;
        }
        void transform_from_ntt(const Ciphertext &encrypted, Ciphertext &destination) const
        {
            destination = encrypted;
            transform_from_ntt_inplace(destination);
        }

        // ---- Galois / rotations ---------------------------------------------------------------------------------
        void apply_galois_inplace(Ciphertext &encrypted, std::uint32_t galois_elt, const GaloisKeys &galois_keys,
                                  MemoryPoolHandle = MemoryPoolHandle()) const
        {
            galois_into(encrypted, encrypted, galois_elt, galois_keys);
        }
        void apply_galois(const Ciphertext &encrypted, std::uint32_t galois_elt, const GaloisKeys &galois_keys,
                          Ciphertext &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // the reference copies and works in place; the device call takes a separate destination
            galois_into(encrypted, destination, galois_elt, galois_keys);
        }
        void rotate_vector_inplace(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys,
                                   MemoryPoolHandle = MemoryPoolHandle()) const
        {
            rotate_internal(encrypted, steps, galois_keys);
        }
        void rotate_vector(const Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys, Ciphertext &destination,
                           MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // one key switch with the key present (the common case): straight into the destination
            std::uint32_t elt = steps != 0 && &destination != &encrypted && context_.get_context_data(encrypted.parms_id()) &&
                                        galois_keys.parms_id() == context_.key_parms_id()
                                    ? moai_galois_elt_from_step(dev(), steps)
                                    : 0;
            if (elt && galois_keys.has_key(elt))
            {
                galois_into(encrypted, destination, elt, galois_keys);
                return;
            }
            destination = encrypted;
            rotate_vector_inplace(destination, steps, galois_keys);
        }
        void complex_conjugate_inplace(Ciphertext &encrypted, const GaloisKeys &galois_keys,
                                       MemoryPoolHandle = MemoryPoolHandle()) const
        {
            // conjugate_internal, SEAL/evaluator.h:1428-1450: Galois element 2N - 1
            apply_galois_inplace(encrypted, static_cast<std::uint32_t>(2 * context_.n() - 1), galois_keys);
        }
        void complex_conjugate(const Ciphertext &encrypted, const GaloisKeys &galois_keys, Ciphertext &destination,
                               MemoryPoolHandle = MemoryPoolHandle()) const
        {
            galois_into(encrypted, destination, static_cast<std::uint32_t>(2 * context_.n() - 1), galois_keys);
        }

        // ---- fork additions (SEAL/evaluator.cpp:395-594) -------------------------------------------------------
        void add_const_inplace(Ciphertext &encrypted, double value) const
        {
            Plaintext const_plain;
            encoder_.encode(value, encrypted.scale(), const_plain);
            mod_switch_to_inplace(const_plain, encrypted.parms_id());
            add_plain_inplace(encrypted, const_plain);
        }
        void add_const(const Ciphertext &encrypted, double value, Ciphertext &destination) const
        {
            destination = encrypted;
            add_const_inplace(destination, value);
        }
        void multiply_const_inplace(Ciphertext &encrypted, double value) const
        {
            Plaintext const_plain;
            encoder_.encode(value, encrypted.scale(), const_plain);
            mod_switch_to_inplace(const_plain, encrypted.parms_id());
            multiply_plain_inplace(encrypted, const_plain);
        }
        void multiply_const(const Ciphertext &encrypted, double value, Ciphertext &destination) const
        {
            Plaintext const_plain;
            encoder_.encode(value, encrypted.scale(), const_plain);
            mod_switch_to_inplace(const_plain, encrypted.parms_id());
            multiply_plain(encrypted, const_plain, destination);
        }
        // multiply_const(encrypted, value, destination) followed by rescale_to_next_inplace(destination) as ONE pass
        // over the ciphertext (moai_mul_scalar_rescale); not part of the reference's interface, same result as the
        // two calls.  forced_scale > 0 replaces the product's scale before the rescale divides it (the
        // *_reduced_error compositions overwrite the scale between the two calls, SEAL/evaluator.cpp:447-452).
        // `addend` (optional; may be `destination` itself): a ciphertext one level below `encrypted` that is added to the result
        // by the same pass (rides_on_rescale_of) -- multiply_const, rescale_to_next_inplace, add_inplace in one call
        void multiply_const_rescale(const Ciphertext &encrypted, double value, Ciphertext &destination, double forced_scale = 0,
                                    const Ciphertext *addend = nullptr) const
        {
            Plaintext const_plain;
            encoder_.encode(value, encrypted.scale(), const_plain);
            mod_switch_to_inplace(const_plain, encrypted.parms_id());
            // multiply_plain's checks (SEAL/evaluator.cpp:2154-2198, 2336-2373)
            check_ct(encrypted, "encrypted");
            if (!const_plain.is_ntt_form() || !encrypted.is_ntt_form())
            {
                throw std::invalid_argument("NTT form mismatch");
            }
            auto cd = context_.get_context_data(encrypted.parms_id());
            const double new_scale = encrypted.scale() * const_plain.scale();
            if (!scale_ok(new_scale, *cd))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            // rescale_to_next's (SEAL/evaluator.cpp:1682-1720)
            if (context_.last_parms_id() == encrypted.parms_id())
            {
                throw std::invalid_argument("end of modulus switching chain reached");
            }
            if (addend && !rides_on_rescale_of(*addend, encrypted))
            {
                throw std::invalid_argument("addend does not sit one level below encrypted with its shape");
            }
            if (!const_plain.is_scalar())
            {
                Ciphertext product;
                multiply_plain(encrypted, const_plain, product);
                if (forced_scale > 0)
                {
                    product.scale() = forced_scale;
                }
                rescale_to_next_inplace(product);
                if (addend)
                {
                    Ciphertext sum = *addend;
                    sum.scale() = product.scale();
                    add_inplace(sum, product);
                    product = std::move(sum);
                }
                destination = std::move(product);
                return;
            }
            auto next = cd->next_context_data();
            const std::size_t L = encrypted.coeff_modulus_size();
            const double out_scale = (forced_scale > 0 ? forced_scale : new_scale) / static_cast<double>(cd->parms().coeff_modulus().back().value());
            if (addend && addend == &destination)
            {
                hip(moai_mul_scalar_rescale_add(dev(), encrypted.device_data(), const_plain.scalar_rows().data(), destination.device_data(),
                                                destination.device_data(), encrypted.size(), L, encrypted.batch(), st()));
                destination.scale() = out_scale;
                return;
            }
            Ciphertext out;
            out.resize_batch(context_, next->parms_id(), encrypted.size(), encrypted.batch());
            if (addend)
            {
                hip(moai_mul_scalar_rescale_add(dev(), encrypted.device_data(), const_plain.scalar_rows().data(), addend->device_data(),
                                                out.device_data(), encrypted.size(), L, encrypted.batch(), st()));
            }
            else
            {
                hip(moai_mul_scalar_rescale(dev(), encrypted.device_data(), const_plain.scalar_rows().data(), out.device_data(), encrypted.size(), L,
                                            encrypted.batch(), st()));
            }
            out.is_ntt_form() = true;
            out.scale() = out_scale;
            destination = std::move(out);
        }
        template <typename T>
        void multiply_vector_inplace(Ciphertext &encrypted, const std::vector<T> &value) const
        {
            Plaintext vector_plain;
            encoder_.encode(value, encrypted.scale(), vector_plain);
            mod_switch_to_inplace(vector_plain, encrypted.parms_id());
            multiply_plain_inplace(encrypted, vector_plain);
        }
        template <typename T>
        void multiply_vector(const Ciphertext &encrypted, const std::vector<T> &value, Ciphertext &destination) const
        {
            Plaintext vector_plain;
            encoder_.encode(value, encrypted.scale(), vector_plain);
            mod_switch_to_inplace(vector_plain, encrypted.parms_id());
            multiply_plain(encrypted, vector_plain, destination);
        }
        // SEAL/evaluator.h:1371-1378
        template <typename T>
        void multiply_vector_inplace_reduced_error(Ciphertext &encrypted, const std::vector<T> &value) const
        {
            Plaintext plain;
            encoder_.encode(value, encrypted.scale(), plain);
            mod_switch_to_inplace(plain, encrypted.parms_id());
            multiply_plain_inplace(encrypted, plain);
        }
        template <typename T>
        void multiply_vector_reduced_error(const Ciphertext &encrypted, const std::vector<T> &value,
                                           Ciphertext &destination) const
        {
            Plaintext plain;
            encoder_.encode(value, encrypted.scale(), plain);
            mod_switch_to_inplace(plain, encrypted.parms_id());
            multiply_plain(encrypted, plain, destination);
        }
        void double_inplace(Ciphertext &encrypted) const
        {
            add_inplace(encrypted, encrypted);
        }
        void add_inplace_reduced_error(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            reduced_error(encrypted1, encrypted2, nullptr, 0);
        }
        void add_reduced_error(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const
        {
            destination = encrypted1;
            add_inplace_reduced_error(destination, encrypted2);
        }
        void sub_inplace_reduced_error(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            reduced_error(encrypted1, encrypted2, nullptr, 1);
        }
        void sub_reduced_error(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const
        {
            destination = encrypted1;
            sub_inplace_reduced_error(destination, encrypted2);
        }
        void multiply_inplace_reduced_error(Ciphertext &encrypted1, const Ciphertext &encrypted2,
                                            const RelinKeys &relin_keys) const
        {
            reduced_error(encrypted1, encrypted2, &relin_keys, 2);
        }
        void multiply_reduced_error(const Ciphertext &encrypted1, const Ciphertext &encrypted2, const RelinKeys &relin_keys,
                                    Ciphertext &destination) const
        {
            destination = encrypted1;
            multiply_inplace_reduced_error(destination, encrypted2, relin_keys);
        }

    private:
        moai_ctx *dev() const
        {
            return context_.device();
        }
        void *st() const
        {
            return context_.stream();
        }
        static void hip(int rc)
        {
            util::hip_check(rc);
        }
        static bool scale_ok(double scale, const SEALContext::ContextData &cd)
        {
            // is_scale_within_bounds, SEAL/evaluator.cpp:29-48
            int bound = cd.total_coeff_modulus_bit_count();
            return !(scale <= 0 || (static_cast<int>(std::log2(scale)) >= bound));
        }
        // give `dst` the shape and metadata of `src` without copying residues
        void like(Ciphertext &dst, const Ciphertext &src) const
        {
            dst.resize_batch(context_, src.parms_id(), src.size(), src.batch());
            dst.is_ntt_form() = src.is_ntt_form();
            dst.scale() = src.scale();
        }
        void check_ct(const Ciphertext &c, const char *name) const
        {
            if (!context_.get_context_data(c.parms_id()) || c.size() < 2 || !c.has_value())
            {
                throw std::invalid_argument(std::string(name) + " is not valid for encryption parameters");
            }
        }

        // the argument checks of add / sub, SEAL/evaluator.cpp:157-180
        void check_pair(const Ciphertext &e1, const Ciphertext &e2) const
        {
            check_ct(e1, "encrypted1");
            check_ct(e2, "encrypted2");
            if (e1.parms_id() != e2.parms_id())
            {
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            }
            if (e1.batch() != e2.batch())
            {
                throw std::invalid_argument("encrypted1 and encrypted2 pack different numbers of ciphertexts");
            }
            if (e1.is_ntt_form() != e2.is_ntt_form())
            {
                throw std::invalid_argument("NTT form mismatch");
            }
            if (!util::are_close(e1.scale(), e2.scale()))
            {
                throw std::invalid_argument("scale mismatch");
            }
        }

        // the members' blocks <-> one packed array, one launch each way (moai_gather_blocks / moai_scatter_blocks)
        static void gather_requests(moai_ctx *d, const std::vector<util::OpCombiner::Request> &reqs, std::uint64_t *packed, std::size_t words,
                                    void *stream)
        {
            const std::uint64_t *ptrs[64];
            for (std::size_t i = 0; i < reqs.size(); i++)
            {
                ptrs[i] = reqs[i].in;
            }
            util::hip_check(moai_gather_blocks(d, ptrs, packed, reqs.size(), words, stream));
        }
        static void scatter_requests(moai_ctx *d, const std::vector<util::OpCombiner::Request> &reqs, const std::uint64_t *packed, std::size_t words,
                                     void *stream)
        {
            std::uint64_t *ptrs[64];
            for (std::size_t i = 0; i < reqs.size(); i++)
            {
                ptrs[i] = reqs[i].out;
            }
            util::hip_check(moai_scatter_blocks(d, packed, ptrs, reqs.size(), words, stream));
        }
        // SEAL/evaluator.cpp:155-240 / :263-350
        // a deferred rotation that is made alone goes through the call combiner, like an eager one
        static void install_single_rotation_hook()
        {
            static const bool once = [] {
                util::single_rotation_hook() = [](moai_ctx *d, const std::uint64_t *in, std::uint64_t *out, std::size_t L, std::uint32_t elt,
                                                  const std::uint64_t *key, void *stream) {
                    util::OpCombiner &comb = util::OpCombiner::instance();
                    if (!comb.enabled())
                    {
                        util::hip_check(moai_apply_galois_to(d, in, out, L, elt, key, 1, stream));
                        return;
                    }
                    const std::size_t words = 2 * L * moai_ctx_coeff_count(d);
                    comb.submit(util::OpCombiner::Key(0, L, elt, key, d), { in, out }, [&](const std::vector<util::OpCombiner::Request> &reqs) {
                        if (reqs.size() == 1)
                        {
                            util::hip_check(moai_apply_galois_to(d, reqs[0].in, reqs[0].out, L, elt, key, 1, stream));
                            return;
                        }
                        util::DeviceArray tmp(reqs.size() * words, stream);
                        gather_requests(d, reqs, tmp.get(), words, stream);
                        util::hip_check(moai_apply_galois(d, tmp.get(), L, elt, key, reqs.size(), stream));
                        scatter_requests(d, reqs, tmp.get(), words, stream);
                    });
                };
                return true;
            }();
            (void)once;
        }
        static bool lazy_products()
        {
            static const bool on = [] {
                const char *e = std::getenv("MOAI_SHIM_LAZY");
                return !(e && e[0] == '0');
            }();
            return on;
        }
        void addsub(Ciphertext &e1, const Ciphertext &e2, bool sub) const
        {
            check_pair(e1, e2);
            if (!sub && e2.is_deferred() && e1.batch() == 1 && e1.size() == e2.size() && (e1.size() == 2 || e1.size() == 3))
            {
                // the addend is a deferred sum of scalar products: its terms join this ciphertext's (Ciphertext::LazyTerm)
                std::shared_ptr<std::vector<Ciphertext::LazyTerm>> theirs;
                std::shared_ptr<util::DeviceArray> their_base;
                {
                    std::lock_guard<std::mutex> g(Ciphertext::lazy_mutex());
                    theirs = e2.lazy_;
                    their_base = e2.buf_;
                }
                if (theirs && !their_base)
                {
                    if (!e1.lazy_ || e1.lazy_.use_count() > 1)
                    {
                        auto mine = std::make_shared<std::vector<Ciphertext::LazyTerm>>();
                        if (e1.lazy_)
                        {
                            *mine = *e1.lazy_;
                        }
                        e1.lazy_ = mine;
                    }
                    e1.lazy_->insert(e1.lazy_->end(), theirs->begin(), theirs->end());
                    e1.deferred_.v.store(true, std::memory_order_release);
                    return;
                }
            }
            const std::size_t L = e1.coeff_modulus_size(), n = e1.poly_modulus_degree();
            const std::size_t min_size = std::min(e1.size(), e2.size());
            const std::size_t max_size = std::max(e1.size(), e2.size());
            if (e1.batch() > 1)
            {
                if (min_size != max_size)
                {
                    throw std::logic_error("packed ciphertexts of different sizes cannot be added");
                }
                if (sub)
                {
                    hip(moai_sub(dev(), e1.device_data(), e2.device_data(), e1.device_data(), min_size * e1.batch(), L, st()));
                }
                else
                {
                    hip(moai_add(dev(), e1.device_data(), e2.device_data(), e1.device_data(), min_size * e1.batch(), L, st()));
                }
                return;
            }
            if (e1.size() < max_size)
            {
                // grow encrypted1, keeping its polynomials
                Ciphertext grown;
                grown.resize(context_, e1.parms_id(), max_size);
                hip(moai_memcpy_d2d(grown.device_data(), e1.device_data(), e1.size() * L * n * 8, st()));
                grown.is_ntt_form() = e1.is_ntt_form();
                grown.scale() = e1.scale();
                e1 = std::move(grown);
            }
            if (sub)
            {
                hip(moai_sub(dev(), e1.device_data(), e2.device_data(), e1.device_data(), min_size, L, st()));
            }
            else
            {
                hip(moai_add(dev(), e1.device_data(), e2.device_data(), e1.device_data(), min_size, L, st()));
            }
            if (e2.size() > min_size)
            {
                // copy (add) or negate (sub) the remaining polynomials of encrypted2
                std::uint64_t *dst = e1.device_data() + min_size * L * n;
                const std::uint64_t *src = e2.device_data() + min_size * L * n;
                if (sub)
                {
                    hip(moai_negate(dev(), src, dst, e2.size() - min_size, L, st()));
                }
                else
                {
                    hip(moai_memcpy_d2d(dst, src, (e2.size() - min_size) * L * n * 8, st()));
                }
            }
        }

        // SEAL/evaluator.cpp:1938-2044 / :2046-2152: touches polynomial 0 only
        void plain_addsub(Ciphertext &encrypted, const Plaintext &plain, bool sub) const
        {
            check_ct(encrypted, "encrypted");
            if (!encrypted.is_ntt_form())
            {
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            }
            if (!plain.is_ntt_form())
            {
                throw std::invalid_argument("plain must be in NTT form");
            }
            if (encrypted.parms_id() != plain.parms_id())
            {
                throw std::invalid_argument("encrypted and plain parameter mismatch");
            }
            if (!util::are_close(encrypted.scale(), plain.scale()))
            {
                throw std::invalid_argument("scale mismatch");
            }
            const std::size_t L = encrypted.coeff_modulus_size();
            const std::size_t ct_words = encrypted.size() * L * encrypted.poly_modulus_degree();
            if (plain.is_scalar())
            {
                std::vector<std::uint64_t> s = plain.scalar_rows();
                if (sub)
                {
                    auto cd = context_.get_context_data(encrypted.parms_id());
                    const auto &cm = cd->parms().coeff_modulus();
                    for (std::size_t r = 0; r < L; r++)
                    {
                        s[r] = s[r] ? cm[r].value() - s[r] : 0;
                    }
                }
                for (std::size_t b = 0; b < encrypted.batch(); b++)
                {
                    std::uint64_t *c0 = encrypted.device_data() + b * ct_words;
                    hip(moai_add_scalar_rows(dev(), c0, s.data(), c0, 1, L, st()));
                }
            }
            else
            {
                // polynomial 0 of every packed ciphertext
                for (std::size_t b = 0; b < encrypted.batch(); b++)
                {
                    std::uint64_t *c0 = encrypted.device_data() + b * ct_words;
                    hip(sub ? moai_sub(dev(), c0, plain.device_data(), c0, 1, L, st())
                            : moai_add(dev(), c0, plain.device_data(), c0, 1, L, st()));
                }
            }
        }

        // SEAL/evaluator.cpp:1483-1546 applied `drop` times in one strided copy
        void drop_levels(const Ciphertext &encrypted, Ciphertext &destination, std::size_t drop) const
        {
            check_ct(encrypted, "encrypted");
            auto cd = context_.get_context_data(encrypted.parms_id());
            if (cd->chain_index() < drop)
            {
                throw std::invalid_argument("end of modulus switching chain reached");
            }
            if (!encrypted.is_ntt_form())
            {
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            }
            const std::size_t L = encrypted.coeff_modulus_size();
            auto target = context_.data_level(L - drop);
            Ciphertext out;
            out.resize_batch(context_, target->parms_id(), encrypted.size(), encrypted.batch());
            hip(moai_mod_drop(dev(), encrypted.device_data(), out.device_data(), encrypted.size(), L, drop, encrypted.batch(),
                              st()));
            out.is_ntt_form() = true;
            out.scale() = encrypted.scale();
            destination = std::move(out);
        }

        // SEAL/evaluator.cpp:1548-1581: a CKKS plaintext just loses its trailing rows
        void drop_plain(Plaintext &plain, const SEALContext::ContextData &target) const
        {
            const std::size_t L = target.parms().coeff_modulus().size();
            if (plain.is_scalar())
            {
                plain.scalar_rows_.resize(L);
            }
            plain.L_ = L;
            plain.parms_id_ = target.parms_id();
        }

        // SEAL/evaluator.cpp:2667-2722
        // Evaluator::apply_galois_inplace, SEAL/evaluator.cpp:2563-2665, reading `src` and writing `dst` (may be the
        // same object): no deep copy of the operand first
        void galois_into(const Ciphertext &src, Ciphertext &dst, std::uint32_t galois_elt, const GaloisKeys &galois_keys) const
        {
            check_ct(src, "encrypted");
            if (galois_keys.parms_id() != context_.key_parms_id())
            {
                throw std::invalid_argument("galois_keys is not valid for encryption parameters");
            }
            if (!(galois_elt & 1) || galois_elt >= 2 * context_.n())
            {
                throw std::invalid_argument("Galois element is not valid");
            }
            if (src.size() > 2)
            {
                throw std::invalid_argument("encrypted size must be 2");
            }
            if (!galois_keys.has_key(galois_elt))
            {
                throw std::invalid_argument("Galois key not present");
            }
            if (!src.is_ntt_form())
            {
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            }
            const std::size_t L = src.coeff_modulus_size();
            if (lazy_products() && src.batch() == 1 && src.size() == 2 && galois_keys.generation() != 0)
            {
                // not made now (util::RotState): recorded, and made when the result is read -- together with the other rotations of
                // the same step this thread has asked for meanwhile.  Everything that can throw has been checked above.
                const std::size_t index = GaloisKeys::get_index(galois_elt);
                std::shared_ptr<util::DeviceArray> kb = galois_keys.key_block(index, L);
                install_single_rotation_hook();
                std::shared_ptr<util::RotState> state;
                if (&dst == &src && dst.rot_ && !dst.lazy_ && !dst.buf_)
                {
                    // one more step of a chain (the non-adjacent form's next power of two, SEAL/evaluator.cpp:2709-2720)
                    if (dst.rot_.use_count() > 1)
                    {
                        auto own = std::make_shared<util::RotState>();
                        {
                            std::lock_guard<std::mutex> g(dst.rot_->mu);
                            own->src = dst.rot_->src;
                            own->elts = dst.rot_->elts;
                            own->keys = dst.rot_->keys;
                            own->key_ids = dst.rot_->key_ids;
                            own->L = dst.rot_->L;
                            own->n = dst.rot_->n;
                            own->dev = dst.rot_->dev;
                            own->stream = dst.rot_->stream;
                        }
                        dst.rot_ = own;
                        util::rot_register(own);
                    }
                    std::lock_guard<std::mutex> g(dst.rot_->mu);
                    dst.rot_->elts.push_back(galois_elt);
                    dst.rot_->keys.push_back(kb);
                    dst.rot_->key_ids.emplace_back(galois_keys.generation(), index);
                    return;
                }
                src.materialize();
                state = std::make_shared<util::RotState>();
                state->src = src.buf_;
                state->elts.push_back(galois_elt);
                state->keys.push_back(kb);
                state->key_ids.emplace_back(galois_keys.generation(), index);
                state->L = L;
                state->n = src.n_;
                state->dev = dev();
                state->stream = st();
                if (&dst != &src)
                {
                    dst.release();
                    dst.parms_id_ = src.parms_id_;
                    dst.is_ntt_form_ = src.is_ntt_form_;
                    dst.size_ = src.size_;
                    dst.batch_ = 1;
                    dst.n_ = src.n_;
                    dst.L_ = src.L_;
                    dst.scale_ = src.scale_;
                }
                dst.stream_ = st();
                dst.dev_ = dev();
                dst.buf_.reset();
                dst.lazy_.reset();
                dst.rot_ = state;
                dst.deferred_.v.store(true, std::memory_order_release);
                util::rot_register(state);
                return;
            }
            const std::uint64_t *key = galois_keys.device_key(GaloisKeys::get_index(galois_elt), L);
            // a rotation of this very block by this element with these keys may have been computed already (util::RotationCache)
            util::RotationCache &cache = util::RotationCache::instance();
            const bool cached = cache.enabled() && src.batch() == 1 && galois_keys.generation() != 0;
            util::RotationCache::Key ckey(src.block_id(), galois_elt, galois_keys.generation(), GaloisKeys::get_index(galois_elt), L);
            std::shared_ptr<util::DeviceArray> src_block = src.buf_;
            if (cached)
            {
                if (auto hit = cache.find(ckey))
                {
                    if (&dst != &src)
                    {
                        dst.parms_id_ = src.parms_id_;
                        dst.is_ntt_form_ = src.is_ntt_form_;
                        dst.size_ = src.size_;
                        dst.batch_ = 1;
                        dst.n_ = src.n_;
                        dst.L_ = src.L_;
                        dst.scale_ = src.scale_;
                        dst.stream_ = context_.stream();
                    }
                    dst.buf_ = hit;
                    return;
                }
                // a miss writes into a block of its own (never into the source's: the cache is about to keep that one)
                if (&dst == &src)
                {
                    dst.buf_ = std::make_shared<util::DeviceArray>(src.words(), st());
                }
                else
                {
                    dst.buf_.reset();
                    like(dst, src);
                }
            }
            else if (&dst != &src)
            {
                like(dst, src);
            }
            const std::uint64_t *in_ptr = src_block->get();
            std::uint64_t *out_ptr = cached ? dst.buf_->get() : dst.device_data();
            if (!cached && &dst == &src)
            {
                in_ptr = out_ptr; // in place on the (now private) block
            }
            struct Remember
            {
                util::RotationCache &cache;
                const util::RotationCache::Key &key;
                const std::shared_ptr<util::DeviceArray> &src, &out;
                bool on;
                ~Remember()
                {
                    if (on && !std::uncaught_exceptions())
                    {
                        cache.insert(key, src, out);
                    }
                }
            } remember{ cache, ckey, src_block, dst.buf_, cached };
            util::OpCombiner &comb = util::OpCombiner::instance();
            if (src.batch() == 1 && comb.enabled())
            {
                // concurrent callers with the same element, level and key share one batched key switch
                const std::size_t words = 2 * L * src.poly_modulus_degree();
                comb.submit(util::OpCombiner::Key(0, L, galois_elt, key, dev()), { in_ptr, out_ptr },
                            [&](const std::vector<util::OpCombiner::Request> &reqs) {
                                if (reqs.size() == 1)
                                {
                                    hip(moai_apply_galois_to(dev(), reqs[0].in, reqs[0].out, L, galois_elt, key, 1, st()));
                                    return;
                                }
                                util::DeviceArray tmp(reqs.size() * words, st());
                                gather_requests(dev(), reqs, tmp.get(), words, st());
                                hip(moai_apply_galois(dev(), tmp.get(), L, galois_elt, key, reqs.size(), st()));
                                scatter_requests(dev(), reqs, tmp.get(), words, st());
                            });
                return;
            }
            if (in_ptr == out_ptr)
            {
                hip(moai_apply_galois(dev(), out_ptr, L, galois_elt, key, src.batch(), st()));
            }
            else
            {
                hip(moai_apply_galois_to(dev(), in_ptr, out_ptr, L, galois_elt, key, src.batch(), st()));
            }
        }

        void rotate_internal(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys) const
        {
            auto cd = context_.get_context_data(encrypted.parms_id());
            if (!cd)
            {
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            }
            if (galois_keys.parms_id() != context_.key_parms_id())
            {
                throw std::invalid_argument("galois_keys is not valid for encryption parameters");
            }
            if (steps == 0)
            {
                return;
            }
            const std::size_t n = context_.n();
            std::uint32_t elt = moai_galois_elt_from_step(dev(), steps);
            if (!elt)
            {
                throw std::invalid_argument("step count too large");
            }
            if (galois_keys.has_key(elt))
            {
                apply_galois_inplace(encrypted, elt, galois_keys);
                return;
            }
            // decompose into power-of-two rotations (non-adjacent form)
            std::vector<int> naf_steps = util::naf(steps);
            if (naf_steps.size() == 1)
            {
                throw std::invalid_argument("Galois key not present");
            }
            for (int s : naf_steps)
            {
                if (static_cast<std::size_t>(std::abs(s)) != (n >> 1))
                {
                    rotate_internal(encrypted, s, galois_keys);
                }
            }
        }

        // the three *_reduced_error compositions (Kim et al., CT-RSA'22), SEAL/evaluator.cpp:419-592
        void reduced_error(Ciphertext &encrypted1, const Ciphertext &encrypted2, const RelinKeys *relin_keys, int op) const
        {
            const std::size_t c1 = encrypted1.coeff_modulus_size();
            const std::size_t c2 = encrypted2.coeff_modulus_size();
            auto combine = [&](Ciphertext &a, const Ciphertext &b) {
                if (op == 0)
                {
                    add_inplace(a, b);
                }
                else if (op == 1)
                {
                    sub_inplace(a, b);
                }
                else
                {
                    multiply_inplace(a, b);
                }
            };
            if (c1 == c2)
            {
                encrypted1.scale() = encrypted2.scale();
                combine(encrypted1, encrypted2);
                if (op == 2)
                {
                    relinearize_inplace(encrypted1, *relin_keys);
                }
                return;
            }
            if (c1 < c2)
            {
                auto cd = context_.get_context_data(encrypted2.parms_id());
                if (!cd)
                {
                    throw std::invalid_argument("encrypted2 is not valid for encryption parameters");
                }
                double q_last = static_cast<double>(cd->parms().coeff_modulus()[c2 - 1].value());
                Ciphertext adjusted;
                double scale_adjust = encrypted1.scale() * q_last / (encrypted2.scale() * encrypted2.scale());
                if (op == 0 && c2 == c1 + 1 && rides_on_rescale_of(encrypted1, encrypted2))
                {
                    // the adjusted operand lands on encrypted1's level: the addition rides on its rescale
                    multiply_const_rescale(encrypted2, scale_adjust, encrypted1, encrypted1.scale() * q_last, &encrypted1);
                    return;
                }
                multiply_const_rescale(encrypted2, scale_adjust, adjusted, encrypted1.scale() * q_last);
                mod_switch_to_inplace(adjusted, encrypted1.parms_id());
                encrypted1.scale() = adjusted.scale();
                combine(encrypted1, adjusted);
            }
            else
            {
                auto cd = context_.get_context_data(encrypted1.parms_id());
                if (!cd)
                {
                    throw std::invalid_argument("encrypted1 is not valid for encryption parameters");
                }
                double q_last = static_cast<double>(cd->parms().coeff_modulus()[c1 - 1].value());
                Ciphertext adjusted;
                double scale_adjust = encrypted2.scale() * q_last / (encrypted1.scale() * encrypted1.scale());
                if (op == 0 && c1 == c2 + 1 && rides_on_rescale_of(encrypted2, encrypted1))
                {
                    multiply_const_rescale(encrypted1, scale_adjust, adjusted, encrypted2.scale() * q_last, &encrypted2);
                    adjusted.scale() = encrypted2.scale();
                    encrypted1 = std::move(adjusted);
                    return;
                }
                multiply_const_rescale(encrypted1, scale_adjust, adjusted, encrypted2.scale() * q_last);
                mod_switch_to_inplace(adjusted, encrypted2.parms_id());
                adjusted.scale() = encrypted2.scale();
                combine(adjusted, encrypted2);
                encrypted1 = std::move(adjusted);
            }
            if (op == 2)
            {
                relinearize_inplace(encrypted1, *relin_keys);
            }
        }

        SEALContext context_;
        CKKSEncoder &encoder_;
    };
} // namespace seal
