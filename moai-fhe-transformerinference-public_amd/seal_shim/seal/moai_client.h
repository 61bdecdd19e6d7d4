// seal/moai_client.h -- client-side classes of the seal:: surface: CKKSEncoder, KeyGenerator,
// Encryptor, Decryptor.  Host code (randomness, FP64 FFT, CRT); out of the hot-path scope
//  but needed so that MOAI's programs link and run.  Their NTTs are
// executed on the device through moai_ntt_forward / moai_ntt_inverse.
#pragma once
#include <cerrno>
#include <exception>
#include <memory>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include <sys/random.h>
#include <sys/types.h>

namespace seal
{
    namespace util
    {
        // Randomness of the client side: ChaCha20 (RFC 8439 block function, 20 rounds) in counter mode, keyed per thread
        // with 256 bits + a 64-bit nonce drawn from the operating system (getrandom(2), /dev/urandom as a fallback).
        // The reference expands a 512-bit OS seed with Blake2xb / Shake256 (SEAL/randomgen.cpp:18-62, 142-169); the
        // construction differs, the property does not: every secret (ternary key, uniform a, noise, u / e0 / e1) comes
        // from a cryptographic stream whose state cannot be recovered from the public outputs.  Satisfies the C++
        // UniformRandomBitGenerator requirements, so the standard distributions below take it.
        class ChaCha20Rng
        {
        public:
            using result_type = std::uint64_t;
            static constexpr result_type min()
            {
                return 0;
            }
            static constexpr result_type max()
            {
                return ~static_cast<result_type>(0);
            }
            ChaCha20Rng()
            {
                unsigned char seed[40];
                os_random(seed, sizeof(seed));
                init(seed);
            }
            // a fixed stream (known-answer test of the block function): 32-byte key + 8-byte nonce
            explicit ChaCha20Rng(const unsigned char (&seed)[40])
            {
                init(seed);
            }
            result_type operator()()
            {
                if (pos_ == 8)
                {
                    refill();
                }
                return buf_[pos_++];
            }
            static void os_random(unsigned char *out, std::size_t len)
            {
                std::size_t got = 0;
                while (got < len)
                {
                    ssize_t r = ::getrandom(out + got, len - got, 0);
                    if (r > 0)
                    {
                        got += static_cast<std::size_t>(r);
                    }
                    else if (r < 0 && errno == EINTR)
                    {
                        continue;
                    }
                    else
                    {
                        break;
                    }
                }
                if (got < len)
                {
                    std::FILE *f = std::fopen("/dev/urandom", "rb");
                    if (f)
                    {
                        got += std::fread(out + got, 1, len - got, f);
                        std::fclose(f);
                    }
                }
                if (got < len)
                {
                    throw std::runtime_error("no operating-system randomness available");
                }
            }

        private:
            static std::uint32_t rotl(std::uint32_t v, int c)
            {
                return (v << c) | (v >> (32 - c));
            }
            static void quarter(std::uint32_t *x, int a, int b, int c, int d)
            {
                x[a] += x[b];
                x[d] = rotl(x[d] ^ x[a], 16);
                x[c] += x[d];
                x[b] = rotl(x[b] ^ x[c], 12);
                x[a] += x[b];
                x[d] = rotl(x[d] ^ x[a], 8);
                x[c] += x[d];
                x[b] = rotl(x[b] ^ x[c], 7);
            }
            void init(const unsigned char *seed)
            {
                static const std::uint32_t sigma[4] = { 0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u }; // "expand 32-byte k"
                std::memcpy(state_, sigma, 16);
                std::memcpy(state_ + 4, seed, 32);      // key
                state_[12] = state_[13] = 0;            // 64-bit block counter
                std::memcpy(state_ + 14, seed + 32, 8); // nonce
                pos_ = 8;
            }
            void refill()
            {
                std::uint32_t x[16];
                std::memcpy(x, state_, sizeof(x));
                for (int i = 0; i < 10; i++)
                {
                    quarter(x, 0, 4, 8, 12);
                    quarter(x, 1, 5, 9, 13);
                    quarter(x, 2, 6, 10, 14);
                    quarter(x, 3, 7, 11, 15);
                    quarter(x, 0, 5, 10, 15);
                    quarter(x, 1, 6, 11, 12);
                    quarter(x, 2, 7, 8, 13);
                    quarter(x, 3, 4, 9, 14);
                }
                for (int i = 0; i < 16; i++)
                {
                    x[i] += state_[i];
                }
                std::memcpy(buf_, x, sizeof(buf_));
                if (++state_[12] == 0)
                {
                    ++state_[13];
                }
                pos_ = 0;
            }
            std::uint32_t state_[16];
            std::uint64_t buf_[8];
            int pos_ = 8;
        };

        inline ChaCha20Rng &thread_rng()
        {
            static thread_local ChaCha20Rng g;
            return g;
        }

        // uniform residues [rows][N], row r under primes[r]
        inline void sample_uniform(const std::vector<std::uint64_t> &primes, std::size_t n, std::vector<std::uint64_t> &out)
        {
            out.resize(primes.size() * n);
            // rows are independent and every thread owns its generator: a switching key at MOAI's size draws 35 x 36 x
            // 65536 residues, which one host thread takes about a second for
#pragma omp parallel for schedule(dynamic)
            for (std::size_t r = 0; r < primes.size(); r++)
            {
                auto &g = thread_rng();
                std::uniform_int_distribution<std::uint64_t> d(0, primes[r] - 1);
                for (std::size_t i = 0; i < n; i++)
                {
                    out[r * n + i] = d(g);
                }
            }
        }

        // small signed polynomial -> RNS rows
        inline void to_rns(const std::vector<std::int64_t> &poly, const std::vector<std::uint64_t> &primes,
                           std::vector<std::uint64_t> &out)
        {
            const std::size_t n = poly.size();
            out.resize(primes.size() * n);
#pragma omp parallel for schedule(static)
            for (std::size_t r = 0; r < primes.size(); r++)
            {
                const std::uint64_t q = primes[r];
                for (std::size_t i = 0; i < n; i++)
                {
                    std::int64_t v = poly[i];
                    out[r * n + i] = v >= 0 ? static_cast<std::uint64_t>(v) % q
                                            : q - (static_cast<std::uint64_t>(-v) % q == 0 ? q : static_cast<std::uint64_t>(-v) % q);
                }
            }
        }

        // centred binomial / clipped normal noise, sigma = 3.2 (SEAL/util/globals.h noise_standard_deviation)
        inline void sample_noise(std::size_t n, std::vector<std::int64_t> &e)
        {
            e.resize(n);
            std::normal_distribution<double> d(0.0, 3.2);
            auto &g = thread_rng();
            for (auto &x : e)
            {
                double v;
                do
                {
                    v = d(g);
                } while (std::fabs(v) > 19.2); // noise_max_deviation = 6 sigma
                x = static_cast<std::int64_t>(std::llround(v));
            }
        }

        // ternary secret; hamming weight hw > 0 gives the fork's sparse secret (SEAL/util/rlwe.cpp:40-97)
        inline void sample_ternary(std::size_t n, std::size_t hw, std::vector<std::int64_t> &s)
        {
            s.assign(n, 0);
            auto &g = thread_rng();
            if (hw == 0 || hw >= n)
            {
                std::uniform_int_distribution<int> d(-1, 1);
                for (auto &x : s)
                {
                    x = d(g);
                }
                return;
            }
            std::size_t placed = 0;
            std::uniform_int_distribution<std::size_t> pos(0, n - 1);
            std::uniform_int_distribution<int> sign(0, 1);
            while (placed < hw)
            {
                std::size_t p = pos(g);
                if (s[p] == 0)
                {
                    s[p] = sign(g) ? 1 : -1;
                    placed++;
                }
            }
        }
    } // namespace util

    // =================================================================================================
    // CKKSEncoder  (SEAL/ckks.{h,cpp})
    // =================================================================================================
    class CKKSEncoder
    {
    public:
        CKKSEncoder(const SEALContext &context) : context_(context)
        {
            const std::size_t n = context_.n();
            logn_ = context_.logn();
            slots_ = n >> 1;
            const std::uint64_t m = static_cast<std::uint64_t>(n) << 1;
            // SEAL/ckks.cpp:36-50 (generator 5 in this fork)
            index_map_.resize(n);
            std::uint64_t gen = 5, pos = 1;
            for (std::size_t i = 0; i < slots_; i++)
            {
                std::uint64_t index1 = (pos - 1) >> 1;
                std::uint64_t index2 = (m - pos - 1) >> 1;
                index_map_[i] = util::reverse_bits(static_cast<std::uint32_t>(index1), logn_);
                index_map_[slots_ + i] = util::reverse_bits(static_cast<std::uint32_t>(index2), logn_);
                pos = (pos * gen) & (m - 1);
            }
            // root_powers_[bitrev(i)] = zeta^i with zeta = exp(2 pi i / 2N); inverse = conjugates
            root_powers_.resize(n);
            inv_root_powers_.resize(n);
            const double pi = 3.14159265358979323846264338327950288;
            for (std::size_t i = 0; i < n; i++)
            {
                double ang = 2.0 * pi * static_cast<double>(i) / static_cast<double>(m);
                std::complex<double> w(std::cos(ang), std::sin(ang));
                std::uint32_t r = util::reverse_bits(static_cast<std::uint32_t>(i), logn_);
                root_powers_[r] = w;
                inv_root_powers_[r] = std::conj(w);
            }
        }

        std::size_t slot_count() const noexcept
        {
            return slots_;
        }

        // ---- vector encodes (SEAL/ckks.h:457-637) -------------------------------------------------
        template <typename T>
        void encode(const std::vector<T> &values, parms_id_type parms_id, double scale, Plaintext &destination,
                    MemoryPoolHandle = MemoryPoolHandle()) const
        {
            encode_vector(values.data(), values.size(), parms_id, scale, destination);
        }
        template <typename T>
        void encode(const std::vector<T> &values, double scale, Plaintext &destination,
                    MemoryPoolHandle = MemoryPoolHandle()) const
        {
            encode_vector(values.data(), values.size(), context_.first_parms_id(), scale, destination);
        }
        // ---- scalar encodes (SEAL/ckks.cpp:77-216): constant rows ---------------------------------
        void encode(double value, parms_id_type parms_id, double scale, Plaintext &destination,
                    MemoryPoolHandle = MemoryPoolHandle()) const
        {
            encode_scalar(value, parms_id, scale, destination);
        }
        void encode(double value, double scale, Plaintext &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            encode_scalar(value, context_.first_parms_id(), scale, destination);
        }
        void encode(std::complex<double> value, parms_id_type parms_id, double scale, Plaintext &destination,
                    MemoryPoolHandle = MemoryPoolHandle()) const
        {
            std::vector<std::complex<double>> v(slots_, value);
            encode_vector(v.data(), v.size(), parms_id, scale, destination);
        }
        void encode(std::complex<double> value, double scale, Plaintext &destination,
                    MemoryPoolHandle = MemoryPoolHandle()) const
        {
            encode(value, context_.first_parms_id(), scale, destination);
        }
        void encode(std::int64_t value, parms_id_type parms_id, Plaintext &destination) const
        {
            encode_scalar(static_cast<double>(value), parms_id, 1.0, destination);
        }
        void encode(std::int64_t value, Plaintext &destination) const
        {
            encode(value, context_.first_parms_id(), destination);
        }

        // ---- decode (SEAL/ckks.h:644-760) -----------------------------------------------------------
        template <typename T>
        void decode(const Plaintext &plain, std::vector<T> &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            auto cd = context_.get_context_data(plain.parms_id());
            if (!cd || !plain.is_ntt_form())
            {
                throw std::invalid_argument("plain is not in NTT form");
            }
            const std::size_t n = context_.n();
            const auto &cm = cd->parms().coeff_modulus();
            const std::size_t L = cm.size();
            std::vector<std::uint64_t> rns(L * n);
            if (plain.is_scalar())
            {
                // constant rows are the NTT of a constant polynomial
                for (std::size_t r = 0; r < L; r++)
                {
                    std::fill(rns.begin() + r * n, rns.begin() + (r + 1) * n, 0);
                    rns[r * n] = plain.scalar_rows()[r];
                }
            }
            else
            {
                util::DeviceArray tmp(L * n, context_.stream());
                util::hip_check(moai_memcpy_d2d(tmp.get(), plain.device_data(), L * n * 8, context_.stream()));
                util::hip_check(moai_ntt_inverse(context_.device(), tmp.get(), 1, L, nullptr, context_.stream()));
                util::hip_check(moai_memcpy_d2h(rns.data(), tmp.get(), L * n * 8, context_.stream()));
                context_.sync();
            }
            std::vector<std::uint64_t> primes;
            for (auto &m : cm)
            {
                primes.push_back(m.value());
            }
            std::vector<double> coeffs(n);
            compose_centered(rns, primes, n, coeffs);
            std::vector<std::complex<double>> res(n);
            const double inv_scale = 1.0 / plain.scale();
            for (std::size_t i = 0; i < n; i++)
            {
                res[i] = std::complex<double>(coeffs[i] * inv_scale, 0.0);
            }
            fft_to_rev(res);
            destination.resize(slots_);
            for (std::size_t i = 0; i < slots_; i++)
            {
                assign(destination[i], res[index_map_[i]]);
            }
        }

    private:
        static void assign(double &d, const std::complex<double> &c)
        {
            d = c.real();
        }
        static void assign(std::complex<double> &d, const std::complex<double> &c)
        {
            d = c;
        }
        static std::complex<double> as_complex(double v)
        {
            return { v, 0.0 };
        }
        static std::complex<double> as_complex(std::complex<double> v)
        {
            return v;
        }

        // forward DWT, natural in -> bit-reversed out (DWTHandler::transform_to_rev with complex roots)
        void fft_to_rev(std::vector<std::complex<double>> &v) const
        {
            const std::size_t n = v.size();
            std::size_t gap = n >> 1, m = 1, root = 0;
            for (; m <= (n >> 1); m <<= 1)
            {
                std::size_t offset = 0;
                for (std::size_t i = 0; i < m; i++)
                {
                    const std::complex<double> r = root_powers_[++root];
                    for (std::size_t j = 0; j < gap; j++)
                    {
                        std::complex<double> u = v[offset + j];
                        std::complex<double> t = v[offset + gap + j] * r;
                        v[offset + j] = u + t;
                        v[offset + gap + j] = u - t;
                    }
                    offset += gap << 1;
                }
                gap >>= 1;
            }
        }

        static const double *as_doubles(const double *v)
        {
            return v;
        }
        static const double *as_doubles(const std::complex<double> *v)
        {
            return reinterpret_cast<const double *>(v); // (re, im) pairs, [complex.numbers.general]
        }

        // SEAL/ckks.h:457-637 on the device (moai_ckks_encode): only the values go over PCIe
        template <typename T>
        void encode_vector(const T *values, std::size_t count, parms_id_type parms_id, double scale,
                           Plaintext &destination) const
        {
            auto cd = context_.get_context_data(parms_id);
            if (!cd)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            if (!values && count > 0)
            {
                throw std::invalid_argument("values cannot be null");
            }
            if (count > slots_)
            {
                throw std::invalid_argument("values_size is too large");
            }
            const auto &cm = cd->parms().coeff_modulus();
            if (scale <= 0 || (static_cast<int>(std::log2(scale)) + 1 >= cd->total_coeff_modulus_bit_count()))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            const bool is_complex = std::is_same<T, std::complex<double>>::value;
            const std::size_t n = context_.n();
            const std::size_t L = cm.size();
            const std::size_t words = count * (is_complex ? 2 : 1);
            void *stream = context_.stream();
            // Every coefficient is (scale / N) * sum of N unit-modulus multiples of the slot values and their
            // conjugates, so |coefficient| <= scale * max |value|.  When that bound already passes the
            // reference's range check (ckks.h:527-538) the check cannot fail and nothing has to come back from
            // the device; otherwise the exact maximum is fetched and tested like the reference does.
            double max_abs = 0;
            for (std::size_t i = 0; i < count; i++)
            {
                max_abs = std::max<>(max_abs, static_cast<double>(std::abs(values[i])));
            }
            const double bound = scale * max_abs * (1.0 + 1e-9);
            const bool conclusive = std::isfinite(bound) && static_cast<int>(std::ceil(std::log2(std::max<>(bound, 1.0)))) + 1 <
                                                                cd->total_coeff_modulus_bit_count();
            // a full vector with one non-zero value at some slots and zero elsewhere -- MOAI's masked weights and biases
            // (Ct_pt_matrix_mul.hpp:124-146, single_att_block.hpp:33-42): recorded, not transformed (Plaintext::mask_).  Only when
            // the range check above cannot fail, so that every exception of the reference is still raised here and now.
            if (!is_complex && conclusive && count == slots_ && masked_constants_enabled())
            {
                if (record_masked_constant(reinterpret_cast<const double *>(values), count, parms_id, scale, L, n, destination))
                {
                    return;
                }
            }
            util::DeviceArray staging(words + 1, stream); // values, then max |coefficient|
            StagingSlot &slot = staging_slot(words * 8);
            if (words)
            {
                // through page-locked memory, so that the copy is asynchronous and the caller's vector is free
                // again when this function returns
                std::memcpy(slot.host, as_doubles(values), words * 8);
                util::hip_check(moai_memcpy_h2d(staging.get(), slot.host, words * 8, stream));
            }
            destination.scalar_rows_.clear();
            destination.parms_id_ = parms_id_zero;
            destination.n_ = n;
            destination.L_ = L;
            destination.stream_ = stream;
            destination.data_.resize(L * n, stream);
            double *max_dev = reinterpret_cast<double *>(staging.get() + words);
            util::hip_check(moai_ckks_encode(context_.device(), reinterpret_cast<const double *>(staging.get()),
                                             is_complex ? 1 : 0, count, 1, destination.data_.get(), L, nullptr, scale,
                                             conclusive ? nullptr : max_dev, stream));
            util::hip_check(moai_event_record(slot.event, stream));
            slot.pending = true;
            if (!conclusive)
            {
                double max_coeff = 0;
                util::hip_check(moai_memcpy_d2h(&max_coeff, max_dev, 8, stream));
                context_.sync();
                // ckks.h:527-538 (the negated comparison also catches NaN)
                int max_coeff_bit_count = static_cast<int>(std::ceil(std::log2(std::max<>(max_coeff, 1.0)))) + 1;
                if (!(max_coeff_bit_count < cd->total_coeff_modulus_bit_count()))
                {
                    destination.data_.release();
                    throw std::invalid_argument("encoded values are too large");
                }
            }
            destination.parms_id_ = parms_id;
            destination.scale_ = scale;
        }

        static bool masked_constants_enabled()
        {
            static const bool on = [] {
                const char *e = std::getenv("MOAI_SHIM_LAZY");
                return !(e && e[0] == '0');
            }();
            return on;
        }
        // values = c * mask with mask in {0, 1}^slots and c != 0?  Then the plaintext records (mask, c, scale).  The mask's device
        // copy is shared by every plaintext with the same pattern this thread encodes in a row (MOAI encodes thousands per mask).
        bool record_masked_constant(const double *values, std::size_t count, parms_id_type parms_id, double scale, std::size_t L, std::size_t n,
                                    Plaintext &destination) const
        {
            double c = 0;
            std::size_t first = count;
            for (std::size_t i = 0; i < count; i++)
            {
                if (values[i] != 0.0)
                {
                    c = values[i];
                    first = i;
                    break;
                }
            }
            if (first == count || !std::isfinite(c))
            {
                return false; // all zero (the reference encodes that too; rare): the ordinary path
            }
            static thread_local std::shared_ptr<const util::SlotMask> last;
            static thread_local moai_ctx *last_dev = nullptr;
            const bool try_last = last && last_dev == context_.device() && last->host.size() == count;
            bool same = try_last;
            for (std::size_t i = 0; i < count; i++)
            {
                const double v = values[i];
                if (v != 0.0 && v != c)
                {
                    return false; // a general vector
                }
                if (same && (v != 0.0) != (last->host[i] != 0))
                {
                    same = false;
                }
            }
            if (!same)
            {
                auto m = std::make_shared<util::SlotMask>();
                m->host.resize(count);
                for (std::size_t i = 0; i < count; i++)
                {
                    m->host[i] = values[i] != 0.0 ? 1 : 0;
                }
                m->dev = std::make_shared<util::DeviceArray>((count + 1) / 2, context_.stream());
                util::hip_check(moai_memcpy_h2d(m->dev->get(), m->host.data(), count * 4, context_.stream()));
                context_.sync(); // once per new pattern
                last = m;
                last_dev = context_.device();
            }
            destination.scalar_rows_.clear();
            destination.data_.release();
            destination.n_ = n;
            destination.L_ = L;
            destination.stream_ = context_.stream();
            destination.dev_ = context_.device();
            destination.parms_id_ = parms_id;
            destination.scale_ = scale;
            {
                std::lock_guard<std::mutex> g(util::lazy_mutex());
                destination.mask_ = last;
                destination.mask_c_ = c;
                destination.mask_scale_ = scale;
                destination.mask_L_ = L;
                destination.owed_.v.store(true, std::memory_order_release);
            }
            return true;
        }

        // page-locked staging buffers for the values of vector encodes: a small ring per host thread; a slot is
        // reused only after the event recorded behind its last copy has completed
        struct StagingSlot
        {
            void *host = nullptr;
            std::size_t bytes = 0;
            void *event = nullptr;
            bool pending = false;
        };
        static StagingSlot &staging_slot(std::size_t bytes)
        {
            static thread_local StagingSlot ring[4];
            static thread_local unsigned next = 0;
            StagingSlot &s = ring[next++ & 3u];
            if (s.pending)
            {
                util::hip_check(moai_event_synchronize(s.event));
                s.pending = false;
            }
            if (!s.event)
            {
                util::hip_check(moai_event_create(&s.event));
            }
            if (s.bytes < bytes)
            {
                if (s.host)
                {
                    util::hip_check(moai_host_free(s.host));
                }
                s.bytes = std::max<std::size_t>(bytes, std::size_t(1) << 16);
                util::hip_check(moai_host_malloc(&s.host, s.bytes));
            }
            return s;
        }

        void encode_scalar(double value, parms_id_type parms_id, double scale, Plaintext &destination) const
        {
            auto cd = context_.get_context_data(parms_id);
            if (!cd)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            const auto &cm = cd->parms().coeff_modulus();
            // SEAL/ckks.cpp:101-115
            if (scale <= 0 || (static_cast<int>(std::log2(scale)) >= cd->total_coeff_modulus_bit_count()))
            {
                throw std::invalid_argument("scale out of bounds");
            }
            value *= scale;
            int coeff_bit_count = value == 0.0 ? 1 : static_cast<int>(std::log2(std::fabs(value))) + 2;
            if (coeff_bit_count >= cd->total_coeff_modulus_bit_count())
            {
                throw std::invalid_argument("encoded value is too large");
            }
            // ckks.cpp:126-211: round, then the exact integer modulo each prime (all three branches of the
            // reference compute that), sign applied by negate_uint_mod; every coefficient of the NTT-form
            // row equals it, so only the L residues are kept
            double c = std::round(value);
            bool neg = std::signbit(c);
            double a = std::fabs(c);
            int e = 0;
            double mant = std::frexp(a, &e); // a = mant * 2^e, mant in [0.5, 1)
            std::uint64_t m53 = a != 0.0 ? static_cast<std::uint64_t>(std::ldexp(mant, 53)) : 0;
            int sh = e - 53;
            destination.scalar_rows_.resize(cm.size());
            for (std::size_t r = 0; r < cm.size(); r++)
            {
                const std::uint64_t q = cm[r].value();
                std::uint64_t v;
                if (sh <= 0)
                {
                    v = (sh > -64 ? (m53 >> (-sh)) : 0) % q;
                }
                else
                {
                    v = m53 % q;
                    for (int left = sh; left > 0;)
                    {
                        int step = left < 63 ? left : 63;
                        v = static_cast<std::uint64_t>((static_cast<util::u128>(v) << step) % q);
                        left -= step;
                    }
                }
                destination.scalar_rows_[r] = (neg && v) ? q - v : v;
            }
            destination.parms_id_ = parms_id;
            destination.scale_ = scale;
            destination.n_ = context_.n();
            destination.L_ = cm.size();
            destination.stream_ = context_.stream();
            destination.data_.release();
        }

    public:
        // centred value of each coefficient mod Q = prod primes, as a double.  Mixed-radix (Garner)
        // digits avoid multi-precision integers: sign from comparing digits with those of floor(Q/2).
        static void compose_centered(const std::vector<std::uint64_t> &rns, const std::vector<std::uint64_t> &primes,
                                     std::size_t n, std::vector<double> &out)
        {
            const std::size_t L = primes.size();
            // inv[i][j] = (q_j)^-1 mod q_i for j < i
            std::vector<std::vector<std::uint64_t>> inv(L);
            for (std::size_t i = 0; i < L; i++)
            {
                inv[i].resize(i);
                for (std::size_t j = 0; j < i; j++)
                {
                    inv[i][j] = util::powmod(primes[j] % primes[i], primes[i] - 2, primes[i]);
                }
            }
            // mixed-radix digits of floor(Q/2): Q/2 = (Q-1)/2 since Q is odd; digits of Q-1 are q_i - 1;
            // halving a mixed-radix number digit by digit from the top
            std::vector<std::uint64_t> half(L);
            {
                std::uint64_t carry = 0; // carry in units of "one of digit i+1" = q_i of digit i
                for (std::size_t ii = L; ii-- > 0;)
                {
                    util::u128 cur = static_cast<util::u128>(carry) * primes[ii] + (primes[ii] - 1);
                    half[ii] = static_cast<std::uint64_t>(cur / 2);
                    carry = static_cast<std::uint64_t>(cur % 2);
                }
            }
            std::vector<long double> weight(L);
            weight[0] = 1.0L;
            for (std::size_t i = 1; i < L; i++)
            {
                weight[i] = weight[i - 1] * static_cast<long double>(primes[i - 1]);
            }
            std::vector<std::uint64_t> d(L);
            out.resize(n);
            for (std::size_t c = 0; c < n; c++)
            {
                for (std::size_t i = 0; i < L; i++)
                {
                    const std::uint64_t q = primes[i];
                    std::uint64_t v = rns[i * n + c] % q;
                    for (std::size_t j = 0; j < i; j++)
                    {
                        std::uint64_t dj = d[j] % q;
                        v = v >= dj ? v - dj : v + q - dj;
                        v = util::mulmod(v, inv[i][j], q);
                    }
                    d[i] = v;
                }
                // compare with half from the top digit
                bool neg = false;
                for (std::size_t ii = L; ii-- > 0;)
                {
                    if (d[ii] != half[ii])
                    {
                        neg = d[ii] > half[ii];
                        break;
                    }
                }
                long double val = 0.0L;
                if (!neg)
                {
                    for (std::size_t ii = L; ii-- > 0;)
                    {
                        val += static_cast<long double>(d[ii]) * weight[ii];
                    }
                }
                else
                {
                    // Q - x in mixed radix: (q_i - 1 - d_i) per digit, plus one
                    long double acc = 1.0L;
                    for (std::size_t ii = L; ii-- > 0;)
                    {
                        acc += static_cast<long double>(primes[ii] - 1 - d[ii]) * weight[ii];
                    }
                    val = -acc;
                }
                out[c] = static_cast<double>(val);
            }
        }

    private:
        SEALContext context_;
        int logn_ = 0;
        std::size_t slots_ = 0;
        std::vector<std::uint32_t> index_map_;
        std::vector<std::complex<double>> root_powers_, inv_root_powers_;
    };

    // =================================================================================================
    // KeyGenerator  (SEAL/keygenerator.cpp)
    // =================================================================================================
    class KeyGenerator
    {
    public:
        KeyGenerator(const SEALContext &context) : context_(context)
        {
            const auto &kp = context_.key_context_data()->parms();
            for (auto &m : kp.coeff_modulus())
            {
                primes_.push_back(m.value());
            }
            n_ = context_.n();
            k_ = primes_.size();
            std::vector<std::int64_t> s;
            util::sample_ternary(n_, kp.secret_key_hamming_weight(), s);
            std::vector<std::uint64_t> rns;
            util::to_rns(s, primes_, rns);
            sk_.ntt_ = std::make_shared<util::DeviceArray>(k_ * n_, context_.stream());
            upload_ntt(rns, *sk_.ntt_, k_);
            sk_.parms_id_ = context_.key_parms_id();
        }
        const SecretKey &secret_key() const
        {
            return sk_;
        }
        void create_public_key(PublicKey &destination) const
        {
            destination.ct_.resize(context_, context_.key_parms_id(), 2);
            encrypt_zero_symmetric(destination.ct_.device_data());
            destination.ct_.is_ntt_form() = true;
            destination.ct_.scale() = 1.0;
        }
        // relinearization key for s^2 (SEAL/keygenerator.cpp:129-168)
        void create_relin_keys(RelinKeys &destination)
        {
            if (!context_.using_keyswitching())
            {
                throw std::logic_error("keyswitching is not supported by the context");
            }
            util::DeviceArray s2(k_ * n_, context_.stream());
            util::hip_check(moai_dyadic_mul(context_.device(), sk_.ntt_->get(), sk_.ntt_->get(), s2.get(), 1, 1, k_,
                                            context_.stream()));
            destination.keys_.assign(1, nullptr);
            destination.hoist_ = std::make_shared<KSwitchKeys::HoistCache>(); // constants derived from the keys this call replaces
            destination.generation_ = KSwitchKeys::next_generation();
            destination.keys_[0] = make_kswitch_key(s2.get());
            destination.parms_id_ = context_.key_parms_id();
            context_.sync();
        }
        // keys for the given Galois elements (SEAL/keygenerator.cpp:170-235)
        void create_galois_keys(const std::vector<std::uint32_t> &galois_elts, GaloisKeys &destination)
        {
            if (!context_.using_keyswitching())
            {
                throw std::logic_error("keyswitching is not supported by the context");
            }
            destination.keys_.assign(n_, nullptr);
            // the hoisted-rotation constants are functions of the key (KSwitchKeys::hoist_correction): regenerating keys into an
            // object that already served hoisted rotations must not leave the old keys' constants behind.  A copy made earlier
            // keeps the old keys together with the old cache; copies made from now on share the new one.
            destination.hoist_ = std::make_shared<KSwitchKeys::HoistCache>();
            destination.generation_ = KSwitchKeys::next_generation();
            util::DeviceArray rotated(k_ * n_, context_.stream());
            for (std::uint32_t elt : galois_elts)
            {
                if (!(elt & 1) || elt >= 2 * n_)
                {
                    throw std::invalid_argument("Galois element is not valid");
                }
                if (destination.keys_[GaloisKeys::get_index(elt)])
                {
                    continue;
                }
                util::hip_check(moai_galois_permute(context_.device(), sk_.ntt_->get(), rotated.get(), 1, k_, elt,
                                                    context_.stream()));
                destination.keys_[GaloisKeys::get_index(elt)] = make_kswitch_key(rotated.get());
            }
            destination.parms_id_ = context_.key_parms_id();
            context_.sync();
        }
        void create_galois_keys(const std::vector<int> &steps, GaloisKeys &destination)
        {
            std::vector<std::uint32_t> elts;
            for (int s : steps)
            {
                std::uint32_t e = moai_galois_elt_from_step(context_.device(), s);
                if (!e)
                {
                    throw std::invalid_argument("step count too large");
                }
                elts.push_back(e);
            }
            create_galois_keys(elts, destination);
        }
        // all power-of-two rotations and the conjugation (GaloisTool::get_elts_all,
        // SEAL/util/galois.cpp:106-131)
        void create_galois_keys(GaloisKeys &destination)
        {
            std::vector<std::uint32_t> elts;
            const std::uint64_t m = static_cast<std::uint64_t>(n_) << 1;
            elts.push_back(static_cast<std::uint32_t>(m - 1));
            std::uint64_t pos = 5, neg = 0;
            for (std::uint64_t x = 1; x < m; x += 2)
            {
                if (((x * 5) & (m - 1)) == 1)
                {
                    neg = x;
                    break;
                }
            }
            for (int i = 0; i < context_.logn() - 1; i++)
            {
                elts.push_back(static_cast<std::uint32_t>(pos));
                pos = (pos * pos) & (m - 1);
                elts.push_back(static_cast<std::uint32_t>(neg));
                neg = (neg * neg) & (m - 1);
            }
            create_galois_keys(elts, destination);
        }

    private:
        void upload_ntt(const std::vector<std::uint64_t> &rns, util::DeviceArray &dst, std::size_t rows) const
        {
            util::hip_check(moai_memcpy_h2d(dst.get(), rns.data(), rows * n_ * 8, context_.stream()));
            context_.sync();
            util::hip_check(moai_ntt_forward(context_.device(), dst.get(), 1, rows, nullptr, context_.stream()));
        }
        // (c0, c1) = (-(a s) + e, a) at the key level, NTT form, written to dst [2][k][N]
        void encrypt_zero_symmetric(std::uint64_t *dst) const
        {
            std::vector<std::uint64_t> a;
            util::sample_uniform(primes_, n_, a);
            std::vector<std::int64_t> e;
            util::sample_noise(n_, e);
            std::vector<std::uint64_t> e_rns;
            util::to_rns(e, primes_, e_rns);
            std::uint64_t *c0 = dst;
            std::uint64_t *c1 = dst + k_ * n_;
            util::hip_check(moai_memcpy_h2d(c1, a.data(), k_ * n_ * 8, context_.stream()));
            util::hip_check(moai_memcpy_h2d(c0, e_rns.data(), k_ * n_ * 8, context_.stream()));
            context_.sync();
            util::hip_check(moai_ntt_forward(context_.device(), c0, 1, k_, nullptr, context_.stream()));
            util::DeviceArray as(k_ * n_, context_.stream());
            util::hip_check(moai_dyadic_mul(context_.device(), c1, sk_.ntt_->get(), as.get(), 1, 1, k_, context_.stream()));
            util::hip_check(moai_sub(context_.device(), c0, as.get(), c0, 1, k_, context_.stream()));
            context_.sync();
        }
        // SEAL/keygenerator.cpp:303-336: key[J] = Enc(0) with c0's row J += (p mod q_J) * new_key[J]
        std::shared_ptr<util::DeviceArray> make_kswitch_key(const std::uint64_t *new_key_ntt) const
        {
            const std::size_t digits = k_ - 1;
            auto key = std::make_shared<util::DeviceArray>(digits * 2 * k_ * n_, context_.stream());
            util::DeviceArray scaled(k_ * n_, context_.stream());
            std::vector<std::uint64_t> factor(k_, 0);
            for (std::size_t j = 0; j < k_; j++)
            {
                factor[j] = primes_[k_ - 1] % primes_[j];
            }
            util::hip_check(moai_mul_scalar_rows(context_.device(), new_key_ntt, factor.data(), scaled.get(), 1, k_,
                                                 context_.stream()));
            // The digits are independent encryptions of zero: host threads draw their randomness side by side (a key at
            // MOAI's size is 35 x 2 x 36 x 65536 samples) and enqueue on the context's one stream, where each digit's own
            // operations keep their order.  Every thread adds through its own addend [k][N], zero except row J =
            // (p mod q_J) * new_key[J].
            std::exception_ptr failure;
#pragma omp parallel
            {
                std::unique_ptr<util::DeviceArray> sparse;
#pragma omp for schedule(dynamic)
                for (std::size_t J = 0; J < digits; J++)
                {
                    try
                    {
                        if (!sparse)
                        {
                            sparse.reset(new util::DeviceArray(k_ * n_, context_.stream()));
                            util::hip_check(moai_memset_zero(sparse->get(), k_ * n_ * 8, context_.stream()));
                        }
                        std::uint64_t *ct = key->get() + J * 2 * k_ * n_;
                        encrypt_zero_symmetric(ct);
                        util::hip_check(moai_memcpy_d2d(sparse->get() + J * n_, scaled.get() + J * n_, n_ * 8, context_.stream()));
                        util::hip_check(moai_add(context_.device(), ct, sparse->get(), ct, 1, k_, context_.stream()));
                        util::hip_check(moai_memset_zero(sparse->get() + J * n_, n_ * 8, context_.stream()));
                    }
                    catch (...)
                    {
#pragma omp critical(moai_keygen_failure)
                        failure = std::current_exception();
                    }
                }
                if (sparse)
                {
                    context_.sync(); // before this thread's addend is released
                }
            }
            if (failure)
            {
                std::rethrow_exception(failure);
            }
            context_.sync();
            return key;
        }

        SEALContext context_;
        std::vector<std::uint64_t> primes_;
        std::size_t n_ = 0, k_ = 0;
        SecretKey sk_;
    };

    // =================================================================================================
    // Encryptor / Decryptor
    // =================================================================================================
    class Encryptor
    {
    public:
        Encryptor(const SEALContext &context, const PublicKey &public_key) : context_(context), pk_(public_key.data())
        {}
        Encryptor(const SEALContext &context, const SecretKey &) : context_(context)
        {
            throw std::logic_error("symmetric Encryptor is not provided");
        }
        // public-key encryption at the level of `plain` (SEAL/encryptor.cpp encrypt_internal; the
        // reference samples at the key level and divides by the special prime, here the public key is
        // restricted to the plaintext's primes -- a fresh RLWE encryption either way)
        void encrypt(const Plaintext &plain, Ciphertext &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            auto cd = context_.get_context_data(plain.parms_id());
            if (!cd || !plain.is_ntt_form())
            {
                throw std::invalid_argument("plain is not valid for encryption parameters");
            }
            encrypt_zero(plain.parms_id(), destination);
            const std::size_t L = cd->parms().coeff_modulus().size();
            if (plain.is_scalar())
            {
                util::hip_check(moai_add_scalar_rows(context_.device(), destination.device_data(), plain.scalar_rows().data(),
                                                     destination.device_data(), 1, L, context_.stream()));
            }
            else
            {
                util::hip_check(moai_add(context_.device(), destination.device_data(), plain.device_data(),
                                         destination.device_data(), 1, L, context_.stream()));
            }
            destination.scale() = plain.scale();
        }
        void encrypt_zero(parms_id_type parms_id, Ciphertext &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            auto cd = context_.get_context_data(parms_id);
            if (!cd)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            const auto &cm = cd->parms().coeff_modulus();
            const std::size_t L = cm.size(), n = context_.n();
            const std::size_t k = context_.key_context_data()->parms().coeff_modulus().size();
            std::vector<std::uint64_t> primes;
            for (auto &m : cm)
            {
                primes.push_back(m.value());
            }
            destination.resize(context_, parms_id, 2);
            destination.is_ntt_form() = true;
            destination.scale() = 1.0;
            // u ternary, e0, e1 noise
            std::vector<std::int64_t> u, e0, e1;
            util::sample_ternary(n, 0, u);
            util::sample_noise(n, e0);
            util::sample_noise(n, e1);
            std::vector<std::uint64_t> u_rns, e_rns(2 * L * n), tmp;
            util::to_rns(u, primes, u_rns);
            util::to_rns(e0, primes, tmp);
            std::copy(tmp.begin(), tmp.end(), e_rns.begin());
            util::to_rns(e1, primes, tmp);
            std::copy(tmp.begin(), tmp.end(), e_rns.begin() + static_cast<std::ptrdiff_t>(L * n));
            util::DeviceArray du(L * n, context_.stream());
            util::hip_check(moai_memcpy_h2d(du.get(), u_rns.data(), L * n * 8, context_.stream()));
            util::hip_check(moai_memcpy_h2d(destination.device_data(), e_rns.data(), 2 * L * n * 8, context_.stream()));
            context_.sync();
            util::hip_check(moai_ntt_forward(context_.device(), du.get(), 1, L, nullptr, context_.stream()));
            util::hip_check(moai_ntt_forward(context_.device(), destination.device_data(), 2, L, nullptr, context_.stream()));
            // c_i = pk_i * u + e_i over the first L primes of the key-level public key
            util::DeviceArray prod(L * n, context_.stream());
            for (int i = 0; i < 2; i++)
            {
                const std::uint64_t *pk_poly = pk_.device_data() + static_cast<std::size_t>(i) * k * n;
                std::uint64_t *c = destination.device_data() + static_cast<std::size_t>(i) * L * n;
                util::hip_check(moai_dyadic_mul(context_.device(), pk_poly, du.get(), prod.get(), 1, 1, L, context_.stream()));
                util::hip_check(moai_add(context_.device(), c, prod.get(), c, 1, L, context_.stream()));
            }
            context_.sync();
        }
        void encrypt_zero(Ciphertext &destination, MemoryPoolHandle = MemoryPoolHandle()) const
        {
            encrypt_zero(context_.first_parms_id(), destination);
        }

    private:
        SEALContext context_;
        Ciphertext pk_;
    };

    class Decryptor
    {
    public:
        Decryptor(const SEALContext &context, const SecretKey &secret_key) : context_(context), sk_(secret_key.ntt_)
        {}
        // c0 + c1 s + c2 s^2 ... (SEAL/decryptor.cpp:131-205)
        void decrypt(const Ciphertext &encrypted, Plaintext &destination)
        {
            auto cd = context_.get_context_data(encrypted.parms_id());
            if (!cd || encrypted.size() < 2)
            {
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            }
            if (!encrypted.is_ntt_form())
            {
                throw std::invalid_argument("encrypted must be in NTT form");
            }
            if (encrypted.batch() != 1)
            {
                throw std::invalid_argument("packed ciphertext: moai_fused::unpack it before decrypting");
            }
            const std::size_t L = encrypted.coeff_modulus_size(), n = context_.n();
            destination.scalar_rows_.clear();
            destination.parms_id_ = encrypted.parms_id();
            destination.scale_ = encrypted.scale();
            destination.n_ = n;
            destination.L_ = L;
            destination.stream_ = context_.stream();
            destination.data_.resize(L * n, context_.stream());
            std::uint64_t *acc = destination.data_.get();
            const std::uint64_t *ct = encrypted.device_data();
            util::DeviceArray spow(L * n, context_.stream()), term(L * n, context_.stream());
            util::hip_check(moai_memcpy_d2d(acc, ct, L * n * 8, context_.stream()));
            util::hip_check(moai_memcpy_d2d(spow.get(), sk_->get(), L * n * 8, context_.stream()));
            for (std::size_t p = 1; p < encrypted.size(); p++)
            {
                util::hip_check(moai_dyadic_mul(context_.device(), ct + p * L * n, spow.get(), term.get(), 1, 1, L,
                                                context_.stream()));
                util::hip_check(moai_add(context_.device(), acc, term.get(), acc, 1, L, context_.stream()));
                if (p + 1 < encrypted.size())
                {
                    util::hip_check(moai_dyadic_mul(context_.device(), spow.get(), sk_->get(), spow.get(), 1, 1, L,
                                                    context_.stream()));
                }
            }
            context_.sync();
        }

    private:
        SEALContext context_;
        std::shared_ptr<util::DeviceArray> sk_;
    };
} // namespace seal
