// seal/seal.h -- the seal:: C++ API surface MOAI consumes (include/include.hpp:10), re-provided over
// the C ABI of libmoai_hip.so (include/moai_hip.h).  MOAI's headers compile against this file
// unchanged; every ciphertext operation is validated on the host exactly where the reference
// validates it (same exception types and messages) and then enqueued as hand-written HIP kernels.
//
// Mirrors (names, argument meaning, error behaviour), SEAL/ = thirdparty/SEAL-4.1-bs/native/src/seal/:
//   EncryptionParameters  SEAL/encryptionparams.h (fork: set_secret_key_hamming_weight :188-200,
//                         set_sparse_slots :217-234)
//   Modulus, CoeffModulus SEAL/modulus.{h,cpp}
//   SEALContext           SEAL/context.{h,cpp} (modulus switching chain :422-522)
//   Plaintext, Ciphertext SEAL/plaintext.h, SEAL/ciphertext.h (layout [poly][prime][coeff] :337-349)
//   KSwitchKeys/RelinKeys/GaloisKeys  SEAL/kswitchkeys.h:340, relinkeys.h:58, galoiskeys.h:48
//   KeyGenerator, Encryptor, Decryptor  SEAL/keygenerator.cpp, encryptor.cpp, decryptor.cpp (client side)
//   CKKSEncoder           SEAL/ckks.{h,cpp}
//   Evaluator             SEAL/evaluator.{h,cpp} incl. the fork's additions :395-594
//
// Residues live in device memory; metadata lives on the host.  All work is enqueued on the context's
// stream in host call order, so results are ordered exactly as the calling program orders them, also
// when MOAI calls from many OpenMP threads.  Client-side pieces (keygen, encrypt, decrypt, encoder
// FFT) are host code and out of the hot-path scope; their NTTs still run on the device.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "moai_hip.h"

namespace seal
{
    using parms_id_type = std::array<std::uint64_t, 4>;
    static const parms_id_type parms_id_zero = { 0, 0, 0, 0 };

    enum class scheme_type : std::uint8_t
    {
        none = 0x0,
        bfv = 0x1,
        ckks = 0x2,
        bgv = 0x3
    };

    enum class sec_level_type : int
    {
        none = 0,
        tc128 = 128,
        tc192 = 192,
        tc256 = 256
    };

    // ---- memory pool handles: accepted and ignored (device memory comes from the moai arena) --------
    class MemoryPoolHandle
    {
    public:
        MemoryPoolHandle() = default;
        explicit operator bool() const noexcept
        {
            return true;
        }
        static MemoryPoolHandle Global()
        {
            return MemoryPoolHandle();
        }
        static MemoryPoolHandle New(bool = false)
        {
            return MemoryPoolHandle();
        }
    };

    enum class mm_prof_opt : std::uint64_t
    {
        mm_default = 0x0,
        mm_force_global = 0x1,
        mm_force_new = 0x2,
        mm_force_thread_local = 0x4
    };

    class MemoryManager
    {
    public:
        template <typename... Args>
        static MemoryPoolHandle GetPool(Args &&...)
        {
            return MemoryPoolHandle();
        }
    };

    namespace util
    {
        typedef unsigned __int128 u128;

        inline void hip_check(int rc)
        {
            if (rc == MOAI_OK)
            {
                return;
            }
            std::string msg = moai_last_error();
            switch (rc)
            {
            case MOAI_EINVAL:
                throw std::invalid_argument(msg);
            case MOAI_ERANGE:
                throw std::out_of_range(msg);
            case MOAI_ELOGIC:
                throw std::logic_error(msg);
            default:
                throw std::runtime_error(msg);
            }
        }

        inline int get_significant_bit_count(std::uint64_t v)
        {
            int n = 0;
            while (v)
            {
                n++;
                v >>= 1;
            }
            return n;
        }

        inline std::uint64_t mulmod(std::uint64_t a, std::uint64_t b, std::uint64_t q)
        {
            return static_cast<std::uint64_t>((static_cast<u128>(a) * b) % q);
        }

        inline std::uint64_t powmod(std::uint64_t a, std::uint64_t e, std::uint64_t q)
        {
            std::uint64_t r = 1;
            a %= q;
            while (e)
            {
                if (e & 1)
                {
                    r = mulmod(r, a, q);
                }
                a = mulmod(a, a, q);
                e >>= 1;
            }
            return r;
        }

        inline bool is_prime(std::uint64_t n)
        {
            static const std::uint64_t bases[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
            if (n < 2)
            {
                return false;
            }
            for (std::uint64_t p : bases)
            {
                if (n % p == 0)
                {
                    return n == p;
                }
            }
            std::uint64_t d = n - 1;
            int r = 0;
            while (!(d & 1))
            {
                d >>= 1;
                r++;
            }
            for (std::uint64_t a : bases)
            {
                std::uint64_t x = powmod(a, d, n);
                if (x == 1 || x == n - 1)
                {
                    continue;
                }
                bool composite = true;
                for (int i = 1; i < r; i++)
                {
                    x = mulmod(x, x, n);
                    if (x == n - 1)
                    {
                        composite = false;
                        break;
                    }
                }
                if (composite)
                {
                    return false;
                }
            }
            return true;
        }

        inline std::uint32_t reverse_bits(std::uint32_t x, int bits)
        {
            std::uint32_t r = 0;
            for (int i = 0; i < bits; i++)
            {
                r = (r << 1) | ((x >> i) & 1u);
            }
            return r;
        }

        // non-adjacent form (SEAL/util/numth.h:22-41)
        inline std::vector<int> naf(int value)
        {
            std::vector<int> res;
            bool sign = value < 0;
            value = std::abs(value);
            for (int i = 0; value; i++)
            {
                int zi = (value & 1) ? 2 - (value & 3) : 0;
                value = (value - zi) >> 1;
                if (zi)
                {
                    res.push_back((sign ? -zi : zi) * (1 << i));
                }
            }
            return res;
        }

        // Stream-ordered cache of device blocks behind DeviceArray (the role of SEAL's MemoryPoolMT,
        // SEAL/util/mempool.h:228).  A released block may be handed to a later request of the same size on the
        // SAME stream only: everything that touched it was enqueued on that stream before the release, so stream
        // order makes the reuse safe without synchronising, and hipFree's implicit device sync is avoided.
        // That invariant -- a block is only ever handed to the library together with the stream it is labelled with -- is
        // enforced by construction where it can be (a DeviceArray cannot be allocated without a stream; every shim object
        // takes SEALContext::stream()) and CHECKED under MOAI_STREAM_AUDIT=1: the pool reports every block's label and
        // state to the library (moai_debug_block_label), and every entry point that enqueues work refuses a pointer whose
        // block carries another stream's label or has been released (include/moai_hip.h, "stream audit").
        class DevicePool
        {
        public:
            static DevicePool &instance()
            {
                static DevicePool p;
                return p;
            }
            // Requests are rounded up to m * 2^k with m in 8..15 (at most 12.5 % slack), so that the blocks of
            // ciphertexts a level or two apart -- what a rescale / mod-switch chain allocates and frees all the
            // time -- are interchangeable instead of each size keeping its own free list.
            static std::size_t size_class(std::size_t bytes)
            {
                if (bytes <= 4096)
                {
                    return 4096;
                }
                int top = 63 - __builtin_clzll(static_cast<unsigned long long>(bytes));
                const std::size_t step = std::size_t(1) << (top - 3);
                return (bytes + step - 1) & ~(step - 1);
            }
            // *granted (optional) receives the size of the block handed out, which may exceed the request
            void *acquire(std::size_t bytes, void *stream, std::size_t *granted = nullptr)
            {
                {
                    // best fit among this stream's cached blocks, up to 1.5 x the request: a rescale / mod-switch
                    // chain asks for slightly smaller blocks at every step and would otherwise grow a free list
                    // per level
                    std::lock_guard<std::mutex> g(mu_);
                    auto it = free_.lower_bound({ stream, bytes });
                    while (it != free_.end() && it->first.first == stream && it->first.second <= bytes + bytes / 2)
                    {
                        if (!it->second.empty())
                        {
                            void *p = it->second.back().ptr; // the most recently released: the old ones age towards a trim
                            it->second.pop_back();
                            last_hit_[it->first] = now_seconds();
                            cached_ -= it->first.second;
                            if (granted)
                            {
                                *granted = it->first.second;
                            }
                            label(p, it->first.second, stream, 1);
                            return p;
                        }
                        ++it;
                    }
                }
                if (granted)
                {
                    *granted = bytes;
                }
                void *p = nullptr;
                const auto t_malloc = std::chrono::steady_clock::now();
                int rc = moai_malloc(&p, bytes);
                fresh_count_.fetch_add(1, std::memory_order_relaxed);
                fresh_us_.fetch_add(static_cast<std::uint64_t>(
                                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_malloc).count()),
                                    std::memory_order_relaxed);
                if (rc != MOAI_OK)
                {
                    // Out of device memory: ONE thread gives cached blocks back (trim() says which), a few GiB at a time: what
                    // stays cached is what later stages find; the other threads wait here and retry once
                    // it is done (a thread that retried while the blocks were still being freed would fail for good:
                    // MOAI's OpenMP loops allocate from 16+ threads).
                    std::lock_guard<std::mutex> g(oom_mu_);
                    rc = moai_malloc(&p, bytes);
                    if (rc != MOAI_OK && pressure_hook())
                    {
                        pressure_hook()(); // caches above the pool (util::RotationCache) let go of their blocks first
                        rc = moai_malloc(&p, bytes);
                    }
                    while (rc != MOAI_OK)
                    {
                        std::size_t want;
                        {
                            std::lock_guard<std::mutex> g2(mu_);
                            want = std::max<std::size_t>(2 * bytes, std::size_t(4) << 30);
                        }
                        if (trim(want) == 0)
                        {
                            hip_check(rc); // nothing left to give back
                        }
                        rc = moai_malloc(&p, bytes);
                    }
                }
                label(p, bytes, stream, 1);
                return p;
            }
            void release(void *p, std::size_t bytes, void *stream)
            {
                if (bytes > cap_)
                {
                    label(p, bytes, stream, 0);
                    moai_free(p); // larger than the whole cache may be (MOAI_POOL_CACHE_MB=0 switches caching off)
                    return;
                }
                label(p, bytes, stream, 2);
                bool full;
                {
                    std::lock_guard<std::mutex> g(mu_);
                    full = cached_ + bytes > cap_;
                }
                if (full)
                {
                    // The cache is at its cap.  The block coming in is the most recently used one there is: room is made by
                    // trim()'s order instead (idle lists first), a couple of GiB at a time.  (Giving back the incoming block --
                    // what this did before -- turned every release into a hipFree and every request into a hipMalloc for as
                    // long as the cache stayed full: 55 000 device allocations in one attention head.)
                    std::lock_guard<std::mutex> g(oom_mu_);
                    trim(std::max<std::size_t>(bytes, std::size_t(2) << 30));
                }
                std::lock_guard<std::mutex> g(mu_);
                free_[{ stream, bytes }].push_back({ p, ++clock_ });
                cached_ += bytes;
            }
            // gives cached blocks back to the device until `at_least` bytes are returned (everything by default), in this order:
            //   1. blocks below 256 MiB whose free list (one stream, one size) has not served a request for a second,
            //      least recently released first -- the leftovers of a stage that is over, or inputs that a running
            //      stage releases and never asks for again;
            //   2. blocks of 256 MiB and more, least recently released first -- packed ciphertexts: getting one from the device
            //      again costs tens of milliseconds (measured: 40 ms per GiB), and the next bootstrapping round wants them back;
            //   3. the small blocks of the lists that are serving requests -- what the running loops are cycling through (a
            //      temporary of MOAI's multiply_plain + add_inplace loops lives for microseconds): giving those back makes the
            //      very next request go to the device again.
            // Returns the bytes freed.
            std::size_t trim(std::size_t at_least = ~std::size_t(0))
            {
                constexpr std::size_t large = std::size_t(256) << 20;
                const double hot_after = now_seconds() - 1.0;
                std::vector<void *> victims;
                std::size_t freed = 0;
                {
                    std::lock_guard<std::mutex> g(mu_);
                    struct Age
                    {
                        int cls;
                        std::uint64_t tick;
                        std::size_t bytes;
                    };
                    auto class_of = [&](const std::pair<void *, std::size_t> &list) {
                        if (list.second >= large)
                        {
                            return 1;
                        }
                        auto hit = last_hit_.find(list);
                        return hit != last_hit_.end() && hit->second >= hot_after ? 2 : 0;
                    };
                    std::vector<Age> ages;
                    for (auto &kv : free_)
                    {
                        const int cls = class_of(kv.first);
                        for (auto &blk : kv.second)
                        {
                            ages.push_back({ cls, blk.tick, kv.first.second });
                        }
                    }
                    std::sort(ages.begin(), ages.end(), [](const Age &a, const Age &b) { return a.cls != b.cls ? a.cls < b.cls : a.tick < b.tick; });
                    std::uint64_t newest_victim[3] = { 0, 0, 0 };
                    for (auto &a : ages)
                    {
                        if (freed >= at_least)
                        {
                            break;
                        }
                        freed += a.bytes;
                        newest_victim[a.cls] = a.tick;
                    }
                    for (auto it = free_.begin(); it != free_.end();)
                    {
                        // one class per list, ticks ascend within it: the victims are a prefix
                        auto &list = it->second;
                        const std::uint64_t cut = newest_victim[class_of(it->first)];
                        std::size_t k = 0;
                        while (k < list.size() && list[k].tick <= cut)
                        {
                            victims.push_back(list[k].ptr);
                            k++;
                        }
                        list.erase(list.begin(), list.begin() + static_cast<std::ptrdiff_t>(k));
                        it = list.empty() ? free_.erase(it) : std::next(it);
                    }
                    cached_ -= freed;
                }
                const auto t0 = std::chrono::steady_clock::now();
                for (void *p : victims)
                {
                    label(p, 0, nullptr, 0);
                    moai_free(p);
                }
                if (std::getenv("MOAI_POOL_DEBUG"))
                {
                    std::fprintf(stderr, "[pool] gave back %zu blocks, %.1f GiB (asked for %.1f) in %.0f ms; %.1f GiB stay cached\n", victims.size(),
                                 freed / 1073741824.0, (at_least == ~std::size_t(0) ? 0.0 : at_least / 1073741824.0),
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), cached_bytes() / 1073741824.0);
                }
                return freed;
            }
            // called when the device refuses an allocation, before cached blocks are given back
            static std::function<void()> &pressure_hook()
            {
                static std::function<void()> h;
                return h;
            }
            static double now_seconds()
            {
                return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            }
            std::size_t cached_bytes()
            {
                std::lock_guard<std::mutex> g(mu_);
                return cached_;
            }
            // requests that went to the device allocator instead of the cache, and the time spent there (diagnostics)
            std::pair<std::uint64_t, double> fresh_allocations() const
            {
                return { fresh_count_.load(), fresh_us_.load() * 1e-3 };
            }
            // cached blocks are deliberately not returned at static destruction: the HIP runtime may already be
            // gone by then, and the process is exiting anyway
            ~DevicePool() = default;

        private:
            // stream audit (debug): tell the library which stream a block belongs to; 1 = in use, 2 = cached, 0 = gone
            static void label(void *p, std::size_t bytes, void *stream, int state)
            {
                static const bool audit = [] {
                    const char *e = std::getenv("MOAI_STREAM_AUDIT");
                    return e && e[0] != '0';
                }();
                if (audit)
                {
                    moai_debug_block_label(p, bytes, stream, state);
                }
            }
            DevicePool()
            {
                const char *e = std::getenv("MOAI_POOL_CACHE_MB");
                cap_ = (e ? static_cast<std::size_t>(std::atoll(e)) : std::size_t(114688)) << 20; // 112 GiB: freed blocks are worth keeping (every hipFree waits for the device); an allocation that fails trims the cache
            }
            std::mutex mu_, oom_mu_;
            struct Block
            {
                void *ptr;
                std::uint64_t tick; // value of clock_ when it was released
            };
            std::map<std::pair<void *, std::size_t>, std::vector<Block>> free_;
            std::map<std::pair<void *, std::size_t>, double> last_hit_; // when a list last served a request
            std::size_t cached_ = 0, cap_ = 0;
            std::uint64_t clock_ = 0;
            std::atomic<std::uint64_t> fresh_count_{ 0 }, fresh_us_{ 0 };
        };

        // RAII device buffer of uint64 words
        class DeviceArray
        {
        public:
            DeviceArray() = default;
            // no default for the stream: the pool's reuse rule rests on the label being the stream the block is used on
            DeviceArray(std::size_t words, void *stream)
            {
                resize(words, stream);
            }
            ~DeviceArray()
            {
                release();
            }
            DeviceArray(const DeviceArray &o) = delete;
            DeviceArray &operator=(const DeviceArray &o) = delete;
            DeviceArray(DeviceArray &&o) noexcept : ptr_(o.ptr_), words_(o.words_), cap_(o.cap_), stream_(o.stream_)
            {
                o.ptr_ = nullptr;
                o.words_ = o.cap_ = 0;
            }
            DeviceArray &operator=(DeviceArray &&o) noexcept
            {
                if (this != &o)
                {
                    release();
                    ptr_ = o.ptr_;
                    words_ = o.words_;
                    cap_ = o.cap_;
                    stream_ = o.stream_;
                    o.ptr_ = nullptr;
                    o.words_ = o.cap_ = 0;
                }
                return *this;
            }
            void release()
            {
                if (ptr_)
                {
                    DevicePool::instance().release(ptr_, cap_ * sizeof(std::uint64_t), stream_);
                }
                ptr_ = nullptr;
                words_ = cap_ = 0;
            }
            // keeps the leading words (like DynArray::resize, SEAL/dynarray.h)
            void resize(std::size_t words, void *stream)
            {
                if (words <= cap_)
                {
                    words_ = words;
                    return;
                }
                if (!stream)
                {
                    // the legacy stream does not order against a context's non-blocking stream: a block labelled with it would
                    // be recycled without any ordering against the kernels that use it
                    throw std::logic_error("DeviceArray: a device block needs the stream it will be used on");
                }
                std::size_t alloc_words = DevicePool::size_class(words * sizeof(std::uint64_t)) / sizeof(std::uint64_t);
                std::size_t granted = 0;
                void *p = DevicePool::instance().acquire(alloc_words * sizeof(std::uint64_t), stream, &granted);
                alloc_words = granted / sizeof(std::uint64_t);
                if (ptr_ && words_)
                {
                    // ordered on `stream`; the old block goes back to the pool of its own stream and can only
                    // be reused behind this copy when both are the same stream, so drain in the other case
                    hip_check(moai_memcpy_d2d(p, ptr_, words_ * sizeof(std::uint64_t), stream));
                    if (stream != stream_)
                    {
                        hip_check(moai_stream_sync(stream));
                    }
                }
                if (ptr_)
                {
                    DevicePool::instance().release(ptr_, cap_ * sizeof(std::uint64_t), stream_);
                }
                ptr_ = static_cast<std::uint64_t *>(p);
                words_ = words;
                cap_ = alloc_words;
                stream_ = stream;
            }
            std::uint64_t *get() const
            {
                return ptr_;
            }
            std::size_t size() const
            {
                return words_;
            }

        private:
            std::uint64_t *ptr_ = nullptr;
            std::size_t words_ = 0;
            std::size_t cap_ = 0;
            void *stream_ = nullptr;
        };
        // Results of single-ciphertext rotations, shared between callers.  MOAI's Q K^T loop (Ct_ct_matrix_mul.hpp:22-31) rotates
        // the SAME 64 ciphertexts by 127 different steps, every step from scratch, and steps without a key of their own take
        // the reference's path through the non-adjacent form -- a chain of power-of-two rotations (SEAL/evaluator.cpp:2699-2721):
        // 330 key switches per ciphertext where a prefix tree of the chains has 127 leaves plus their shared prefixes.
        // An entry maps (block of the source, Galois element, key generation, key index, level) to the block of the result and
        // keeps BOTH blocks alive: a Ciphertext never writes into a block it shares (copy on write), so a cached block cannot
        // change, and a live block's address cannot be reused.  The result of a hit is the same block a miss computed: same
        // bits by construction.  Least recently used entries go when the cap is reached (MOAI_SHIM_ROTCACHE_MB, default 24576;
        // 0 switches the cache off).
        class RotationCache
        {
        public:
            using Key = std::tuple<const void *, std::uint32_t, std::uint64_t, std::size_t, std::size_t>;
            static RotationCache &instance()
            {
                static RotationCache c;
                return c;
            }
            bool enabled() const
            {
                return cap_ > 0;
            }
            std::shared_ptr<DeviceArray> find(const Key &k)
            {
                std::lock_guard<std::mutex> g(mu_);
                auto it = map_.find(k);
                if (it == map_.end())
                {
                    misses_++;
                    return nullptr;
                }
                hits_++;
                it->second.tick = ++clock_;
                return it->second.out;
            }
            void insert(const Key &k, const std::shared_ptr<DeviceArray> &src, const std::shared_ptr<DeviceArray> &out)
            {
                const std::size_t bytes = out->size() * sizeof(std::uint64_t);
                if (2 * bytes > cap_)
                {
                    return;
                }
                std::lock_guard<std::mutex> g(mu_);
                auto it = map_.find(k);
                if (it != map_.end())
                {
                    return; // another thread computed the same rotation meanwhile: keep the first (equal bits)
                }
                while (bytes_ + bytes > cap_ && !map_.empty())
                {
                    auto oldest = map_.begin();
                    for (auto jt = map_.begin(); jt != map_.end(); ++jt)
                    {
                        if (jt->second.tick < oldest->second.tick)
                        {
                            oldest = jt;
                        }
                    }
                    bytes_ -= oldest->second.bytes;
                    map_.erase(oldest);
                }
                map_[k] = Entry{ src, out, bytes, ++clock_ };
                bytes_ += bytes;
            }
            void clear()
            {
                std::lock_guard<std::mutex> g(mu_);
                map_.clear();
                bytes_ = 0;
            }
            std::pair<std::uint64_t, std::uint64_t> statistics() const
            {
                return { hits_.load(), misses_.load() };
            }

        private:
            RotationCache()
            {
                const char *e = std::getenv("MOAI_SHIM_ROTCACHE_MB");
                cap_ = (e ? static_cast<std::size_t>(std::atoll(e)) : std::size_t(24576)) << 20;
                DevicePool::pressure_hook() = [this] { clear(); };
            }
            struct Entry
            {
                std::shared_ptr<DeviceArray> src, out;
                std::size_t bytes;
                std::uint64_t tick;
            };
            std::mutex mu_;
            std::map<Key, Entry> map_;
            std::size_t cap_ = 0, bytes_ = 0;
            std::uint64_t clock_ = 0;
            std::atomic<std::uint64_t> hits_{ 0 }, misses_{ 0 };
        };
        // a flag that can be read without a lock and copied with its owner: "this value is still owed" (Ciphertext / Plaintext)
        struct CopyableFlag
        {
            std::atomic<bool> v{ false };
            CopyableFlag() = default;
            CopyableFlag(const CopyableFlag &o) : v(o.v.load(std::memory_order_acquire))
            {}
            CopyableFlag &operator=(const CopyableFlag &o)
            {
                v.store(o.v.load(std::memory_order_acquire), std::memory_order_release);
                return *this;
            }
        };
        inline std::mutex &lazy_mutex()
        {
            static std::mutex m;
            return m;
        }
        // Small host -> device copies whose source must be free again when the call returns (a constant, a few hundred of them):
        // through a per-thread ring of page-locked slots; a slot is reused only after the event recorded behind its last copy
        // has completed, so nobody synchronises the stream for eight bytes.
        inline void upload_small(void *dst, const void *src, std::size_t bytes, void *stream)
        {
            struct Slot
            {
                void *host = nullptr;
                std::size_t bytes = 0;
                void *event = nullptr;
                bool pending = false;
            };
            static thread_local Slot ring[8];
            static thread_local unsigned next = 0;
            Slot &sl = ring[next++ & 7u];
            if (sl.pending)
            {
                hip_check(moai_event_synchronize(sl.event));
                sl.pending = false;
            }
            if (!sl.event)
            {
                hip_check(moai_event_create(&sl.event));
            }
            if (sl.bytes < bytes)
            {
                if (sl.host)
                {
                    hip_check(moai_host_free(sl.host));
                }
                sl.bytes = std::max<std::size_t>(bytes, std::size_t(1) << 14);
                hip_check(moai_host_malloc(&sl.host, sl.bytes));
            }
            std::memcpy(sl.host, src, bytes);
            hip_check(moai_memcpy_h2d(dst, sl.host, bytes, stream));
            hip_check(moai_event_record(sl.event, stream));
            sl.pending = true;
        }
        // the 0/1 slot pattern of MOAI's masked plaintexts (its bias_vec, Batch_encode_encrypt.hpp:40-49) on host and device
        struct SlotMask
        {
            std::vector<std::int32_t> host;
            std::shared_ptr<DeviceArray> dev; // int32 [slots]
        };

        // A single-ciphertext rotation that has been asked for but not made: the chain of Galois elements still to apply to the
        // block `src` (one element for a step with a key of its own, several for a step that goes through the non-adjacent form).
        // MOAI's Q K^T loop asks for the rotation of 64 ciphertexts by the same step one call after the other and only then reads
        // the first of them (Ct_ct_matrix_mul.hpp:24-32): when one pending rotation has to be made, every other pending rotation
        // of the same thread with the same next element, level and key is made WITH it -- one batched key switch (0.12 ms per
        // ciphertext at l = 15 instead of 0.22) -- and the rotation cache is consulted and fed per member.  Each ciphertext gets
        // exactly the key switches it asked for; the batched kernels compute every member independently (same bits as single
        // calls, tests/test_gpu_parity.py).
        struct RotState
        {
            std::mutex mu;
            std::shared_ptr<DeviceArray> src;
            std::vector<std::uint32_t> elts;
            std::vector<std::shared_ptr<DeviceArray>> keys; // the device block of each element's key as it serves level L
            std::vector<std::pair<std::uint64_t, std::size_t>> key_ids; // (generation, index): the rotation cache's name for it
            std::size_t L = 0, n = 0;
            moai_ctx *dev = nullptr;
            void *stream = nullptr;
        };
        // how a rotation that found no company in its own thread is issued: the evaluator routes it through the call combiner
        // (moai_combiner.h), where concurrent callers of OTHER threads with the same element, level and key share a batched call
        inline std::function<void(moai_ctx *, const std::uint64_t *, std::uint64_t *, std::size_t, std::uint32_t, const std::uint64_t *, void *)> &
        single_rotation_hook()
        {
            static std::function<void(moai_ctx *, const std::uint64_t *, std::uint64_t *, std::size_t, std::uint32_t, const std::uint64_t *, void *)> h;
            return h;
        }
        inline std::vector<std::weak_ptr<RotState>> &rot_registry()
        {
            static thread_local std::vector<std::weak_ptr<RotState>> r;
            return r;
        }
        inline void rot_register(const std::shared_ptr<RotState> &st)
        {
            auto &r = rot_registry();
            if (r.size() >= 512)
            {
                r.erase(std::remove_if(r.begin(), r.end(),
                                       [](const std::weak_ptr<RotState> &w) {
                                           auto p = w.lock();
                                           return !p || p->elts.empty();
                                       }),
                        r.end());
                if (r.size() >= 512)
                {
                    r.erase(r.begin(), r.begin() + 256);
                }
            }
            r.push_back(st);
        }
        // applies every element of `me` (and, step by step, of the calling thread's other pending rotations that can share a
        // batched call with it); returns the block that holds me's result
        inline std::shared_ptr<DeviceArray> rot_resolve(const std::shared_ptr<RotState> &me)
        {
            std::unique_lock<std::mutex> mine(me->mu);
            RotationCache &cache = RotationCache::instance();
            while (!me->elts.empty())
            {
                const std::uint32_t e = me->elts.front();
                const DeviceArray *key = me->keys.front().get();
                // peers: pending rotations registered by this thread with the same next step
                std::vector<std::shared_ptr<RotState>> group{ me };
                std::vector<std::unique_lock<std::mutex>> held;
                for (auto &w : rot_registry())
                {
                    auto p = w.lock();
                    if (!p || p == me || group.size() >= 64)
                    {
                        continue;
                    }
                    std::unique_lock<std::mutex> lk(p->mu, std::try_to_lock);
                    if (lk.owns_lock() && !p->elts.empty() && p->elts.front() == e && p->keys.front().get() == key && p->L == me->L && p->dev == me->dev &&
                        p->n == me->n)
                    {
                        group.push_back(p);
                        held.push_back(std::move(lk));
                    }
                }
                const std::size_t words = 2 * me->L * me->n;
                // what the cache already holds needs no work
                std::vector<std::size_t> todo;
                std::vector<RotationCache::Key> ckeys(group.size());
                std::vector<std::shared_ptr<DeviceArray>> next(group.size()); // committed together below: a failure half-way changes nothing
                for (std::size_t i = 0; i < group.size(); i++)
                {
                    RotState &g = *group[i];
                    ckeys[i] = RotationCache::Key(g.src.get(), e, g.key_ids.front().first, g.key_ids.front().second, g.L);
                    next[i] = cache.enabled() ? cache.find(ckeys[i]) : nullptr;
                    if (!next[i])
                    {
                        todo.push_back(i);
                    }
                }
                if (todo.size() == 1)
                {
                    RotState &g = *group[todo[0]];
                    auto out = std::make_shared<DeviceArray>(words, g.stream);
                    if (single_rotation_hook())
                    {
                        single_rotation_hook()(g.dev, g.src->get(), out->get(), g.L, e, key->get(), g.stream);
                    }
                    else
                    {
                        hip_check(moai_apply_galois_to(g.dev, g.src->get(), out->get(), g.L, e, key->get(), 1, g.stream));
                    }
                    next[todo[0]] = out;
                }
                else if (todo.size() > 1)
                {
                    const std::size_t m = todo.size();
                    DeviceArray tmp(m * words, me->stream);
                    for (std::size_t t = 0; t < m; t++)
                    {
                        hip_check(moai_memcpy_d2d(tmp.get() + t * words, group[todo[t]]->src->get(), words * 8, me->stream));
                    }
                    hip_check(moai_apply_galois(me->dev, tmp.get(), me->L, e, key->get(), m, me->stream));
                    for (std::size_t t = 0; t < m; t++)
                    {
                        RotState &g = *group[todo[t]];
                        auto out = std::make_shared<DeviceArray>(words, g.stream);
                        hip_check(moai_memcpy_d2d(out->get(), tmp.get() + t * words, words * 8, me->stream));
                        next[todo[t]] = out;
                    }
                }
                for (std::size_t t : todo)
                {
                    if (cache.enabled())
                    {
                        cache.insert(ckeys[t], group[t]->src, next[t]);
                    }
                }
                for (std::size_t i = 0; i < group.size(); i++)
                {
                    RotState &g = *group[i];
                    g.src = next[i];
                    g.elts.erase(g.elts.begin());
                    g.keys.erase(g.keys.begin());
                    g.key_ids.erase(g.key_ids.begin());
                }
            }
            return me->src;
        }
    } // namespace util

    // =================================================================================================
    // Modulus / CoeffModulus  (SEAL/modulus.{h,cpp})
    // =================================================================================================
    class Modulus
    {
    public:
        Modulus(std::uint64_t value = 0)
        {
            set_value(value);
        }
        Modulus &operator=(std::uint64_t value)
        {
            set_value(value);
            return *this;
        }
        std::uint64_t value() const noexcept
        {
            return value_;
        }
        int bit_count() const noexcept
        {
            return bit_count_;
        }
        bool is_zero() const noexcept
        {
            return value_ == 0;
        }
        bool is_prime() const noexcept
        {
            return util::is_prime(value_);
        }
        bool operator==(const Modulus &o) const noexcept
        {
            return value_ == o.value_;
        }
        bool operator!=(const Modulus &o) const noexcept
        {
            return value_ != o.value_;
        }
        bool operator<(const Modulus &o) const noexcept
        {
            return value_ < o.value_;
        }

    private:
        void set_value(std::uint64_t value)
        {
            if (value != 0 && ((value >> 61) != 0 || value == 1))
            {
                throw std::invalid_argument("value can be at most 61-bit and cannot be 1");
            }
            value_ = value;
            bit_count_ = util::get_significant_bit_count(value);
        }
        std::uint64_t value_ = 0;
        int bit_count_ = 0;
    };

    class CoeffModulus
    {
    public:
        // SEAL/modulus.cpp:142-183: primes = 1 mod 2N, found descending from 2^bits, handed out
        // smallest-first within each bit size
        static std::vector<Modulus> Create(std::size_t poly_modulus_degree, std::vector<int> bit_sizes)
        {
            if (poly_modulus_degree > 131072 || poly_modulus_degree < 2 ||
                (poly_modulus_degree & (poly_modulus_degree - 1)) != 0)
            {
                throw std::invalid_argument("poly_modulus_degree is invalid");
            }
            if (bit_sizes.size() > 256)
            {
                throw std::invalid_argument("bit_sizes is invalid");
            }
            std::map<int, std::vector<std::uint64_t>> table;
            std::map<int, std::size_t> count;
            for (int b : bit_sizes)
            {
                if (b > 60 || b < 2)
                {
                    throw std::invalid_argument("bit_sizes is invalid");
                }
                ++count[b];
            }
            std::uint64_t factor = 2 * static_cast<std::uint64_t>(poly_modulus_degree);
            for (auto &kv : count)
            {
                int bits = kv.first;
                std::uint64_t value = ((std::uint64_t(1) << bits) - 1) / factor * factor + 1;
                std::uint64_t lower = std::uint64_t(1) << (bits - 1);
                std::vector<std::uint64_t> found;
                while (found.size() < kv.second && value > lower)
                {
                    if (util::is_prime(value))
                    {
                        found.push_back(value);
                    }
                    value -= factor;
                }
                if (found.size() < kv.second)
                {
                    throw std::logic_error("failed to find enough qualifying primes");
                }
                table[bits] = found;
            }
            std::vector<Modulus> result;
            for (int b : bit_sizes)
            {
                result.emplace_back(table[b].back());
                table[b].pop_back();
            }
            return result;
        }
    };

    // =================================================================================================
    // EncryptionParameters  (SEAL/encryptionparams.h)
    // =================================================================================================
    namespace util
    {
        // SEAL/util/iterator.h iter(...) as MOAI uses it on the modulus vector (include/source/bootstrapping/
        // Bootstrapper.cpp:2344, :2481, :3235: `iter(...coeff_modulus())[i].value()`): indexed read access.  The raw
        // residue iterators over a Ciphertext (only Bootstrapper::modraise_inplace, :2973-2975) are not offered: the
        // data lives on the device, and that routine is the C ABI's moai_modraise.
        inline const std::vector<Modulus> &iter(const std::vector<Modulus> &v) noexcept
        {
            return v;
        }
    } // namespace util

    class EncryptionParameters
    {
    public:
        EncryptionParameters(scheme_type scheme = scheme_type::none) : scheme_(scheme)
        {
            if (scheme != scheme_type::ckks && scheme != scheme_type::none)
            {
                throw std::invalid_argument("unsupported scheme (this build provides CKKS only)");
            }
        }
        void set_poly_modulus_degree(std::size_t n)
        {
            if (scheme_ == scheme_type::none && n)
            {
                throw std::logic_error("poly_modulus_degree is not supported for this scheme");
            }
            poly_modulus_degree_ = n;
        }
        void set_coeff_modulus(const std::vector<Modulus> &m)
        {
            if (scheme_ == scheme_type::none)
            {
                if (!m.empty())
                {
                    throw std::logic_error("coeff_modulus is not supported for this scheme");
                }
            }
            else if (m.size() > 256 || m.size() < 1)
            {
                throw std::invalid_argument("coeff_modulus is invalid");
            }
            coeff_modulus_ = m;
        }
        // fork additions (sparse ternary secret, sparse slot packing)
        void set_secret_key_hamming_weight(std::size_t hw)
        {
            secret_key_hamming_weight_ = hw;
        }
        void set_sparse_slots(std::size_t s)
        {
            sparse_slots_ = s;
        }
        scheme_type scheme() const noexcept
        {
            return scheme_;
        }
        std::size_t poly_modulus_degree() const noexcept
        {
            return poly_modulus_degree_;
        }
        const std::vector<Modulus> &coeff_modulus() const noexcept
        {
            return coeff_modulus_;
        }
        // BFV / BGV only (SEAL/encryptionparams.h:330-333): zero for CKKS, as in the reference
        const Modulus &plain_modulus() const noexcept
        {
            return plain_modulus_;
        }
        std::size_t secret_key_hamming_weight() const noexcept
        {
            return secret_key_hamming_weight_;
        }
        std::size_t sparse_slots() const noexcept
        {
            return sparse_slots_;
        }

    private:
        scheme_type scheme_;
        std::size_t poly_modulus_degree_ = 0;
        std::vector<Modulus> coeff_modulus_;
        Modulus plain_modulus_;
        std::size_t secret_key_hamming_weight_ = 0;
        std::size_t sparse_slots_ = 0;
    };

    // =================================================================================================
    // SEALContext  (SEAL/context.{h,cpp})
    // =================================================================================================
    class SEALContext
    {
    public:
        class ContextData
        {
        public:
            const EncryptionParameters &parms() const noexcept
            {
                return parms_;
            }
            const parms_id_type &parms_id() const noexcept
            {
                return parms_id_;
            }
            std::size_t chain_index() const noexcept
            {
                return chain_index_;
            }
            int total_coeff_modulus_bit_count() const noexcept
            {
                return total_bits_;
            }
            std::shared_ptr<const ContextData> prev_context_data() const noexcept
            {
                return prev_.lock();
            }
            std::shared_ptr<const ContextData> next_context_data() const noexcept
            {
                return next_;
            }

        private:
            friend class SEALContext;
            EncryptionParameters parms_;
            parms_id_type parms_id_ = parms_id_zero;
            std::size_t chain_index_ = 0;
            int total_bits_ = 0;
            std::weak_ptr<const ContextData> prev_;
            std::shared_ptr<const ContextData> next_;
        };

        SEALContext(const EncryptionParameters &parms, bool expand_mod_chain = true,
                    sec_level_type sec_level = sec_level_type::tc128)
        {
            (void)sec_level; // MOAI runs with sec_level_type::none (test_full_scheme.hpp:389)
            if (parms.scheme() != scheme_type::ckks)
            {
                throw std::invalid_argument("only the CKKS scheme is provided");
            }
            const std::size_t n = parms.poly_modulus_degree();
            int logn = 0;
            while ((std::size_t(1) << logn) < n)
            {
                logn++;
            }
            if (n < 2 || (std::size_t(1) << logn) != n)
            {
                throw std::invalid_argument("poly_modulus_degree is invalid");
            }
            const auto &cm = parms.coeff_modulus();
            std::vector<std::uint64_t> primes;
            for (auto &m : cm)
            {
                primes.push_back(m.value());
            }
            impl_ = std::make_shared<Impl>();
            impl_->logn = logn;
            impl_->n = n;
            moai_ctx *c = nullptr;
            util::hip_check(moai_ctx_create(logn, primes.data(), primes.size(), 0, &c));
            impl_->ctx = c;
            util::hip_check(moai_stream_create(&impl_->stream));
            const std::size_t k = cm.size();
            // chain: key level (k primes), then data levels with k-1 ... 1 primes
            std::shared_ptr<ContextData> prev;
            std::size_t levels = expand_mod_chain ? k : (k > 1 ? 2 : 1);
            for (std::size_t lvl = 0; lvl < levels; lvl++)
            {
                std::size_t count = k - lvl;
                if (count < 1)
                {
                    break;
                }
                auto cd = std::make_shared<ContextData>();
                EncryptionParameters p(parms);
                std::vector<Modulus> sub(cm.begin(), cm.begin() + static_cast<std::ptrdiff_t>(count));
                p.set_coeff_modulus(sub);
                cd->parms_ = p;
                cd->chain_index_ = count - 1;
                int bits = 0;
                std::uint64_t x = 0x4d4f4149ull;
                for (auto &m : sub)
                {
                    bits += m.bit_count();
                    x = (x ^ m.value()) * 0x9E3779B97F4A7C15ull;
                }
                cd->total_bits_ = bits;
                cd->parms_id_ = { 0x4d4f41495345414cull, static_cast<std::uint64_t>(n), static_cast<std::uint64_t>(count), x };
                if (prev)
                {
                    cd->prev_ = prev;
                    prev->next_ = cd;
                }
                impl_->by_id[cd->parms_id_] = cd;
                impl_->by_count[count] = cd;
                if (lvl == 0)
                {
                    impl_->key = cd;
                }
                if ((lvl == 1) || (k == 1 && lvl == 0))
                {
                    impl_->first = cd;
                }
                impl_->last = cd;
                prev = cd;
            }
            if (!impl_->first)
            {
                impl_->first = impl_->key;
            }
        }

        std::shared_ptr<const ContextData> get_context_data(const parms_id_type &id) const
        {
            auto it = impl_->by_id.find(id);
            return it == impl_->by_id.end() ? nullptr : it->second;
        }
        std::shared_ptr<const ContextData> key_context_data() const
        {
            return impl_->key;
        }
        std::shared_ptr<const ContextData> first_context_data() const
        {
            return impl_->first;
        }
        std::shared_ptr<const ContextData> last_context_data() const
        {
            return impl_->last;
        }
        const parms_id_type &key_parms_id() const
        {
            return impl_->key->parms_id();
        }
        const parms_id_type &first_parms_id() const
        {
            return impl_->first->parms_id();
        }
        const parms_id_type &last_parms_id() const
        {
            return impl_->last->parms_id();
        }
        bool using_keyswitching() const
        {
            return impl_->key->parms().coeff_modulus().size() > 1;
        }
        bool parameters_set() const
        {
            return true;
        }

        // ---- device side (not part of the reference API) -------------------------------------------
        moai_ctx *device() const
        {
            return impl_->ctx;
        }
        void *stream() const
        {
            return impl_->stream;
        }
        void sync() const
        {
            util::hip_check(moai_stream_sync(impl_->stream));
        }
        std::size_t n() const
        {
            return impl_->n;
        }
        int logn() const
        {
            return impl_->logn;
        }
        std::shared_ptr<const ContextData> data_level(std::size_t prime_count) const
        {
            auto it = impl_->by_count.find(prime_count);
            return it == impl_->by_count.end() ? nullptr : it->second;
        }

    private:
        struct Impl
        {
            moai_ctx *ctx = nullptr;
            void *stream = nullptr;
            int logn = 0;
            std::size_t n = 0;
            std::map<parms_id_type, std::shared_ptr<ContextData>> by_id;
            std::map<std::size_t, std::shared_ptr<ContextData>> by_count;
            std::shared_ptr<ContextData> key, first, last;
            ~Impl()
            {
                if (stream)
                {
                    moai_stream_sync(stream);
                    moai_stream_destroy(stream);
                }
                if (ctx)
                {
                    moai_ctx_destroy(ctx);
                }
            }
        };
        std::shared_ptr<Impl> impl_;
    };

    // =================================================================================================
    // Plaintext / Ciphertext
    // =================================================================================================
    // A CKKS plaintext is kept in NTT form over the primes of its level, [L][N] on the device.  A
    // plaintext produced by the scalar encode overloads has constant rows (SEAL/ckks.cpp:131-150); it is
    // kept as L scalars and never materialised, so multiply_plain / add_plain become one scalar op per row.
    class Plaintext
    {
    public:
        Plaintext(MemoryPoolHandle = MemoryPoolHandle())
        {}
        Plaintext(const Plaintext &o)
        {
            *this = o;
        }
        Plaintext(Plaintext &&) = default;
        Plaintext &operator=(Plaintext &&) = default;
        Plaintext &operator=(const Plaintext &o)
        {
            if (this == &o)
            {
                return *this;
            }
            parms_id_ = o.parms_id_;
            scale_ = o.scale_;
            n_ = o.n_;
            L_ = o.L_;
            scalar_rows_ = o.scalar_rows_;
            stream_ = o.stream_;
            dev_ = o.dev_;
            {
                std::lock_guard<std::mutex> g(util::lazy_mutex());
                mask_ = o.mask_;
                mask_c_ = o.mask_c_;
                mask_scale_ = o.mask_scale_;
                mask_L_ = o.mask_L_;
                owed_.v.store(static_cast<bool>(mask_), std::memory_order_release);
            }
            if (mask_)
            {
                data_.release();
                return *this;
            }
            data_.resize(o.data_.size(), stream_);
            if (o.data_.size())
            {
                util::hip_check(moai_memcpy_d2d(data_.get(), o.data_.get(), o.data_.size() * 8, stream_));
            }
            return *this;
        }
        parms_id_type &parms_id() noexcept
        {
            return parms_id_;
        }
        const parms_id_type &parms_id() const noexcept
        {
            return parms_id_;
        }
        double &scale() noexcept
        {
            return scale_;
        }
        const double &scale() const noexcept
        {
            return scale_;
        }
        bool is_ntt_form() const noexcept
        {
            return parms_id_ != parms_id_zero;
        }
        std::size_t coeff_count() const noexcept
        {
            return n_ * L_;
        }
        bool is_zero() const
        {
            return coeff_count() == 0;
        }
        // ---- device side ---------------------------------------------------------------------------
        bool is_scalar() const
        {
            return !scalar_rows_.empty();
        }
        const std::vector<std::uint64_t> &scalar_rows() const
        {
            return scalar_rows_;
        }
        std::uint64_t *device_data() const
        {
            materialize();
            return data_.get();
        }
        std::size_t coeff_modulus_size() const
        {
            return L_;
        }
        // a vector plaintext of the form  c * (0/1 mask)  whose residues have not been produced yet (see `mask_`)
        bool is_masked_constant() const
        {
            return owed_.v.load(std::memory_order_acquire);
        }

    private:
        friend class CKKSEncoder;
        friend class Evaluator;
        friend class Decryptor;
        friend class Ciphertext;
        // MOAI's masked matrix products encode every weight as the vector  w * bias_vec  (Ct_pt_matrix_mul.hpp:124-146): 2.4 M
        // FP64 transforms per layer that each serve ONE multiply_plain.  CKKSEncoder::encode recognises such a vector and records
        // (mask, constant, scale) instead of transforming; a product with it is recorded in turn (Ciphertext::LazyTerm), and the
        // transforms of a whole chain are made by one moai_ckks_encode_masked call when the sum is needed -- the same residues as the
        // single encodes (tests/test_gpu_encoder.py), so the same bits.  Anything else that reads the plaintext produces it here.
        void materialize() const
        {
            if (!owed_.v.load(std::memory_order_acquire))
            {
                return;
            }
            std::lock_guard<std::mutex> g(util::lazy_mutex());
            if (!mask_)
            {
                owed_.v.store(false, std::memory_order_release);
                return;
            }
            const std::size_t full_L = mask_L_;
            data_.resize(full_L * n_, stream_);
            util::DeviceArray staging(2, stream_);
            const double c = mask_c_;
            util::upload_small(staging.get(), &c, 8, stream_);
            util::hip_check(moai_ckks_encode_masked(dev_, reinterpret_cast<const double *>(staging.get()),
                                                    reinterpret_cast<const std::int32_t *>(mask_->dev->get()), mask_->host.size(), 1, data_.get(),
                                                    full_L, nullptr, mask_scale_, nullptr, stream_));
            mask_keep_ = mask_; // the kernel reads the mask's device copy in stream order; this object may outlive its last other owner
            mask_.reset();
            owed_.v.store(false, std::memory_order_release);
        }
        parms_id_type parms_id_ = parms_id_zero;
        double scale_ = 1.0;
        std::size_t n_ = 0, L_ = 0;
        std::vector<std::uint64_t> scalar_rows_;
        mutable util::DeviceArray data_;
        mutable std::shared_ptr<const util::SlotMask> mask_; // set: the residues are owed
        mutable std::shared_ptr<const util::SlotMask> mask_keep_;
        mutable double mask_c_ = 0, mask_scale_ = 0;         // the constant, and the scale the vector was encoded at
        std::size_t mask_L_ = 0;                               // primes at encode time (a later mod switch only drops rows)
        mutable util::CopyableFlag owed_;
        moai_ctx *dev_ = nullptr;
        void *stream_ = nullptr;
    };

    class Ciphertext
    {
    public:
        Ciphertext(MemoryPoolHandle = MemoryPoolHandle())
        {}
        explicit Ciphertext(const SEALContext &context, MemoryPoolHandle = MemoryPoolHandle())
        {
            resize(context, context.first_parms_id(), 2);
        }
        Ciphertext(const Ciphertext &o)
        {
            *this = o;
        }
        Ciphertext(Ciphertext &&) = default;
        Ciphertext &operator=(Ciphertext &&) = default;
        // A copy has the value semantics of the reference's deep copy (SEAL/ciphertext.cpp:16-37) without its cost: the two
        // objects share one immutable device block until either is written through -- device_data() on a non-const object,
        // resize, upload -- which gives the writer a private copy first (copy on write).  MOAI copies ciphertexts freely
        // (`copy_w[j] = enc_W[j]` 8192 times per Q K^T, Ct_ct_matrix_mul.hpp:26; `vector<Ciphertext> c_g(g, enc_W[i])`, :108).
        Ciphertext &operator=(const Ciphertext &o)
        {
            if (this == &o)
            {
                return *this;
            }
            parms_id_ = o.parms_id_;
            is_ntt_form_ = o.is_ntt_form_;
            size_ = o.size_;
            batch_ = o.batch_;
            n_ = o.n_;
            L_ = o.L_;
            scale_ = o.scale_;
            stream_ = o.stream_;
            dev_ = o.dev_;
            buf_ = o.words() ? o.buf_ : nullptr;
            lazy_ = o.lazy_;
            rot_ = o.rot_;
            deferred_.v.store(lazy_ || rot_, std::memory_order_release);
            return *this;
        }
        void resize(const SEALContext &context, parms_id_type parms_id, std::size_t size)
        {
            resize_batch(context, parms_id, size, 1);
        }
        void resize_batch(const SEALContext &context, parms_id_type parms_id, std::size_t size, std::size_t batch)
        {
            if (batch < 1)
            {
                throw std::invalid_argument("invalid batch");
            }
            auto cd = context.get_context_data(parms_id);
            if (!cd)
            {
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            }
            if ((size < 2 && size != 0) || size > 6)
            {
                throw std::invalid_argument("invalid size");
            }
            materialize();
            stream_ = context.stream();
            dev_ = context.device();
            parms_id_ = parms_id;
            n_ = cd->parms().poly_modulus_degree();
            L_ = cd->parms().coeff_modulus().size();
            const std::size_t old_words = words();
            size_ = size;
            batch_ = batch;
            reshape(old_words);
        }
        void release()
        {
            buf_.reset();
            lazy_.reset();
            rot_.reset();
            deferred_.v.store(false, std::memory_order_release);
            size_ = 0;
            batch_ = 1;
            parms_id_ = parms_id_zero;
        }
        // ---- packed ciphertexts (moai_fused::pack / unpack, seal/moai_fused.h) -----------------------------
        // A Ciphertext may hold `batch` ciphertexts of one size, level and scale back to back,
        // [batch][size][L][N].  Every Evaluator method then performs its operation on each of them -- with the
        // device's batched kernels instead of `batch` separate calls -- so MOAI's per-ciphertext routines
        // (gelu_v2, exp, invert_sqrt, ...) process a whole batch when handed a packed ciphertext, unchanged.
        // batch() is 1 for everything the reference API creates.
        std::size_t batch() const noexcept
        {
            return batch_;
        }

        std::size_t size() const noexcept
        {
            return size_;
        }
        std::size_t coeff_modulus_size() const noexcept
        {
            return L_;
        }
        std::size_t poly_modulus_degree() const noexcept
        {
            return n_;
        }
        bool &is_ntt_form() noexcept
        {
            return is_ntt_form_;
        }
        bool is_ntt_form() const noexcept
        {
            return is_ntt_form_;
        }
        parms_id_type &parms_id() noexcept
        {
            return parms_id_;
        }
        const parms_id_type &parms_id() const noexcept
        {
            return parms_id_;
        }
        double &scale() noexcept
        {
            return scale_;
        }
        const double &scale() const noexcept
        {
            return scale_;
        }
        // ---- device side ---------------------------------------------------------------------------
        // reading: the block as it is (possibly shared with copies)
        const std::uint64_t *device_data() const
        {
            materialize();
            return buf_ ? buf_->get() : nullptr;
        }
        // writing (any access through a non-const object counts): a private block
        std::uint64_t *device_data()
        {
            materialize();
            unshare();
            return buf_ ? buf_->get() : nullptr;
        }
        // identity of the block behind a value: two ciphertexts with the same block hold the same residues (used as a cache key)
        const void *block_id() const
        {
            materialize();
            return buf_.get();
        }
        // has residues, on the device or still owed (see `lazy_`)
        bool has_value() const
        {
            return deferred_.v.load(std::memory_order_acquire) || buf_;
        }
        bool is_deferred() const
        {
            return deferred_.v.load(std::memory_order_acquire);
        }
        // host copy of the residues [size][L][N] (the reference exposes data(); MOAI itself only needs it
        // inside Bootstrapper::modraise_inplace, which maps to moai_modraise)
        std::vector<std::uint64_t> download() const
        {
            std::vector<std::uint64_t> h(batch_ * size_ * L_ * n_);
            if (!h.empty())
            {
                util::hip_check(moai_memcpy_d2h(h.data(), device_data(), h.size() * 8, stream_));
                util::hip_check(moai_stream_sync(stream_));
            }
            return h;
        }
        void upload(const std::vector<std::uint64_t> &h)
        {
            if (h.size() != batch_ * size_ * L_ * n_)
            {
                throw std::invalid_argument("size mismatch");
            }
            util::hip_check(moai_memcpy_h2d(device_data(), h.data(), h.size() * 8, stream_));
            util::hip_check(moai_stream_sync(stream_));
        }

    private:
        friend class Evaluator;
        friend class Encryptor;
        friend class Decryptor;
        friend class KeyGenerator;
        // metadata-only change of level / size after the device op produced the new layout
        void set_layout(void *stream, parms_id_type id, std::size_t size, std::size_t L, std::size_t n)
        {
            stream_ = stream;
            parms_id_ = id;
            size_ = size;
            L_ = L;
            n_ = n;
        }
        std::size_t words() const
        {
            return batch_ * size_ * L_ * n_;
        }
        // ---- deferred scalar products -------------------------------------------------------------------------------------
        // MOAI's column-packed ct x pt product is, per output column, 768 (or 3072) times
        //     encoder.encode(w, ...); evaluator.multiply_plain(X[j], pt, temp); evaluator.add_inplace(out, temp);
        // (Ct_pt_matrix_mul.hpp:19-42), one kernel and one temporary per call when executed as written.  A product with a
        // SCALAR-encoded plaintext is therefore not computed when it is asked for: the destination records (block of X[j], the
        // plaintext's constant rows); add_inplace of such a ciphertext appends its terms; the residues are produced when
        // someone needs them (device_data(), i.e. any other operation) by moai_scalar_dot, sixteen terms per pass.
        // value = residues of buf_ (if any) + sum of the terms.  The recorded blocks cannot change under the record (copy on
        // write), so the value is the one the eager sequence gives, bit for bit.  MOAI_SHIM_LAZY=0 computes eagerly.
        struct LazyTerm
        {
            std::shared_ptr<util::DeviceArray> src;
            std::vector<std::uint64_t> scalars;         // [L]: a scalar-encoded plaintext's constant rows, or empty:
            std::shared_ptr<const util::SlotMask> mask; // a masked-constant vector plaintext (Plaintext::mask_), or unset:
            double c = 0, pscale = 0;
            std::shared_ptr<util::DeviceArray> src2;    // a second size-2 CIPHERTEXT: the term is multiply(src, src2), size 3
        };
        static std::mutex &lazy_mutex()
        {
            return util::lazy_mutex();
        }
        typedef util::CopyableFlag Flag;
        void materialize() const
        {
            if (!deferred_.v.load(std::memory_order_acquire))
            {
                return;
            }
            // a pending rotation (util::RotState): made now, together with the thread's other pending rotations of the same step.
            // Not under the lock below: a rotation made alone waits in the call combiner for callers of OTHER threads, who must be
            // able to get here; the state has a lock of its own (a second reader of this object waits there and finds the result).
            std::shared_ptr<util::RotState> pending;
            {
                std::lock_guard<std::mutex> g(lazy_mutex());
                pending = rot_;
            }
            if (pending)
            {
                std::shared_ptr<util::DeviceArray> block = util::rot_resolve(pending);
                std::lock_guard<std::mutex> g(lazy_mutex());
                if (rot_ == pending)
                {
                    buf_ = block;
                    rot_.reset();
                    deferred_.v.store(static_cast<bool>(lazy_), std::memory_order_release);
                }
            }
            std::lock_guard<std::mutex> g(lazy_mutex());
            if (rot_)
            {
                return; // replaced meanwhile by the owner (not a const use): the owner's business
            }
            if (!lazy_)
            {
                deferred_.v.store(false, std::memory_order_release);
                return;
            }
            const std::size_t w = words();
            std::shared_ptr<util::DeviceArray> out = buf_;
            if (!out || out.use_count() > 2) // shared with another ciphertext (`out` itself is the second owner): not ours to write
            {
                out = std::make_shared<util::DeviceArray>(w, stream_);
            }
            // runs of terms of one kind: scalar plaintexts -> moai_scalar_dot; masked-constant vector plaintexts of one mask and
            // scale -> their transforms in one moai_ckks_encode_masked call per chunk, then moai_vector_dot
            const std::uint64_t *base = buf_ ? buf_->get() : nullptr;
            const std::vector<LazyTerm> &terms = *lazy_;
            for (std::size_t i = 0; i < terms.size();)
            {
                std::vector<const std::uint64_t *> ptrs;
                std::size_t j = i;
                if (terms[i].src2)
                {
                    std::vector<const std::uint64_t *> ptrs2;
                    for (; j < terms.size() && terms[j].src2; j++)
                    {
                        ptrs.push_back(terms[j].src->get());
                        ptrs2.push_back(terms[j].src2->get());
                    }
                    util::hip_check(moai_ct_dot_ptrs(dev_, ptrs.data(), ptrs2.data(), ptrs.size(), base, out->get(), L_, stream_));
                }
                else if (!terms[i].mask)
                {
                    std::vector<std::uint64_t> sc;
                    for (; j < terms.size() && !terms[j].mask && !terms[j].src2; j++)
                    {
                        ptrs.push_back(terms[j].src->get());
                        sc.insert(sc.end(), terms[j].scalars.begin(), terms[j].scalars.end());
                    }
                    util::hip_check(moai_scalar_dot(dev_, ptrs.data(), sc.data(), ptrs.size(), base, out->get(), size_, L_, stream_));
                }
                else
                {
                    const std::size_t chunk = std::max<std::size_t>(16, (std::size_t(1) << 30) / (L_ * n_ * 8));
                    std::vector<double> cs;
                    for (; j < terms.size() && j - i < chunk && !terms[j].src2 && terms[j].mask == terms[i].mask && terms[j].pscale == terms[i].pscale; j++)
                    {
                        ptrs.push_back(terms[j].src->get());
                        cs.push_back(terms[j].c);
                    }
                    const std::size_t T = ptrs.size();
                    util::DeviceArray dconst(T, stream_), dP(T * L_ * n_, stream_);
                    util::upload_small(dconst.get(), cs.data(), T * 8, stream_);
                    util::hip_check(moai_ckks_encode_masked(dev_, reinterpret_cast<const double *>(dconst.get()),
                                                            reinterpret_cast<const std::int32_t *>(terms[i].mask->dev->get()),
                                                            terms[i].mask->host.size(), T, dP.get(), L_, nullptr, terms[i].pscale, nullptr, stream_));
                    util::hip_check(moai_vector_dot(dev_, ptrs.data(), dP.get(), T, base, out->get(), size_, L_, stream_));
                }
                base = out->get();
                i = j;
            }
            buf_ = out;
            lazy_.reset(); // the terms' blocks are released behind the kernel that read them (stream order)
            deferred_.v.store(false, std::memory_order_release);
        }
        // a private block for this object: copies the residues over when the block is shared
        void unshare()
        {
            if (buf_ && buf_.use_count() > 1)
            {
                const std::size_t w = words();
                auto fresh = std::make_shared<util::DeviceArray>(w, stream_);
                if (w)
                {
                    util::hip_check(moai_memcpy_d2d(fresh->get(), buf_->get(), w * 8, stream_));
                }
                buf_ = fresh; // the old block lives on with its other owners; stream order protects the copy
            }
        }
        // after the metadata changed: a block of the new size that keeps the leading residues (DynArray::resize, SEAL/dynarray.h)
        void reshape(std::size_t old_words)
        {
            const std::size_t w = words();
            if (!buf_)
            {
                buf_ = std::make_shared<util::DeviceArray>(w, stream_);
            }
            else if (buf_.use_count() > 1)
            {
                auto fresh = std::make_shared<util::DeviceArray>(w, stream_);
                const std::size_t keep = std::min(old_words, w);
                if (keep)
                {
                    util::hip_check(moai_memcpy_d2d(fresh->get(), buf_->get(), keep * 8, stream_));
                }
                buf_ = fresh;
            }
            else
            {
                buf_->resize(w, stream_);
            }
        }
        parms_id_type parms_id_ = parms_id_zero;
        bool is_ntt_form_ = false;
        std::size_t size_ = 0, n_ = 0, L_ = 0;
        std::size_t batch_ = 1;
        double scale_ = 1.0;
        mutable std::shared_ptr<util::DeviceArray> buf_;
        mutable std::shared_ptr<std::vector<LazyTerm>> lazy_;
        mutable std::shared_ptr<util::RotState> rot_; // a pending rotation of rot_->src (then buf_ is empty and lazy_ unset)
        mutable Flag deferred_; // lazy_ or rot_ is set
        moai_ctx *dev_ = nullptr;
        void *stream_ = nullptr;
    };

    // =================================================================================================
    // keys
    // =================================================================================================
    class SecretKey
    {
    public:
        const parms_id_type &parms_id() const
        {
            return parms_id_;
        }

    private:
        friend class KeyGenerator;
        friend class Decryptor;
        friend class Encryptor;
        parms_id_type parms_id_ = parms_id_zero;
        std::shared_ptr<util::DeviceArray> ntt_; // [k][N], NTT form at the key level
    };

    class PublicKey
    {
    public:
        const parms_id_type &parms_id() const
        {
            return ct_.parms_id();
        }
        const Ciphertext &data() const
        {
            return ct_;
        }

    private:
        friend class KeyGenerator;
        friend class Encryptor;
        Ciphertext ct_;
    };

    // one entry = the reference's vector<PublicKey> of k-1 size-2 key-level ciphertexts, flattened to
    // uint64[k-1][2][k][N] on the device (SEAL/kswitchkeys.h:340)
    class KSwitchKeys
    {
    public:
        const parms_id_type &parms_id() const
        {
            return parms_id_;
        }
        parms_id_type &parms_id()
        {
            return parms_id_;
        }
        std::size_t size() const
        {
            std::size_t c = 0;
            for (auto &k : keys_)
            {
                c += k ? 1 : 0;
            }
            return c;
        }
        const std::uint64_t *device_key(std::size_t index) const
        {
            if (index >= keys_.size() || !keys_[index])
            {
                return nullptr;
            }
            if (res_)
            {
                std::lock_guard<std::mutex> g(res_->mu);
                if (index < res_->regrown.size() && res_->regrown[index])
                {
                    return res_->regrown[index]->get();
                }
            }
            return keys_[index]->get();
        }
        // the key for a switch at L data primes: the same pointer unless the key was trimmed below L (limit_to_chain_index),
        // in which case the full key comes back from its host copy first
        const std::uint64_t *device_key(std::size_t index, std::size_t L) const
        {
            auto b = key_block(index, L);
            return b ? b->get() : nullptr;
        }
        // the same as an owner of the block (a deferred rotation keeps its key alive until it is made)
        std::shared_ptr<util::DeviceArray> key_block(std::size_t index, std::size_t L) const;

        // ---- level-trimmed residency (not part of the reference API) ---------------------------------------------------
        // A key switch at l data primes reads the digits J < l and the rows {0 .. l-1, special prime} of a key
        // (SEAL/evaluator.cpp:2818, 2831); the reference keeps all of every key resident (1.32 GB each at MOAI's parameters).
        // limit_to_chain_index keeps on the device only what ciphertexts at chain index <= `chain_index` need (moai_key_trim:
        // (c+1) (c+2) / (35 * 36) of the key) and, unless told otherwise, parks the full key in host memory; a later switch at
        // a higher level brings the full key back (device_key(index, L)), so results never depend on the call.  MOAI's 31
        // default rotation keys serve Q K^T and softmax . V at chain index <= 14 only (Ct_ct_matrix_mul.hpp:29,95,112,147).
        void limit_to_chain_index(const SEALContext &context, std::size_t chain_index, bool keep_host_copy = true);
        std::size_t device_bytes() const
        {
            std::size_t b = 0;
            for (std::size_t i = 0; i < keys_.size(); i++)
            {
                if (!keys_[i])
                {
                    continue;
                }
                std::shared_ptr<util::DeviceArray> cur = keys_[i];
                if (res_)
                {
                    std::lock_guard<std::mutex> g(res_->mu);
                    if (i < res_->regrown.size() && res_->regrown[i])
                    {
                        cur = res_->regrown[i];
                    }
                }
                b += cur->size() * sizeof(std::uint64_t);
            }
            return b;
        }
        // how many trimmed keys had to come back whole so far
        std::size_t regrown_count() const
        {
            if (!res_)
            {
                return 0;
            }
            std::lock_guard<std::mutex> g(res_->mu);
            std::size_t c = 0;
            for (auto &r : res_->regrown)
            {
                c += r ? 1 : 0;
            }
            return c;
        }
        // the per-(key, level) constant of hoisted rotations (moai_hoist_correction), computed on first use and kept for
        // the lifetime of the key object; [2][L+1][N] on the device
        const std::uint64_t *hoist_correction(const SEALContext &context, std::size_t index, std::uint32_t galois_elt, std::size_t L) const;

        // changes whenever the key material does (KeyGenerator::create_*_keys): caches of results computed WITH a key name it
        std::uint64_t generation() const
        {
            return generation_;
        }

    protected:
        friend class KeyGenerator;
        static std::uint64_t next_generation()
        {
            static std::atomic<std::uint64_t> counter{ 1 };
            return counter.fetch_add(1);
        }
        std::uint64_t generation_ = 0;
        parms_id_type parms_id_ = parms_id_zero;
        mutable std::vector<std::shared_ptr<util::DeviceArray>> keys_;
        struct HoistCache
        {
            std::mutex mu;
            std::map<std::pair<std::size_t, std::size_t>, std::shared_ptr<util::DeviceArray>> blocks;
        };
        mutable std::shared_ptr<HoistCache> hoist_ = std::make_shared<HoistCache>();
        struct Residency
        {
            std::mutex mu;
            std::unique_ptr<SEALContext> context;                        // keeps the device context alive for the records below
            std::vector<std::size_t> levels;                             // data primes a trimmed key serves (0 = not trimmed)
            std::vector<std::shared_ptr<std::vector<std::uint64_t>>> host; // the full key, parked
            std::vector<std::shared_ptr<util::DeviceArray>> regrown;     // the full key, back on the device
        };
        std::shared_ptr<Residency> res_;
    };

    class RelinKeys : public KSwitchKeys
    {
    public:
        static std::size_t get_index(std::size_t key_power)
        {
            if (key_power < 2)
            {
                throw std::invalid_argument("key_power cannot be less than 2");
            }
            return key_power - 2;
        }
        bool has_key(std::size_t key_power) const
        {
            return device_key(get_index(key_power)) != nullptr;
        }
    };

    class GaloisKeys : public KSwitchKeys
    {
    public:
        static std::size_t get_index(std::uint32_t galois_elt)
        {
            if (!(galois_elt & 1))
            {
                throw std::invalid_argument("galois_elt is not valid");
            }
            return (galois_elt - 1) >> 1; // SEAL/galoiskeys.h:48
        }
        bool has_key(std::uint32_t galois_elt) const
        {
            return device_key(get_index(galois_elt)) != nullptr;
        }
    };
    inline void KSwitchKeys::limit_to_chain_index(const SEALContext &context, std::size_t chain_index, bool keep_host_copy)
    {
        const std::size_t k = context.key_context_data()->parms().coeff_modulus().size(), n = context.n();
        const std::size_t levels = chain_index + 1;
        if (k < 2 || levels >= k - 1)
        {
            return; // nothing to drop
        }
        if (!res_)
        {
            res_ = std::make_shared<Residency>();
            res_->context.reset(new SEALContext(context));
        }
        context.sync();
        std::lock_guard<std::mutex> g(res_->mu);
        res_->levels.resize(keys_.size(), 0);
        res_->host.resize(keys_.size());
        res_->regrown.resize(keys_.size());
        const std::size_t full_words = (k - 1) * 2 * k * n;
        for (std::size_t i = 0; i < keys_.size(); i++)
        {
            if (!keys_[i] || res_->regrown[i] || (res_->levels[i] && res_->levels[i] <= levels))
            {
                continue; // absent, already whole again (it was needed), or already trimmed at least this far
            }
            if (res_->levels[i])
            {
                continue; // trimmed to more levels than asked: cutting further needs the full key; leave it
            }
            if (keys_[i]->size() != full_words)
            {
                throw std::logic_error("limit_to_chain_index: key is not in the reference's layout");
            }
            if (keep_host_copy)
            {
                auto h = std::make_shared<std::vector<std::uint64_t>>(full_words);
                util::hip_check(moai_memcpy_d2h(h->data(), keys_[i]->get(), full_words * 8, context.stream()));
                context.sync();
                res_->host[i] = h;
            }
            // the library keeps the layout of the trimmed block by its address: forget it before the block goes back to the pool
            std::shared_ptr<SEALContext> keep(new SEALContext(context));
            std::shared_ptr<util::DeviceArray> t(new util::DeviceArray(moai_key_words(context.device(), levels), context.stream()),
                                                 [keep](util::DeviceArray *p) {
                                                     moai_key_forget(keep->device(), p->get());
                                                     delete p;
                                                 });
            util::hip_check(moai_key_trim(context.device(), keys_[i]->get(), levels, t->get(), context.stream()));
            context.sync(); // the full block is released behind the copy
            keys_[i] = t;
            res_->levels[i] = levels;
        }
    }
    inline std::shared_ptr<util::DeviceArray> KSwitchKeys::key_block(std::size_t index, std::size_t L) const
    {
        if (index >= keys_.size() || !keys_[index])
        {
            return nullptr;
        }
        if (!res_)
        {
            return keys_[index];
        }
        std::lock_guard<std::mutex> g(res_->mu);
        if (index < res_->regrown.size() && res_->regrown[index])
        {
            return res_->regrown[index];
        }
        if (index >= res_->levels.size() || res_->levels[index] == 0 || L <= res_->levels[index])
        {
            return keys_[index];
        }
        // a switch above the level the key was trimmed to: the full key comes back.  The trimmed block stays where it is
        // (callers on other threads may hold its address); it is dropped with the key object.
        if (!res_->host[index])
        {
            throw std::logic_error("key was trimmed to " + std::to_string(res_->levels[index]) + " levels without a host copy and is asked for " +
                                   std::to_string(L));
        }
        const SEALContext &context = *res_->context;
        auto full = std::make_shared<util::DeviceArray>(res_->host[index]->size(), context.stream());
        util::hip_check(moai_memcpy_h2d(full->get(), res_->host[index]->data(), res_->host[index]->size() * 8, context.stream()));
        context.sync();
        res_->regrown[index] = full;
        return full;
    }
    inline const std::uint64_t *KSwitchKeys::hoist_correction(const SEALContext &context, std::size_t index, std::uint32_t galois_elt,
                                                              std::size_t L) const
    {
        const std::uint64_t *key = device_key(index, L);
        if (!key)
        {
            throw std::invalid_argument("Galois key not present");
        }
        std::lock_guard<std::mutex> g(hoist_->mu);
        auto &slot = hoist_->blocks[std::make_pair(index, L)];
        if (!slot)
        {
            auto block = std::make_shared<util::DeviceArray>(2 * (L + 1) * context.n(), context.stream());
            util::hip_check(moai_hoist_correction(context.device(), key, galois_elt, L, block->get(), context.stream()));
            slot = block;
        }
        return slot->get();
    }
} // namespace seal

#include "seal/moai_combiner.h"
#include "seal/moai_client.h"
#include "seal/moai_evaluator.h"
