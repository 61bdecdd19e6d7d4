// seal/moai_combiner.h -- coalesces concurrent single-ciphertext key switches into batched device calls
// ("collect the per-thread rotate_vector calls by Galois element").
//
// MOAI calls the evaluator from OpenMP loops in which every thread runs the same sequence of operations on
// its own ciphertext (include/source/matrix_mul/Ct_ct_matrix_mul.hpp:22-49, the bootstrapping loops of
// include/test/test_full_scheme.hpp:654-660).  One key switch of one ciphertext leaves much of the GPU idle
// (1.06 ms at l = 35 against 0.62 ms per ciphertext in a batch; 0.35 against 0.15 ms at l = 15).  Calls that
// arrive within a short window with the same operation, level and key are therefore executed as ONE batched call:
// the first caller becomes the group's leader, waits up to `window` for the number of threads recently seen
// calling (never when only one thread is calling), then gathers the members' ciphertexts into one buffer, issues
// the batched call and scatters the results; the other members block until that is enqueued.  Every ciphertext
// gets exactly the operation it asked for, so results do not change; only the grouping of launches does.
// All waits are bounded; a leader that keeps waiting in vain stops waiting for a while.
// MOAI_SHIM_COMBINE_US sets the window in microseconds (default 500, 0 turns combining off).
#pragma once
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

namespace seal
{
    namespace util
    {
        class OpCombiner
        {
        public:
            struct Request
            {
                const std::uint64_t *in;
                std::uint64_t *out;
            };
            // operation kind, level (RNS rows), Galois element or 0, key pointer, device context
            using Key = std::tuple<int, std::size_t, std::uint32_t, const void *, const void *>;
            using Exec = std::function<void(const std::vector<Request> &)>;

            static OpCombiner &instance()
            {
                static OpCombiner c;
                return c;
            }
            bool enabled() const
            {
                return window_us_ > 0;
            }
            // Runs `exec` on a group that contains `r`; returns once that call has been enqueued.
            void submit(const Key &key, const Request &r, const Exec &exec)
            {
                const std::int64_t now = now_us();
                note_caller(now);
                std::unique_lock<std::mutex> lk(mu_);
                auto it = open_.find(key);
                if (it != open_.end())
                {
                    // join the open group
                    std::shared_ptr<Group> g = it->second;
                    g->reqs.push_back(r);
                    if (g->reqs.size() >= g->target)
                    {
                        g->cv_leader.notify_one();
                    }
                    g->cv_done.wait(lk, [&] { return g->done; });
                    if (g->error)
                    {
                        std::rethrow_exception(g->error);
                    }
                    return;
                }
                // lead a new group
                auto g = std::make_shared<Group>();
                g->reqs.push_back(r);
                const std::size_t expected = recent_callers(now);
                g->target = expected < max_batch_ ? expected : max_batch_;
                const bool wait = g->target > 1 && now >= cooldown_until_;
                if (wait)
                {
                    open_[key] = g;
                    g->cv_leader.wait_for(lk, std::chrono::microseconds(window_us_), [&] { return g->reqs.size() >= g->target; });
                    open_.erase(key);
                }
                std::vector<Request> reqs = g->reqs; // closed: nobody can find the group any more
                if (wait)
                {
                    if (reqs.size() == 1)
                    {
                        if (++misses_ >= 8)
                        {
                            // nobody shares our operations at the moment: stop paying the window for 20 ms
                            cooldown_until_ = now_us() + 20000;
                            misses_ = 0;
                        }
                    }
                    else
                    {
                        misses_ = 0;
                    }
                }
                lk.unlock();
                std::exception_ptr err;
                try
                {
                    exec(reqs);
                }
                catch (...)
                {
                    err = std::current_exception();
                }
                lk.lock();
                g->done = true;
                g->error = err;
                g->cv_done.notify_all();
                lk.unlock();
                if (err)
                {
                    std::rethrow_exception(err);
                }
            }

        private:
            struct Group
            {
                std::vector<Request> reqs;
                std::size_t target = 1;
                bool done = false;
                std::exception_ptr error;
                std::condition_variable cv_leader, cv_done;
            };
            OpCombiner()
            {
                const char *e = std::getenv("MOAI_SHIM_COMBINE_US");
                window_us_ = e ? std::atol(e) : 500;
                for (auto &t : seen_)
                {
                    t.store(0);
                }
            }
            static std::int64_t now_us()
            {
                return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
            }
            void note_caller(std::int64_t now)
            {
                const std::size_t h = std::hash<std::thread::id>()(std::this_thread::get_id()) % seen_.size();
                seen_[h].store(now, std::memory_order_relaxed);
            }
            // threads that called in within the last 5 ms (hash collisions only make the estimate smaller)
            std::size_t recent_callers(std::int64_t now) const
            {
                std::size_t c = 0;
                for (auto &t : seen_)
                {
                    std::int64_t v = t.load(std::memory_order_relaxed);
                    c += (v != 0 && now - v < 5000) ? 1 : 0;
                }
                return c ? c : 1;
            }
            long window_us_ = 500;
            std::size_t max_batch_ = 64;
            std::mutex mu_;
            std::map<Key, std::shared_ptr<Group>> open_;
            std::array<std::atomic<std::int64_t>, 257> seen_;
            int misses_ = 0;
            std::int64_t cooldown_until_ = 0;
        };
    } // namespace util
} // namespace seal
