// bench_streams.cpp -- how much do single-ciphertext operations gain when independent host threads issue them
// on their own HIP streams?  T threads, each with a private stream and ciphertext, call moai_apply_galois
// (one key switch) in a loop through the C ABI; aggregate operations per second against T = 1.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "moai_hip.h"
#include "seal/seal.h"

using namespace std;

int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 15;
    const size_t n = 65536;
    vector<int> bits{ 51 };
    for (int i = 0; i < 20; i++) bits.push_back(46);
    for (int i = 0; i < 14; i++) bits.push_back(51);
    bits.push_back(58);
    auto mods = seal::CoeffModulus::Create(n, bits);
    vector<uint64_t> primes;
    for (auto &m : mods) primes.push_back(m.value());
    moai_ctx *ctx = nullptr;
    if (moai_ctx_create(16, primes.data(), primes.size(), 0, &ctx)) { printf("ctx: %s\n", moai_last_error()); return 1; }
    const size_t k = primes.size();
    void *key = nullptr;
    moai_malloc(&key, (k - 1) * 2 * k * n * 8);
    moai_memset_zero(key, (k - 1) * 2 * k * n * 8, nullptr);
    const uint32_t elt = moai_galois_elt_from_step(ctx, 1);
    for (int T : { 1, 2, 4, 8, 16 })
    {
        vector<void *> streams(T), cts(T);
        for (int t = 0; t < T; t++)
        {
            moai_stream_create(&streams[t]);
            moai_malloc(&cts[t], 2 * L * n * 8);
            moai_memset_zero(cts[t], 2 * L * n * 8, streams[t]);
            moai_apply_galois(ctx, (uint64_t *)cts[t], L, elt, (const uint64_t *)key, 1, streams[t]); // warm
            moai_stream_sync(streams[t]);
        }
        const int reps = 200;
        auto t0 = chrono::steady_clock::now();
#pragma omp parallel for num_threads(T)
        for (int t = 0; t < T; t++)
        {
            for (int r = 0; r < reps; r++)
                moai_apply_galois(ctx, (uint64_t *)cts[t], L, elt, (const uint64_t *)key, 1, streams[t]);
            moai_stream_sync(streams[t]);
        }
        double s = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        printf("L=%d, %2d threads/streams: %.3f ms per key switch aggregate (%.0f per second)\n", L, T, s * 1e3 / (reps * T), reps * T / s);
        for (int t = 0; t < T; t++)
        {
            moai_stream_destroy(streams[t]);
            moai_free(cts[t]);
        }
    }
    return 0;
}
