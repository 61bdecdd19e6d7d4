// valu_bench.hip -- measures the integer / fp64 VALU issue rates that bound a 64-bit modular
// butterfly on gfx950 (MI355X).  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench
// Output: one line per instruction: lane-ops per clock per CU (128 = full rate on 4 x SIMD32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint32_t y = seed * 2654435761u + 12345u;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, dy = 1.0000001;
    for (int i = 0; i < ITER; ++i)
    {
        if (OP == 0)
        {
#define A(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
            A(a0) A(a1) A(a2) A(a3) A(a4) A(a5) A(a6) A(a7)
#undef A
        }
        else if (OP == 1)
        {
#define A(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));
            A(a0) A(a1) A(a2) A(a3) A(a4) A(a5) A(a6) A(a7)
#undef A
        }
        else if (OP == 2)
        {
#define A(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y));
            A(a0) A(a1) A(a2) A(a3) A(a4) A(a5) A(a6) A(a7)
#undef A
        }
        else if (OP == 3)
        {
#define A(x, w) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(x), "v"(y) : "vcc");
            A(a0, w0) A(a1, w1) A(a2, w2) A(a3, w3) A(a4, w4) A(a5, w5) A(a6, w6) A(a7, w7)
#undef A
        }
        else if (OP == 4)
        {
#define A(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
            A(a0) A(a1) A(a2) A(a3) A(a4) A(a5) A(a6) A(a7)
#undef A
        }
        else if (OP == 5)
        {
#define A(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(dy));
            A(d0) A(d1) A(d2) A(d3) A(d4) A(d5) A(d6) A(d7)
#undef A
        }
        else if (OP == 6)
        {
#define A(w) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(w7));
            A(w0) A(w1) A(w2) A(w3) A(w4) A(w5) A(w6)
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w7) : "v"(w0));
#undef A
        }
        else if (OP == 7)
        {
            // full 64x64 -> high 64 (compiler-generated)
#define A(w) w = __umul64hi(w, w7 | 1);
            A(w0) A(w1) A(w2) A(w3) A(w4) A(w5) A(w6)
#undef A
            w7 += w0;
        }
        else if (OP == 8)
        {
            // Harvey/Shoup butterfly as used by the NTT kernels, 4 per iteration
            const uint64_t q = 1152921504606584833ull, q2 = q * 2, tw = 288794978602139552ull, twq = 4620693217682128896ull;
#define B(x, yv) { uint64_t u = x >= q2 ? x - q2 : x; uint64_t t = __umul64hi(yv, twq + w7); uint64_t v = yv * tw - t * q; x = u + v; yv = u + q2 - v; }
            B(w0, w1) B(w2, w3) B(w4, w5) B(w6, w7)
#undef B
        }
        else if (OP == 10)
        {
#define A(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(dy));
            A(d0) A(d1) A(d2) A(d3) A(d4) A(d5) A(d6) A(d7)
#undef A
        }
        else if (OP == 11)
        {
#define A(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(dy));
            A(d0) A(d1) A(d2) A(d3) A(d4) A(d5) A(d6) A(d7)
#undef A
        }
        else if (OP == 12)
        {
#define A(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
            A(d0) A(d1) A(d2) A(d3) A(d4) A(d5) A(d6) A(d7)
#undef A
        }
        else if (OP == 13)
        {
            // FP64 butterfly for primes below 2^50: v = y*w - rint(y*(w/q))*q exactly (FMA error-free
            // product), x' = u + v, y' = u - v; 4 per iteration
            const double q = 1125899906826241.0, tw = 288794978602139.0, twq = tw / q;
#define B(x, yv) { double h = yv * (tw + dy); double l = __builtin_fma(yv, tw + dy, -h); double c = __builtin_rint(yv * twq); double v = __builtin_fma(-c, q, h) + l; double u = x; x = u + v; yv = u - v; }
            B(d0, d1) B(d2, d3) B(d4, d5) B(d6, d7)
#undef B
        }
        else if (OP == 14)
        {
            // the M_LAZY8 butterfly of the NTT kernels (csrc/modarith.hip.h ct_bfly_lazy8): approximate Shoup quotient from the two
            // high cross terms, remainder as one multiply-add chain modulo 2^64, sign-tested guard; 4 per iteration
            const uint64_t q = 1152921504606584833ull, nq = 0 - q, n4q = 0 - 4 * q, tw = 288794978602139552ull, twq = 4620693217682128896ull;
#define B(x, yv) { uint64_t d = x + n4q; uint64_t u = (int64_t)d < 0 ? x : d; const uint64_t wq_ = twq + w7; \
                   const uint32_t y0 = (uint32_t)yv, y1 = (uint32_t)(yv >> 32), q0 = (uint32_t)wq_, q1 = (uint32_t)(wq_ >> 32); \
                   uint64_t t = (uint64_t)y1 * q1 + __umulhi(y1, q0); t += __umulhi(y0, q1); \
                   const uint32_t t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32), w0_ = (uint32_t)tw, w1_ = (uint32_t)(tw >> 32), n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32); \
                   uint64_t p = (uint64_t)y0 * w0_; p += (uint64_t)t0 * n0; \
                   const uint32_t hi = (uint32_t)(p >> 32) + y0 * w1_ + y1 * w0_ + t0 * n1 + t1 * n0; \
                   const uint64_t v = ((uint64_t)hi << 32) | (uint32_t)p; x = u + v; yv = u - (v + n4q); }
            B(w0, w1) B(w2, w3) B(w4, w5) B(w6, w7)
#undef B
        }
        else if (OP == 9)
        {
#define A(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(y));
            A(a0) A(a1) A(a2) A(a3) A(a4) A(a5) A(a6) A(a7)
#undef A
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) ^
                 (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (r == 0x12345678u)
    {
        out[0] = r;
    }
}

template <int NT>
__global__ __launch_bounds__(256) void copyk(const ulonglong2 *__restrict__ a, ulonglong2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        if (NT)
        {
            ulonglong2 v;
            v.x = __builtin_nontemporal_load(&a[i].x);
            v.y = __builtin_nontemporal_load(&a[i].y);
            __builtin_nontemporal_store(v.x, &b[i].x);
            __builtin_nontemporal_store(v.y, &b[i].y);
        }
        else
        {
            b[i] = a[i];
        }
    }
}

// in-place read-modify-write of 32 KiB tiles, one tile per workgroup iteration (the NTT's access shape)
template <int UNROLL>
__global__ __launch_bounds__(256) void rmwk(ulonglong2 *__restrict__ a, size_t ntiles)
{
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x)
    {
        ulonglong2 *p = a + t * 2048;
        ulonglong2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            v[j] = p[j * 256 + threadIdx.x];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            v[j].x += 1;
            p[j * 256 + threadIdx.x] = v[j];
        }
    }
}


// store-only kernels in the access shape of the strided NTT pass (a workgroup owns G columns of a 2^16-coefficient row and stores
// 256 segments of G * 8 bytes, 2 KiB apart) against the same bytes stored contiguously.  ORDER 0: consecutive workgroups take
// consecutive tiles of one row; 1: the same tile of consecutive rows (the production order: the twiddle slice is shared).
template <int GB, int ORDER>
__global__ __launch_bounds__(256) void wpat(uint64_t *__restrict__ out, uint32_t rows)
{
    constexpr uint32_t G = 1u << GB;            // columns per workgroup
    constexpr uint32_t TILES = 256u / G;        // workgroups per row
    constexpr uint32_t PER = (256u * G) / 256u; // stores per thread
    const uint32_t w = blockIdx.x;
    const uint32_t row = ORDER ? w % rows : w / TILES;
    const uint32_t tile = ORDER ? w / rows : w % TILES;
    uint64_t *__restrict__ r = out + ((size_t)row << 16) + tile * G;
    const uint32_t g = threadIdx.x & (G - 1), th = threadIdx.x >> GB;
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j)
    {
        const uint32_t t_ = th * PER + j;
        r[(t_ << 8) + g] = (uint64_t)t_ * 0x9e3779b97f4a7c15ull + g;
    }
}
__global__ __launch_bounds__(256) void wlin(ulonglong2 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        ulonglong2 v;
        v.x = i;
        v.y = ~i;
        out[i] = v;
    }
}

template <int OP>
int run(const char *name, double ops_per_iter, int cus, uint32_t *d)
{
    const int blocks = cus * 8;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
    {
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
    }
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double total = (double)reps * blocks * 256.0 * ITER * ops_per_iter;
    double per_s = total / (ms * 1e-3);
    printf("%-28s %8.3f ms  %9.2f Gop/s  %7.2f lane-ops/clk/CU @2.4GHz\n", name, ms / reps, per_s * 1e-9,
           per_s / (2.4e9 * cus));
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("device: %s %s, %d CUs, clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    uint32_t *d;
    CHK(hipMalloc(&d, 4096));
    int cus = p.multiProcessorCount;
    run<0>("v_add_u32", 8, cus, d);
    run<1>("v_mul_lo_u32", 8, cus, d);
    run<2>("v_mul_hi_u32", 8, cus, d);
    run<3>("v_mad_u64_u32", 8, cus, d);
    run<4>("v_mul_u32_u24", 8, cus, d);
    run<9>("v_mad_u32_u24", 8, cus, d);
    run<5>("v_fma_f64", 8, cus, d);
    run<6>("v_lshl_add_u64 (64-bit add)", 8, cus, d);
    run<7>("__umul64hi", 7, cus, d);
    run<8>("shoup butterfly (64-bit)", 4, cus, d);
    run<14>("lazy8 butterfly (q < 2^60)", 4, cus, d);
    run<10>("v_mul_f64", 8, cus, d);
    run<11>("v_add_f64", 8, cus, d);
    run<12>("v_rndne_f64", 8, cus, d);
    run<13>("fp64 butterfly (q < 2^50)", 4, cus, d);
    // HBM copy reference
    {
        size_t bytes = (size_t)2 << 30;
        void *a, *b;
        CHK(hipMalloc(&a, bytes));
        CHK(hipMalloc(&b, bytes));
        CHK(hipMemset(a, 1, bytes));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice));
        CHK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i)
        {
            CHK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
        }
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpy D2D 2 GiB: %.3f ms -> %.2f TB/s (read+write)\n", ms / 5, 2.0 * bytes * 5 / (ms * 1e-3) * 1e-12);
    }
    {
        size_t bytes = (size_t)4 << 30;
        ulonglong2 *a, *b;
        CHK(hipMalloc(&a, bytes));
        CHK(hipMalloc(&b, bytes));
        CHK(hipMemset(a, 1, bytes));
        CHK(hipMemset(b, 1, bytes));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        size_t n = bytes / 16;
        const char *names[] = { "copy 16B/lane grid 2048", "copy nontemporal grid 2048", "copy 16B/lane grid 8192",
                                "in-place rmw 32KiB tiles, persistent 1024 WGs", "in-place rmw 32KiB tiles, one WG per tile" };
        for (int variant = 0; variant < 5; ++variant)
        {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep)
            {
                CHK(hipEventRecord(e0, 0));
                if (variant == 0) hipLaunchKernelGGL(copyk<0>, dim3(256 * 8), dim3(256), 0, 0, a, b, n);
                if (variant == 1) hipLaunchKernelGGL(copyk<1>, dim3(256 * 8), dim3(256), 0, 0, a, b, n);
                if (variant == 2) hipLaunchKernelGGL(copyk<0>, dim3(256 * 32), dim3(256), 0, 0, a, b, n);
                if (variant == 3) hipLaunchKernelGGL(rmwk<1>, dim3(256 * 4), dim3(256), 0, 0, a, bytes / 32768);
                if (variant == 4) hipLaunchKernelGGL(rmwk<1>, dim3((unsigned)(bytes / 32768)), dim3(256), 0, 0, a, bytes / 32768);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-48s 4 GiB: %.3f ms -> %.2f TB/s (read+write)\n", names[variant], best, 2.0 * bytes / (best * 1e-3) * 1e-12);
        }
    }
    {
        const uint32_t rows = 96 * 30; // one mode group of a pack of 48 at l = 30
        const size_t bytes = (size_t)rows << 19;
        uint64_t *o;
        CHK(hipMalloc(&o, bytes));
        CHK(hipMemset(o, 0, bytes));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        const char *names[] = { "store only, contiguous",
                                "store only, 128 B segments 2 KiB apart, tiles of a row together",
                                "store only, 128 B segments 2 KiB apart, one tile of many rows together",
                                "store only, 256 B segments 2 KiB apart, tiles of a row together",
                                "store only, 256 B segments 2 KiB apart, one tile of many rows together",
                                "store only, 512 B segments 2 KiB apart, one tile of many rows together" };
        for (int variant = 0; variant < 6; ++variant)
        {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep)
            {
                CHK(hipEventRecord(e0, 0));
                if (variant == 0) hipLaunchKernelGGL(wlin, dim3(256 * 32), dim3(256), 0, 0, (ulonglong2 *)o, bytes / 16);
                if (variant == 1) hipLaunchKernelGGL((wpat<4, 0>), dim3(rows * 16), dim3(256), 0, 0, o, rows);
                if (variant == 2) hipLaunchKernelGGL((wpat<4, 1>), dim3(rows * 16), dim3(256), 0, 0, o, rows);
                if (variant == 3) hipLaunchKernelGGL((wpat<5, 0>), dim3(rows * 8), dim3(256), 0, 0, o, rows);
                if (variant == 4) hipLaunchKernelGGL((wpat<5, 1>), dim3(rows * 8), dim3(256), 0, 0, o, rows);
                if (variant == 5) hipLaunchKernelGGL((wpat<6, 1>), dim3(rows * 4), dim3(256), 0, 0, o, rows);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-72s %.2f GiB: %.3f ms -> %.2f TB/s (write)\n", names[variant], bytes / 1073741824.0, best, bytes / (best * 1e-3) * 1e-12);
        }
    }
    return 0;
}
