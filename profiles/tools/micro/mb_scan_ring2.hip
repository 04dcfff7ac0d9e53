// Round 3: does a DEEPER register prefetch help the streamed scan?  k_zpropagate4's step takes its A operands from 25
// registers per lane that were loaded ONE step ahead; two wavefronts per SIMD share the MFMA pipe (8 per CU).  Here:
//   mode 0: that loop (one register set, refilled tile-row by tile-row with the next step's operator)
//   mode 1: TWO register sets, each refilled with the operator two steps ahead (needs ~50 more registers: one wavefront
//           per SIMD, 4 per CU, twice the steps per wavefront)
//   mode 2: mode 0 with 4 wavefronts per CU (one per SIMD), for reference
// Tokens: uniform over A entries (every step cold once A * 3.2 KB exceeds the 4 MB L2) or 90 % from 256 hot entries.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NT = 5, TOK = 400, NTE = 4;
__device__ __forceinline__ void grow(double (&a)[NT], const double *Gz, int I, int lo, int lx)
{
    const double2 *m = reinterpret_cast<const double2 *>(Gz + I * 16 * NTE + lo);
    const double2 v0 = m[0], v1 = m[1];
    a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
    a[4] = Gz[320 + I * 16 + lx];
}
template <int DEPTH, int WAVES, bool TOUCH = false>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(const double *table, int steps, const int *toks, double *out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * NTE, lx = q * 4 + r;
    double P[NT][NT], Q[NT][NT], pre[DEPTH][NT][NT];
    for (int i = 0; i < NT; ++i) for (int j = 0; j < NT; ++j) P[i][j] = 1e-3 * (i + j + lane);
    const int *tp = toks + ((blockIdx.x * WAVES + wave) * 4 + bq) * (steps + 4);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int I = 0; I < NT; ++I) grow(pre[d][I], table + (size_t)tp[d] * TOK, I, lo, lx);
    auto step = [&](const double (&Pin)[NT][NT], double (&Pout)[NT][NT], double (&A)[NT][NT], int tnext) __attribute__((always_inline)) {
        const double *Gn = table + (size_t)tnext * TOK;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            double av[NT];
#pragma unroll
            for (int K = 0; K < NT; ++K) av[K] = A[I][K];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            grow(A[I], Gn, I, lo, lx);
        }
    };
    // TOUCH: an L2 warm-up of the operator TWO steps ahead - one 8-byte load per lane and 128-byte line (lane lx of the
    // segment's 16 takes lines lx and 16 + lx of the entry's 25), issued at the top of a step and only waited for two steps
    // later (the values are thrown away) - in front of the one-step register refill
    double tch[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    auto touch = [&](int tok, double (&t)[2]) __attribute__((always_inline)) {
        const double *G = table + (size_t)tok * TOK;
        t[0] = G[lx * 16];
        t[1] = G[(lx < 9 ? 16 + lx : 24) * 16];
    };
    auto sink = [&](double (&t)[2]) __attribute__((always_inline)) { asm volatile("" ::"v"(t[0]), "v"(t[1])); };
    for (int s = 0; s < steps; s += 2) {
        if constexpr (DEPTH == 1 && TOUCH) {
            sink(tch[0]); touch(tp[s + 2], tch[0]);
            step(P, Q, pre[0], tp[s + 1]);
            sink(tch[1]); touch(tp[s + 3], tch[1]);
            step(Q, P, pre[0], tp[s + 2]);
        } else if constexpr (DEPTH == 1) {
            step(P, Q, pre[0], tp[s + 1]);
            step(Q, P, pre[0], tp[s + 2]);
        } else {
            step(P, Q, pre[0], tp[s + 2]);
            step(Q, P, pre[1], tp[s + 3]);
        }
        if ((s & 15) == 14) {
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J < NT; ++J) P[I][J] *= 0.25;
        }
    }
    double acc = 0;
    for (int i = 0; i < NT; ++i) for (int j = 0; j < NT; ++j) acc += P[i][j];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int DEPTH, int WAVES, bool TOUCH = false>
int run(const char *name, const double *tab, int steps_per_slot8, const int *toks, double *out)
{
    const int steps = steps_per_slot8 * 8 / WAVES;          // same work per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<DEPTH, WAVES, TOUCH>), dim3(255), dim3(WAVES * 64), 0, 0, tab, steps, toks, out);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-64s %7.1f us  %.2f ns per MFMA per SIMD\n", name, best * 1e3, best * 1e6 / ((double)steps * 125 * (WAVES / 4.0)));
    return 0;
}
int main()
{
    const int AMAX = 4096, steps8 = 84;                     // 84 steps per wavefront at 8 wavefronts per CU = BASELINE config[1]
    double *tab, *out; int *toks;
    CHECK(hipMalloc(&tab, (size_t)(AMAX + 64) * TOK * 8)); CHECK(hipMalloc(&out, 256 * 512 * 8));
    std::vector<double> h((size_t)(AMAX + 64) * TOK);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001 + 1e-6 * (double)(i % 977);
    CHECK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const size_t per = 2 * steps8 + 4;
    std::vector<int> ht((size_t)255 * 32 * per);
    CHECK(hipMalloc(&toks, ht.size() * 4));
    for (int skew = 0; skew < 2; ++skew) {
        unsigned x = 12345u;
        for (auto &t : ht) {
            x = x * 1664525u + 1013904223u; const unsigned a = x >> 8;
            x = x * 1664525u + 1013904223u; const unsigned b = x >> 8;
            t = skew ? ((a % 100) < 91 ? (int)(b % 1024) : (int)(b % AMAX)) : (int)(b % AMAX);   // skewed: 91 % of the steps inside 3.3 MB
        }
        CHECK(hipMemcpy(toks, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
        printf("tokens: %s\n", skew ? "91 % of the steps from 1024 entries (3.3 MB, L2 resident), the rest uniform over 4096" : "uniform over 4096 entries (13 MB: most steps miss the 4 MB L2)");
        if (run<1, 8>("one register set, 8 wavefronts per CU (k_zpropagate4's loop)", tab, steps8, toks, out)) return 1;
        if (run<1, 8, true>("  + L2 warm-up touch two steps ahead", tab, steps8, toks, out)) return 1;
        if (run<1, 4>("one register set, 4 wavefronts per CU", tab, steps8, toks, out)) return 1;
        if (run<2, 4>("two register sets (two steps ahead), 4 wavefronts per CU", tab, steps8, toks, out)) return 1;
        if (run<2, 8>("two register sets, 8 wavefronts per CU (if the registers allow)", tab, steps8, toks, out)) return 1;
    }
    return 0;
}
