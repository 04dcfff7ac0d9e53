// Does a prefetch distance of TWO steps hide the operator fetches that miss the XCD's L2?  Same step as mb_scan_step.hip
// (125 MFMA 4x4x4 per step, operands of a 4096-entry 13 MB table, uniformly random tokens = the worst case), 4 wavefronts
// per CU (one per SIMD).  Variant A: registers refilled from global one step ahead (the streamed k_zpropagate4).
// Variant B: the operator of step t+2 is loaded into registers during step t and parked in an LDS ring (2 slots per
// block) at the end of the step; step t+2 reads its A rows from LDS one tile-row ahead (the LDS-table path).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NT = 5, TOK = 400, NTE = 4;
__device__ __forceinline__ void row_from(double (&a)[NT], const double *Cz, int I, int lo, int lx)
{
    const double2 *m = reinterpret_cast<const double2 *>(Cz + I * 16 * NTE + lo);
    const double2 v0 = m[0], v1 = m[1];
    a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
    a[4] = Cz[320 + I * 16 + lx];
}
template <int RING>
__global__ __launch_bounds__(256) void k(const double *table, int A, int steps, const int *toks, double *out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];      // RING: [wave][block][2 slots][TOK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * NTE, lx = q * 4 + r;
    double P[NT][NT], Q[NT][NT], pre[NT][NT];
    for (int i = 0; i < NT; ++i) for (int j = 0; j < NT; ++j) { P[i][j] = 1e-3 * (i + j + lane); pre[i][j] = 0.01; }
    const int *tp = toks + ((blockIdx.x * 4 + wave) * 4 + bq) * (steps + 4);
    double *ring = lds + (size_t)((wave * 4 + bq) * 2) * TOK;
    auto gload = [&](double (&dst)[NT][NT], int tok) __attribute__((always_inline)) {
        const double *G = table + (size_t)(tok % A) * TOK;
#pragma unroll
        for (int I = 0; I < NT; ++I) row_from(dst[I], G, I, lo, lx);
    };
    auto park = [&](const double (&src)[NT][NT], int slot) __attribute__((always_inline)) {   // this lane's 25 values, table layout
        double *S = ring + slot * TOK;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            double2 *m = reinterpret_cast<double2 *>(S + I * 16 * NTE + lo);
            m[0] = double2{src[I][0], src[I][1]}; m[1] = double2{src[I][2], src[I][3]};
            S[320 + I * 16 + lx] = src[I][4];
        }
    };
    auto step_regs = [&](const double (&Pin)[NT][NT], double (&Pout)[NT][NT], int tnext) __attribute__((always_inline)) {
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            double av[NT];
#pragma unroll
            for (int K = 0; K < NT; ++K) av[K] = pre[I][K];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            row_from(pre[I], table + (size_t)(tnext % A) * TOK, I, lo, lx);
        }
    };
    auto step_ring = [&](const double (&Pin)[NT][NT], double (&Pout)[NT][NT], int slot, int tnext2) __attribute__((always_inline)) {
        const double *S = ring + slot * TOK;
        double al[NT];
        row_from(al, S, 0, lo, lx);
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            double an[NT];
            if (I + 1 < NT) row_from(an, S, I + 1, lo, lx);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(al[K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            row_from(pre[I], table + (size_t)(tnext2 % A) * TOK, I, lo, lx);     // token t+2 -> registers during step t
            if (I + 1 < NT) {
#pragma unroll
                for (int K = 0; K < NT; ++K) al[K] = an[K];
            }
        }
        park(pre, slot);                                   // ... and into the slot step t has just finished reading
    };
    if constexpr (RING) {
        gload(pre, tp[0]); park(pre, 0);
        gload(pre, tp[1]); park(pre, 1);
    } else {
        gload(pre, tp[0]);
    }
    for (int s = 0; s < steps; s += 2) {
        if constexpr (RING) {
            step_ring(P, Q, 0, tp[s + 2]);
            step_ring(Q, P, 1, tp[s + 3]);
        } else {
            step_regs(P, Q, tp[s + 1]);
            step_regs(Q, P, tp[s + 2]);
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
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int RING>
int run(const char *name, const double *tab, int A, int steps, const int *toks, double *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = RING ? (size_t)4 * 4 * 2 * TOK * 8 : 1024;
    CHECK(hipFuncSetAttribute((const void *)k<RING>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<RING>, dim3(248), dim3(256), lds, 0, tab, A, steps, toks, out);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s A=%5d: %.3f us per wave-step, %.2f ns per MFMA per SIMD\n", name, A, ms * 1e3 / steps, ms * 1e6 / ((double)steps * 125));
    return 0;
}
int main()
{
    const int AMAX = 4096, steps = 512;
    double *tab, *out; int *toks;
    CHECK(hipMalloc(&tab, (size_t)(AMAX + 64) * TOK * 8)); CHECK(hipMalloc(&out, 256 * 512 * 8));
    std::vector<double> h((size_t)(AMAX + 64) * TOK);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001 + 1e-6 * (double)(i % 977);
    CHECK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<int> ht((size_t)248 * 16 * (steps + 4));
    unsigned x = 12345u;
    for (auto &t : ht) { x = x * 1664525u + 1013904223u; t = (int)(x >> 8); }
    CHECK(hipMalloc(&toks, ht.size() * 4)); CHECK(hipMemcpy(toks, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    for (int A : {44, 1024, 4096}) {
        run<0>("registers refilled one step ahead (4 waves per CU)", tab, A, steps, toks, out);
        run<1>("LDS ring, operands requested two steps ahead", tab, A, steps, toks, out);
    }
    return 0;
}
