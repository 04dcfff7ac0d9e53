// what does each ingredient of the hybrid scan's step cost?  125 MFMA 4x4x4 per step, 8 waves/CU (2 per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NT = 5, TOK = 400, NTE = 4;
__device__ __forceinline__ void lds_row(double (&a)[NT], const double *Cz, int I, int lo, int lx)
{
    const double2 *m = reinterpret_cast<const double2 *>(Cz + I * 16 * NTE + lo);
    const double2 v0 = m[0], v1 = m[1];
    a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
    a[4] = Cz[320 + I * 16 + lx];
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const double *table, int A, int steps, const int *toks, double *out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * NTE, lx = q * 4 + r;
    for (int i = tid; i < 44 * TOK; i += 512) lds[i] = table[i];
    __syncthreads();
    double P[NT][NT], Q[NT][NT], pre[NT][NT], al[NT];
    for (int i = 0; i < NT; ++i) for (int j = 0; j < NT; ++j) { P[i][j] = 1e-3 * (i + j + lane); pre[i][j] = 0.01; }
    const int *tp = toks + ((blockIdx.x * 8 + wave) * 4 + bq) * (steps + 2);
    auto step = [&](const double (&Pin)[NT][NT], double (&Pout)[NT][NT], int tc, int tn) __attribute__((always_inline)) {
        const int sc = tc % 44, sn = tn % 44;
        const bool cur_cold = (MODE & 2) ? (tc & 64) != 0 : false;
        const double *Cz = lds + sc * TOK, *Cn = lds + sn * TOK;
        const double *Gn = (MODE & 8) ? table + (size_t)(tn % A) * TOK : table + 44 * TOK;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            double an[NT], av[NT];
            if constexpr (MODE & 1) {
                if (I + 1 < NT) lds_row(an, Cz, I + 1, lo, lx); else lds_row(an, Cn, 0, lo, lx);
            }
#pragma unroll
            for (int K = 0; K < NT; ++K) av[K] = (MODE & 1) ? ((MODE & 2) ? (cur_cold ? pre[I][K] : al[K]) : al[K]) : pre[I][K];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 4) {
                const double2 *m = reinterpret_cast<const double2 *>(Gn + I * 16 * NTE + lo);
                const double2 v0 = m[0], v1 = m[1];
                pre[I][0] = v0.x; pre[I][1] = v0.y; pre[I][2] = v1.x; pre[I][3] = v1.y;
                pre[I][4] = Gn[320 + I * 16 + lx];
            }
            if constexpr (MODE & 1) {
#pragma unroll
                for (int K = 0; K < NT; ++K) al[K] = an[K];
            }
        }
    };
    int t0 = tp[0];
    if constexpr (MODE & 1) lds_row(al, lds + (t0 % 44) * TOK, 0, lo, lx);
    for (int s = 0; s < steps; s += 2) {
        const int t1 = tp[s + 1], t2 = tp[s + 2];
        step(P, Q, t0, t1);
        step(Q, P, t1, t2);
        t0 = t2;
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
template <int MODE>
int run(const char *name, const double *tab, int A, int steps, const int *toks, double *out, int waves)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    CHECK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(248), dim3(waves * 64), 150 * 1024, 0, tab, A, steps, toks, out);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_mfma_simd = ms * 1e6 / ((double)steps * 125 * (waves / 4.0));
    printf("%-44s A=%5d waves=%d: %.3f us per wave-step, %.2f ns per MFMA per SIMD (%.1f TF/s chip)\n", name, A, waves, ms * 1e3 / steps, ns_per_mfma_simd,
           248.0 * waves * steps * 125 * 512 / (ms * 1e-3) * 1e-12);
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
    std::vector<int> ht((size_t)248 * 32 * (steps + 2));
    unsigned x = 12345u;
    for (auto &t : ht) { x = x * 1664525u + 1013904223u; t = (int)(x >> 8); }
    CHECK(hipMalloc(&toks, ht.size() * 4)); CHECK(hipMemcpy(toks, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    for (int waves : {8}) {
        run<1>("A operands from LDS, one row ahead", tab, 44, steps, toks, out, waves);
        run<15>("LDS + select + refill loads from the table", tab, 4096, steps, toks, out, waves);
        for (int A : {44, 256, 1024, 4096}) run<12>("operands from the global table only", tab, A, steps, toks, out, waves);
    }
    return 0;
}
