// microbenchmarks (gfx950): fp64 FMA / MFMA issue rates, co-issue, 4x4x4 f64 layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// mode 0: 8 independent v_fma_f64 per iter; mode 1: 8 independent mfma 4x4x4 per iter; mode 2: 8 mfma 16x16x4
// mode 3: waves with odd wave-id run mfma4x4x4, even run fma (co-issue test); mode 4: v_fmac_f64_dpp row_newbcast
// mode 5: dependent mfma4x4x4 chain (latency); mode 6: odd waves mfma 16x16x4, even fma
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(int mode, int iters, double *out, long long *cyc)
{
    const int wave = threadIdx.x >> 6;
    double c = 1.0 + threadIdx.x * 1e-9, p = 0.999999;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    v4d q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
    int m = mode;
    if (mode == 3) m = (wave & 1) ? 1 : 0;
    if (mode == 6) m = (wave & 1) ? 2 : 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (m == 0) {
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                         "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(p));
        }
    } else if (m == 1) {
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a4, 0, 0, 0);
            a5 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a6, 0, 0, 0);
            a7 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a7, 0, 0, 0);
        }
    } else if (m == 2) {
        for (int it = 0; it < iters; ++it) {
            q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q2, 0, 0, 0);
            q3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q3, 0, 0, 0);
            q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q2, 0, 0, 0);
            q3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, p, q3, 0, 0, 0);
        }
    } else if (m == 4) {
#define FB(acc, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(p))
        for (int it = 0; it < iters; ++it) { FB(a0, 0); FB(a1, 1); FB(a2, 2); FB(a3, 3); FB(a4, 4); FB(a5, 5); FB(a6, 6); FB(a7, 7); }
    } else if (m == 5) {
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, p, a0, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q0[0] + q1[1] + q2[2] + q3[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// layout probe for v_mfma_f64_4x4x4f64: lane l supplies a = A_val(l), b = B_val(l); D returned per lane.
__global__ void k_layout(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
// dpp newbcast correctness: out[l] = acc where acc = c[l] * p[row*16 + K]
__global__ void k_bcast(const double *C, const double *P, double *O)
{
    const int l = threadIdx.x;
    double acc = 0.0, c = C[l], p = P[l];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(p));
    O[l] = acc;
    double acc2 = 0.0;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(p), "v"(c));
    O[64 + l] = acc2;
}

int main()
{
    double *d; long long *dc;
    CHECK(hipMalloc(&d, 256 * 1024 * 8)); CHECK(hipMalloc(&dc, 256 * 16 * 8));
    const int iters = 20000;
    const char *names[] = {"v_fma_f64 x8", "mfma_f64_4x4x4 x8 indep", "mfma_f64_16x16x4 x8 (4 acc)", "co-issue: odd waves mfma4x4x4, even fma", "v_fmac_f64_dpp row_newbcast x8", "mfma_f64_4x4x4 dependent chain x8", "co-issue: odd waves mfma16x16x4, even fma"};
    for (int mode = 0; mode < 7; ++mode)
        for (int wps = 1; wps <= 2; ++wps) {
            if ((mode == 3 || mode == 6) && wps == 1) continue;
            const int threads = 256 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, mode, iters, d, dc);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, mode, iters, d, dc);
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(256 * 4 * wps);
            CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
            double avg_even = 0, avg_odd = 0; int ne = 0, no = 0;
            for (size_t i = 0; i < h.size(); ++i) { if ((i % (4 * wps)) & 1) { avg_odd += h[i]; ++no; } else { avg_even += h[i]; ++ne; } }
            avg_even /= ne; avg_odd /= (no ? no : 1);
            // s_memtime ticks at 100 MHz constant? report both: ticks per instr, and derived ns
            printf("%-44s %d wave/SIMD: %.3f ms | memtime ticks/instr even-waves %.3f odd-waves %.3f | ns per instr per wave %.3f\n",
                   names[mode], wps, ms, avg_even / (iters * 8.0), avg_odd / (iters * 8.0), ms * 1e6 / (iters * 8.0));
        }
    // layout probe
    {
        std::vector<double> A(64), B(64), D(64);
        double *dA, *dB, *dD; CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dD, 512));
        // A[l] = 2^(l) distinct?  use: A one-hot at lane la, B one-hot at lane lb; D lanes nonzero tell the mapping
        printf("layout probe (la, lb) -> D lanes nonzero:\n");
        for (int la = 0; la < 64; ++la) {
            for (int lb = 0; lb < 64; ++lb) {
                for (int i = 0; i < 64; ++i) { A[i] = 0; B[i] = 0; }
                A[la] = 1.0; B[lb] = 1.0;
                hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
                hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
                bool any = false;
                for (int i = 0; i < 64; ++i) if (D[i] != 0.0) { if (!any) printf("  A@%2d B@%2d ->", la, lb); any = true; printf(" %d", i); }
                if (any) printf("\n");
            }
        }
        std::vector<double> C(64), P(64), O(128);
        for (int i = 0; i < 64; ++i) { C[i] = 1.0 + i; P[i] = 100.0 + i; }
        double *dO; CHECK(hipMalloc(&dO, 1024));
        hipMemcpy(dA, C.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, P.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_bcast, dim3(1), dim3(64), 0, 0, dA, dB, dO);
        hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
        printf("newbcast:5 src0=c src1=p: lane0 %.1f lane17 %.1f lane40 %.1f  (c*p[row*16+5] would be %.1f %.1f %.1f; c[row*16+5]*p would be %.1f %.1f %.1f)\n",
               O[0], O[17], O[40], C[0] * P[5], C[17] * P[21], C[40] * P[37], C[5] * P[0], C[21] * P[17], C[37] * P[40]);
        printf("swapped operands: lane0 %.1f lane17 %.1f lane40 %.1f\n", O[64], O[64 + 17], O[64 + 40]);
    }
    return 0;
}
