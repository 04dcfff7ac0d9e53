// kernels_zip2.hpp - register-blocked token kernel: P <- C_tok * P with the segment's whole transfer
// operator P (N x N) held in the registers of ONE 16-lane DPP row.  Included by imcoal_fwd.hip only.
//
// k_zpropagate (kernels_zip.hpp) moves every vector through LDS once per token and re-reads the token
// operator per vector: ~0.6 LDS-read instructions per fp64 FMA, which makes it LDS-bound (measured
// 69 % LDS-busy, 21 % VALU-busy at N=20).  Here the N basis vectors of a segment are advanced together:
//   * 16 lanes form a 4 x 4 grid (rb, cb); a lane owns the RB x RB block P[rb*RB.., cb*RB..] (NP = 4 RB);
//   * per token it needs C_tok[its RB rows][all j] - RB*NP doubles from the LDS operator table, the same
//     for the 4 lanes of a grid row (broadcast) - and P_old[all j][its RB columns], i.e. its own block plus
//     the blocks of the 3 other lanes of its grid column, which arrive by DPP row rotations (row_ror 4/8/12:
//     pure VALU moves, no LDS);
//   * RB*RB*NP FMAs per lane per token against RB*NP LDS doubles: RB FMAs per LDS operand instead of ~1.6.
// Four segments ride in one wavefront; each keeps one common power-of-two exponent for its operator.
#pragma once
#include "kernels_big.hpp"
#include "kernels_zip.hpp"

#ifndef IMC_Z2WAVES
#define IMC_Z2WAVES 8
#endif
static constexpr int Z2WAVES = IMC_Z2WAVES;   // wavefronts per workgroup
static constexpr int Z2SLOTS = Z2WAVES * 4;    // segments (16-lane rows) per workgroup


template <int RB>
struct Zip2Geom {
    static constexpr int NP = 4 * RB;
    static constexpr int NPS = NP + 2;   // operator row stride in LDS (doubles): rows of one grid column land on distinct banks
    // the operator table doubles as the exchange area of the end-of-kernel fold (Z2SLOTS operators, unpadded rows)
    static constexpr size_t op_doubles(int A)
    {
        return (size_t)A * NP * NPS > (size_t)Z2SLOTS * NP * NP ? (size_t)A * NP * NPS : (size_t)Z2SLOTS * NP * NP;
    }
    static constexpr int ints(int A) { return ((A > Z2SLOTS ? A : Z2SLOTS) + 1) & ~1; }
    static constexpr size_t lds_bytes(int A) { return op_doubles(A) * 8 + (size_t)ints(A) * 4 + 16; }
};

// acc[ii][cc] += sum_jj C[row0+ii][j0+jj] * src[jj][cc], where src is P itself (CTRL = 0) or the P block of
// another lane of the grid column, fetched one row at a time by DPP row rotation (keeps only RB doubles of the
// partner block live, which is what lets three wavefronts share a SIMD at N=20).
template <int RB, int NPS, int CTRL>
__device__ __forceinline__ void zip2_block(double (&acc)[RB][RB], const double (&P)[RB][RB], const double *Crow)
{
#pragma unroll
    for (int jj = 0; jj < RB; ++jj) {
        double q[RB];
#pragma unroll
        for (int cc = 0; cc < RB; ++cc) {
            if constexpr (CTRL == 0) q[cc] = P[jj][cc];
            else q[cc] = dpp_f64<CTRL>(P[jj][cc]);
        }
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
            const double c = Crow[ii * NPS + jj];
#pragma unroll
            for (int cc = 0; cc < RB; ++cc) acc[ii][cc] = fma(c, q[cc], acc[ii][cc]);
        }
    }
}

template <int RB, bool PRED, int NPS = Zip2Geom<RB>::NPS>
__device__ __forceinline__ void zip2_step(double (&P)[RB][RB], const double *C, const int *cex, int tok, int rb,
                                          const int (&rbs)[3], bool act, int &ex)
{
    constexpr int NP = 4 * RB;
    const int ce = cex[tok];
    const double *Cz = C + (size_t)tok * (NP * NPS) + (size_t)(rb * RB) * NPS;
    double acc[RB][RB];
#pragma unroll
    for (int ii = 0; ii < RB; ++ii)
#pragma unroll
        for (int cc = 0; cc < RB; ++cc) acc[ii][cc] = 0.0;
    zip2_block<RB, NPS, 0>(acc, P, Cz + rb * RB);
    zip2_block<RB, NPS, DPP_ROW_ROR4>(acc, P, Cz + rbs[0] * RB);
    zip2_block<RB, NPS, DPP_ROW_ROR8>(acc, P, Cz + rbs[1] * RB);
    zip2_block<RB, NPS, DPP_ROW_ROR12>(acc, P, Cz + rbs[2] * RB);
#pragma unroll
    for (int ii = 0; ii < RB; ++ii)
#pragma unroll
        for (int cc = 0; cc < RB; ++cc) P[ii][cc] = (PRED && !act) ? P[ii][cc] : acc[ii][cc];
    ex += (PRED && !act) ? 0 : ce;
}

// common power-of-two rescale of a segment's operator: exponent of its largest entry over the 16 lanes
template <int RB>
__device__ __forceinline__ void zip2_rescale(double (&P)[RB][RB], int &ex)
{
    double mx = 0.0;
#pragma unroll
    for (int ii = 0; ii < RB; ++ii)
#pragma unroll
        for (int cc = 0; cc < RB; ++cc) mx = (P[ii][cc] > mx || P[ii][cc] != P[ii][cc]) ? P[ii][cc] : mx;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        const double o = __shfl_xor(mx, m, 16);
        mx = (o > mx || o != o) ? o : mx;
    }
    int e = 0;
    (void)frexp(mx, &e);
    e = (mx > 0.0 && mx < INFINITY) ? e : 0;
#pragma unroll
    for (int ii = 0; ii < RB; ++ii)
#pragma unroll
        for (int cc = 0; cc < RB; ++cc) P[ii][cc] = ldexp(P[ii][cc], -e);
    ex += e;
}

template <int RB>
__global__ __launch_bounds__(Z2WAVES * 64, Z2WAVES / 4) void k_zpropagate2(BigArgs a)
{
    using Geo = Zip2Geom<RB>;
    constexpr int NP = Geo::NP, NPS = Geo::NPS, NT = Z2WAVES * 64;
    constexpr int EPT = (NP * NP + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *C = lds;                                                   // [A][NP][NPS]
    int *cex = reinterpret_cast<int *>(C + Geo::op_doubles(a.A));    // [max(A, Z2SLOTS)]
    unsigned long long *smax = reinterpret_cast<unsigned long long *>(cex + Geo::ints(a.A));   // [2]

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *pi_p = pp;
    const double *Tp = pp + a.PP;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;

    // ---- operator table (as in k_zpropagate) ----
    for (int idx = tid; idx < a.S * NP * NP; idx += NT) {
        const int sidx = idx / (NP * NP);
        const int rem = idx - sidx * NP * NP;
        const int i = rem / NP, j = rem - i * NP;
        C[((size_t)sidx * NP + i) * NPS + j] = Etg[(size_t)sidx * a.PP + i] * Tp[(size_t)j * a.PP + i];
    }
    if (tid < a.S) cex[tid] = 0;
    if (tid < 2) smax[tid] = 0ull;
    __syncthreads();
    for (int z = a.S; z < a.A; ++z) {
        const int zl = a.tok_left[z], zr = a.tok_right[z];
        const double *Cl = C + (size_t)zl * NP * NPS, *Cr = C + (size_t)zr * NP * NPS;
        double *Cz = C + (size_t)z * NP * NPS;
        double vals[EPT];
        double mx = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = tid + e * NT;
            double acc = 0.0;
            if (idx < NP * NP) {
                const int i = idx / NP, j = idx - i * NP;
#pragma unroll 4
                for (int k = 0; k < NP; ++k) acc = fma(Cr[i * NPS + k], Cl[k * NPS + j], acc);
            }
            vals[e] = acc;
            mx = (acc > mx || acc != acc) ? acc : mx;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double o = __shfl_xor(mx, m, 64);
            mx = (o > mx || o != o) ? o : mx;
        }
        if ((tid & 63) == 0) atomicMax(&smax[z & 1], (unsigned long long)__double_as_longlong(mx));
        __syncthreads();
        const double m = __longlong_as_double((long long)smax[z & 1]);
        int e2 = 0;
        (void)frexp(m, &e2);
        e2 = (m > 0.0 && m < INFINITY) ? e2 : 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = tid + e * NT;
            if (idx < NP * NP) {
                const int i = idx / NP, j = idx - i * NP;
                Cz[i * NPS + j] = ldexp(vals[e], -e2);
            }
        }
        if (tid == 0) {
            cex[z] = cex[zl] + cex[zr] + e2;
            smax[(z + 1) & 1] = 0ull;
        }
        __syncthreads();
    }

    // ---- scan: one segment per 16-lane row ----
    const int lane = tid & 63;
    const int q16 = lane & 15, rb = q16 >> 2, cb = q16 & 3;
    const int rbs[3] = {dpp_i32<DPP_ROW_ROR4>(rb), dpp_i32<DPP_ROW_ROR8>(rb), dpp_i32<DPP_ROW_ROR12>(rb)};
    const Z2Block blk = a.blocks[blockIdx.x];
    const int slot = (tid >> 6) * 4 + (lane >> 4);          // 0..Z2SLOTS-1 within the workgroup
    const bool valid = slot < (int)blk.n;
    const uint32_t seg = blk.seg0 + (valid ? slot : 0);
    const SegDesc sd = a.segs[seg];
    const int len = valid ? (int)sd.len : 0;
    const bool first = (sd.first & SEG_FIRST) != 0;
    const uint8_t *tokp = sd.obs;

    double P[RB][RB];
    {
        const int tok0 = (valid && first) ? (int)tokp[0] : 0;
#pragma unroll
        for (int ii = 0; ii < RB; ++ii)
#pragma unroll
            for (int cc = 0; cc < RB; ++cc) {
                const int i = rb * RB + ii, c = cb * RB + cc;
                double v;
                if (first) v = (c == 0 && i < a.N) ? pi_p[i] * Etg[(size_t)tok0 * a.PP + i] : 0.0;
                else v = (i == c && i < a.N) ? 1.0 : 0.0;
                P[ii][cc] = valid ? v : 0.0;
            }
    }
    int ex = 0;
    const int maxlen = wave_max_i32(len);
    const int nfull = maxlen == 0 ? 0 : wave_min_i32(valid ? len / RESCALE_EVERY : INT_MAX);   // idle wavefronts still join the fold's barriers
    const int head_end = min(RESCALE_EVERY, maxlen);
    for (int t = 0; t < head_end; ++t) {
        const bool act = t < len && !(first && t == 0);   // token 0 of a first segment went into the initial P
        const int tok = (t < len) ? (int)tokp[t] : 0;
        zip2_step<RB, true>(P, C, cex, act ? tok : 0, rb, rbs, act, ex);
    }
    zip2_rescale<RB>(P, ex);
    for (int bi = 1; bi < nfull; ++bi) {
        const uint4 ob = *reinterpret_cast<const uint4 *>(tokp + (size_t)bi * RESCALE_EVERY);
        uint32_t w0 = ob.x, w1 = ob.y, w2 = ob.z, w3 = ob.w;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const uint32_t w = w0;
            w0 = w1; w1 = w2; w2 = w3;
#pragma unroll 1
            for (int u = 0; u < 4; ++u)
                zip2_step<RB, false>(P, C, cex, (w >> (8 * u)) & 0xffu, rb, rbs, true, ex);
        }
        zip2_rescale<RB>(P, ex);
    }
    for (int t = max(RESCALE_EVERY, nfull * RESCALE_EVERY); t < maxlen; ++t) {
        const bool act = t < len;
        const int tok = act ? (int)tokp[t] : 0;
        zip2_step<RB, true>(P, C, cex, tok, rb, rbs, act, ex);
        if (((t + 1) & (RESCALE_EVERY - 1)) == 0) zip2_rescale<RB>(P, ex);
    }
    zip2_rescale<RB>(P, ex);

    // ---- fold the workgroup's segments into one: P_0 <- P_{n-1} ... P_1 P_0 (binary tree through LDS) ----
    // The operator table is dead once every wavefront is here; its space becomes the exchange area, and a
    // fold step is an ordinary token step whose "token operator" is the partner slot's P.
    for (int stride = 1; stride < Z2SLOTS; stride <<= 1) {
        if ((int)blk.n <= stride) break;          // workgroup-uniform
        __syncthreads();                          // table (or previous level's exchange data) no longer read
        if (valid && (slot & stride) && !(slot & (stride - 1))) {   // this slot is a partner ("hi") at this level
            double *dst = C + (size_t)slot * (NP * NP) + (size_t)(rb * RB) * NP + cb * RB;   // exchange rows are unpadded
#pragma unroll
            for (int ii = 0; ii < RB; ++ii)
#pragma unroll
                for (int cc = 0; cc < RB; ++cc) dst[ii * NP + cc] = P[ii][cc];
            if (q16 == 0) cex[slot] = ex;
        }
        __syncthreads();
        const bool act = valid && !(slot & (2 * stride - 1)) && slot + stride < (int)blk.n;
        zip2_step<RB, true, NP>(P, C, cex, act ? slot + stride : 0, rb, rbs, act, ex);
        zip2_rescale<RB>(P, ex);
    }

    if (slot == 0) {
        const size_t gv = (size_t)b * a.n_vecs_total + blk.out_vec0;
        double *Pout = a.P + gv * NP;
        if (blk.first) {
            if (cb == 0) {
#pragma unroll
                for (int ii = 0; ii < RB; ++ii)
                    if (rb * RB + ii < a.N) Pout[rb * RB + ii] = P[ii][0];
            }
            if (q16 == 0) a.EX[gv] = ex;
        } else {
#pragma unroll
            for (int ii = 0; ii < RB; ++ii)
#pragma unroll
                for (int cc = 0; cc < RB; ++cc) {
                    const int i = rb * RB + ii, c = cb * RB + cc;
                    if (i < a.N && c < a.N) Pout[(size_t)i * NP + c] = P[ii][cc];
                }
            if (rb == 0) {
#pragma unroll
                for (int cc = 0; cc < RB; ++cc)
                    if (cb * RB + cc < a.N) a.EX[gv + cb * RB + cc] = ex;
            }
        }
    }
}
