// kernels_zip.hpp - forward propagate over the pair-compressed token stream (zipHMM-style).
// Included by imcoal_fwd.hip only.
//
// Every token z of the stream stands for a fixed run of raw columns and owns one N x N operator
//     C_s = diag(E[:,s]) T'            for a raw symbol s        (C_s[i][j] = E[i][s] T[j][i])
//     C_z = C_right(z) * C_left(z)     for a merged token z       (left is applied first)
// (what ziphmm.zip_forward builds per evaluation from sym2pair, reference call site
// src/IMCoalHMM/hmm.py:20-21).  The whole table (A operators, each rescaled by an exact power of
// two whose exponent is kept in cex[z]) is built by every workgroup in its prologue and stays in the
// CU's LDS for the launch; the scan then costs one dense N x N mat-vec per TOKEN instead of per column.
//
// Lane layout as in the plain kernel (G lanes per vector, 64/G vectors per wavefront), except that a
// lane owns the interleaved states {r, r+G, r+2G, ...}: the G lanes of a vector then read G consecutive
// operator rows per instruction, which spreads over the LDS banks without padding for N=20.
#pragma once
#include "kernels_plain.hpp"

static constexpr int ZWAVES = 16;   // wavefronts per workgroup (1024 threads, one workgroup per CU)

// Row stride (doubles) of an operator in LDS: smallest even pad that keeps the G row starts of one
// ds_read_b128 on distinct 16-byte bank slots (64 banks x 4 B).
constexpr int zip_row_stride(int NP, int G)
{
    int best = NP, best_worst = 1 << 30;
    for (int pad = 0; pad <= 6; pad += 2) {
        const int nps = NP + pad;
        int worst = 0;
        for (int slot = 0; slot < 16; ++slot) {
            int cnt = 0;
            for (int r = 0; r < G && r < 16; ++r)
                if ((((r * nps * 2) % 64) / 4) == slot) ++cnt;
            if (cnt > worst) worst = cnt;
        }
        if (worst < best_worst) { best_worst = worst; best = nps; }
    }
    return best;
}

template <int R, int G>
struct ZipGeom {
    static constexpr int NP = R * G;
    static constexpr int NPS = zip_row_stride(NP, G);
    static constexpr int VPW = 64 / G;
    // LDS bytes for an alphabet of A tokens
    static constexpr size_t lds_bytes(int A)
    {
        return ((size_t)A * NP * NPS + (size_t)ZWAVES * VPW * NP) * 8 + (size_t)((A + 1) & ~1) * 4 + 16;
    }
};

// One token for the R interleaved states of this lane:  x <- C_tok x.
template <int R, int G, bool PRED, bool SUM>
__device__ __forceinline__ void zip_step(double (&xo)[R], double *xw, int r, const double *C, const int *cex,
                                         int tok, bool act, int &ex, double &s)
{
    constexpr int NP = R * G;
    constexpr int NPS = ZipGeom<R, G>::NPS;
    const int ce = cex[tok];
#pragma unroll
    for (int k = 0; k < R; ++k) xw[r + G * k] = xo[k];
    wave_fence();
    const double *Cz = C + (size_t)tok * (NP * NPS) + r * NPS;
    double acc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0.0;
    if (SUM) s = 0.0;
    const double2 *xv = reinterpret_cast<const double2 *>(xw);
#pragma unroll
    for (int m = 0; m < NP / 2; ++m) {
        const double2 t = xv[m];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double2 c = reinterpret_cast<const double2 *>(Cz + G * k * NPS)[m];
            acc[k] = fma(c.x, t.x, acc[k]);
            acc[k] = fma(c.y, t.y, acc[k]);
        }
        if (SUM) s += t.x;
        if (SUM) s += t.y;
    }
    wave_fence();
#pragma unroll
    for (int k = 0; k < R; ++k) xo[k] = (PRED && !act) ? xo[k] : acc[k];
    ex += (PRED && !act) ? 0 : ce;
}

template <int R, int G>
__global__ __launch_bounds__(ZWAVES * 64) void k_zpropagate(PropArgs a)
{
    using Geo = ZipGeom<R, G>;
    constexpr int NP = Geo::NP, NPS = Geo::NPS, VPW = Geo::VPW, NT = ZWAVES * 64;
    constexpr int EPT = (NP * NP + NT - 1) / NT;   // operator elements per thread in the table build
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *C = lds;                                         // [A][NP][NPS]
    double *xb = C + (size_t)a.A * NP * NPS;                 // [ZWAVES][VPW][NP]
    int *cex = reinterpret_cast<int *>(xb + ZWAVES * VPW * NP);   // [A]
    unsigned long long *smax = reinterpret_cast<unsigned long long *>(cex + ((a.A + 1) & ~1));   // [2]

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *pi_p = pp;
    const double *Tp = pp + NP;
    const double *Etg = pp + NP + NP * NP;

    // ---- operator table: raw symbols, then merged tokens in dictionary order ----
    for (int idx = tid; idx < a.S * NP * NP; idx += NT) {
        const int sidx = idx / (NP * NP);
        const int rem = idx - sidx * NP * NP;
        const int i = rem / NP, j = rem - i * NP;
        C[((size_t)sidx * NP + i) * NPS + j] = Etg[sidx * NP + i] * Tp[j * NP + i];
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
            mx = (acc > mx || acc != acc) ? acc : mx;   // NaN propagates into the max
        }
        // wavefront max, then one LDS atomic per wavefront (bit pattern order == value order for x >= 0)
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
            smax[(z + 1) & 1] = 0ull;   // the other slot is idle during this iteration
        }
        __syncthreads();
    }

    // ---- scan ----
    const int lane = tid & 63, wave = tid >> 6;
    int v = lane / G;
    const int r = lane - v * G;
    const bool spare = v >= VPW;
    v = spare ? VPW - 1 : v;
    double *xw = xb + (wave * VPW + v) * NP;

    const uint32_t vid = (blockIdx.x * ZWAVES + wave) * VPW + v;
    const bool valid = vid < a.n_vecs;
    const VecDesc vd = a.vecs[min(vid, a.n_vecs - 1u)];
    const SegDesc sd = a.segs[vd.seg];
    const int len = valid ? (int)sd.len : 0;
    const bool first = (sd.first & SEG_FIRST) != 0;
    const uint8_t *tokp = sd.obs;

    double xo[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = r + G * k;
        xo[k] = valid ? (first ? pi_p[i] : (i == (int)vd.c ? 1.0 : 0.0)) : 0.0;
    }
    int ex = 0;
    double s = 0.0;
    const int maxlen = wave_max_i32(len);
    const int nfull = wave_min_i32(valid ? len / RESCALE_EVERY : INT_MAX);
    if (maxlen == 0) return;

    // head: first 16 tokens one by one; token 0 of a first segment is the raw first column: x = pi .* E[:,o_0]
    const int head_end = min(RESCALE_EVERY, maxlen);
    for (int t = 0; t < head_end; ++t) {
        const bool act = t < len;
        const int tok = act ? (int)tokp[t] : 0;
        if (t == 0) {
            double y[R];
            const int sym0 = (act && first) ? tok : 0;   // only a first segment's token 0 is a raw symbol
#pragma unroll
            for (int k = 0; k < R; ++k) y[k] = xo[k] * Etg[sym0 * NP + r + G * k];
            double dummy = 0.0;
            int ex0 = 0;
            zip_step<R, G, true, true>(xo, xw, r, C, cex, tok, act && !first, ex0, dummy);
            ex += (act && !first) ? ex0 : 0;
#pragma unroll
            for (int k = 0; k < R; ++k) xo[k] = (act && first) ? y[k] : xo[k];
        } else {
            zip_step<R, G, true, true>(xo, xw, r, C, cex, tok, act, ex, s);
            rescale<R>(xo, s, ex);
        }
    }
    // body: full 16-token blocks common to all vectors of the wavefront
    for (int blk = 1; blk < nfull; ++blk) {
        const uint4 ob = *reinterpret_cast<const uint4 *>(tokp + (size_t)blk * RESCALE_EVERY);
        uint32_t w0 = ob.x, w1 = ob.y, w2 = ob.z, w3 = ob.w;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const uint32_t w = w0;
            w0 = w1; w1 = w2; w2 = w3;
            zip_step<R, G, false, false>(xo, xw, r, C, cex, w & 0xffu, true, ex, s);
            zip_step<R, G, false, false>(xo, xw, r, C, cex, (w >> 8) & 0xffu, true, ex, s);
            zip_step<R, G, false, false>(xo, xw, r, C, cex, (w >> 16) & 0xffu, true, ex, s);
            zip_step<R, G, false, true>(xo, xw, r, C, cex, w >> 24, true, ex, s);
        }
        rescale<R>(xo, s, ex);
    }
    // tail
    for (int t = max(RESCALE_EVERY, nfull * RESCALE_EVERY); t < maxlen; ++t) {
        const bool act = t < len;
        const int tok = act ? (int)tokp[t] : 0;
        zip_step<R, G, true, true>(xo, xw, r, C, cex, tok, act, ex, s);
        rescale<R>(xo, s, ex);
    }
    // final normalisation
#pragma unroll
    for (int k = 0; k < R; ++k) xw[r + G * k] = xo[k];
    wave_fence();
    s = sum_vec<NP>(xw);
    wave_fence();
    rescale<R>(xo, s, ex);

    if (valid && !spare) {
        const size_t gv = (size_t)b * a.n_vecs_total + a.vec_base + vid;
        double *Pout = first ? a.P + gv * NP : a.P + (gv - vd.c) * NP + vd.c;   // see kernels_plain.hpp
        const int st = first ? 1 : NP;
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (r + G * k < a.N) Pout[(size_t)(r + G * k) * st] = xo[k];
        if (r == 0) a.EX[gv] = ex;
    }
}
