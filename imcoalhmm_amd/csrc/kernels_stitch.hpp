// kernels_stitch.hpp - combine the per-segment results of the propagate kernels into per-chunk
// log-likelihoods.  Included by imcoal_fwd.hip only.
//
// A level is a list of segments per chunk; a chunk's first segment is a VECTOR (N values + one
// exponent), every later one an OPERATOR (N x NP block stored state-major [i][c], one exponent per
// column c).  k_chain folds a run of g consecutive segments into one segment of the next level: a
// run that starts the chunk yields a vector (one chain), any other run yields an operator (N chains,
// one per basis vector).  Applied with g ~ sqrt(K) until every chunk holds a single vector, the
// serial depth drops from K steps to ~2 sqrt(K).  k_finish turns the last vectors into log-likelihoods.
#pragma once
#include "kernels_plain.hpp"

struct ChainDesc {
    uint32_t seg_begin, seg_end;   // input segments [begin, end) of the previous level
    uint32_t c;                    // basis index when the run does not start its chunk
    uint32_t out_vec;              // output vector index in the next level
    uint32_t first;                // 1: the run starts with its chunk's first segment (a vector)
    uint32_t chunk1;               // last level only: 1 + the chunk whose final vector this chain produces (0: none)
};

// EMAX[b][seg] = max_c EX[b][vec0(seg)+c]
__global__ void k_emax(const uint32_t *seg_vec0, const uint8_t *seg_first, uint32_t n_segs, uint32_t n_vecs, int N,
                       const int *EX, int *EMAX)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (s >= n_segs) return;
    const int *e = EX + (size_t)b * n_vecs + seg_vec0[s];
    int m = e[0];
    if (!seg_first[s])
        for (int c = 1; c < N; ++c) m = max(m, e[c]);
    EMAX[(size_t)b * n_segs + s] = m;
}

// One workgroup per (chain, parameter set); thread i owns state i:
//     a <- Op_k * (a .* 2^(ex_k - emax_k))      for k in the run, rescaled by exact powers of two.
// The operators of a run are consecutive N-vector blocks, so their addresses are known up front: the
// rows and exponents of the next D operators are kept in flight in a register ring (D-deep prefetch),
// which takes the L2/HBM latency (~0.7 us) off the serial chain.  With D > 0 (one wavefront, NP <= 64) the
// kernel also finds each unit's largest exponent itself and EMin is not read.
template <int NP, int D>
__global__ __launch_bounds__(((NP + 63) / 64) * 64) void k_chain(
    const ChainDesc *chains, int N,
    const uint32_t *in_vec0, uint32_t in_n_segs, uint32_t in_n_vecs, const double *Pin, const int *EXin, const int *EMin,
    uint32_t out_n_vecs, double *Pout, int *EXout, double *fin_out, int n_chunks)
{
    __shared__ __attribute__((aligned(16))) double w[2][NP];
    // a single-wavefront workgroup needs no s_barrier (LDS is in-order within a wavefront), and a
    // __syncthreads() would drain the prefetch loads with its vmcnt(0)
    constexpr bool ONE_WAVE = NP <= 64;
    auto sync = [&]() { if (ONE_WAVE) wave_fence(); else __syncthreads(); };
    const ChainDesc cd = chains[blockIdx.x];
    const int b = blockIdx.y, i = threadIdx.x;
    const double *Pb = Pin + (size_t)b * in_n_vecs * NP;
    const int *EXb = EXin + (size_t)b * in_n_vecs;
    const int *EMb = EMin + (size_t)b * in_n_segs;
    const bool mine = i < N;
    const int ii = mine ? i : 0;

    double a;
    int etot;
    uint32_t k0 = cd.seg_begin;
    if (cd.first) {
        const uint32_t v = in_vec0[k0];
        a = mine ? Pb[(size_t)v * NP + i] : 0.0;
        etot = EXb[v];
        ++k0;
    } else {
        a = (i == (int)cd.c) ? 1.0 : 0.0;
        etot = 0;
    }
    const int nsteps = (int)cd.seg_end - (int)k0;
    // operator k0+t starts at vector vbase + t*N (every segment after a chunk's first is an operator)
    const uint32_t vbase = nsteps > 0 ? in_vec0[k0] : 0u;

    if constexpr (D == 0) {
        // large NP: no register ring (it would not fit); stream the operator row straight from L2
        int buf = 0;
        for (int t = 0; t < nsteps; ++t) {
            const size_t v0 = (size_t)vbase + (size_t)t * N;
            const int em = EMb[k0 + t];
            if (mine) w[buf][i] = ldexp(a, EXb[v0 + ii] - em);
            else if (i < NP) w[buf][i] = 0.0;
            sync();
            const double2 *row = reinterpret_cast<const double2 *>(Pb + (v0 + ii) * NP);
            const double2 *wv = reinterpret_cast<const double2 *>(&w[buf][0]);
            double acc0 = 0.0, acc1 = 0.0, s0 = 0.0, s1 = 0.0;
#pragma unroll 4
            for (int m = 0; m < NP / 2; ++m) {
                const double2 wc = wv[m], pc = row[m];
                acc0 = fma(pc.x, wc.x, acc0);
                acc1 = fma(pc.y, wc.y, acc1);
                s0 += wc.x;
                s1 += wc.y;
            }
            const double sm = s0 + s1;
            int e = 0;
            (void)frexp(sm, &e);
            e = (sm > 0.0 && sm < INFINITY) ? e : 0;
            a = mine ? ldexp(acc0 + acc1, -e) : 0.0;
            etot += em + e;
            buf ^= 1;
        }
    } else {
    double2 pk[D > 0 ? D : 1][NP / 2];
    int exr[D > 0 ? D : 1];
    auto fetch = [&](int t, int slot) {   // slot is a compile-time constant at every call site
        const size_t v0 = (size_t)vbase + (size_t)t * N;
        // Loads only, D steps ahead of their use.  (Round 2 reduced the exponents to their maximum right here: that
        // waited for the load it had just issued and then ran six ds_bpermute round trips - ~0.5 us in front of every
        // step of the serial chain, most of the 0.65 us a step took.)  Lanes beyond N repeat column 0.
        exr[slot] = EXb[v0 + ii];
        const double2 *row = reinterpret_cast<const double2 *>(Pb + (v0 + ii) * NP);
#pragma unroll
        for (int m = 0; m < NP / 2; ++m) pk[slot][m] = row[m];
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nsteps) fetch(s, s);   // (D > 0 here)
    int buf = 0;
    for (int t0 = 0; t0 < nsteps; t0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int t = t0 + s;
            if (t < nsteps) {   // wave-uniform
                // the unit's largest column exponent, found here instead of by a k_emax launch per level: seven DPP moves
                const int em = wave_max_i32_dpp(exr[s]);
                if (mine) w[buf][i] = ldexp(a, exr[s] - em);
                else if (i < NP) w[buf][i] = 0.0;
                double2 cur[NP / 2];
#pragma unroll
                for (int m = 0; m < NP / 2; ++m) cur[m] = pk[s][m];
                if (t + D < nsteps) fetch(t + D, s);
                sync();
                double acc0 = 0.0, acc1 = 0.0, s0 = 0.0, s1 = 0.0;
                const double2 *wv = reinterpret_cast<const double2 *>(&w[buf][0]);
#pragma unroll
                for (int m = 0; m < NP / 2; ++m) {
                    const double2 wc = wv[m];
                    acc0 = fma(cur[m].x, wc.x, acc0);
                    acc1 = fma(cur[m].y, wc.y, acc1);
                    s0 += wc.x;
                    s1 += wc.y;
                }
                const double sm = s0 + s1;
                int e = 0;
                (void)frexp(sm, &e);
                e = (sm > 0.0 && sm < INFINITY) ? e : 0;
                a = mine ? ldexp(acc0 + acc1, -e) : 0.0;
                etot += em + e;
                buf ^= 1;   // the next step writes the other buffer: one fence per step suffices
            }
        }
    }
    }
    // normalise the result by its total and store it as a segment of the next level
    sync();
    if (i < NP) w[0][i] = mine ? a : 0.0;
    sync();
    double tot = 0.0;
    for (int c = 0; c < N; ++c) tot += w[0][c];
    int e = 0;
    (void)frexp(tot, &e);
    e = (tot > 0.0 && tot < INFINITY) ? e : 0;
    const size_t gv = (size_t)b * out_n_vecs + cd.out_vec;
    if (mine) {
        double *dst = cd.first ? Pout + gv * NP + i : Pout + (gv - cd.c) * NP + (size_t)i * NP + cd.c;
        *dst = ldexp(a, -e);
    }
    if (i == 0) {
        EXout[gv] = etot + e;
        // last level: the chunk's log-likelihood straight from here (what k_finish computes from the stored vector:
        // the sum of the normalised entries is this sum scaled by the same exact power of two)
        if (fin_out && cd.chunk1)
            fin_out[(size_t)b * n_chunks + (cd.chunk1 - 1)] = (double)(etot + e) * 0.693147180559945309417232121458 + log(ldexp(tot, -e));
    }
}

// out[b][f] = log-likelihood of chunk f from its single remaining vector (0.0 for an empty chunk).
__global__ void k_finish(const int32_t *final_vec, int n_chunks, int N, int NP, uint32_t n_vecs, const double *P,
                         const int *EX, double *out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (f >= n_chunks) return;
    const int32_t v = final_vec[f];
    double r = 0.0;
    if (v >= 0) {
        const double *p = P + ((size_t)b * n_vecs + v) * NP;
        double tot = 0.0;
        for (int c = 0; c < N; ++c) tot += p[c];
        r = (double)EX[(size_t)b * n_vecs + v] * 0.693147180559945309417232121458 + log(tot);
    }
    out[(size_t)b * n_chunks + f] = r;
}

// State export: chunk f's single remaining unit, unreduced.  op == 0: its vector a[i] (N doubles, one exponent);
// op == 1: its transfer operator P[i][c] (N x N row-major, one exponent per column c).  Value = stored * 2^exponent.
__global__ void k_export(const int32_t *final_vec, int n_chunks, int N, int NP, uint32_t n_vecs, const double *P,
                         const int *EX, int op, double *out_state, int *out_exp)
{
    const int f = blockIdx.x, b = blockIdx.y;
    const int32_t v = final_vec[f];
    if (v < 0) return;
    const double *src = P + ((size_t)b * n_vecs + v) * NP;
    const int *ex = EX + (size_t)b * n_vecs + v;
    if (op) {
        double *dst = out_state + ((size_t)b * n_chunks + f) * N * N;
        int *de = out_exp + ((size_t)b * n_chunks + f) * N;
        for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
            const int i = idx / N, c = idx - i * N;
            dst[idx] = src[(size_t)i * NP + c];
        }
        for (int c = threadIdx.x; c < N; c += blockDim.x) de[c] = ex[c];
    } else {
        double *dst = out_state + ((size_t)b * n_chunks + f) * N;
        for (int i = threadIdx.x; i < N; i += blockDim.x) dst[i] = src[i];
        if (threadIdx.x == 0) out_exp[(size_t)b * n_chunks + f] = ex[0];
    }
}

// Parameter upload by kernel: `src` is the device-visible alias of a mapped, pinned staging slot on the host.
__global__ void k_stage_params(const double2 *__restrict__ src, double2 *__restrict__ dst, unsigned n2)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}

// partial[b] = sum_f per_chunk[b][f], left to right from 0.0 (likelihood.py:33)
__global__ void k_sum_chunks(const double *per_chunk, int n_chunks, int B, double *partial)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double t = 0.0;
    for (int f = 0; f < n_chunks; ++f) t += per_chunk[(size_t)b * n_chunks + f];
    partial[b] = t;
}
