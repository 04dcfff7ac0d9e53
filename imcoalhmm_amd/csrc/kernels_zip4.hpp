// kernels_zip4.hpp - the register-blocked MFMA kernel (kernels_zip3.hpp) with a HYBRID operator table: a dictionary far
// larger than LDS can hold (up to 256 tokens at N = 20, ~100 alignment columns per token instead of ~61 with the 44
// that fit), its operators in a global table that stays in L2 (256 x 3.2 KB = 0.8 MB), the most frequent ones also
// cached in LDS.  Included by imcoal_fwd.hip only.
//
//   * k_z4_raw / k_z4_level<NT>: the table in global memory, one launch per dictionary depth, each token the MFMA step
//     itself on its children's entries (four tokens per wavefront, as in k_zpropagate3's prologue).
//   * k_zpropagate4<NT>: k_zpropagate3's scan.  Every workgroup copies the H hottest operators (+ the identity) from
//     the global table into LDS.  A step whose token is hot reads its A operands from LDS one tile-row ahead, exactly
//     as k_zpropagate3; a step whose token is cold takes them from 25 registers per lane that were loaded from the
//     global table ONE STEP (NT^3 MFMAs, ~1 us) ahead - tile-row I of the next cold operator is requested right after
//     tile-row I of the current one has been consumed, so a single register set suffices and an L2 hit is never
//     waited for.  Measured feasibility (scratch microbenchmark, 3.3 MB table, 85 % of the steps cold): +10 % per step.
#pragma once
#include "kernels_zip3.hpp"

template <int NT>
struct Zip4Geom {
    using G3 = Zip3Geom<NT>;
    static constexpr int TOK = G3::TOK;
    // LDS: `slots` operator entries (hot operators + identity; later the fold's exchange area), the slot map and the
    // exponents of all A + 1 table entries
    static constexpr int slots(int H) { return (H + 1 > Z2SLOTS + 1 ? H + 1 : Z2SLOTS + 1); }
    static constexpr size_t lds_bytes(int A, int H) { return (size_t)slots(H) * TOK * 8 + (size_t)(2 * (A + 2) + 64) * 4 + 16; }
    // how many hot operators fit beside the identity
    static constexpr int max_hot(int A, size_t budget)
    {
        const size_t fixed = (size_t)(2 * (A + 2) + 64) * 4 + 16;
        const size_t n = budget > fixed ? (budget - fixed) / ((size_t)TOK * 8) : 0;
        return n > 1 ? (int)n - 1 : 0;
    }
};

// Pout <- C * Pin with the A operands of all tile-rows in registers (Areg[I][K] = this lane's C[4I + r][4K + q]).
template <int NT>
__device__ __forceinline__ void zip4_step_regs(const double (&Pin)[NT][NT], double (&Pout)[NT][NT], const double (&Areg)[NT][NT])
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J)
                Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(Areg[I][K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
}

// this lane's A operands of tile-row I of the table entry at Gz (global memory, the LDS entries' own layout)
template <int NT>
__device__ __forceinline__ void zip4_load_row_global(double (&a)[NT], const double *Gz, int I, int lo, int lx)
{
    using Geo = Zip3Geom<NT>;
    const double2 *m = reinterpret_cast<const double2 *>(Gz + I * 16 * Geo::NTE + lo);
#pragma unroll
    for (int k2 = 0; k2 < Geo::NTE / 2; ++k2) {
        const double2 v = m[k2];
        a[2 * k2] = v.x;
        a[2 * k2 + 1] = v.y;
    }
    if constexpr (NT & 1) a[NT - 1] = Gz[Geo::MAIN + I * 16 + lx];
}

// ---- the global table ---------------------------------------------------------------------------------------------
// a.Ctab: [B][A + 1][TOK] (entry A = identity), a.cex: [B][A + 1].  k_z4_raw writes the raw symbols' operators and the
// identity; k_z4_level, launched once per dictionary depth, builds that depth's merged tokens - one MFMA block per
// token, four per wavefront, one wavefront per workgroup so that a depth's ~10-70 tokens spread over as many CUs - from
// its children's entries, which earlier launches left in L2.  A depth is latency, not work (children in, 125 MFMAs,
// rescale, entry out: ~3.5 us), and plain kernel boundaries turned out to be the cheapest way to order the depths:
// a single-workgroup build with barriers (with or without forwarding fresh operators through LDS) took 72-82 us for
// 253 tokens in 10 depths, because every pass then serialises behind the slowest of eight wavefronts' latency chains.
template <int NT>
__global__ __launch_bounds__(256) void k_z4_raw(BigArgs a)
{
    using Geo = Zip3Geom<NT>;
    constexpr int NP = Geo::NP, TOK = Geo::TOK;
    const int b = blockIdx.y, sidx = blockIdx.x;                    // one workgroup per raw symbol, one more for the identity
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *Tp = pp + a.PP;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    double *Gt = a.Ctab + (size_t)b * (a.A + 1) * TOK;
    int *Gc = a.cex + (size_t)b * (a.A + 1);
    const int entry = sidx < a.S ? sidx : a.A;
    for (int idx = threadIdx.x; idx < NP * NP; idx += blockDim.x) {
        const int i = idx / NP, j = idx - i * NP;
        Gt[(size_t)entry * TOK + Geo::idx(i, j)] = sidx < a.S ? Etg[(size_t)sidx * a.PP + i] * Tp[(size_t)j * a.PP + i] : (i == j ? 1.0 : 0.0);
    }
    if (threadIdx.x == 0) Gc[entry] = 0;
}

template <int NT>
__global__ __launch_bounds__(64) void k_z4_level(BigArgs a, int first, int count)
{
    using Geo = Zip3Geom<NT>;
    constexpr int TOK = Geo::TOK;
    const int b = blockIdx.y, lane = threadIdx.x;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * Geo::NTE, lx = q * 4 + r;
    double *Gt = a.Ctab + (size_t)b * (a.A + 1) * TOK;
    int *Gc = a.cex + (size_t)b * (a.A + 1);
    const int ti = blockIdx.x * 4 + bq;
    const bool have = ti < count;
    const int4 td = have ? a.tab_desc[first + ti] : make_int4(a.A, a.A, a.A, 0);   // (an idle block multiplies identities)
    const int z = td.x, zl = td.y, zr = td.z;
    const double *Gl = Gt + (size_t)zl * TOK, *Gr = Gt + (size_t)zr * TOK;
    double Bt[NT][NT], Ar[NT][NT], Out[NT][NT];
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J) Bt[K][J] = Gl[Geo::idx(4 * K + q, 4 * J + r)];
#pragma unroll
    for (int I = 0; I < NT; ++I) zip4_load_row_global<NT>(Ar[I], Gr, I, lo, lx);
    const int cel = Gc[zl], cer = Gc[zr];
    zip4_step_regs<NT>(Bt, Out, Ar);
    int e2 = 0;
    zip3_rescale<NT>(Out, e2);
    if (have) {
        double *Gz = Gt + (size_t)z * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) Gz[Geo::idx(4 * K + q, 4 * J + r)] = Out[K][J];
        if (q == 0 && r == 0) Gc[z] = cel + cer + e2;
    }
}

// TWO dictionary depths per launch.  A token of the second depth has at least one child in the first, which a sibling
// workgroup of the same launch is only just building - so the wavefront RECOMPUTES that child from the grandchildren
// (both older than the launch), keeps the product in registers (a left child: a result tile is already in the layout
// of a B operand) or turns it round through LDS (a right child is needed as A operand rows), and goes on to the token
// itself.  Scaling by powers of two is exact, so the un-normalised recomputed child gives the same bits as the stored,
// normalised one (the exponents are carried separately).  An entry is two int4: {token, left, right, flags} and the
// grandchildren {ll, lr, rl, rr}; flags bit 0 / 1: recompute the left / right child; 0: a first-depth token, built as
// k_z4_level does.  The host sorts a launch's entries so that the four of a wavefront share their flags (padded with
// idle entries), which keeps the three phases wavefront-uniform; a mixed wavefront is still correct (a child that
// needs no recomputing is "recomputed" as child x identity, which is exact).
// Saves a kernel boundary (~4.9 us) per pair of depths for ~1.5-3 us of extra products: 12 -> 7 launches at 4096 tokens.
// FIRST = true: the evaluation's first launch (depths 1 and 2).  Every leaf of those tokens is a raw symbol or the
// identity, so the wavefront builds the S raw operators C_s = diag(E[:,s]) T' (and the identity) itself, in LDS, straight
// from the caller's parameters - `params_src` is the device-visible alias of the mapped staging slot on the host - and
// never reads the table; workgroup 0 also writes the raw entries and the identity into the table and the parameters into
// a.params for the launches that follow.  This replaces k_stage_params + k_z4_raw + the two host-paced gaps between
// three very short kernels (rocprofv3 timeline, profiles/r03_trace_config1.txt: 3.4 + 4.0 + 2.4 + 3.8 us).
template <int NT, bool FIRST>
__global__ __launch_bounds__(64) void k_z4_level2(BigArgs a, const int4 *desc2, int first, int count, const double *params_src)
{
    using Geo = Zip3Geom<NT>;
    constexpr int TOK = Geo::TOK, NP = Geo::NP;
    __shared__ __attribute__((aligned(16))) double turn[4 * TOK];          // one entry per MFMA block: D layout in, A rows out
    extern __shared__ __attribute__((aligned(16))) double rawtab[];        // FIRST: the parameter set (a.pstride doubles)
    const int b = blockIdx.y, lane = threadIdx.x;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * Geo::NTE, lx = q * 4 + r;
    double *Gt = a.Ctab + (size_t)b * (a.A + 1) * TOK;
    int *Gc = a.cex + (size_t)b * (a.A + 1);
    const int IDENT = a.A;
    if constexpr (FIRST) {
        const double *src = params_src + (size_t)b * a.pstride;
        double *lp = rawtab;                                               // a.pstride doubles (even)
        // (all of a lane's loads in flight together: `src` is host memory, ~2 us per round trip over PCIe)
        for (int k0 = 0; k0 < (int)a.pstride; k0 += 8 * 128) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * 128 + lane * 2;
                v[u] = k < (int)a.pstride ? *reinterpret_cast<const double2 *>(src + k) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * 128 + lane * 2;
                if (k < (int)a.pstride) *reinterpret_cast<double2 *>(lp + k) = v[u];
            }
        }
        wave_fence();
        if (blockIdx.x == 0) {                                            // (wavefront-uniform)
            double *dp = const_cast<double *>(a.params) + (size_t)b * a.pstride;
            for (int k = lane * 2; k < (int)a.pstride; k += 128) *reinterpret_cast<double2 *>(dp + k) = *reinterpret_cast<const double2 *>(lp + k);
        }
        // the table's raw entries and the identity, dealt over the launch's workgroups (later launches read them)
        const double *Tp = lp + a.PP, *Etg = lp + a.PP + (size_t)a.PP * a.PP;
        for (int s = blockIdx.x; s <= a.S; s += gridDim.x) {
            double *Gz = Gt + (size_t)(s < a.S ? s : IDENT) * TOK;
            for (int idx = lane; idx < NP * NP; idx += 64) {
                const int i = idx / NP, j = idx - i * NP;
                Gz[Geo::idx(i, j)] = s < a.S ? Etg[(size_t)s * a.PP + i] * Tp[(size_t)j * a.PP + i] : (i == j ? 1.0 : 0.0);
            }
            if (lane == 0) Gc[s < a.S ? s : IDENT] = 0;
        }
    }
    const int ti = blockIdx.x * 4 + bq;
    int4 d0 = ti < count ? desc2[2 * (first + ti)] : make_int4(-1, 0, 0, 0);
    int4 d1 = ti < count ? desc2[2 * (first + ti) + 1] : make_int4(0, 0, 0, 0);
    const bool have = d0.x >= 0;                                       // (token -1: a padding entry - multiplies identities, stores nothing)
    if (!have) { d0 = make_int4(IDENT, IDENT, IDENT, 0); d1 = make_int4(IDENT, IDENT, IDENT, IDENT); }
    const int z = d0.x, zl = d0.y, zr = d0.z;
    const bool needL = (d0.w & 1) != 0, needR = (d0.w & 2) != 0;
    const bool anyL = __any(needL), anyR = __any(needR);              // wavefront-uniform

    // a leaf operand: its table entry; in the first launch every leaf is a raw symbol's operator C_s[i][j] = E[s][i] T[j][i]
    // or the identity, formed on the fly from the parameters in LDS
    const double *lp = rawtab;
    auto leaf = [&](int tok, int i, int j) __attribute__((always_inline)) {
        const double *Tp = lp + a.PP, *Etg = lp + a.PP + (size_t)a.PP * a.PP;
        const double v = Etg[(size_t)(tok == IDENT ? 0 : tok) * a.PP + i] * Tp[(size_t)j * a.PP + i];
        return tok == IDENT ? (i == j ? 1.0 : 0.0) : v;
    };
    auto expo = [&](int tok) __attribute__((always_inline)) { if constexpr (FIRST) return 0; else return Gc[tok]; };
    auto load_B = [&](double (&Bt)[NT][NT], int tok) __attribute__((always_inline)) {
        const double *G = Gt + (size_t)tok * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                if constexpr (FIRST) Bt[K][J] = leaf(tok, 4 * K + q, 4 * J + r);
                else Bt[K][J] = G[Geo::idx(4 * K + q, 4 * J + r)];
            }
    };
    auto load_A = [&](double (&Ar)[NT][NT], int tok) __attribute__((always_inline)) {     // Ar[I][K] = C[4I + r][4K + q]
        const double *G = Gt + (size_t)tok * TOK;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            if constexpr (FIRST) {
#pragma unroll
                for (int K = 0; K < NT; ++K) Ar[I][K] = leaf(tok, 4 * I + r, 4 * K + q);
            } else zip4_load_row_global<NT>(Ar[I], G, I, lo, lx);
        }
    };

    // ---- left child as the B operand ----
    // (hoisting the right child's loads above the left child's 125 MFMAs was tried in round 3: 324 registers, one
    // wavefront per SIMD, launches 7.4-10.4 us instead of 6.8-9.4 - no gain)
    double L[NT][NT];
    int eL;
    if (anyL) {
        const int t0 = needL ? d1.x : zl, t1 = needL ? d1.y : IDENT;      // L = C_t1 * C_t0
        double Bt[NT][NT], Ar[NT][NT];
        load_B(Bt, t0);
        load_A(Ar, t1);
        eL = expo(t0) + expo(t1);
        zip4_step_regs<NT>(Bt, L, Ar);
    } else {
        load_B(L, zl);
        eL = expo(zl);
    }
    // ---- right child as A operand rows ----
    double R[NT][NT];
    int eR;
    if (anyR) {
        const int t0 = needR ? d1.z : zr, t1 = needR ? d1.w : IDENT;      // R = C_t1 * C_t0
        double Bt[NT][NT], Ar[NT][NT], Out[NT][NT];
        load_B(Bt, t0);
        load_A(Ar, t1);
        eR = expo(t0) + expo(t1);
        zip4_step_regs<NT>(Bt, Out, Ar);
        double *mine = turn + bq * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) mine[Geo::idx(4 * K + q, 4 * J + r)] = Out[K][J];
        wave_fence();
#pragma unroll
        for (int I = 0; I < NT; ++I) zip3_load_row<NT>(R[I], mine, I, lo, lx);
    } else {
        load_A(R, zr);
        eR = expo(zr);
    }
    // ---- the token itself ----
    double Out[NT][NT];
    zip4_step_regs<NT>(L, Out, R);
    int e2 = 0;
    zip3_rescale<NT>(Out, e2);
    if (have) {
        double *Gz = Gt + (size_t)z * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) Gz[Geo::idx(4 * K + q, 4 * J + r)] = Out[K][J];
        if (q == 0 && r == 0) Gc[z] = eL + eR + e2;
    }
}

// THREE dictionary depths per launch (k_z4_level3): one WAVEFRONT per token.  A token of depths D+1 .. D+3 is the ordered
// product of at most eight entries of depth <= D (its children, grandchildren or great-grandchildren - whichever are
// already in the table; the host lists them as the eight leaves of a complete binary tree, padded with the identity:
// a ready node sits in the first leaf of its range).  The wavefront's four MFMA blocks form the four leaf pairs at once,
// then fold 1 -> 0 and 3 -> 2, then 2 -> 0 through LDS (zip3_fold's in-wavefront stage): three dependent steps for three
// depths, where k_z4_level2 spends three dependent products on two.  The bracketing is the dictionary's own, and
// powers of two scale exactly, so every entry has the bits the one-depth-per-launch build gives it.
// FIRST: the evaluation's first launch - every leaf is a raw symbol's operator or the identity, formed on the fly from
// the parameters, which the workgroup fetches from the caller's mapped staging slot (see k_z4_level2<., true>).
constexpr int Z4L3_WAVES = 2;       // wavefronts (= tokens) per workgroup: 32 KB of exchange areas at N = 20, five workgroups per CU
template <int NT>
struct Z4L3Geom {
    static constexpr int AREAS = Z4L3_WAVES * 4 + 2;
    static constexpr size_t PARAMS_AT = (size_t)AREAS * Zip3Geom<NT>::TOK + ((AREAS + 3) / 4) * 2;   // doubles, 16-byte aligned
    static constexpr size_t lds_bytes(size_t pstride) { return (PARAMS_AT + pstride) * 8; }
};
template <int NT, bool FIRST>
__global__ __launch_bounds__(Z4L3_WAVES * 64, 3) void k_z4_level3(BigArgs a, const int4 *desc3, int first, int count, const double *params_src)
{
    using Geo = Zip3Geom<NT>;
    constexpr int TOK = Geo::TOK, NP = Geo::NP, THREADS = Z4L3_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) double l3lds[];         // [waves][4][TOK] exchange areas | [waves][4] ints | FIRST: parameters
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * Geo::NTE, lx = q * 4 + r;
    double *Gt = a.Ctab + (size_t)b * (a.A + 1) * TOK;
    int *Gc = a.cex + (size_t)b * (a.A + 1);
    const int IDENT = a.A;
    // (two spare areas: zip3_fold's idle blocks read - and discard - up to two areas beyond their wavefront's four)
    double *X = l3lds + (size_t)wv * 4 * TOK;
    int *xe = reinterpret_cast<int *>(l3lds + (size_t)Z4L3Geom<NT>::AREAS * TOK) + wv * 4;
    const double *lp = l3lds + Z4L3Geom<NT>::PARAMS_AT;                   // FIRST: a.pstride doubles behind the ints
    if constexpr (FIRST) {
        const double *src = params_src + (size_t)b * a.pstride;
        double *lpw = const_cast<double *>(lp);
        // (all of a lane's loads in flight together: `src` is host memory, ~2 us per round trip over PCIe)
        for (int k0 = 0; k0 < (int)a.pstride; k0 += 2 * 2 * THREADS) {
            double2 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + (u * THREADS + tid) * 2;
                v[u] = k < (int)a.pstride ? *reinterpret_cast<const double2 *>(src + k) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + (u * THREADS + tid) * 2;
                if (k < (int)a.pstride) *reinterpret_cast<double2 *>(lpw + k) = v[u];
            }
        }
        __syncthreads();
        if (blockIdx.x == 0) {
            double *dp = const_cast<double *>(a.params) + (size_t)b * a.pstride;
            for (int k = tid * 2; k < (int)a.pstride; k += 2 * THREADS) *reinterpret_cast<double2 *>(dp + k) = *reinterpret_cast<const double2 *>(lp + k);
        }
        // the table's raw entries and the identity, dealt over the launch's workgroups (later launches read them)
        const double *Tp = lp + a.PP, *Etg = lp + a.PP + (size_t)a.PP * a.PP;
        for (int s = blockIdx.x; s <= a.S; s += gridDim.x) {
            double *Gz = Gt + (size_t)(s < a.S ? s : IDENT) * TOK;
            for (int idx = tid; idx < NP * NP; idx += THREADS) {
                const int i = idx / NP, j = idx - i * NP;
                Gz[Geo::idx(i, j)] = s < a.S ? Etg[(size_t)s * a.PP + i] * Tp[(size_t)j * a.PP + i] : (i == j ? 1.0 : 0.0);
            }
            if (tid == 0) Gc[s < a.S ? s : IDENT] = 0;
        }
    }
    const int ti = blockIdx.x * Z4L3_WAVES + wv;                          // (wavefront-uniform)
    if (ti >= count) return;
    const int4 d0 = desc3[3 * (first + ti)], d1 = desc3[3 * (first + ti) + 1], d2 = desc3[3 * (first + ti) + 2];
    const int z = d0.x;
    const int leaves[8] = {d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x};
    int t0 = leaves[0], t1 = leaves[1];                                   // this block's pair: leaves 2 bq, 2 bq + 1
#pragma unroll
    for (int k = 1; k < 4; ++k) { t0 = bq == k ? leaves[2 * k] : t0; t1 = bq == k ? leaves[2 * k + 1] : t1; }
    auto leaf = [&](int tok, int i, int j) __attribute__((always_inline)) {
        const double *Tp = lp + a.PP, *Etg = lp + a.PP + (size_t)a.PP * a.PP;
        const double v = Etg[(size_t)(tok == IDENT ? 0 : tok) * a.PP + i] * Tp[(size_t)j * a.PP + i];
        return tok == IDENT ? (i == j ? 1.0 : 0.0) : v;
    };
    double Bt[NT][NT], P[NT][NT];
    int ex;
    {
        const double *G0 = Gt + (size_t)t0 * TOK, *G1 = Gt + (size_t)t1 * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                if constexpr (FIRST) Bt[K][J] = leaf(t0, 4 * K + q, 4 * J + r);
                else Bt[K][J] = G0[Geo::idx(4 * K + q, 4 * J + r)];
            }
        if constexpr (FIRST) ex = 0;
        else ex = Gc[t0] + Gc[t1];
        // P = C_t1 * C_t0, the A operand's tile-rows one row ahead of their use (ten registers instead of fifty)
        auto row = [&](double (&av)[NT], int I) __attribute__((always_inline)) {
            if constexpr (FIRST) {
#pragma unroll
                for (int K = 0; K < NT; ++K) av[K] = leaf(t1, 4 * I + r, 4 * K + q);
            } else zip4_load_row_global<NT>(av, G1, I, lo, lx);
        };
        double av[NT], an[NT];
        row(av, 0);
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            if (I + 1 < NT) row(an, I + 1);
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    P[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[K], Bt[K][J], K == 0 ? 0.0 : P[I][J], 0, 0, 0);
#pragma unroll
            for (int K = 0; K < NT; ++K) av[K] = an[K];
        }
    }
    // (no rescale between the three levels: products of a few normalised operators stay far inside the exponent range,
    // and powers of two scale exactly - the entry's bits are those of the one-depth-per-launch build)
    zip3_fold<NT, false, false>(P, ex, X, xe, 4, bq, 0, lo, lx);          // 1 -> 0, 3 -> 2, then 2 -> 0
    zip3_rescale<NT>(P, ex);
    if (bq == 0) {
        double *Gz = Gt + (size_t)z * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) Gz[Geo::idx(4 * K + q, 4 * J + r)] = P[K][J];
        if (q == 0 && r == 0) Gc[z] = ex;
    }
}

// ---- the scan -----------------------------------------------------------------------------------------------------
struct Z4Tok {              // one step's operator for this lane's segment (kept small: five of them are live in the loop)
    int s;                  // LDS slot, or -1: not cached in LDS - operands come from the registers loaded from the global table
    int ce;                 // its power-of-two exponent
    int tok;                // its token id = index of its global table entry
};

// Pout <- C_cur * Pin.  `al` holds tile-row 0 of cur's LDS entry on entry and of nxt's on exit (as zip3_step);
// `pre` holds ALL tile-rows of cur's global entry if cur is cold, and is refilled tile-row by tile-row with nxt's
// entry (if that is cold) as soon as each row has been consumed.
// `refill`: the token whose global entry goes into `pre` behind the rows just consumed - the next step's (one register
// set) or the one after (two sets: see Zip4Ring).
template <int NT, bool HYB>
__device__ __forceinline__ void zip4_step(const double (&Pin)[NT][NT], double (&Pout)[NT][NT], const double *C, const double *Gt, const double *Gi,
                                          const Z4Tok &cur, const Z4Tok &nxt, const Z4Tok &refill, double (&al)[NT], double (&pre)[NT][NT], int lo, int lx)
{
    constexpr int TOK = Zip3Geom<NT>::TOK;
    // The refill is issued for EVERY step - a hot next token fetches the identity entry `Gi` instead (the same few
    // cache lines for every such lane: L1 hits) - so that the number of loads in flight is the same on every path and
    // the compiler can wait for exactly the tile-row it needs (with the loads under a branch it waited for all of
    // them, vmcnt(0), at the top of every step: the L2 round trip of the last tile-row, issued 25 MFMAs earlier, was
    // exposed on every cold step).
    // HYB = false (streamed table): no LDS copies at all - every operand comes from `pre`.
    const bool cur_cold = !HYB || cur.s < 0, nxt_cold = !HYB || refill.s < 0;
    const double *Gn = nxt_cold ? Gt + (size_t)refill.tok * TOK : Gi;
    // (a cold token reads LDS slot 0: a valid address whose data is not used)
    const double *Cz = C + (size_t)max(cur.s, 0) * TOK, *Cn = C + (size_t)max(nxt.s, 0) * TOK;
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        double an[NT], av[NT];
        if constexpr (HYB) {
            if (I + 1 < NT) zip3_load_row<NT>(an, Cz, I + 1, lo, lx);
            else zip3_load_row<NT>(an, Cn, 0, lo, lx);
        }
#pragma unroll
        for (int K = 0; K < NT; ++K) av[K] = cur_cold ? pre[I][K] : al[K];
        __builtin_amdgcn_sched_barrier(0);         // keep the LDS prefetch ahead of this tile-row's MFMAs
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J)
                Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        zip4_load_row_global<NT>(pre[I], Gn, I, lo, lx);                 // (per lane; a full step ahead of its use)
        if constexpr (HYB) {
#pragma unroll
            for (int K = 0; K < NT; ++K) al[K] = an[K];
        }
    }
}

#ifdef IMC_Z4_TIMING      // diagnostics build only: phase timestamps (100 MHz clock) of wavefront 0 of every workgroup
__device__ long long g_z4_dbg[1024 * 8 * 3];        // [0]: phases of wavefront 0; [1]: scan end per wavefront; [2]: fold levels
#define Z4_STAMP(k_) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_z4_dbg[blockIdx.x * 8 + (k_)] = wall_clock64(); } while (0)
#else
#define Z4_STAMP(k_) do { } while (0)
#endif

// WIDE: the token stream holds 16-bit ids (dictionary levels beyond 256 tokens) instead of bytes.
// HYB = false: the STREAMED table - nothing is cached in LDS, every step's operands arrive from the global table
// (L1 / L2 / Infinity Cache) a step ahead.  Without the per-row LDS reads and the LDS-or-register selects a step is
// ~10 % faster as long as the tables of a launch stay cache resident (the host decides: Z4_STREAM_MAX_BYTES).
template <int NT, bool WIDE, bool HYB>
__global__ __launch_bounds__(Z2WAVES * 64, Z2WAVES / 4) void k_zpropagate4(BigArgs a)
{
    constexpr int TB = WIDE ? 2 : 1;                                   // bytes per token
    using Geo = Zip3Geom<NT>;
    using G4 = Zip4Geom<NT>;
    constexpr int NP = Geo::NP, TOK = Geo::TOK, THREADS = Z2WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int H = HYB ? a.n_hot : 0;                                   // hot operators; LDS slot H = the identity
    double *C = lds;                                                   // [slots(H)][TOK]
    int *cex = reinterpret_cast<int *>(C + (size_t)G4::slots(H) * TOK);   // [A + 1] exponents of the table entries
    int *slot_of = cex + a.A + 2;                                      // [A + 1] LDS slot of a token, -1 = cold
    int *xex = slot_of + a.A + 2;                                      // [Z2SLOTS + 1] the fold's exchange exponents

    const int tid = threadIdx.x;
    int b = (int)blockIdx.y, bx = (int)blockIdx.x;                     // parameter set, block (BigArgs::n_phases)
    if (a.n_phases > 0) {
        int ph = 0;
        while (ph + 1 < a.n_phases && (int)blockIdx.x >= a.ph_begin[ph + 1]) ++ph;
        const int local = (int)blockIdx.x - a.ph_begin[ph], xcd = local & 7, turn = local >> 3, nb = (int)a.n_group_segs;
        const int sets = a.ph_sets[ph];
        if (sets >= 8) {                                               // eight sets at a time, one per XCD
            b = a.ph_first[ph] + xcd + 8 * (turn / nb);
            bx = turn % nb;
        } else {                                                       // sets = 4, 2, 1: each on 8 / sets XCDs
            b = a.ph_first[ph] + xcd % sets;
            bx = turn * (8 / sets) + xcd / sets;
            if (bx >= nb) return;                                      // (workgroup-uniform, before any barrier)
        }
    }
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *pi_p = pp;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    const double *Gt = a.Ctab + (size_t)b * (a.A + 1) * TOK;
    const int *Gc = a.cex + (size_t)b * (a.A + 1);
    const int IDENT = a.A;
    const double *Gi = Gt + (size_t)IDENT * TOK;

    // ---- LDS: exponents, slot map, the hot operators and the identity ----
    Z4_STAMP(0);
    for (int z = tid; z <= a.A; z += THREADS) { cex[z] = Gc[z]; slot_of[z] = -1; }
    if constexpr (HYB) {
        __syncthreads();
        for (int k = tid; k <= H; k += THREADS) slot_of[k < H ? (int)a.hot[k] : IDENT] = k;
        for (int idx = tid; idx < (H + 1) * (TOK / 2); idx += THREADS) {
            const int k = idx / (TOK / 2), w = idx - k * (TOK / 2);
            const int z = k < H ? (int)a.hot[k] : IDENT;
            reinterpret_cast<double2 *>(C + (size_t)k * TOK)[w] = reinterpret_cast<const double2 *>(Gt + (size_t)z * TOK)[w];
        }
    }
    __syncthreads();

    // ---- scan: one segment per MFMA block ----
    Z4_STAMP(1);
    const int lane = tid & 63;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * Geo::NTE, lx = q * 4 + r;
    const Z2Block blk = a.blocks[bx];
    const int slot = (tid >> 6) * 4 + bq;                   // 0..Z2SLOTS-1 within the workgroup
    const bool valid = slot < (int)blk.n;
    // A lane without a segment runs the wavefront's unconditional token loads on another segment's stream: the FIRST
    // slot of its own wavefront (whose length bounds the wavefront's full blocks: nfull is a minimum over the valid
    // lanes), not the workgroup's segment 0 - in a packed block (Z2Block::first == 2) that may be a one-column chunk
    // beside a thousand-column one, and the loads then ran hundreds of bytes past its buffer (found under IMC_GUARD=1).
    const uint32_t seg = blk.seg0 + (valid ? slot : ((slot & ~3) < (int)blk.n ? (slot & ~3) : 0));
    const SegDesc sd = a.segs[seg];
    const int len = valid ? (int)sd.len : 0;
    const bool first = (sd.first & SEG_FIRST) != 0;
    const uint8_t *tokp = sd.obs;
    // (token loads as GLOBAL loads: through the generic pointer of the segment descriptor they are flat loads, which
    // count against the LDS counter too and can return out of order with the operand refills - every wait behind one
    // becomes a full one)
    typedef const __attribute__((address_space(1))) uint8_t *gptr;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4 *gptr4;
    typedef const __attribute__((address_space(1))) uint32_t *gptr1;
    const gptr tokg = (gptr)tokp;

    double P[NT][NT], Q[NT][NT];
    {
        const int tok0 = (valid && first) ? seg_token(tokp, WIDE, 0) : 0;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                const int i = 4 * K + q, c = 4 * J + r;
                double v;
                if (first) v = (c == 0 && i < a.N) ? pi_p[i] * Etg[(size_t)tok0 * a.PP + i] : 0.0;
                else v = (i == c && i < a.N) ? 1.0 : 0.0;
                P[K][J] = valid ? v : 0.0;
            }
    }
    int ex = 0;
    const int maxlen = wave_max_i32(len);
    const int nfull = maxlen == 0 ? 0 : wave_min_i32(valid ? len / RESCALE_EVERY : INT_MAX);   // idle wavefronts still join the fold's barriers

    // Register sets of global-table operands.  One set, refilled with the NEXT step's operator, hides an L2 round trip
    // behind a step of 125 MFMAs (20 states); at 16 states and below a step (<= 64 MFMAs, <= 0.5 us) is shorter than that
    // round trip and the streamed form keeps TWO sets, each refilled with the operator two steps ahead (microbenchmark
    // profiles/tools/micro/mb_scan_ring3.hip, 10 states: 60.0 -> 35.4 us for the planner's 100 x 1e6-column shape, the
    // MFMA instruction's own rate; at 20 states the same change is slower: mb_scan_ring2.hip).
    constexpr int RING = (!HYB && NT <= 4) ? 2 : 1;
    double al[NT] = {}, pre[RING][NT][NT];
#pragma unroll
    for (int d = 0; d < RING; ++d)
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int K = 0; K < NT; ++K) pre[d][I][K] = 0.0;
    auto mk = [&](int tok) __attribute__((always_inline)) {
        Z4Tok t;
        t.s = HYB ? slot_of[tok] : -1;
        t.ce = cex[tok];
        t.tok = tok;
        return t;
    };
    // start (or restart) the pipeline at `c`: its LDS row 0 and, if it is cold, all of its global rows - exposed latency,
    // paid at the start of a segment and where the token of the next position is not known a step ahead
    // (c1: the token of the position after c - only the two-set form loads it)
    auto prime = [&](const Z4Tok &c, const Z4Tok &c1) __attribute__((always_inline)) {
        if constexpr (HYB) zip3_load_row<NT>(al, C + (size_t)max(c.s, 0) * TOK, 0, lo, lx);
        const double *Gc0 = c.s < 0 ? Gt + (size_t)c.tok * TOK : Gi;        // (unconditional, as zip4_step's refill)
#pragma unroll
        for (int I = 0; I < NT; ++I) zip4_load_row_global<NT>(pre[0][I], Gc0, I, lo, lx);
        if constexpr (RING == 2) {
            const double *Gc1 = Gt + (size_t)c1.tok * TOK;
#pragma unroll
            for (int I = 0; I < NT; ++I) zip4_load_row_global<NT>(pre[1][I], Gc1, I, lo, lx);
        }
    };
    // two steps on t0, t1; tn, tn1: the tokens of the two positions behind them (tn1 only matters to the two-set form)
    auto two_steps = [&](const Z4Tok &t0, const Z4Tok &t1, const Z4Tok &tn, const Z4Tok &tn1) __attribute__((always_inline)) {
        if constexpr (RING == 2) {
            zip4_step<NT, HYB>(P, Q, C, Gt, Gi, t0, t1, tn, al, pre[0], lo, lx);
            zip4_step<NT, HYB>(Q, P, C, Gt, Gi, t1, tn, tn1, al, pre[1], lo, lx);
        } else {
            zip4_step<NT, HYB>(P, Q, C, Gt, Gi, t0, t1, t1, al, pre[0], lo, lx);
            zip4_step<NT, HYB>(Q, P, C, Gt, Gi, t1, tn, tn, al, pre[0], lo, lx);
        }
        ex += t0.ce + t1.ce;
    };
    // A block of 16 positions in which not every lane's segment has a token (see k_zpropagate3): per run of PER positions
    // (one 16-byte load: 16 byte tokens or 8 16-bit ones) the lane loads where its segment still has tokens and steps
    // with the identity elsewhere; the pipeline is primed at the start of every run.
    constexpr int PER = WIDE ? 8 : 16;
    auto masked_run = [&](int bi, int u0, int n) __attribute__((always_inline)) {   // positions u0 .. u0 + n - 1 of block bi, n even
        uint4 ob = make_uint4(0u, 0u, 0u, 0u);
        if (bi * RESCALE_EVERY + u0 < len) {
            const u32x4 v = *(gptr4)(tokg + ((size_t)bi * RESCALE_EVERY + u0) * TB);
            ob = make_uint4(v.x, v.y, v.z, v.w);
        }
        const int live = len - bi * RESCALE_EVERY - u0;                 // run positions v < live hold a token
        const int dead0 = (first && bi == 0 && u0 == 0) ? 0 : -1;
        const unsigned long long lo64 = (unsigned long long)ob.y << 32 | ob.x, hi64 = (unsigned long long)ob.w << 32 | ob.z;
        auto tok_of = [&](int v) __attribute__((always_inline)) {
            const unsigned long long h = v < PER / 2 ? lo64 : hi64;
            const int tk = WIDE ? (int)((h >> (16 * (v & 3))) & 0xffffull) : (int)((h >> (8 * (v & 7))) & 0xffull);
            return (v < live && v != dead0 && v < PER) ? tk : IDENT;
        };
        Z4Tok c0 = mk(tok_of(0)), c1 = mk(tok_of(1));
        prime(c0, c1);
#pragma unroll 1
        for (int v = 0; v < n; v += 2) {
            const Z4Tok c2 = mk(tok_of(v + 2)), c3 = mk(tok_of(v + 3));
            two_steps(c0, c1, c2, c3);
            c0 = c2;
            c1 = c3;
        }
    };
    auto masked_block = [&](int bi, int npos) __attribute__((always_inline)) {
#pragma unroll 1
        for (int u0 = 0; u0 < npos; u0 += PER) masked_run(bi, u0, min(PER, npos - u0));
        zip3_rescale<NT>(P, ex);
    };
    // Block 0 runs in the fast loop too when every segment of the wavefront has 16 tokens (position 0 of a chunk's first
    // segment went into the start vector: that lane steps with the identity there); otherwise it is a masked block.
    if (nfull == 0 && maxlen > 0) masked_block(0, min(RESCALE_EVERY, (maxlen + 1) & ~1));
    Z4_STAMP(2);
    if (nfull > 0) {
        // full blocks: the tokens of block bi + 1 are fetched while block bi runs, so the first token of the next block
        // is known a step ahead and the cold-operand pipeline never drains inside this loop.  The block's tokens sit
        // in a shift register of 32-bit words: four tokens - one word of bytes, two words of 16-bit ids - per iteration.
        constexpr int NW = 4 * TB;                                        // words per block
        constexpr uint32_t TM = WIDE ? 0xffffu : 0xffu;
        auto load_words = [&](int bi, uint32_t (&w)[NW]) __attribute__((always_inline)) {
            gptr4 src = (gptr4)(tokg + (size_t)bi * RESCALE_EVERY * TB);
            const u32x4 x = src[0];
            w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
            if constexpr (WIDE) {
                const u32x4 y = src[1];
                w[4] = y.x; w[5] = y.y; w[6] = y.z; w[7] = y.w;
            }
        };
        uint32_t cw[NW];
        load_words(0, cw);
        Z4Tok c0 = mk(first ? IDENT : (int)(cw[0] & TM));
        prime(c0, mk(WIDE ? (int)(cw[0] >> 16) : (int)((cw[0] >> 8) & TM)));
        for (int bi = 0; bi < nfull; ++bi) {
            // Straight-line body (no load under a branch: the compiler then waits for exactly the loads it needs): the
            // next block's first token now (one word), its other words once this block's have been consumed; behind
            // the last full block both re-read the current block and the pipeline is pointed at the identity (the
            // block that follows, if any, is a masked one and primes itself).
            const bool more = bi + 1 < nfull;
            const int bn = more ? bi + 1 : bi;
            const uint32_t nfirst = *(gptr1)(tokg + (size_t)bn * RESCALE_EVERY * TB);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                int t1, t2, t3, t4;
                if constexpr (WIDE) {
                    const uint32_t wa = cw[0], wb = cw[1];
#pragma unroll
                    for (int k = 0; k + 2 < NW; ++k) cw[k] = cw[k + 2];
                    t1 = (int)(wa >> 16); t2 = (int)(wb & TM); t3 = (int)(wb >> 16);
                } else {
                    const uint32_t wa = cw[0];
#pragma unroll
                    for (int k = 0; k + 1 < NW; ++k) cw[k] = cw[k + 1];
                    t1 = (int)((wa >> 8) & TM); t2 = (int)((wa >> 16) & TM); t3 = (int)(wa >> 24);
                }
                t4 = g4 < 3 ? (int)(cw[0] & TM) : more ? (int)(nfirst & TM) : IDENT;
                // (the token behind t4: only the two-set form looks two steps ahead)
                const uint32_t w5 = g4 < 3 ? cw[0] : nfirst;
                const int t5 = (g4 < 3 || more) ? (WIDE ? (int)(w5 >> 16) : (int)((w5 >> 8) & TM)) : IDENT;
                if (g4 == 3) load_words(bn, cw);                          // lands during the last four steps of this block
                const Z4Tok c1 = mk(t1), c2 = mk(t2), c3 = mk(t3);
                two_steps(c0, c1, c2, c3);
                const Z4Tok c4 = mk(t4), c5 = mk(t5);
                two_steps(c2, c3, c4, c5);
                c0 = c4;
            }
            zip3_rescale<NT>(P, ex);
        }
    }
    Z4_STAMP(3);
    for (int bi = max(1, nfull); bi * RESCALE_EVERY < maxlen; ++bi)
        masked_block(bi, min(RESCALE_EVERY, (maxlen - bi * RESCALE_EVERY + 1) & ~1));
    zip3_rescale<NT>(P, ex);
    Z4_STAMP(4);
#ifdef IMC_Z4_TIMING
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_z4_dbg[1024 * 8 + blockIdx.x * 8 + (threadIdx.x >> 6)] = wall_clock64();
#endif

    // ---- packed block (Z2Block::first == 2): every slot is a whole one-segment chunk - its vector goes out as it is ----
    if (blk.first == 2) {                                   // (workgroup-uniform: nobody reaches the fold's barriers)
        if (valid && r == 0) {
            const size_t gvp = (size_t)b * a.n_vecs_total + blk.out_vec0 + slot;
#pragma unroll
            for (int K = 0; K < NT; ++K)
                if (4 * K + q < a.N) a.P[gvp * NP + 4 * K + q] = P[K][0];
            if (q == 0) a.EX[gvp] = ex;
        }
        return;
    }

    // ---- fold the workgroup's segments into one (zip3_fold; the LDS entries become the exchange area; the streamed
    // form keeps nothing in LDS during the scan, so its wavefronts start folding as they finish) ----
    zip3_fold<NT, HYB>(P, ex, C, xex, (int)blk.n, slot, tid >> 6, lo, lx);

    Z4_STAMP(5);
#ifdef IMC_Z4_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_z4_dbg[blockIdx.x * 8 + 7] = (long long)nfull * 1000000 + maxlen;
#endif
    if (slot == 0) {
        const size_t gv = (size_t)b * a.n_vecs_total + blk.out_vec0;
        double *Pout = a.P + gv * NP;
        if (blk.first) {
            if (r == 0) {
#pragma unroll
                for (int K = 0; K < NT; ++K)
                    if (4 * K + q < a.N) Pout[4 * K + q] = P[K][0];
            }
            if (q == 0 && r == 0) a.EX[gv] = ex;
        } else {
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    const int i = 4 * K + q, c = 4 * J + r;
                    if (i < a.N && c < a.N) Pout[(size_t)i * NP + c] = P[K][J];
                }
            if (q == 0) {
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    if (4 * J + r < a.N) a.EX[gv + 4 * J + r] = ex;
            }
        }
    }
    if (a.tail) zip3_tail<NT>(a, P, ex, C, xex, b, bx, slot, lo, lx);
}
