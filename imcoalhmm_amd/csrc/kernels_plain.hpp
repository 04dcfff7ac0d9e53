// kernels_plain.hpp - the per-column forward propagate kernel (uncompressed observation stream).
// Included by imcoal_fwd.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cmath>
#include <cstdint>

// =====================================================================================================
// Device side
// =====================================================================================================

// cross-lane moves inside a 16-lane DPP row (no LDS traffic)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    // bound_ctrl with full row/bank masks: every lane receives data, so no pre-initialising v_mov of the destination
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}
static constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128, DPP_ROW_ROR12 = 0x12c;

struct SegDesc {          // one segment of one chunk
    const uint8_t *obs;   // first column of the segment (16-byte aligned - 4-byte for the blocked kernels -, padded past the end)
    uint32_t len;         // columns in this segment
    uint32_t first;       // bit 0: first segment of its chunk (single vector, starts from pi); bit 1: 16-bit tokens
};
static constexpr uint32_t SEG_FIRST = 1u, SEG_WIDE = 2u;

// token t of a segment: bytes, or 16-bit ids on the wide dictionary levels (large-N kernels only)
__device__ __forceinline__ int seg_token(const uint8_t *p, bool wide, int t)
{
    return wide ? (int)reinterpret_cast<const uint16_t *>(p)[t] : (int)p[t];
}

struct VecDesc {          // one propagated vector
    uint32_t seg;         // segment id
    uint32_t c;           // basis index (0 for a first segment)
};

struct PropArgs {
    const SegDesc *segs;   // all segments of the plan
    const VecDesc *vecs;   // this launch group's vectors
    uint32_t n_vecs;       // vectors in this group
    uint32_t vec_base;     // index of the group's first vector in the plan-wide arrays
    uint32_t n_vecs_total; // vectors in the whole plan (stride of P / EX per parameter set)
    int N;                 // true number of states
    int S;                 // alphabet size
    const double *params;  // per parameter set: pi[NP] | Tp[NP*NP] (Tp[j*NP+i]=T[j][i]) | Et[S*NP]
    size_t pstride;        // doubles per parameter set
    double *P;             // [B][n_vecs_total][NP]  normalised end vectors
    int *EX;               // [B][n_vecs_total]      power-of-two exponents
    // compressed path only
    int A;                 // alphabet of the token stream (raw symbols + merges)
    const uint16_t *tok_left, *tok_right;  // [A] merge table (token z = left[z] then right[z])
};

static constexpr int WPB = 4;            // wavefronts per workgroup (256 threads)
static constexpr int RESCALE_EVERY = 16; // columns between power-of-two rescales (= one 16-byte obs load)

__device__ __forceinline__ void wave_fence()
{
    // A vector lives inside one wavefront and LDS executes a wavefront's DS instructions in order,
    // so only the compiler has to be stopped from moving LDS reads across the preceding writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One column for the R states of this lane:  x <- E[:,sym] .* (T' x)   (or E .* x when skipT).
// The vector's NP values are streamed from LDS (broadcast ds_read_b128) straight into the FMA chains;
// with SUM their total (identical in all G lanes of the vector) is returned for the rescale.
template <int R, int NP, bool PRED, bool SUM>
__device__ __forceinline__ void column_step(double (&xo)[R], const double (&Tb)[R][NP], double *xw, int own,
                                            const double *Et, int sym, bool act, bool skipT, double &s)
{
    // emission factors first: their LDS latency hides under the FMA chains (Et is read-only)
    double ev[R];
    {
        const double *e = Et + sym * NP + own;
#pragma unroll
        for (int k = 0; k < R; ++k) ev[k] = e[k];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) xw[own + k] = xo[k];
    wave_fence();
    double acc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0.0;
    if (SUM) s = 0.0;
    const double2 *xv = reinterpret_cast<const double2 *>(xw);
#pragma unroll
    for (int m = 0; m < NP / 2; ++m) {
        const double2 t = xv[m];
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = fma(Tb[k][2 * m], t.x, acc[k]);
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = fma(Tb[k][2 * m + 1], t.y, acc[k]);
        if (SUM) s += t.x;
        if (SUM) s += t.y;
    }
    if (NP & 1) {
        const double t = xw[NP - 1];
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = fma(Tb[k][NP - 1], t, acc[k]);
        if (SUM) s += t;
    }
    wave_fence();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (PRED) {
            const double y = (skipT ? xo[k] : acc[k]) * ev[k];
            xo[k] = act ? y : xo[k];
        } else {
            xo[k] = acc[k] * ev[k];
        }
    }
}

// Power-of-two rescale by the exponent of s (the vector's total, identical in all G lanes).
template <int R>
__device__ __forceinline__ void rescale(double (&xo)[R], double s, int &ex)
{
    int e = 0;
    (void)frexp(s, &e);
    e = (s > 0.0 && s < INFINITY) ? e : 0;   // 0, inf and NaN: leave the vector alone
#pragma unroll
    for (int k = 0; k < R; ++k) xo[k] = ldexp(xo[k], -e);
    ex += e;
}

template <int NP>
__device__ __forceinline__ double sum_vec(const double *xw)
{
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) s += xw[j];
    return s;
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
// The same maximum by DPP moves (row shifts inside the 16-lane rows, then the two row broadcasts of gfx9): seven VALU
// instructions instead of six ds_bpermute round trips (~80 instead of ~500 cycles of dependent latency) - for use on a
// serial chain (k_chain's per-step exponent maximum).
__device__ __forceinline__ int wave_max_i32_dpp(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of a row: the row's maximum
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

template <int R, int G, int MINW>
__global__ __launch_bounds__(WPB * 64, MINW) void k_propagate(PropArgs a)
{
    constexpr int NP = R * G;      // padded state count
    constexpr int VPW = 64 / G;    // vectors per wavefront
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Et = lds + WPB * VPW * NP;   // [S][NP]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int v = lane / G;
    const int r = lane - v * G;
    const bool spare = v >= VPW;            // leftover lanes mirror the last vector, never store
    v = spare ? VPW - 1 : v;
    const int own = r * R;
    double *xw = lds + (wave * VPW + v) * NP;

    const int b = blockIdx.y;
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *pi_p = pp;
    const double *Tp = pp + NP;
    const double *Etg = pp + NP + NP * NP;

    // stage E' once per workgroup, T' block once per lane
    for (int i = threadIdx.x; i < a.S * NP; i += WPB * 64) Et[i] = Etg[i];
    double Tb[R][NP];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int k = 0; k < R; ++k) Tb[k][j] = Tp[j * NP + own + k];
    __syncthreads();

    const uint32_t vid = (blockIdx.x * WPB + wave) * VPW + v;
    const bool active = !spare && vid < a.n_vecs;
    // lanes past the last vector shadow it (valid addresses, len 0, never stored)
    const VecDesc vd = a.vecs[min(vid, a.n_vecs - 1u)];
    const SegDesc sd = a.segs[vd.seg];
    const int len = (vid < a.n_vecs) ? (int)sd.len : 0;
    const bool first = (sd.first & SEG_FIRST) != 0;
    const bool wide = (sd.first & SEG_WIDE) != 0;      // 16-bit symbols (raw alphabets beyond 256)
    const uint8_t *obs = sd.obs;

    double xo[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = own + k;
        xo[k] = (vid < a.n_vecs) ? (first ? pi_p[i] : (i == (int)vd.c ? 1.0 : 0.0)) : 0.0;
    }
    int ex = 0;
    double s = 0.0;

    const int maxlen = wave_max_i32(len);
    const int nfull = wave_min_i32(vid < a.n_vecs ? len / RESCALE_EVERY : INT_MAX);
    if (maxlen == 0) return;   // wavefront entirely past the last vector (wave-uniform exit)

    // ---- head: first block, column by column (column 0 of a first segment skips T') ----
    const int head_end = min(RESCALE_EVERY, maxlen);
    for (int t = 0; t < head_end; ++t) {
        const bool act = t < len;
        const int sym = act ? seg_token(obs, wide, t) : 0;
        column_step<R, NP, true, true>(xo, Tb, xw, own, Et, sym, act, first && t == 0, s);
        rescale<R>(xo, s, ex);
    }
    // ---- body: full 16-column blocks common to every vector of this wavefront ----
    for (int blk = 1; blk < nfull; ++blk) {
        // 16 symbols: one 16-byte load of bytes, or two of 16-bit values; either way the loop below sees four words
        // of four symbols each, a symbol occupying SH = 8 or 16 bits of a 64-bit pair (per-lane, no divergence)
        const uint4 *src = reinterpret_cast<const uint4 *>(obs + (size_t)blk * RESCALE_EVERY * (wide ? 2 : 1));
        const uint4 oa = src[0];
        const uint4 ob = wide ? src[1] : oa;
        unsigned long long v0, v1, v2, v3;
        if (wide) {
            v0 = (unsigned long long)oa.y << 32 | oa.x; v1 = (unsigned long long)oa.w << 32 | oa.z;
            v2 = (unsigned long long)ob.y << 32 | ob.x; v3 = (unsigned long long)ob.w << 32 | ob.z;
        } else {
            v0 = oa.x; v1 = oa.y; v2 = oa.z; v3 = oa.w;
        }
        const int sh = wide ? 16 : 8;
        const unsigned mask = wide ? 0xffffu : 0xffu;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const unsigned long long w = v0;
            v0 = v1; v1 = v2; v2 = v3;
            column_step<R, NP, false, false>(xo, Tb, xw, own, Et, (int)((unsigned)w & mask), true, false, s);
            column_step<R, NP, false, false>(xo, Tb, xw, own, Et, (int)((unsigned)(w >> sh) & mask), true, false, s);
            column_step<R, NP, false, false>(xo, Tb, xw, own, Et, (int)((unsigned)(w >> (2 * sh)) & mask), true, false, s);
            column_step<R, NP, false, true>(xo, Tb, xw, own, Et, (int)((unsigned)(w >> (3 * sh)) & mask), true, false, s);
        }
        rescale<R>(xo, s, ex);
    }
    // ---- tail: ragged remainder, column by column ----
    for (int t = max(RESCALE_EVERY, nfull * RESCALE_EVERY); t < maxlen; ++t) {
        const bool act = t < len;
        const int sym = act ? seg_token(obs, wide, t) : 0;
        column_step<R, NP, true, true>(xo, Tb, xw, own, Et, sym, act, false, s);
        rescale<R>(xo, s, ex);
    }
    // ---- final normalisation: one more LDS round to see the finished vector ----
#pragma unroll
    for (int k = 0; k < R; ++k) xw[own + k] = xo[k];
    wave_fence();
    s = sum_vec<NP>(xw);
    wave_fence();
    rescale<R>(xo, s, ex);

    if (active) {
        // a first segment's vector is stored [i]; an operator's column c goes into the segment's
        // N x NP block state-major ([i][c]) so that the chain kernel reads a state's row contiguously
        const size_t gv = (size_t)b * a.n_vecs_total + a.vec_base + vid;
        double *Pout = first ? a.P + gv * NP : a.P + (gv - vd.c) * NP + vd.c;
        const int st = first ? 1 : NP;
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (own + k < a.N) Pout[(size_t)(own + k) * st] = xo[k];
        if (r == 0) a.EX[gv] = ex;
    }
}

