// kernels_big.hpp - the kernels whose operator table lives in global memory.  Included by imcoal_fwd.hip.
// Used for 64 < N <= 256 and, when chunks are long, for 24 < N <= 64.
//
// For N ~ 150 neither the N x N operators (180 KB each) nor a segment's transfer operator fit one wavefront's
// registers, and the work per token step, P <- C_tok * P (N x N by N x N), is a genuine dense fp64 GEMM.  This is
// the one place where the matrix cores are the right tool ("MFMA becomes relevant only if the
// N^3 transfer-operator formulation is adopted"): v_mfma_f64_16x16x4_f64 tiles, NP = 16 NT.
//   * k_big_table_raw / k_big_table_level build the per-evaluation operator table (up to 16384 tokens), one launch
//     per dictionary depth; big_gemm is their workgroup-wide product (k panels staged through LDS).
//   * k_big_propagate (one tile-row per wavefront) and k_big_propagate_s (NT = 6, 10, 14: tiles dealt over eight
//     wavefronts so that the four SIMDs carry equal MFMA work) keep a column slab of P resident in LDS for the whole
//     launch and stream only the token operator's A panels: the GEMM chain.
//   * k_big_vector is the mat-vec chain x <- C_tok x for segments that are vectors: every chunk when a call has enough
//     (chunk, parameter set) chains to fill the chip, first segments, and - TAIL variant - the tails of operator
//     segments whose head k_rank1_check has certified to be rank one (the rank-one hand-off).
// Every step is rescaled by one exact power of two (the exponent of the largest entry).
#pragma once
#include "kernels_plain.hpp"

typedef double v4f64 __attribute__((ext_vector_type(4)));

struct Z2Block {                  // one workgroup of k_zpropagate2: up to Z2SLOTS consecutive segments of one chunk
    uint32_t seg0, n;             // first segment id, number of segments
    uint32_t out_vec0;            // where the block's combined result goes (level-0 vector index)
    uint32_t first;               // 1: seg0 is its chunk's first segment -> the result is a vector; 2 (MFMA forms): a PACKED block -
                                  // every one of the n segments is a whole chunk, slot s writes its vector to out_vec0 + s, no fold
};

struct Z2Tail { uint32_t chunk, unit, n_units, pad; };

struct BigArgs {
    const SegDesc *segs;          // all segments of the plan
    const uint32_t *seg_ids;      // k_big_propagate: segments of this launch group
    const uint32_t *seg_vec0;     // k_big_propagate: level-0 vector index per entry of seg_ids
    const Z2Block *blocks;        // k_zpropagate2: one entry per workgroup
    uint32_t n_group_segs;
    uint32_t n_vecs_total;
    int N, S, A;
    const double *params;         // per parameter set: pi[PP] | Tp[PP*PP] | Et[S*PP]   (PP = params' padding)
    const double *params_src;     // k_zpropagate3, small launches: non-null = fetch the parameters from here (the mapped staging slot on the
                                  // host) into LDS instead of reading a.params - the evaluation then has no k_stage_params launch
    size_t pstride;
    int PP;
    const uint16_t *tok_left, *tok_right;
    double *Ctab;                 // [B][A][NP][NP] operator table (row-major, zero padded)
    int *cex;                     // [B][A] power-of-two exponents of the table entries
    double *P;                    // level-0 results (see kernels_stitch.hpp for the layout)
    int *EX;
    // rank-one hand-off (see k_rank1_check).  phase 0: whole segments.  phase 1: one ROUND of the GEMM chain over the
    // operator segments (first segments run on the mat-vec chain kernel): tokens [t_from, t_to) of every segment that
    // is longer than t_from and not yet certified; t_from > 0 continues from the block stored by the previous round.
    int phase;
    int t_from, t_to;
    int *r1flag;                  // [B][n_segs] 1: the operator after r1at tokens is rank one within R1_TOL
    int *r1at;                    // [B][n_segs] token count at which that was certified (the tail starts there)
    double *r1u, *r1alpha;        // [B][n_segs][NP] its column direction u and column scales alpha
    uint32_t n_segs;
    // k_zpropagate3's table build: the merged tokens of this launch's alphabet sorted by dictionary depth, tokens of
    // depth d at tab_order[tab_lvl[d] .. tab_lvl[d+1]) - a token only depends on tokens of smaller depth
    const uint16_t *tab_order;
    const int *tab_lvl;
    int tab_nlvl;
    // k_zpropagate4 (hybrid table): the n_hot most frequent tokens of the launch's chunks, cached in LDS (slot k = hot[k])
    const uint16_t *hot;
    int n_hot;
    const int4 *tab_desc;         // k_z4_level: per entry of tab_order {token, left child, right child, 0} (one load instead of three dependent ones)
    // k_big_vector reads a PACKED copy of the operator table when there is one: [B][A][N][TS], TS = N rounded up to even
    // (150 x 150 doubles instead of the 160-double rows the GEMM kernels want: the mat-vec chains are bound by these
    // bytes and a padded row costs them ten cache lines instead of 9.4); written by the table kernels beside Ctab
    double *Cpack;
    int TS;
    // fused tail of the blocked MFMA kernels (zip3_tail): every chunk of the launch is at most Z2SLOTS workgroups, each
    // publishes its folded operator, and the chunk's LAST workgroup to arrive folds them and writes the chunk's
    // log-likelihood - the evaluation has no stitch launches.  tail == nullptr: off.
    const Z2Tail *tail;           // per workgroup: {chunk, unit within the chunk, units of the chunk}
    double *tailX;                // [B][n_chunks][tail_stride][TOK] published operators (an LDS table entry's layout)
    int *tailE;                   // [B][n_chunks][tail_stride][32] their exponents (one 128-byte line each)
    int *tail_arrive;             // [B][n_chunks] arrival counters (zero between evaluations: the last arriver resets its own)
    double *tail_out;             // [B][n_chunks] log-likelihoods (device memory or the mapped result slots on the host)
    int tail_stride, n_chunks;
    // k_zpropagate4's workgroup -> (parameter set, block) map.  n_phases = 0: grid (blocks, B).  Otherwise a one-dimensional
    // grid, dealt round-robin over the eight XCDs by the dispatcher (workgroup L runs on XCD L % 8), cut into phases
    // that each load all XCDs evenly: first the B / 8 * 8 sets of the "eight at a time" phase - set L % 8 + 8 * round
    // on XCD L % 8, all of a set's blocks on that one XCD - then, for the remaining B % 8 = 4 a + 2 b + c sets, a phase of
    // four sets on two XCDs each, one of two sets on four XCDs each, one of a single set on all eight.  A set's operator
    // table is then read through the L2 of one (two, four) XCDs, and an XCD works on one or two sets at a time, instead
    // of every XCD's 4 MB L2 seeing the tables of all B sets.
    int n_phases, ph_begin[4], ph_first[4], ph_sets[4];
};


// acc = A(rows row0.., all k) * B(all k, cols col0..) for one workgroup-wide GEMM of size NP; A and B are
// row-major NP x NP in (L2-resident) global memory.  k is consumed in panels of 16: all NT wavefronts stage
// the A panel (NP x 16) and the B panel (16 x NP) through LDS, double buffered - the global loads of panel
// kb+1 are in flight while the MFMAs of panel kb run - so every operand byte is fetched once per
// workgroup instead of once per wavefront, and one barrier per panel is the only synchronisation.
// Lane l of a wavefront takes k = 4*(l>>4) + s (s = 0..3) of the panel from both operands.
template <int NT>
struct BigLds {
    static constexpr int NP = 16 * NT;
    static constexpr int APS = 18;        // A panel row stride (doubles): 16 + 2, keeps ds_read_b128 rows on distinct banks
    static constexpr int BPS = NP + 4;    // B panel row stride (doubles): rows 4 apart land 32 banks apart
    static constexpr int A_DOUBLES = NP * APS, B_DOUBLES = 16 * BPS;
    static constexpr size_t bytes = (size_t)2 * (A_DOUBLES + B_DOUBLES) * 8;
};

template <int NT, int TC = NT>
__device__ __forceinline__ void big_gemm(const double *__restrict__ A, const double *__restrict__ B, int row0, int col0,
                                         int tid, double *lds, v4f64 (&acc)[1][TC])
{
    using L = BigLds<NT>;
    constexpr int NP = 16 * NT, TR = 1, THREADS = NT * 64;   // wavefront w computes tile-row w, TC tile-columns from col0
    constexpr int APS = L::APS, BPS = L::BPS;
    const int lane = tid & 63, lm = lane & 15, lg = lane >> 4;
    // staging assignment: two double2 of the A panel (NP rows x 8 double2) and two of the B panel
    // (16 rows x NP/2 double2) per thread; plain scalars so that nothing is demoted to scratch/LDS
    const int e0 = tid, e1 = tid + THREADS;
    const int ar0 = e0 >> 3, ac0 = e0 & 7, ar1 = e1 >> 3, ac1 = e1 & 7;
    const int br0 = e0 / (NP / 2), bc0 = e0 - br0 * (NP / 2), br1 = e1 / (NP / 2), bc1 = e1 - br1 * (NP / 2);
    const double *gA0 = A + (size_t)ar0 * NP + 2 * ac0, *gA1 = A + (size_t)ar1 * NP + 2 * ac1;
    const double *gB0 = B + (size_t)br0 * NP + 2 * bc0, *gB1 = B + (size_t)br1 * NP + 2 * bc1;
    const int lA0 = ar0 * APS + 2 * ac0, lA1 = ar1 * APS + 2 * ac1;
    const int lB0 = L::A_DOUBLES + br0 * BPS + 2 * bc0, lB1 = L::A_DOUBLES + br1 * BPS + 2 * bc1;
    double2 sa0, sa1, sb0, sb1;
#define BIG_FETCH(kb_)                                                                 \
    do {                                                                               \
        sa0 = *reinterpret_cast<const double2 *>(gA0 + (kb_) * 16);                    \
        sa1 = *reinterpret_cast<const double2 *>(gA1 + (kb_) * 16);                    \
        sb0 = *reinterpret_cast<const double2 *>(gB0 + (size_t)(kb_) * 16 * NP);       \
        sb1 = *reinterpret_cast<const double2 *>(gB1 + (size_t)(kb_) * 16 * NP);       \
    } while (0)
#define BIG_STAGE(buf_)                                                                \
    do {                                                                               \
        double *base_ = lds + (buf_) * (L::A_DOUBLES + L::B_DOUBLES);                  \
        *reinterpret_cast<double2 *>(base_ + lA0) = sa0;                               \
        *reinterpret_cast<double2 *>(base_ + lA1) = sa1;                               \
        *reinterpret_cast<double2 *>(base_ + lB0) = sb0;                               \
        *reinterpret_cast<double2 *>(base_ + lB1) = sb1;                               \
    } while (0)
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) acc[tr][tc] = v4f64{0.0, 0.0, 0.0, 0.0};
    BIG_FETCH(0);
    BIG_STAGE(0);
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < NP / 16; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < NP / 16) BIG_FETCH(kb + 1);
        const double *Al = lds + buf * (L::A_DOUBLES + L::B_DOUBLES), *Bl = Al + L::A_DOUBLES;
        double a[TR][4], bq[TC][4];
#pragma unroll
        for (int tr = 0; tr < TR; ++tr) {
            const double2 *ap = reinterpret_cast<const double2 *>(Al + (row0 + tr * 16 + lm) * APS + 4 * lg);
            const double2 a01 = ap[0], a23 = ap[1];
            a[tr][0] = a01.x; a[tr][1] = a01.y; a[tr][2] = a23.x; a[tr][3] = a23.y;
        }
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int s = 0; s < 4; ++s) bq[tc][s] = Bl[(4 * lg + s) * BPS + col0 + tc * 16 + lm];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tr = 0; tr < TR; ++tr)
#pragma unroll
                for (int tc = 0; tc < TC; ++tc)
                    acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tr][s], bq[tc][s], acc[tr][tc], 0, 0, 0);
        if (kb + 1 < NP / 16) BIG_STAGE(buf ^ 1);   // nobody reads buf^1 during this iteration
        __syncthreads();
    }
#undef BIG_FETCH
#undef BIG_STAGE
}

// Workgroup-wide maximum of the (non-negative) accumulator entries -> exponent e with max in [2^(e-1), 2^e).
// smax: two LDS slots used alternately (slot `which`); the other slot is cleared for the next call.
template <int TC>
__device__ __forceinline__ double big_tile_max(const v4f64 (&acc)[1][TC])
{
    double mx = 0.0;
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = acc[0][tc][q];
            mx = (v > mx || v != v) ? v : mx;
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = __shfl_xor(mx, m, 64);
        mx = (o > mx || o != o) ? o : mx;
    }
    return mx;
}

template <int NT>
__device__ __forceinline__ int big_exponent(const v4f64 (&acc)[1][NT], unsigned long long *smax, int which, int tid)
{
    constexpr int TR = 1, TC = NT;
    double mx = 0.0;
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double v = acc[tr][tc][q];
                mx = (v > mx || v != v) ? v : mx;
            }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = __shfl_xor(mx, m, 64);
        mx = (o > mx || o != o) ? o : mx;
    }
    if ((tid & 63) == 0) atomicMax(&smax[which], (unsigned long long)__double_as_longlong(mx));
    __syncthreads();
    const double m = __longlong_as_double((long long)smax[which]);
    if (tid == 0) smax[which ^ 1] = 0ull;
    int e = 0;
    (void)frexp(m, &e);
    return (m > 0.0 && m < INFINITY) ? e : 0;
}

// Store the scaled accumulator tiles: D layout of v_mfma_f64_16x16x4_f64 is row = (lane>>4) + 4*reg, col = lane&15.
template <int TC>
__device__ __forceinline__ void big_store(double *__restrict__ D, int row0, int col0, int lane, const v4f64 (&acc)[1][TC],
                                          int e, int row_limit, int col_limit, size_t ld)
{
    constexpr int TR = 1;
    const int lm = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + tr * 16 + lg + 4 * q, col = col0 + tc * 16 + lm;
                if (row < row_limit && col < col_limit) D[(size_t)row * ld + col] = ldexp(acc[tr][tc][q], -e);
            }
}

// Operator table.  k_big_table_raw: raw symbols C_s[i][j] = E[i][s] T[j][i] (one workgroup per symbol and
// parameter set).  k_big_table_level: merged tokens C_z = C_right * C_left (left is applied first), each
// normalised by a power of two; a token only depends on tokens of smaller depth, so the host launches one
// grid per dictionary depth and all tokens of that depth are built concurrently.
template <int NT>
__global__ __launch_bounds__(NT * 64) void k_big_table_raw(BigArgs a)
{
    constexpr int NP = 16 * NT, BIG_THREADS = NT * 64;
    const int tid = threadIdx.x, s = blockIdx.x, b = blockIdx.y;
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *Tp = pp + a.PP;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    double *Cs = a.Ctab + ((size_t)b * a.A + s) * NP * NP;
    for (int idx = tid; idx < NP * NP; idx += BIG_THREADS) {
        const int i = idx / NP, j = idx - i * NP;
        Cs[idx] = (i < a.N && j < a.N) ? Etg[(size_t)s * a.PP + i] * Tp[(size_t)j * a.PP + i] : 0.0;
    }
    if (a.Cpack) {
        double *Cp = a.Cpack + ((size_t)b * a.A + s) * a.N * a.TS;
        for (int idx = tid; idx < a.N * a.TS; idx += BIG_THREADS) {
            const int i = idx / a.TS, j = idx - i * a.TS;
            Cp[idx] = j < a.N ? Etg[(size_t)s * a.PP + i] * Tp[(size_t)j * a.PP + i] : 0.0;
        }
    }
    if (tid == 0) a.cex[(size_t)b * a.A + s] = 0;
}

template <int NT>
__global__ __launch_bounds__(NT * 64) void k_big_table_level(BigArgs a, const uint16_t *order, int first)
{
    constexpr int NP = 16 * NT;
    __shared__ unsigned long long smax[2];
    __shared__ __attribute__((aligned(16))) double panels[BigLds<NT>::bytes / 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int z = order[first + blockIdx.x];
    double *Ct = a.Ctab + (size_t)b * a.A * NP * NP;
    int *cex = a.cex + (size_t)b * a.A;
    if (tid < 2) smax[tid] = 0ull;
    __syncthreads();
    const int row0 = wave * 16, col0 = 0;
    const int zl = a.tok_left[z], zr = a.tok_right[z];
    if constexpr (NT <= 12) {
        v4f64 acc[1][NT];
        big_gemm<NT>(Ct + (size_t)zr * NP * NP, Ct + (size_t)zl * NP * NP, row0, col0, tid, panels, acc);
        const int e = big_exponent<NT>(acc, smax, 0, tid);
        big_store<NT>(Ct + (size_t)z * NP * NP, row0, col0, lane, acc, e, NP, NP, NP);
        if (a.Cpack)   // (the padded rows / columns of the product are exact zeros: columns N .. TS-1 get them)
            big_store<NT>(a.Cpack + ((size_t)b * a.A + z) * a.N * a.TS, row0, col0, lane, acc, e, a.N, a.TS, (size_t)a.TS);
        if (tid == 0) cex[z] = cex[zl] + cex[zr] + e;
    } else {
        // 14-16 wavefronts leave 128 registers per lane: the product is formed in two column halves, stored
        // unscaled, and rescaled in place once the maximum over both halves is known (every lane rewrites
        // exactly the elements it stored)
        constexpr int HALF = NT / 2;
        static_assert(NT % 2 == 0, "column halves");
        double *D = Ct + (size_t)z * NP * NP;
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            v4f64 acc[1][HALF];
            big_gemm<NT, HALF>(Ct + (size_t)zr * NP * NP, Ct + (size_t)zl * NP * NP, row0, h * HALF * 16, tid, panels, acc);
            const double mx = big_tile_max<HALF>(acc);
            if (lane == 0) atomicMax(&smax[0], (unsigned long long)__double_as_longlong(mx));
            big_store<HALF>(D, row0, h * HALF * 16, lane, acc, 0, NP, NP, NP);
            __syncthreads();   // the LDS panels are reused by the next half
        }
        const double m = __longlong_as_double((long long)smax[0]);
        int e = 0;
        (void)frexp(m, &e);
        e = (m > 0.0 && m < INFINITY) ? e : 0;
        const int lm = lane & 15, lg = lane >> 4;
#pragma unroll 1
        for (int tc = 0; tc < NT; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + lg + 4 * q, col = tc * 16 + lm;
                double *p = D + (size_t)row * NP + col;
                const double v = ldexp(*p, -e);
                *p = v;
                if (a.Cpack && row < a.N && col < a.TS) a.Cpack[((size_t)b * a.A + z) * a.N * a.TS + (size_t)row * a.TS + col] = v;
            }
        if (tid == 0) cex[z] = cex[zl] + cex[zr] + e;
    }
}

// One workgroup per (segment, column slab, parameter set): P[:, slab] <- C_tok * P[:, slab] over the segment's
// tokens.  Columns of P are independent, so a segment is split into NSLAB column slabs of SC = NP/NSLAB
// columns that live in LDS for the whole launch (the B operand is read straight from the slab, the result
// is written back into it after each step): P never touches global memory, and the only streamed operand
// is the token operator's 16-deep A panels (double-buffered through LDS).  Wavefront w owns tile-row w:
// NT/NSLAB accumulator tiles.  Slabs of one segment are given block ids 8 apart so that they land on the
// same XCD and share the operator rows in its L2.
struct BigBlock {
    uint32_t seg;        // segment id
    uint32_t slab;       // column slab index
    uint32_t out_vec0;   // level-0 vector index of the segment's result
    uint32_t pad;
};

template <int NT, int NSLAB>
struct BigSlab {
    static constexpr int NP = 16 * NT;
    static constexpr int SC = NP / NSLAB;       // columns per slab
    static constexpr int TCS = NT / NSLAB;      // tile-columns per slab
    static constexpr int SPS = SC + 4;          // slab row stride (doubles): k rows 4 apart land 32 banks apart
    static constexpr int APS = 18;              // A panel row stride
    static constexpr int SLAB_DOUBLES = NP * SPS, A_DOUBLES = NP * APS;
    static constexpr size_t bytes = (size_t)(SLAB_DOUBLES + 2 * A_DOUBLES) * 8;
    static_assert(NT % NSLAB == 0, "tile columns must divide evenly over the slabs");
};

template <int NT, int NSLAB>
__global__ __launch_bounds__(NT * 64) void k_big_propagate(BigArgs a, const BigBlock *blocks)
{
    using G = BigSlab<NT, NSLAB>;
    constexpr int NP = G::NP, SC = G::SC, TCS = G::TCS, SPS = G::SPS, APS = G::APS, THREADS = NT * 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *slab = lds;                              // [NP][SPS]  P[k][c - c0]
    double *Apan = lds + G::SLAB_DOUBLES;            // [2][NP][APS]
    __shared__ unsigned long long smax[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y;
    const BigBlock bk = blocks[blockIdx.x];
    const SegDesc sd = a.segs[bk.seg];
    const bool first = (sd.first & SEG_FIRST) != 0, wide = (sd.first & SEG_WIDE) != 0;
    const int len = (int)sd.len;
    const uint8_t *tokp = sd.obs;
    const int c0 = (int)bk.slab * SC;                // first global column of this slab
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    const double *Ct = a.Ctab + (size_t)b * a.A * NP * NP;
    const int *cex = a.cex + (size_t)b * a.A;

    // rounds of the rank-one hand-off: which tokens this launch covers
    const bool tail = a.phase == 1 && a.t_from > 0;   // continue from the stored block
    if (a.phase != 0 && first) return;             // hand-off mode: first segments run on the mat-vec chain kernel
    if (tail && (len <= a.t_from || a.r1flag[(size_t)b * a.n_segs + bk.seg])) return;   // finished, or certified: the mat-vec chain takes over
    const int t_end = (a.phase == 1 && len > a.t_to) ? a.t_to : len;
    const size_t gv = (size_t)b * a.n_vecs_total + bk.out_vec0;
    double *Pout = a.P + gv * NP;
    // initial slab: identity columns, or (first segment, slab 0) column 0 = pi .* E[:,o_0], or (tail) the stored block
    const int tok0 = first ? seg_token(tokp, wide, 0) : 0;
    for (int idx = tid; idx < NP * SC; idx += THREADS) {
        const int k = idx / SC, c = idx - k * SC;
        double v;
        if (tail) v = (k < a.N && c0 + c < NP) ? Pout[(size_t)k * NP + c0 + c] : 0.0;
        else if (first) v = (c0 + c == 0 && k < a.N) ? pp[k] * Etg[(size_t)tok0 * a.PP + k] : 0.0;
        else v = (k == c0 + c && k < a.N) ? 1.0 : 0.0;
        slab[k * SPS + c] = v;
    }
    if (tid < 2) smax[tid] = 0ull;
    __syncthreads();

    // A-panel staging: NP rows x 8 double2 per panel, two per thread
    const int e0 = tid, e1 = tid + THREADS;
    const int ar0 = e0 >> 3, ac0 = e0 & 7, ar1 = e1 >> 3, ac1 = e1 & 7;
    const int lA0 = ar0 * APS + 2 * ac0, lA1 = ar1 * APS + 2 * ac1;
    const int row0 = wave * 16;
    long long ex = tail ? (long long)a.EX[gv + (c0 < a.N ? c0 : 0)] : 0;
    int which = 0;
    // the first A panel of the NEXT token is fetched while the current step finishes (exponent, write-back),
    // so no step starts with an exposed L2 round trip
    const int t_begin = tail ? a.t_from : first ? 1 : 0;
    const size_t aoff0 = (size_t)ar0 * NP + 2 * ac0, aoff1 = (size_t)ar1 * NP + 2 * ac1;
    double2 sa0 = double2{0.0, 0.0}, sa1 = double2{0.0, 0.0};
    if (t_begin < t_end) {
        const double *A0 = Ct + (size_t)seg_token(tokp, wide, t_begin) * NP * NP;
        sa0 = *reinterpret_cast<const double2 *>(A0 + aoff0);
        sa1 = *reinterpret_cast<const double2 *>(A0 + aoff1);
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int tok = seg_token(tokp, wide, t);
        const double *A = Ct + (size_t)tok * NP * NP;
        const double *gA0 = A + aoff0, *gA1 = A + aoff1;
        v4f64 acc[TCS];
#pragma unroll
        for (int tc = 0; tc < TCS; ++tc) acc[tc] = v4f64{0.0, 0.0, 0.0, 0.0};
        *reinterpret_cast<double2 *>(Apan + lA0) = sa0;
        *reinterpret_cast<double2 *>(Apan + lA1) = sa1;
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < NP / 16; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < NP / 16) {
                sa0 = *reinterpret_cast<const double2 *>(gA0 + (kb + 1) * 16);
                sa1 = *reinterpret_cast<const double2 *>(gA1 + (kb + 1) * 16);
            }
            const double *Al = Apan + buf * G::A_DOUBLES;
            const double2 *ap = reinterpret_cast<const double2 *>(Al + (row0 + lm) * APS + 4 * lg);
            const double2 a01 = ap[0], a23 = ap[1];
            const double av[4] = {a01.x, a01.y, a23.x, a23.y};
            const double *Bk = slab + (size_t)(kb * 16 + 4 * lg) * SPS + lm;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int tc = 0; tc < TCS; ++tc)
                    acc[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], Bk[s2 * SPS + tc * 16], acc[tc], 0, 0, 0);
            if (kb + 1 < NP / 16) {
                double *An = Apan + (buf ^ 1) * G::A_DOUBLES;
                *reinterpret_cast<double2 *>(An + lA0) = sa0;
                *reinterpret_cast<double2 *>(An + lA1) = sa1;
            }
            __syncthreads();
        }
        if (t + 1 < t_end) {   // prefetch the next token's first panel
            const double *An = Ct + (size_t)seg_token(tokp, wide, t + 1) * NP * NP;
            sa0 = *reinterpret_cast<const double2 *>(An + aoff0);
            sa1 = *reinterpret_cast<const double2 *>(An + aoff1);
        }
        // one power-of-two scale for the slab: exponent of its largest entry
        double mx = 0.0;
#pragma unroll
        for (int tc = 0; tc < TCS; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double v = acc[tc][q];
                mx = (v > mx || v != v) ? v : mx;
            }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double o = __shfl_xor(mx, m, 64);
            mx = (o > mx || o != o) ? o : mx;
        }
        if (lane == 0) atomicMax(&smax[which], (unsigned long long)__double_as_longlong(mx));
        __syncthreads();   // also: every wavefront has finished reading the slab
        const double m = __longlong_as_double((long long)smax[which]);
        if (tid == 0) smax[which ^ 1] = 0ull;
        which ^= 1;
        int e = 0;
        (void)frexp(m, &e);
        e = (m > 0.0 && m < INFINITY) ? e : 0;
#pragma unroll
        for (int tc = 0; tc < TCS; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) slab[(size_t)(row0 + lg + 4 * q) * SPS + tc * 16 + lm] = ldexp(acc[tc][q], -e);
        ex += cex[tok] + e;
        __syncthreads();
    }
    // results -> level 0: operator block state-major [i][c] (N x NP), or the vector [i] for a first segment
    if (first) {
        if (bk.slab == 0) {
            for (int i = tid; i < a.N; i += THREADS) Pout[i] = slab[i * SPS];
            if (tid == 0) a.EX[gv] = (int)ex;
        }
    } else {
        for (int idx = tid; idx < a.N * SC; idx += THREADS) {
            const int i = idx / SC, c = idx - i * SC;
            if (c0 + c < NP) Pout[(size_t)i * NP + c0 + c] = (c0 + c < a.N) ? slab[i * SPS + c] : 0.0;
        }
        for (int c = tid; c < SC; c += THREADS)
            if (c0 + c < a.N) a.EX[gv + c0 + c] = (int)ex;
    }
}

// Mat-vec chain for plans in which every chunk is ONE segment (many chunks x parameter sets, e.g. 64 proposals x
// 32 one-Mbp chunks at N = 150): no transfer operators are needed at all, so the work per token step drops from a
// GEMM (2 NP^3 flops) to x <- C_tok x (2 NP^2) and the kernel is bound by streaming the token operator
// (NP^2 x 8 B per step) from L2 / MALL / HBM.  One workgroup of 8 or 16 wavefronts per (chunk, parameter set):
//   * 16 lanes share an operator row: lane lm takes two adjacent k per 16-byte load, so one load instruction of a
//     wavefront covers four rows x 256 contiguous bytes; the 16 partial sums are folded with four DPP row rotations (no LDS);
//   * x lives in LDS, double buffered; the power-of-two scale of step t is applied when step t+1 reads x, which
//     leaves ONE barrier per step (three rotating slots hold the per-step maximum);
//   * workgroup ids are dealt so that all chunks of one parameter set run on one XCD back to back: they share that
//     set's operator table in the XCD's L2.
template <int NT>
struct BigVec {
    // every pass covers 4 WAVES rows and the passes tile NP = 16 NT rows exactly; <= 48 operator doubles per lane
    static constexpr int WAVES = NT == 3 ? 12 : (NT == 7 || NT == 9 || NT == 14) ? NT : NT > 12 ? 16 : 8;
    static constexpr bool PIPELINED = NT <= 10;      // beyond that the double set of operator registers would spill
    // not pipelined: all of a step's loads are issued before its first use while they fit the registers (NT = 12;
    // measured at NT = 10: loading pass by pass instead costs 50 %), pass by pass beyond
    static constexpr bool LOAD_AHEAD = !PIPELINED && NT <= 12;
};

// TAIL = true is the second half of the rank-one hand-off (k_rank1_check): the chain starts from the collapsed
// operator's column direction u after r1at tokens and ends by writing the operator block u' alpha^T.
template <int NT, bool TAIL = false>
__global__ __launch_bounds__(BigVec<NT>::WAVES * 64) void k_big_vector(BigArgs a, const BigBlock *blocks, int n_blocks, int B)
{
    constexpr int BV_WAVES = BigVec<NT>::WAVES;
    constexpr int NP = 16 * NT, THREADS = BV_WAVES * 64, ROWS_PER_PASS = BV_WAVES * 4;
    constexpr int PASSES = NP / ROWS_PER_PASS;
    static_assert(PASSES * ROWS_PER_PASS == NP, "passes must tile the operator rows exactly");
    constexpr int W = NT % 2 == 0 ? 2 : 1, NJ = NT / W;   // lane lm takes k = W lm + 16 W jj + (0..W-1): 16-byte loads when NT is even
    __shared__ __attribute__((aligned(16))) double xs[2][NP];
    __shared__ unsigned long long smax[3];
    // B >= 8: all chains of one parameter set on one XCD (they share its table in that L2); fewer sets: spread the
    // chains over every XCD (grid = n_blocks * B, consecutive ids go round the XCDs)
    int b, blk;
    if (B >= 8) {
        const int xcd = blockIdx.x & 7, turn = blockIdx.x >> 3;
        b = xcd + 8 * (turn / n_blocks);
        blk = turn % n_blocks;
    } else {
        b = blockIdx.x % B;
        blk = blockIdx.x / B;
    }
    if (b >= B || blk >= n_blocks) return;
    const BigBlock bk = blocks[blk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lg = lane >> 4;
    const SegDesc sd = a.segs[bk.seg];
    const int len = (int)sd.len;
    const bool wide = (sd.first & SEG_WIDE) != 0;
    const uint8_t *tokp = sd.obs;
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    const bool packed = a.Cpack != nullptr;
    const int ld = packed ? a.TS : NP;                               // row stride and entry size of the table that is read
    const size_t esz = packed ? (size_t)a.N * a.TS : (size_t)NP * NP;
    const double *Ct = packed ? a.Cpack + (size_t)b * a.A * esz : a.Ctab + (size_t)b * a.A * esz;
    const int *cex = a.cex + (size_t)b * a.A;

    const size_t r1 = ((size_t)b * a.n_segs + bk.seg) * NP;
    // TAIL launches cover every segment of the hand-off group: a chunk's first segment is a plain vector chain from
    // pi, an operator segment continues from u if (and only if) its head was certified rank one
    const bool from_u = TAIL && !(sd.first & SEG_FIRST);
    if (from_u) {
        if (!a.r1flag[(size_t)b * a.n_segs + bk.seg]) return;   // (workgroup-uniform)
        for (int k = tid; k < NP; k += THREADS) xs[0][k] = k < a.N ? a.r1u[r1 + k] : 0.0;
    } else {
        const int tok0 = seg_token(tokp, wide, 0);
        for (int k = tid; k < NP; k += THREADS) xs[0][k] = k < a.N ? pp[k] * Etg[(size_t)tok0 * a.PP + k] : 0.0;
    }
    const int t0 = from_u ? a.r1at[(size_t)b * a.n_segs + bk.seg] : 1;        // first token applied as an operator
    if (tid < 3) smax[tid] = 0ull;
    __syncthreads();

    // The operator rows of step t+1 do not depend on x: as soon as a pass has consumed its registers they are
    // refilled with the same rows of the NEXT token's operator, so a full operator (NP^2 x 8 B) is always in
    // flight and no step starts with an exposed memory round trip.
    long long ex = 0;
    int e_prev = 0, cur = 0, slot = 0;
    int rowoff[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * ROWS_PER_PASS + wave * 4 + lg;
        rowoff[ps] = row * ld + W * lm;
    }
    // The table is zero beyond N: the padded rows (last pass) and the padded columns (last piece of a row) are not
    // fetched at all - their registers stay 0 - which saves up to 12 % of the streamed bytes (N = 150 in NP = 160).
    const bool last_rows_live = (PASSES - 1) * ROWS_PER_PASS + wave * 4 + lg < a.N;
    const bool last_cols_live = W * lm + 16 * W * (NJ - 1) < a.N;
    double av[PASSES][NT];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
        for (int j = 0; j < NT; ++j) av[ps][j] = 0.0;
#define BV_LOAD(ps_, src_)                                                                          \
    do {                                                                                            \
        _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                         \
            if (((ps_) < PASSES - 1 || last_rows_live) && (jj < NJ - 1 || last_cols_live)) {        \
                if constexpr (W == 2) {                                                             \
                    const double2 v_ = *reinterpret_cast<const double2 *>((src_) + rowoff[ps_] + 32 * jj); \
                    av[ps_][2 * jj] = v_.x;                                                         \
                    av[ps_][2 * jj + 1] = v_.y;                                                     \
                } else {                                                                            \
                    av[ps_][jj] = (src_)[rowoff[ps_] + 16 * jj];                                    \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    } while (0)
    int tok = len > t0 ? seg_token(tokp, wide, t0) : 0;
    int tok_next = len > t0 + 1 ? seg_token(tokp, wide, t0 + 1) : tok;
    if constexpr (BigVec<NT>::PIPELINED) {
        const double *A = Ct + (size_t)tok * esz;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) BV_LOAD(ps, A);
    }
    for (int t = t0; t < len; ++t) {
        const double *An = Ct + (size_t)(BigVec<NT>::PIPELINED ? tok_next : tok) * esz;
        if constexpr (BigVec<NT>::LOAD_AHEAD) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) BV_LOAD(ps, An);
        }
        const int tok_after = t + 2 < len ? seg_token(tokp, wide, t + 2) : tok_next;
        const double *xc = &xs[cur][W * lm];   // x of step t-1 as stored (before its scale): the scale is applied to the dot products
        double mx = 0.0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = ps * ROWS_PER_PASS + wave * 4 + lg;
            if constexpr (!BigVec<NT>::PIPELINED && !BigVec<NT>::LOAD_AHEAD) BV_LOAD(ps, An);
            double p = 0.0;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                if constexpr (W == 2) {
                    const double2 xv = *reinterpret_cast<const double2 *>(xc + 32 * jj);
                    p = fma(av[ps][2 * jj], xv.x, p);
                    p = fma(av[ps][2 * jj + 1], xv.y, p);
                } else {
                    p = fma(av[ps][jj], xc[16 * jj], p);
                }
            }
            __builtin_amdgcn_sched_barrier(0);     // refill the registers just consumed, not fresh ones
            if constexpr (BigVec<NT>::PIPELINED) BV_LOAD(ps, An);   // (last step: a harmless reload)
            p += dpp_f64<DPP_ROW_ROR8>(p);
            p += dpp_f64<DPP_ROW_ROR4>(p);
            p += dpp_f64<DPP_ROW_ROR2>(p);
            p += dpp_f64<DPP_ROW_ROR1>(p);
            p = ldexp(p, -e_prev);
            if (lm == 0) xs[cur ^ 1][row] = p;
            mx = (p > mx || p != p) ? p : mx;
        }
#pragma unroll
        for (int m = 32; m >= 16; m >>= 1) {
            const double o = __shfl_xor(mx, m, 64);
            mx = (o > mx || o != o) ? o : mx;
        }
        if (lane == 0) atomicMax(&smax[slot], (unsigned long long)__double_as_longlong(mx));
        const int next_slot = slot == 2 ? 0 : slot + 1;
        if (tid == 0) smax[next_slot] = 0ull;      // last read two steps ago, next written after this step's barrier
        __syncthreads();
        const double m = __longlong_as_double((long long)smax[slot]);
        int e = 0;
        (void)frexp(m, &e);
        e = (m > 0.0 && m < INFINITY) ? e : 0;
        ex += cex[tok] + e;
        e_prev = e;
        cur ^= 1;
        slot = next_slot;
        tok = tok_next;
        tok_next = tok_after;
    }
    const size_t gv = (size_t)b * a.n_vecs_total + bk.out_vec0;
    double *Pout = a.P + gv * NP;
    if (from_u) {   // the operator block u' alpha^T, state-major [i][c]; column exponents grow by the tail's
        for (int idx = tid; idx < a.N * NP; idx += THREADS) {
            const int i = idx / NP, c = idx - i * NP;
            Pout[idx] = c < a.N ? ldexp(xs[cur][i], -e_prev) * a.r1alpha[r1 + c] : 0.0;
        }
        for (int c = tid; c < a.N; c += THREADS) a.EX[gv + c] += (int)ex;
    } else {
        for (int i = tid; i < a.N; i += THREADS) Pout[i] = ldexp(xs[cur][i], -e_prev);
        if (tid == 0) a.EX[gv] = (int)ex;
    }
#undef BV_LOAD
}

// k_big_propagate with the output tiles of a step dealt round-robin over EIGHT wavefronts instead of one tile-row per
// wavefront.  With NT = 6, 10 or 14 tile-rows the row-per-wavefront layout loads the four SIMDs unevenly (10
// wavefronts = 3/3/2/2: the MFMA pipe of two SIMDs idles a third of the time); 8 wavefronts x (NT * TCS / 8) tiles
// puts 13/13/12/12 tiles per step on the SIMDs at NT = 10.  Price: a wavefront's tiles no longer share a tile-row,
// so the A fragment is read from the LDS panel once per tile instead of once per tile-row (1.5 LDS reads per MFMA
// instead of 1.1) - still well inside the LDS bandwidth.  Everything else (slab resident in LDS, double-buffered
// 16-deep A panels, one power-of-two scale per step) is k_big_propagate's.
static constexpr int BS_WAVES = 8;

template <int NT, int NSLAB>
__global__ __launch_bounds__(BS_WAVES * 64) void k_big_propagate_s(BigArgs a, const BigBlock *blocks)
{
    using G = BigSlab<NT, NSLAB>;
    constexpr int NP = G::NP, SC = G::SC, TCS = G::TCS, SPS = G::SPS, APS = G::APS, THREADS = BS_WAVES * 64;
    // NT >= 8: wavefront w owns tile-row w (TCS tiles sharing one A fragment) plus its share of the rows beyond 8,
    // dealt tile by tile; NT < 8: all tiles dealt round-robin
    constexpr bool ROWS = NT >= BS_WAVES;
    constexpr int NTILES = NT * TCS, NEXTRA = ROWS ? (NT - BS_WAVES) * TCS : NTILES;
    constexpr int JROW = ROWS ? TCS : 0, JMAX = JROW + (NEXTRA + BS_WAVES - 1) / BS_WAVES;
    constexpr int JMAX_ALL = JROW + NEXTRA / BS_WAVES;   // tiles every wavefront owns
    constexpr int PANEL_D2 = NP * 8, EPT = (PANEL_D2 + THREADS - 1) / THREADS;   // double2 per panel, per thread
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *slab = lds;                              // [NP][SPS]  P[k][c - c0]
    double *Apan = lds + G::SLAB_DOUBLES;            // [2][NP][APS]
    __shared__ unsigned long long smax[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y;
    const BigBlock bk = blocks[blockIdx.x];
    const SegDesc sd = a.segs[bk.seg];
    const bool first = (sd.first & SEG_FIRST) != 0, wide = (sd.first & SEG_WIDE) != 0;
    const int len = (int)sd.len;
    const uint8_t *tokp = sd.obs;
    const int c0 = (int)bk.slab * SC;                // first global column of this slab
    const double *pp = a.params + (size_t)b * a.pstride;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    const double *Ct = a.Ctab + (size_t)b * a.A * NP * NP;
    const int *cex = a.cex + (size_t)b * a.A;

    // rounds of the rank-one hand-off: which tokens this launch covers
    const bool tail = a.phase == 1 && a.t_from > 0;   // continue from the stored block
    if (a.phase != 0 && first) return;             // hand-off mode: first segments run on the mat-vec chain kernel
    if (tail && (len <= a.t_from || a.r1flag[(size_t)b * a.n_segs + bk.seg])) return;   // finished, or certified: the mat-vec chain takes over
    const int t_end = (a.phase == 1 && len > a.t_to) ? a.t_to : len;
    const size_t gv = (size_t)b * a.n_vecs_total + bk.out_vec0;
    double *Pout = a.P + gv * NP;
    // initial slab: identity columns, or (first segment, slab 0) column 0 = pi .* E[:,o_0], or (tail) the stored block
    const int tok0 = first ? seg_token(tokp, wide, 0) : 0;
    for (int idx = tid; idx < NP * SC; idx += THREADS) {
        const int k = idx / SC, c = idx - k * SC;
        double v;
        if (tail) v = (k < a.N && c0 + c < NP) ? Pout[(size_t)k * NP + c0 + c] : 0.0;
        else if (first) v = (c0 + c == 0 && k < a.N) ? pp[k] * Etg[(size_t)tok0 * a.PP + k] : 0.0;
        else v = (k == c0 + c && k < a.N) ? 1.0 : 0.0;
        slab[k * SPS + c] = v;
    }
    if (tid < 2) smax[tid] = 0ull;
    __syncthreads();

    // this wavefront's tiles: q = wave + 8 j  ->  (tile-row q / TCS, tile-column q % TCS)
    int a_off[JMAX], b_off[JMAX], w_off[JMAX];
    const int n_own = JROW + (NEXTRA - wave + BS_WAVES - 1) / BS_WAVES;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        int tr, tc;
        if (j < JROW) { tr = wave; tc = j; }
        else {
            const int x = wave + BS_WAVES * (j - JROW);
            const int q = x < NEXTRA ? x : 0;                      // (unused slots repeat a valid tile)
            tr = (ROWS ? BS_WAVES : 0) + q / TCS;
            tc = q % TCS;
        }
        a_off[j] = (tr * 16 + lm) * APS + 4 * lg;          // A fragment in the LDS panel
        b_off[j] = 4 * lg * SPS + tc * 16 + lm;            // B fragment in the slab (+ kb * 16 * SPS + s2 * SPS)
        w_off[j] = (tr * 16 + lg) * SPS + tc * 16 + lm;    // D tile write-back (+ 4 q * SPS)
    }
    // A-panel staging: NP rows x 8 double2 per panel, EPT per thread (the last one guarded)
    int lA[EPT];
    size_t gAo[EPT];
    bool act[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * THREADS;
        act[k] = e < PANEL_D2;
        const int ar = (act[k] ? e : 0) >> 3, ac = e & 7;
        lA[k] = ar * APS + 2 * ac;
        gAo[k] = (size_t)ar * NP + 2 * ac;
    }
    long long ex = tail ? (long long)a.EX[gv + (c0 < a.N ? c0 : 0)] : 0;
    int which = 0;
    const int t_begin = tail ? a.t_from : first ? 1 : 0;
    // Shapes with NT >= 8 take their A fragments straight from global memory (below); the smaller ones, whose tiles are
    // all dealt round-robin (one fragment per tile: too many registers), keep the LDS-staged panels.
    constexpr bool DIRECT_A = ROWS;
    // A fragments straight from the (L2-resident) operator table into registers, one 16-deep panel ahead: a tile-row
    // belongs to ONE wavefront (only the rows beyond 8 are dealt tile by tile), so nothing is shared through LDS and
    // the k loop runs without a barrier - the LDS-staged panels cost one __syncthreads per panel, ten per step at
    // N = 150, and a step ran at 72 % of its MFMA time.  Fragment f: f = 0 the wavefront's own tile-row (ROWS), then one
    // per dealt tile; lane (lm, lg) takes A[16 tr + lm][16 kb + 4 lg .. +3] (32 contiguous bytes, four lanes a line).
    constexpr int NF = DIRECT_A ? 1 + (JMAX - JROW) : 1;
    size_t fo[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int j = ROWS ? (f == 0 ? 0 : JROW + f - 1) : f;
        fo[f] = (size_t)(a_off[j] / APS) * NP + 4 * lg;          // a_off[j] = (16 tr + lm) * APS + 4 lg
    }
    // THREE register sets: panel kb is consumed from set kb % 3 while the loads of panels kb + 1 and kb + 2 are in
    // flight (every step's operator is new to the chip - 205 KB per segment and step, 2.7 TB/s over the launch - so a
    // fetch is an HBM round trip, longer than the 1.5 us one panel's MFMAs take; with a single set ahead every panel
    // ended in a wait).  The k loop is unrolled over the step's NT panels so that the set of a panel is a compile-time
    // name; between steps the two sets in flight are renamed (register moves behind the step's barriers).
    double2 fr[3][NF][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int f = 0; f < NF; ++f) fr[d][f][0] = fr[d][f][1] = double2{0.0, 0.0};
    auto fetch = [&](double2 (&dst)[NF][2], const double *A, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double2 *src = reinterpret_cast<const double2 *>(A + fo[f] + kb * 16);
            dst[f][0] = src[0];
            dst[f][1] = src[1];
        }
    };
    int tok = t_begin < t_end ? seg_token(tokp, wide, t_begin) : 0;
    double2 sa[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) sa[k] = double2{0.0, 0.0};
    if (t_begin < t_end) {
        const double *A0 = Ct + (size_t)tok * NP * NP;
        if constexpr (DIRECT_A) { fetch(fr[0], A0, 0); fetch(fr[1], A0, 1); }
        else {
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (act[k]) sa[k] = *reinterpret_cast<const double2 *>(A0 + gAo[k]);
        }
    }
    for (int t = t_begin; t < t_end; ++t) {
        int cex_cur = 0;
        v4f64 acc[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) acc[j] = v4f64{0.0, 0.0, 0.0, 0.0};
        if constexpr (DIRECT_A) {
            const double *A = Ct + (size_t)tok * NP * NP;
            cex_cur = cex[tok];
            const int tok_next = t + 1 < t_end ? seg_token(tokp, wide, t + 1) : tok;
            const double *An = Ct + (size_t)tok_next * NP * NP;
#pragma unroll
            for (int kb = 0; kb < NT; ++kb) {
                // (behind the token's last panels: the next token's first two; last step: a harmless reload)
                if (kb + 2 < NT) fetch(fr[(kb + 2) % 3], A, kb + 2);
                else fetch(fr[(kb + 2) % 3], An, kb + 2 - NT);
                const double2 (&fc)[NF][2] = fr[kb % 3];
                const double *Bk = slab + (size_t)kb * 16 * SPS;
                // One pass over k = 4 s2 .. 4 s2 + 3 for ALL of the wavefront's tiles: the four products of a tile are
                // dependent (same accumulator) and stay a tile-count apart, and every B operand of the panel can be read
                // ahead of its product (the dealt tiles used to run one after the other, each behind its own LDS read).
                // Only the product of the LAST dealt tile is conditional (waves below NEXTRA % 8 own one more tile; its
                // operands are valid for every wavefront - unused slots repeat tile 0).
                double bv[4][JMAX];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int j = 0; j < JMAX; ++j)
                        bv[s2][j] = (ROWS && j < JROW) ? Bk[b_off[0] + s2 * SPS + j * 16] : Bk[b_off[j] + s2 * SPS];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
                    for (int j = 0; j < JMAX; ++j) {
                        const int f = (ROWS && j < JROW) ? 0 : (ROWS ? 1 : 0) + j - JROW;
                        const double2 ah = fc[f][s2 >> 1];
                        const double av = (s2 & 1) ? ah.y : ah.x;
                        if (j < JMAX_ALL || j < n_own)     // (compile-time true except for the last dealt tile)
                            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[s2][j], acc[j], 0, 0, 0);
                    }
                }
            }
            tok = tok_next;
        } else {
            const int tok_l = seg_token(tokp, wide, t);
            const double *A = Ct + (size_t)tok_l * NP * NP;
            cex_cur = cex[tok_l];
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (act[k]) *reinterpret_cast<double2 *>(Apan + lA[k]) = sa[k];
            __syncthreads();
#pragma unroll 1
            for (int kb = 0; kb < NP / 16; ++kb) {
                const int buf = kb & 1;
                if (kb + 1 < NP / 16) {
#pragma unroll
                    for (int k = 0; k < EPT; ++k)
                        if (act[k]) sa[k] = *reinterpret_cast<const double2 *>(A + gAo[k] + (kb + 1) * 16);
                }
                const double *Al = Apan + buf * G::A_DOUBLES;
                const double *Bk = slab + (size_t)kb * 16 * SPS;
                if constexpr (ROWS) {   // the wavefront's own tile-row: one A fragment for TCS tiles
                    const double2 *ap = reinterpret_cast<const double2 *>(Al + a_off[0]);
                    const double2 a01 = ap[0], a23 = ap[1];
                    const double av[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int j = 0; j < JROW; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], Bk[b_off[0] + s2 * SPS + j * 16], acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = JROW; j < JMAX; ++j) {
                    if (j < n_own) {   // wave-uniform
                        const double2 *ap = reinterpret_cast<const double2 *>(Al + a_off[j]);
                        const double2 a01 = ap[0], a23 = ap[1];
                        const double av[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2)
                            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], Bk[b_off[j] + s2 * SPS], acc[j], 0, 0, 0);
                    }
                }
                if (kb + 1 < NP / 16) {
                    double *An = Apan + (buf ^ 1) * G::A_DOUBLES;
#pragma unroll
                    for (int k = 0; k < EPT; ++k)
                        if (act[k]) *reinterpret_cast<double2 *>(An + lA[k]) = sa[k];
                }
                __syncthreads();
            }
            if (t + 1 < t_end) {   // prefetch the next token's first panel
                const double *An = Ct + (size_t)seg_token(tokp, wide, t + 1) * NP * NP;
#pragma unroll
                for (int k = 0; k < EPT; ++k)
                    if (act[k]) sa[k] = *reinterpret_cast<const double2 *>(An + gAo[k]);
            }
        }
        // One power-of-two scale for the slab: the exponent of its largest entry - read off the HIGH words alone (entries
        // are non-negative, so the sign-stripped high 32 bits order them as the values do, and a NaN / infinity has the
        // largest exponent field of all): 32-bit maxima and DPP moves instead of 28 fp64 compare-and-select pairs and
        // six cross-lane round trips per wavefront and step.  (A largest entry below 2^-1022 counts as zero: no scale.)
        int hmax = 0;
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j < JMAX_ALL || j < n_own) hmax = max(hmax, __double2hiint(acc[j][q]) & 0x7fffffff);
        hmax = wave_max_i32_dpp(hmax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int *>(&smax[which]), (unsigned int)hmax);
        __syncthreads();   // also: every wavefront has finished reading the slab
        const int field = (int)(*reinterpret_cast<const unsigned int *>(&smax[which]) >> 20);
        if (tid == 0) smax[which ^ 1] = 0ull;
        which ^= 1;
        const int e = (field > 0 && field < 0x7ff) ? field - 1022 : 0;   // frexp's exponent: largest entry in [2^(e-1), 2^e)
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            if (j < n_own) {
#pragma unroll
                for (int q = 0; q < 4; ++q) slab[w_off[j] + 4 * q * SPS] = ldexp(acc[j][q], -e);
            }
        ex += cex_cur + e;
        __syncthreads();
        if constexpr (DIRECT_A && NT % 3 != 0) {   // the next token's panels 0 and 1 sit in sets NT % 3 and (NT + 1) % 3
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double2 p0 = fr[NT % 3][f][h], p1 = fr[(NT + 1) % 3][f][h];
                    fr[0][f][h] = p0;
                    fr[1][f][h] = p1;
                }
        }
    }
    // results -> level 0: operator block state-major [i][c] (N x NP), or the vector [i] for a first segment
    if (first) {
        if (bk.slab == 0) {
            for (int i = tid; i < a.N; i += THREADS) Pout[i] = slab[i * SPS];
            if (tid == 0) a.EX[gv] = (int)ex;
        }
    } else {
        for (int idx = tid; idx < a.N * SC; idx += THREADS) {
            const int i = idx / SC, c = idx - i * SC;
            if (c0 + c < NP) Pout[(size_t)i * NP + c0 + c] = (c0 + c < a.N) ? slab[i * SPS + c] : 0.0;
        }
        for (int c = tid; c < SC; c += THREADS)
            if (c0 + c < a.N) a.EX[gv + c0 + c] = (int)ex;
    }
}

// Rank-one hand-off.  A product of many positive operators forgets its input: after enough columns every column of a
// segment's transfer operator P points in the same direction (Birkhoff contraction), P = u alpha^T, and the rest of
// the segment only has to propagate the VECTOR u - N^2 instead of N^3 work per step.  The hand-off is certified,
// not assumed: at a checkpoint (t_to tokens into the segment) this kernel tests, component-wise,
//     | P[i][c] s_* / (P[i][c*] s_c) - 1 | <= R1_TOL      for every i, c   (s_c = column sums, c* = the largest column)
// and only then records u = P[:, c*], alpha_c = s_c / s_*.  A component-wise relative bound survives every later
// non-negative linear map unchanged, so the log-likelihood moves by at most R1_TOL (2^-42 = 2.3e-13) per collapsed
// segment - absolute, in nats, on segments whose own log-likelihood is in the thousands.  Segments that fail the
// test (slow mixing, structural zeros) simply continue on the GEMM chain.
// The test is repeated at a fixed schedule of token counts (t_chk = a.t_to of the round that just ran): a segment
// hands off at the first checkpoint it passes, so the head length is decided per segment and per evaluation from
// the data alone - identical inputs give identical decisions (and bits), there is no state carried between calls.
static constexpr double R1_TOL = 2.2737367544323206e-13;   // 2^-42

__global__ __launch_bounds__(1024) void k_rank1_check(BigArgs a, const BigBlock *blocks, int NP)
{
    __shared__ double s_sum[256];
    __shared__ int s_star;
    __shared__ unsigned int s_bad;
    const int tid = threadIdx.x, b = blockIdx.y;
    const BigBlock bk = blocks[blockIdx.x];
    const SegDesc sd = a.segs[bk.seg];
    if (bk.slab != 0 || (sd.first & SEG_FIRST) || (int)sd.len <= a.t_to) return;
    if (a.r1flag[(size_t)b * a.n_segs + bk.seg]) return;      // certified at an earlier checkpoint (workgroup-uniform)
    const size_t gv = (size_t)b * a.n_vecs_total + bk.out_vec0;
    const double *P = a.P + gv * NP;
    const int N = a.N;
    // Quick reject (most checkpoints of most segments are too early): two columns of a certifiable operator are
    // proportional row by row to ~5e-13, so a spread of the row ratios P[i][N/2] / P[i][0] beyond 1e-9 (relative) -
    // three thousand times the tolerance of the real test below - means "not yet" without reading the whole block.
    if (N >= 2) {
        __shared__ double s_lo[256], s_hi[256];
        double lo = INFINITY, hi = 0.0;
        for (int i = tid; i < N; i += 256) {
            const double p0 = P[(size_t)i * NP], p1 = P[(size_t)i * NP + N / 2];
            if (p0 > 0.0 && p1 > 0.0 && p0 < INFINITY && p1 < INFINITY) {
                const double q = p1 / p0;
                lo = fmin(lo, q);
                hi = fmax(hi, q);
            }
        }
        if (tid < 256) { s_lo[tid] = lo; s_hi[tid] = hi; }
        __syncthreads();
        for (int h = 128; h >= 1; h >>= 1) {
            if (tid < h) { s_lo[tid] = fmin(s_lo[tid], s_lo[tid + h]); s_hi[tid] = fmax(s_hi[tid], s_hi[tid + h]); }
            __syncthreads();
        }
        if (s_hi[0] > 0.0 && s_hi[0] - s_lo[0] > 1e-9 * s_hi[0]) return;      // (workgroup-uniform)
    }
    // column sums: four row quarters per column in parallel, added in a fixed order (bit-identical repeats)
    {
        __shared__ double s_part[4][256];
        const int c = tid & 255, part = tid >> 8;                     // blockDim.x = 1024
        if (c < N) {
            const int i0 = (N * part) / 4, i1 = (N * (part + 1)) / 4;
            double t = 0.0;
            for (int i = i0; i < i1; ++i) t += P[(size_t)i * NP + c];
            s_part[part][c] = t;
        }
        __syncthreads();
        if (tid < N) s_sum[tid] = ((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid];
    }
    if (tid == 0) s_bad = 0u;
    __syncthreads();
    // reference column: the largest one (exponent first, then stored sum; ties: the lowest index) - an argmax over
    // N <= 256 columns, one per thread, folded through LDS
    {
        __shared__ int s_idx[256], s_ex[256];
        if (tid < 256) {
            s_idx[tid] = tid < N ? tid : -1;
            s_ex[tid] = tid < N ? a.EX[gv + tid] : 0;
        }
        __syncthreads();
        for (int h = 128; h >= 1; h >>= 1) {
            if (tid < h) {
                const int ia = s_idx[tid], ib = s_idx[tid + h];
                if (ib >= 0 && (ia < 0 || s_ex[tid + h] > s_ex[tid] || (s_ex[tid + h] == s_ex[tid] &&
                                (s_sum[ib] > s_sum[ia] || (s_sum[ib] == s_sum[ia] && ib < ia))))) {
                    s_idx[tid] = ib;
                    s_ex[tid] = s_ex[tid + h];
                }
            }
            __syncthreads();
        }
        if (tid == 0) s_star = s_idx[0];
    }
    __syncthreads();
    const int cs = s_star;
    const double ss = s_sum[cs];
    unsigned int bad = !(ss > 0.0 && ss < INFINITY);
    for (int idx = tid; idx < N * N; idx += blockDim.x) {
        const int i = idx / N, c = idx - i * N;
        const double u = P[(size_t)i * NP + cs], v = P[(size_t)i * NP + c], sc = s_sum[c];
        if (u == 0.0) bad |= (v != 0.0);
        else {
            const double r = (v * ss) / (u * sc);
            bad |= !(fabs(r - 1.0) <= R1_TOL);     // NaN / inf / zero column -> not certified
        }
    }
    if (bad) atomicOr(&s_bad, 1u);
    __syncthreads();
    const bool ok = s_bad == 0u;
    const size_t r1 = ((size_t)b * a.n_segs + bk.seg) * NP;
    if (!ok) return;                                           // the flag stays 0 (cleared at the start of the evaluation)
    if (tid == 0) {
        a.r1flag[(size_t)b * a.n_segs + bk.seg] = 1;
        a.r1at[(size_t)b * a.n_segs + bk.seg] = a.t_to;
    }
    {
        for (int k = tid; k < NP; k += blockDim.x) {
            a.r1u[r1 + k] = k < N ? P[(size_t)k * NP + cs] : 0.0;
            a.r1alpha[r1 + k] = k < N ? s_sum[k] / ss : 0.0;
        }
    }
}
