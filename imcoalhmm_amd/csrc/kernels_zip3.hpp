// kernels_zip3.hpp - register-blocked token kernel on the fp64 matrix instruction: P <- C_tok * P with the segment's whole
// transfer operator P (N x N, N <= 24) in registers and the step issued as v_mfma_f64_4x4x4_4b_f64.  Included by imcoal_fwd.hip.
//
// Why the matrix instruction here.  Measured on the MI355X (scratch microbenchmark): the DP FMA
// units are shared by v_fma_f64 and v_mfma_f64 (a wavefront of each on one SIMD do not overlap), but the same FMAs cost
// a quarter of the issue slots as MFMAs and draw less power: a bare v_fma_f64 loop sustains 50 TFLOP/s (4.7 cycles per
// wavefront instruction at two wavefronts per SIMD, clock held at ~1.8 GHz), v_mfma_f64_4x4x4_4b 68 TFLOP/s (256 FMAs
// per 12.75 cycles), and the MFMA form needs no cross-lane operand moves at all - k_zpropagate2 spends a fifth of its
// VALU issue slots on DPP moves and ran at 36 TFLOP/s.
//
// v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4x4x4 blocks.  Lane l = 16 q + 4 blk + r supplies
//     A[row r][k q] and B[k q][col r] of block blk and receives D[row q][col r]            (layout probed on the device)
// so a result tile is already in the layout of a B operand: the new P feeds the next step with no data movement.
//   * block blk of a wavefront = one segment (4 segments per wavefront, Z2SLOTS per workgroup, as in k_zpropagate2);
//   * P is NT x NT tiles of 4 x 4 (NP = 4 NT): lane (q, blk, r) holds element (4K + q, 4J + r) of every tile (K, J);
//   * per step, for every tile-row I the lane reads its NT A operands C_tok[4I + r][4K + q] from the LDS table (stored
//     so that these are NT consecutive doubles) and issues NT*NT MFMAs: NT^3 per step, no other vector work;
//   * a step that must leave a segment untouched (ragged ends, idle slots) uses the IDENTITY entry of the table, so
//     the loop has no predication; the workgroup's segments are folded at the end exactly as in k_zpropagate2.
#pragma once
#include "kernels_zip2.hpp"

template <int NT>
struct Zip3Geom {
    static constexpr int NP = 4 * NT;
    static constexpr int NTE = NT & ~1;                 // K values a lane reads as 16-byte pairs
    static constexpr int MAIN = NT * 16 * NTE;          // doubles: [I][q][r][K < NTE]
    static constexpr int EXTRA = (NT & 1) ? NT * 16 : 0;   // doubles: [I][q][r] for K = NT - 1 (odd NT)
#ifndef IMC_Z3PAD
#define IMC_Z3PAD 0
#endif
    static constexpr int TOK = MAIN + EXTRA + IMC_Z3PAD;   // doubles per table entry: NP * NP (+ padding that staggers the banks of consecutive entries)
    // where element (row, col) of an operator lives inside its TOK doubles
    static __host__ __device__ constexpr int idx(int row, int col)
    {
        return (col >> 2) < NTE ? ((((row >> 2) * 4 + (col & 3)) * 4 + (row & 3)) * NTE + (col >> 2))
                                : (MAIN + ((row >> 2) * 4 + (col & 3)) * 4 + (row & 3));
    }
    // table entries: A tokens + the identity; the same space later holds the Z2SLOTS exchange operators + the identity
    static constexpr int entries(int A) { return (A > Z2SLOTS ? A : Z2SLOTS) + 1; }
    static constexpr size_t op_doubles(int A) { return (size_t)entries(A) * TOK; }
    static constexpr int ints(int A) { return (entries(A) + 1) & ~1; }
    // + the dictionary's merge lists for the table build: left[A], right[A], depth order[A], level offsets[A + 2]
    static constexpr int meta_ints(int A) { return 4 * A + 4; }
    static constexpr size_t lds_bytes(int A) { return op_doubles(A) * 8 + (size_t)(ints(A) + meta_ints(A)) * 4 + 16; }
};

// The NT A operands of tile-row I of operator Cz for this lane: C[4I + r][4K + q], K = 0..NT-1 (lane offsets
// lo = (q*4 + r) * NTE into the main part of a tile-row and lx = q*4 + r into the odd part).
template <int NT>
__device__ __forceinline__ void zip3_load_row(double (&a)[NT], const double *Cz, int I, int lo, int lx)
{
    using Geo = Zip3Geom<NT>;
    const double2 *m = reinterpret_cast<const double2 *>(Cz + I * 16 * Geo::NTE + lo);
#pragma unroll
    for (int k2 = 0; k2 < Geo::NTE / 2; ++k2) {
        const double2 v = m[k2];
        a[2 * k2] = v.x;
        a[2 * k2 + 1] = v.y;
    }
    if constexpr (NT & 1) a[NT - 1] = Cz[Geo::MAIN + I * 16 + lx];
}

// Pout <- C * Pin for the four segments of the wavefront; Cz = this lane's segment's operator (table entry).  `a` holds
// tile-row 0 of Cz on entry and tile-row 0 of Cn (the operator of the NEXT step) on exit: the LDS reads of a tile-row
// are issued one tile-row (NT*NT MFMAs) ahead of their use, so no MFMA waits for LDS.
template <int NT>
__device__ __forceinline__ void zip3_step(const double (&Pin)[NT][NT], double (&Pout)[NT][NT], const double *Cz, const double *Cn,
                                          double (&a)[NT], int lo, int lx)
{
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        double an[NT];
        if (I + 1 < NT) zip3_load_row<NT>(an, Cz, I + 1, lo, lx);
        else zip3_load_row<NT>(an, Cn, 0, lo, lx);
        __builtin_amdgcn_sched_barrier(0);         // keep the prefetch ahead of this tile-row's MFMAs
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J)
                Pout[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[K], Pin[K][J], K == 0 ? 0.0 : Pout[I][J], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int K = 0; K < NT; ++K) a[K] = an[K];
    }
}

// common power-of-two rescale of a segment's operator: exponent of its largest entry over the segment's 16 lanes
// (lanes 16 q + 4 blk + r: the xor partners 1, 2, 16, 32 stay inside the block).  The largest entry's exponent is the
// largest exponent, so the cross-lane part reduces one int per lane instead of a double; a lane whose maximum is not a
// positive finite number (0, inf, NaN) votes "leave the operator alone", which wins (same rule as before).
template <int NT>
__device__ __forceinline__ void zip3_rescale(double (&P)[NT][NT], int &ex)
{
    // The exponent of the lane's largest entry from the HIGH words alone (entries are non-negative - or NaN, whose exponent
    // field is the largest of all - so the sign-stripped high 32 bits order them as the values do): NT^2 32-bit maxima instead
    // of NT^2 fp64 compare / compare-unordered / select triples.  The fp64 VALU shares the DP units with the matrix
    // instruction, so those triples came straight out of the MFMA pipe's time, every 16 steps.
    // (A largest entry below 2^-1022 counts as zero - the lane abstains - where frexp would have normalised it.)
    int hmax = 0;
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J) hmax = max(hmax, __double2hiint(P[K][J]) & 0x7fffffff);
    const int field = hmax >> 20;
    int e = field == 0 ? INT_MIN : field == 0x7ff ? INT_MAX : field - 1022;   // all-zero lane abstains; inf / NaN: leave alone
#pragma unroll
    for (int m = 1; m <= 32; m = (m == 2 ? 16 : m * 2)) e = max(e, __shfl_xor(e, m, 64));
    e = (e == INT_MAX || e == INT_MIN) ? 0 : e;
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J) P[K][J] = ldexp(P[K][J], -e);
    ex += e;
}

// ---- fold the workgroup's n segments into one: P_0 <- P_{n-1} ... P_1 P_0 ------------------------------------------------
// Slot s = 4 * wavefront + block holds segment s's operator P (registers, the layout of an MFMA B operand) and exponent.
// X: exchange area of Z2SLOTS entries of TOK doubles (the operator table's space once it is dead), xe: Z2SLOTS ints.
//   stage A - inside each wavefront, no workgroup barrier: block 1 -> 0 and 3 -> 2 (one step), then 2 -> 0 (one step);
//             a partner hands its operator over through X in the table's layout (an A operand's rows) - LDS executes a
//             wavefront's own instructions in order - and the receiving block multiplies it on: an ordinary step.
//   stage B - the (up to) eight wavefront products meet in X behind ONE workgroup barrier and wavefront 0 alone folds
//             them: pairs (2b, 2b + 1) in its four blocks, then 1 -> 0 and 3 -> 2, then 2 -> 0 - three steps with the
//             MFMA pipe to itself.
// A block without a partner (ragged n) still issues the step - the matrix instruction is wavefront-wide - on whatever X
// holds and keeps its own P.  Round 2's fold was five workgroup-wide levels (two barriers and a step by all eight
// wavefronts each): 16 us at N = 20 (in-kernel timestamps, profiles/r02_z4_phase_times.txt).
// TABLE_LIVE: X overlays an LDS table that other wavefronts may still be reading -> one more barrier in front.
template <int NT, bool TABLE_LIVE, bool RESCALE = true>
__device__ __forceinline__ void zip3_fold(double (&P)[NT][NT], int &ex, double *X, int *xe, int n, int slot, int wv, int lo, int lx)
{
    using Geo = Zip3Geom<NT>;
    constexpr int TOK = Geo::TOK;
    if (n <= 1) return;                                     // workgroup-uniform
    const int lane = threadIdx.x & 63;                      // (wv: this wavefront's index among those that fold together)
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    double Q[NT][NT], al[NT];
    auto put = [&](int area) __attribute__((always_inline)) {
        double *dst = X + (size_t)area * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) dst[Geo::idx(4 * K + q, 4 * J + r)] = P[K][J];
        if (q == 0 && r == 0) xe[area] = ex;
    };
    auto take = [&](int area, bool act) __attribute__((always_inline)) {     // P <- X[area] * P where act
        const double *src = X + (size_t)area * TOK;
        zip3_load_row<NT>(al, src, 0, lo, lx);
        zip3_step<NT>(P, Q, src, src, al, lo, lx);
        const int e = xe[area];
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) P[K][J] = act ? Q[K][J] : P[K][J];
        ex += act ? e : 0;
        if constexpr (RESCALE) zip3_rescale<NT>(P, ex);   // (a caller that folds at most a few normalised operators rescales once, at the end)
    };
    if constexpr (TABLE_LIVE) __syncthreads();
    // ---- stage A ----
    if (wv * 4 + 1 < n) {                                   // (wavefront-uniform: this wavefront holds at least two segments)
        if ((bq & 1) && slot < n) put(slot);
        wave_fence();
        take(slot + 1 < Z2SLOTS ? slot + 1 : slot, !(bq & 1) && slot + 1 < n);
        if (wv * 4 + 2 < n) {
            wave_fence();                                   // (the reads of level 1 are done before area slot 2 is rewritten)
            if (bq == 2 && slot < n) put(slot);
            wave_fence();
            take(slot + 2 < Z2SLOTS ? slot + 2 : slot, bq == 0 && slot + 2 < n);
        }
    }
    if (n <= 4) return;
    // ---- stage B ----
    const int nw = (n + 3) / 4;                             // wavefronts that hold segments
    if (bq == 0 && wv < nw && wv > 0) put(wv * 4);          // (wavefront 0's own product stays in its registers ...)
    if (wv == 0) { wave_fence(); if (bq == 0) put(0); }     // (... but blocks 1..3 of wavefront 0 need it from X as well)
    __syncthreads();
    if (wv != 0) return;
    {   // level 3: block b takes the pair (2b, 2b + 1)
        const int lo_w = 2 * bq, hi_w = 2 * bq + 1;
        const double *src = X + (size_t)(lo_w < nw ? lo_w : 0) * 4 * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) P[K][J] = src[Geo::idx(4 * K + q, 4 * J + r)];
        ex = xe[(lo_w < nw ? lo_w : 0) * 4];
        take((hi_w < nw ? hi_w : 0) * 4, hi_w < nw);
    }
    if (nw > 2) {                                           // level 4: 1 -> 0, 3 -> 2 (blocks hold wavefronts 2b, 2b + 1)
        wave_fence();
        if (bq & 1) put(bq);                                // areas 1 and 3: wavefront 0's own slots, free since stage A
        wave_fence();
        take(bq + 1 < 4 ? bq + 1 : bq, !(bq & 1) && 2 * (bq + 1) < nw);
    }
    if (nw > 4) {                                           // level 5: 2 -> 0
        wave_fence();
        if (bq == 2) put(2);
        wave_fence();
        take(bq + 2 < 4 ? bq + 2 : bq, bq == 0);
    }
}

// ---- fused tail: the chunk's workgroups meet, the last one to arrive finishes the chunk ----------------------------------
// After zip3_fold slot 0 of a workgroup holds the product of its segments (the chunk's first workgroup: the forward
// vector in column 0).  With at most Z2SLOTS workgroups per chunk the stitch is one more zip3_fold: every workgroup
// publishes its operator in an LDS table entry's layout (sc1 stores: written through, the publishing wavefront drains
// them with s_waitcnt vmcnt(0) before it signals), then adds one to the chunk's arrival counter (an agent-scope atomic);
// the workgroup whose add returns n_units - 1 is the last: its 32 slots reload the chunk's operators (sc1 loads: past
// this CU's L1; nobody on this XCD has read those lines before), fold them as the scan's own fold does, and slot 0
// turns column 0 of the product into the chunk's log-likelihood, ln2 * exponent + log(sum), written straight to the
// result slot.  (MI355X_MICROARCH.md, "Valid forms": one lane signals for all of its workgroup's stores, the consumer is
// told by the value its own add returned, payload stores and loads all sc1.)  The counter is set back to zero by the
// last arriver, so the next evaluation of the plan finds it clean.  A chunk of ONE workgroup skips the meeting.
// Every published operator and every exponent sits on cache lines of its own (TOK * 8 is a multiple of 128 bytes, an
// exponent has a 128-byte slot): a last arriver that shares an XCD with an earlier one must not find a line in that
// L2 that was fetched before all of its words were written.
template <int NT>
__device__ __forceinline__ void zip3_tail(const BigArgs &a, double (&P)[NT][NT], int &ex, double *X, int *xe, int b, int bx, int slot, int lo, int lx)
{
    using Geo = Zip3Geom<NT>;
    constexpr int TOK = Geo::TOK;
    __shared__ int s_last;
    const Z2Tail td = a.tail[bx];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 4, r = lane & 3;
    const size_t cb = (size_t)b * a.n_chunks + td.chunk;
    if (td.n_units > 1) {                                    // (workgroup-uniform)
        if (slot == 0) {
            double *dst = a.tailX + (cb * a.tail_stride + td.unit) * TOK;
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    __hip_atomic_store(&dst[Geo::idx(4 * K + q, 4 * J + r)], P[K][J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q == 0 && r == 0) __hip_atomic_store(&a.tailE[(cb * a.tail_stride + td.unit) * 32], ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {                                      // wavefront 0 holds slot 0: it alone stored
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0)
                s_last = __hip_atomic_fetch_add(&a.tail_arrive[cb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)td.n_units - 1;
        }
        __syncthreads();
        if (!s_last) return;
        const bool valid = slot < (int)td.n_units;
        const double *src = a.tailX + (cb * a.tail_stride + (valid ? slot : 0)) * TOK;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                const double v = __hip_atomic_load(&src[Geo::idx(4 * K + q, 4 * J + r)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                P[K][J] = valid ? v : 0.0;
            }
        ex = __hip_atomic_load(&a.tailE[(cb * a.tail_stride + (valid ? slot : 0)) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        zip3_fold<NT, true>(P, ex, X, xe, (int)td.n_units, slot, tid >> 6, lo, lx);
        if (tid == 0) __hip_atomic_store(&a.tail_arrive[cb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (slot == 0) {
        double part = 0.0;
#pragma unroll
        for (int K = 0; K < NT; ++K) part += (r == 0 && 4 * K + q < a.N) ? P[K][0] : 0.0;
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (lane == 0) a.tail_out[cb] = (double)ex * 0.693147180559945309417232121458 + log(part);
    }
}

template <int NT>
__global__ __launch_bounds__(Z2WAVES * 64, Z2WAVES / 4) void k_zpropagate3(BigArgs a)
{
    using Geo = Zip3Geom<NT>;
    constexpr int NP = Geo::NP, TOK = Geo::TOK, THREADS = Z2WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *C = lds;                                                   // [entries][TOK]
    int *cex = reinterpret_cast<int *>(C + Geo::op_doubles(a.A));    // [entries]
    int *m_left = cex + Geo::ints(a.A), *m_right = m_left + a.A, *m_order = m_right + a.A, *m_lvl = m_order + a.A;   // [A], [A], [A], [A + 2]

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const double *pp = a.params + (size_t)b * a.pstride;
    if (a.params_src) {
        // small launch: every workgroup fetches the parameter set straight from the caller's mapped staging slot (all of
        // a lane's loads in flight together: host memory, ~2 us per round trip) into LDS behind the table's lists
        double *lp = reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + ((Geo::lds_bytes(a.A) + 15) & ~(size_t)15));
        const double *src = a.params_src + (size_t)b * a.pstride;
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
                if (k < (int)a.pstride) *reinterpret_cast<double2 *>(lp + k) = v[u];
            }
        }
        __syncthreads();
        pp = lp;
    }
    const double *pi_p = pp;
    const double *Tp = pp + a.PP;
    const double *Etg = pp + a.PP + (size_t)a.PP * a.PP;
    const int IDENT = a.A;                                             // table entry of the identity operator

    // ---- operator table: raw symbols, the identity, then merged tokens in dictionary order ----
    for (int idx = tid; idx < (a.S + 1) * NP * NP; idx += THREADS) {
        const int sidx = idx / (NP * NP);
        const int rem = idx - sidx * NP * NP;
        const int i = rem / NP, j = rem - i * NP;
        if (sidx < a.S) C[(size_t)sidx * TOK + Geo::idx(i, j)] = Etg[(size_t)sidx * a.PP + i] * Tp[(size_t)j * a.PP + i];
        else C[(size_t)IDENT * TOK + Geo::idx(i, j)] = i == j ? 1.0 : 0.0;
    }
    if (tid < a.S) cex[tid] = 0;
    if (tid == 0) cex[IDENT] = 0;
    // the merge lists go to LDS in one sweep: the level loop below then has no dependent global loads
    for (int z = a.S + tid; z < a.A; z += THREADS) { m_left[z] = a.tok_left[z]; m_right[z] = a.tok_right[z]; }
    for (int k = tid; k < a.A - a.S && a.tab_nlvl > 0; k += THREADS) m_order[k] = a.tab_order[k];
    for (int k = tid; k <= a.tab_nlvl && a.tab_nlvl > 0; k += THREADS) m_lvl[k] = a.tab_lvl[k];   // (raw stream: no lists at all)
    __syncthreads();
    // Merged tokens C_z = C_right * C_left, one dictionary depth at a time: the tokens of a depth are independent, each
    // goes to one MFMA block (wavefront w, block b takes the (4w + b)-th token of the depth), and the product is the
    // scan's own step with C_left in the role of P - NT^3 MFMAs per wavefront for four tokens, against ~4800 cycles
    // per token for an element-per-thread product (measured: 2.2 us per token, a fifth of the whole kernel at 44 tokens).
    {
        const int lane0 = tid & 63;
        const int q0 = lane0 >> 4, b0 = (lane0 >> 2) & 3, r0 = lane0 & 3, wv = tid >> 6;
        const int lo0 = (q0 * 4 + r0) * Geo::NTE, lx0 = q0 * 4 + r0;
        for (int d = 0; d < a.tab_nlvl; ++d) {
            const int o0 = m_lvl[d], o1 = m_lvl[d + 1];
            for (int base = o0; base < o1; base += Z2SLOTS) {
                if (base + wv * 4 >= o1) continue;               // nothing for this wavefront (wavefront-uniform)
                const int ti = base + wv * 4 + b0;
                const bool have = ti < o1;
                const int z = have ? m_order[ti] : IDENT;
                const int zl = have ? m_left[z] : IDENT, zr = have ? m_right[z] : IDENT;
                const double *Cl = C + (size_t)zl * TOK, *Cr = C + (size_t)zr * TOK;
                double Bt[NT][NT], Out[NT][NT], arow0[NT];
#pragma unroll
                for (int K = 0; K < NT; ++K)
#pragma unroll
                    for (int J = 0; J < NT; ++J) Bt[K][J] = Cl[Geo::idx(4 * K + q0, 4 * J + r0)];
                zip3_load_row<NT>(arow0, Cr, 0, lo0, lx0);
                zip3_step<NT>(Bt, Out, Cr, Cr, arow0, lo0, lx0);
                int e2 = 0;
                zip3_rescale<NT>(Out, e2);
                if (have) {
                    double *Cz = C + (size_t)z * TOK;
#pragma unroll
                    for (int K = 0; K < NT; ++K)
#pragma unroll
                        for (int J = 0; J < NT; ++J) Cz[Geo::idx(4 * K + q0, 4 * J + r0)] = Out[K][J];
                    if (q0 == 0 && r0 == 0) cex[z] = cex[zl] + cex[zr] + e2;
                }
            }
            __syncthreads();
        }
    }

    // ---- scan: one segment per MFMA block ----
    const int lane = tid & 63;
    const int q = lane >> 4, bq = (lane >> 2) & 3, r = lane & 3;
    const int lo = (q * 4 + r) * Geo::NTE, lx = q * 4 + r;
    const Z2Block blk = a.blocks[blockIdx.x];
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
    // (token loads as GLOBAL loads: through the descriptor's generic pointer they are flat loads, which also count
    // against the LDS counter and turn every LDS wait behind them into a full one)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4 *gptr4;
    const __attribute__((address_space(1))) uint8_t *tokg = (const __attribute__((address_space(1))) uint8_t *)tokp;

    double P[NT][NT], Q[NT][NT];
    {
        const int tok0 = (valid && first) ? (int)tokp[0] : 0;
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
    // a step pair; `arow` carries the prefetched first tile-row of t0's operator in and of tn's operator out
    double arow[NT];
    auto two_steps = [&](int t0, int t1, int tn) __attribute__((always_inline)) {
        zip3_step<NT>(P, Q, C + (size_t)t0 * TOK, C + (size_t)t1 * TOK, arow, lo, lx);
        zip3_step<NT>(Q, P, C + (size_t)t1 * TOK, C + (size_t)tn * TOK, arow, lo, lx);
        ex += cex[t0] + cex[t1];
    };
    // A block of 16 positions in which not every lane's segment has a token (the first block: token 0 of a first
    // segment went into the initial P, short segments; the blocks after the last full one: ragged ends, idle slots):
    // one 16-byte load per lane - only where the segment still has tokens, so nothing is read past a stream's padded
    // end - and the identity entry wherever there is no token; `npos` positions are stepped (even, <= 16).
    auto masked_block = [&](int bi, int npos) __attribute__((always_inline)) {
        uint4 ob = make_uint4(0u, 0u, 0u, 0u);
        if (bi * RESCALE_EVERY < len) {
            const u32x4 v = *(gptr4)(tokg + (size_t)bi * RESCALE_EVERY);
            ob = make_uint4(v.x, v.y, v.z, v.w);
        }
        const int live = len - bi * RESCALE_EVERY;              // positions u < live hold a token
        const int dead0 = (first && bi == 0) ? 0 : -1;
        const unsigned long long lo64 = (unsigned long long)ob.y << 32 | ob.x, hi64 = (unsigned long long)ob.w << 32 | ob.z;
        auto tok_of = [&](int u) __attribute__((always_inline)) {
            const unsigned long long h = u < 8 ? lo64 : hi64;
            const int tk = (int)((h >> (8 * (u & 7))) & 0xffull);
            return (u < live && u != dead0 && u < RESCALE_EVERY) ? tk : IDENT;
        };
        zip3_load_row<NT>(arow, C + (size_t)tok_of(0) * TOK, 0, lo, lx);
#pragma unroll 1
        for (int u = 0; u < npos; u += 2) two_steps(tok_of(u), tok_of(u + 1), tok_of(u + 2));
        zip3_rescale<NT>(P, ex);
    };
    if (maxlen > 0) masked_block(0, min(RESCALE_EVERY, (maxlen + 1) & ~1));
    // (the tokens of block bi + 1 are requested while block bi runs; behind the last full block the load re-reads it)
    u32x4 obn = nfull > 1 ? *(gptr4)(tokg + (size_t)RESCALE_EVERY) : u32x4{0u, 0u, 0u, 0u};
    for (int bi = 1; bi < nfull; ++bi) {
        const u32x4 ob = obn;
        obn = *(gptr4)(tokg + (size_t)min(bi + 1, nfull - 1) * RESCALE_EVERY);
        uint32_t w0 = ob.x, w1 = ob.y, w2 = ob.z, w3 = ob.w;
        zip3_load_row<NT>(arow, C + (size_t)(w0 & 0xffu) * TOK, 0, lo, lx);
#pragma unroll 1
        for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t w = w0;
            w0 = w1; w1 = w2; w2 = w3; w3 = (uint32_t)IDENT;   // (after the block's last word the prefetch reads the identity)
            two_steps(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu);
            two_steps((w >> 16) & 0xffu, w >> 24, w0 & 0xffu);
        }
        zip3_rescale<NT>(P, ex);
    }
    for (int bi = max(1, nfull); bi * RESCALE_EVERY < maxlen; ++bi)
        masked_block(bi, min(RESCALE_EVERY, (maxlen - bi * RESCALE_EVERY + 1) & ~1));
    zip3_rescale<NT>(P, ex);

    // ---- packed block (Z2Block::first == 2): every slot is a whole one-segment chunk - its vector goes out as it is ----
    if (blk.first == 2) {                                   // (workgroup-uniform: nobody reaches the fold's barriers)
        if (slot < (int)blk.n && r == 0) {
            const size_t gvp = (size_t)b * a.n_vecs_total + blk.out_vec0 + slot;
#pragma unroll
            for (int K = 0; K < NT; ++K)
                if (4 * K + q < a.N) a.P[gvp * NP + 4 * K + q] = P[K][0];
            if (q == 0) a.EX[gvp] = ex;
        }
        return;
    }

    // ---- fold the workgroup's segments into one (zip3_fold): the operator table is dead once every wavefront is
    // here; its space becomes the exchange area ----
    zip3_fold<NT, true>(P, ex, C, cex, (int)blk.n, slot, tid >> 6, lo, lx);

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
    if (a.tail) zip3_tail<NT>(a, P, ex, C, cex, b, (int)blockIdx.x, slot, lo, lx);
}
