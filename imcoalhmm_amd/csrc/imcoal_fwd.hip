// imcoal_fwd.hip - MI355X (gfx950 / CDNA4) HMM forward log-likelihood engine behind the C ABI of
// include/imcoal_fwd.h.  Replaces ziphmm.preprocess_raw_observations / ziphmm.zip_forward as called
// from the reference at src/IMCoalHMM/hmm.py:16,20-21 and the sum at likelihood.py:33.
//
// Algorithm (see DESIGN.md for the derivation and the roofline accounting)
// ---------------------------------------------------------------------
// The forward recursion  a_t = (T' a_{t-1}) .* E[:,o_t]  is a serial chain per alignment file, so a
// file is cut into K segments ("parallel in time").  Segment 0 propagates the single vector
// pi .* E[:,o_0]; every later segment propagates the N unit vectors e_c, which yields the segment's
// exact N x N transfer operator column by column.  A short second kernel stitches the operators in
// order.  All rescaling is by exact powers of two (integer exponents are summed), so the only
// difference from the textbook recursion is fp64 rounding order.
//
// propagate kernel (the hot kernel): lanes are grouped G lanes per vector, each lane owns R
// consecutive states of that vector and keeps the R x N block of T' it needs in VGPRs for the whole
// launch (T is read from HBM exactly once per lane).  Per column a lane publishes its R values to
// LDS, reads the vector's N values back as broadcast ds_read_b128, and runs R independent fp64 FMA
// chains.  64/G vectors ride in one wavefront; no workgroup barrier is needed because a vector never
// leaves its wavefront.  MFMA is deliberately not used (north_star): the matrices are tiny and the
// dependent-FMA chain, not matrix throughput, is the limit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../include/imcoal_fwd.h"

// =====================================================================================================
// Device side
// =====================================================================================================

struct SegDesc {          // one segment of one chunk
    const uint8_t *obs;   // first column of the segment (16-byte aligned, padded past the end)
    uint32_t len;         // columns in this segment
    uint32_t first;       // 1 = first segment of its chunk (single vector, starts from pi)
};

struct VecDesc {          // one propagated vector
    uint32_t seg;         // segment id
    uint32_t c;           // basis index (0 for a first segment)
};

struct PropArgs {
    const SegDesc *segs;
    const VecDesc *vecs;
    uint32_t n_vecs;
    int N;                 // true number of states
    int S;                 // alphabet size
    const double *params;  // per parameter set: pi[NP] | Tp[NP*NP] (Tp[j*NP+i]=T[j][i]) | Et[S*NP]
    size_t pstride;        // doubles per parameter set
    double *P;             // [B][n_vecs][NP]  normalised end vectors
    int *EX;               // [B][n_vecs]      power-of-two exponents
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
    const bool first = sd.first != 0;
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
        const int sym = act ? (int)obs[t] : 0;
        column_step<R, NP, true, true>(xo, Tb, xw, own, Et, sym, act, first && t == 0, s);
        rescale<R>(xo, s, ex);
    }
    // ---- body: full 16-column blocks common to every vector of this wavefront ----
    for (int blk = 1; blk < nfull; ++blk) {
        const uint4 ob = *reinterpret_cast<const uint4 *>(obs + (size_t)blk * RESCALE_EVERY);
        uint32_t w0 = ob.x, w1 = ob.y, w2 = ob.z, w3 = ob.w;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const uint32_t w = w0;
            w0 = w1; w1 = w2; w2 = w3;
            column_step<R, NP, false, false>(xo, Tb, xw, own, Et, w & 0xffu, true, false, s);
            column_step<R, NP, false, false>(xo, Tb, xw, own, Et, (w >> 8) & 0xffu, true, false, s);
            column_step<R, NP, false, false>(xo, Tb, xw, own, Et, (w >> 16) & 0xffu, true, false, s);
            column_step<R, NP, false, true>(xo, Tb, xw, own, Et, w >> 24, true, false, s);
        }
        rescale<R>(xo, s, ex);
    }
    // ---- tail: ragged remainder, column by column ----
    for (int t = max(RESCALE_EVERY, nfull * RESCALE_EVERY); t < maxlen; ++t) {
        const bool act = t < len;
        const int sym = act ? (int)obs[t] : 0;
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
        double *Pout = a.P + ((size_t)b * a.n_vecs + vid) * NP + own;
#pragma unroll
        for (int k = 0; k < R; ++k) Pout[k] = xo[k];
        if (r == 0) a.EX[(size_t)b * a.n_vecs + vid] = ex;
    }
}

// EMAX[b][seg] = max_c EX[b][vec0(seg)+c]  (segments after the first of a chunk only)
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

// Stitch the segment operators of one chunk in order:  a <- P_k * (a .* 2^(ex_k - emax_k)).
// One workgroup per (chunk, parameter set); thread i owns state i.
__global__ void k_stitch(const uint32_t *chunk_seg, const uint32_t *seg_vec0, uint32_t n_segs, uint32_t n_vecs,
                         int N, int NP, const double *P, const int *EX, const int *EMAX, double *out, int n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) double w[];   // [2][NPW] double buffered weights
    const int NPW = (N + 1) & ~1;
    const int f = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
    const uint32_t s0 = chunk_seg[f], s1 = chunk_seg[f + 1];
    if (s0 == s1) {   // empty chunk
        if (i == 0) out[(size_t)b * n_chunks + f] = 0.0;
        return;
    }
    const double *Pb = P + (size_t)b * n_vecs * NP;
    const int *EXb = EX + (size_t)b * n_vecs;
    const int *EMb = EMAX + (size_t)b * n_segs;
    double a = (i < N) ? Pb[(size_t)seg_vec0[s0] * NP + i] : 0.0;
    long long etot = EXb[seg_vec0[s0]];
    int buf = 0;
    for (uint32_t k = s0 + 1; k < s1; ++k) {
        const uint32_t v0 = seg_vec0[k];
        const int em = EMb[k];
        double *wb = w + buf * NPW;
        if (i < N) wb[i] = ldexp(a, EXb[v0 + i] - em);
        __syncthreads();
        double acc = 0.0, s = 0.0;
        if (i < N) {
            const double *Pk = Pb + (size_t)v0 * NP + i;
            for (int c = 0; c < N; ++c) {
                const double wc = wb[c];
                acc = fma(Pk[(size_t)c * NP], wc, acc);
                s += wc;
            }
        } else {
            for (int c = 0; c < N; ++c) s += wb[c];
        }
        int e = 0;
        (void)frexp(s, &e);
        e = (s > 0.0 && s < INFINITY) ? e : 0;
        a = ldexp(acc, -e);
        etot += (long long)em + e;
        buf ^= 1;   // next iteration writes the other buffer: one barrier per step suffices
    }
    // total = sum_i a_i, by thread 0 through LDS
    __syncthreads();
    if (i < N) w[i] = a;
    __syncthreads();
    if (i == 0) {
        double tot = 0.0;
        for (int c = 0; c < N; ++c) tot += w[c];
        out[(size_t)b * n_chunks + f] = (double)etot * 0.693147180559945309417232121458 + log(tot);
    }
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

// =====================================================================================================
// Host side
// =====================================================================================================

namespace {

thread_local std::string g_err;
std::mutex g_mu;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(_e == hipErrorOutOfMemory ? IMC_ERR_OOM : IMC_ERR_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                             \
    } while (0)

struct Ctx {
    pid_t pid = 0;
    int device = -1;          // -1: use the thread's current device at first use
    bool ready = false;
    hipStream_t stream = nullptr;
    int cus = 256;
    uint64_t next_obs_id = 1;
    size_t seg_override = 0;
    bool profile = false;
    struct Ev3 { hipEvent_t a, b, c; };   // a: before propagate, b: after propagate, c: after stitch
    std::vector<Ev3> events;
    uint64_t last_segments = 0, last_vectors = 0, last_seglen = 0, last_vcols = 0;
} g;

int ensure_ctx()
{
    const pid_t me = getpid();
    if (g.ready && g.pid == me) return IMC_OK;
    if (g.ready && g.pid != me) {
        // forked child: the parent's HIP state is not usable here; start over (handles leak, by design)
        g.ready = false;
        g.stream = nullptr;
        g.events.clear();
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(IMC_ERR_NODEVICE, "no HIP device available (libimcoal_fwd has no CPU fallback)");
    if (g.device < 0) {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) cur = 0;
        g.device = cur;
    }
    if (g.device >= n) return fail(IMC_ERR_NODEVICE, "requested device index out of range");
    HIP_TRY(hipSetDevice(g.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g.device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(IMC_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    g.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    g.pid = me;
    g.ready = true;
    return IMC_OK;
}

}  // namespace

struct imc_obs {
    uint64_t id;
    pid_t pid;
    int device;
    int nsym;
    size_t L;
    uint8_t *d_sym;   // L bytes + zero padding
};

namespace {

constexpr size_t OBS_PAD = 256;

int obs_upload(const uint8_t *host, size_t L, int nsym, imc_obs **out)
{
    if (int rc = ensure_ctx()) return rc;
    auto *o = new (std::nothrow) imc_obs();
    if (!o) return fail(IMC_ERR_OOM, "host allocation failed");
    o->id = g.next_obs_id++;
    o->pid = g.pid;
    o->device = g.device;
    o->nsym = nsym;
    o->L = L;
    o->d_sym = nullptr;
    const size_t bytes = ((L + OBS_PAD - 1) / OBS_PAD) * OBS_PAD + OBS_PAD;
    hipError_t e = hipMalloc((void **)&o->d_sym, bytes);
    if (e != hipSuccess) {
        delete o;
        return fail(IMC_ERR_OOM, std::string("hipMalloc(observations): ") + hipGetErrorString(e));
    }
    e = hipMemsetAsync(o->d_sym, 0, bytes, g.stream);
    if (e == hipSuccess && L) e = hipMemcpyAsync(o->d_sym, host, L, hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) {
        (void)hipFree(o->d_sym);
        delete o;
        return fail(IMC_ERR_HIP, std::string("observation upload: ") + hipGetErrorString(e));
    }
    *out = o;
    return IMC_OK;
}

// ---- launch plan: segmentation of a list of chunks for given (N, B) -------------------------------

struct KernelChoice {
    int R, G, NP, VPW, minw;
    void (*fn)(PropArgs);
};

#define KC(R_, G_, MW_) KernelChoice{R_, G_, (R_) * (G_), 64 / (G_), MW_, k_propagate<R_, G_, MW_>}
const KernelChoice kChoices[] = {
    KC(4, 1, 2), KC(4, 2, 2), KC(4, 3, 2), KC(4, 4, 2), KC(4, 5, 2), KC(3, 8, 2), KC(2, 14, 2), KC(2, 16, 2),
    KC(2, 20, 1), KC(1, 48, 1), KC(1, 56, 1), KC(1, 64, 1),
};
#undef KC
constexpr int IMC_MAX_N = 64;

const KernelChoice *choose_kernel(int N)
{
    for (const auto &k : kChoices)
        if (k.NP >= N) return &k;
    return nullptr;
}

struct Plan {
    std::vector<uint64_t> key;   // obs ids..., N, S, B, seg_override
    const KernelChoice *kc = nullptr;
    int N = 0, S = 0, B = 0, n_chunks = 0;
    uint32_t n_segs = 0, n_vecs = 0;
    size_t seglen = 0;
    uint64_t vcols = 0;          // executed vector-columns per parameter set
    size_t pstride = 0;
    SegDesc *d_segs = nullptr;
    VecDesc *d_vecs = nullptr;
    uint32_t *d_seg_vec0 = nullptr, *d_chunk_seg = nullptr;
    uint8_t *d_seg_first = nullptr;
    double *d_params = nullptr, *d_P = nullptr, *d_out = nullptr, *d_partial = nullptr;
    int *d_EX = nullptr, *d_EMAX = nullptr;
    double *h_params = nullptr, *h_out = nullptr;   // pinned
    void release()
    {
        (void)hipFree(d_segs); (void)hipFree(d_vecs); (void)hipFree(d_seg_vec0); (void)hipFree(d_chunk_seg);
        (void)hipFree(d_seg_first); (void)hipFree(d_params); (void)hipFree(d_P); (void)hipFree(d_out);
        (void)hipFree(d_partial); (void)hipFree(d_EX); (void)hipFree(d_EMAX);
        (void)hipHostFree(h_params); (void)hipHostFree(h_out);
    }
};

std::list<std::unique_ptr<Plan>> g_plans;   // most recent first
constexpr size_t MAX_PLANS = 4;

void drop_plans()
{
    for (auto &p : g_plans) p->release();
    g_plans.clear();
}

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// Pick the segment length that minimises a simple machine model: equal-length wavefront tasks are
// executed in rounds of `resident` wavefronts; the serial stitch adds ~stitch_cost per segment of the
// longest chunk.
size_t choose_seglen(const std::vector<size_t> &lens, int N, int B, const KernelChoice *kc, int cus)
{
    size_t maxlen = 0;
    for (size_t L : lens) maxlen = std::max(maxlen, L);
    if (maxlen <= 1024) return std::max<size_t>(maxlen, 16);
    const double resident = (double)cus * 4.0 * kc->minw;
    const double col_cost = 4.0 * kc->R * kc->NP + 120.0;   // cycles per column per wavefront (issue + LDS)
    const double stitch_cost = 4.0 * N + 400.0;              // cycles per stitched segment
    double best = 1e300;
    size_t best_seg = maxlen;
    for (double s = 1024.0; ; s *= 1.189207115) {
        size_t seg = std::min(round_up((size_t)s, 16), round_up(maxlen, 16));
        double vecs = 0.0, kmax = 0.0;
        for (size_t L : lens) {
            if (!L) continue;
            const double K = std::ceil((double)L / (double)seg);
            vecs += 1.0 + (K - 1.0) * N;
            kmax = std::max(kmax, K);
        }
        const double waves = std::ceil(vecs * B / kc->VPW);
        const double rounds = std::ceil(waves / resident);
        const double cost = rounds * (double)seg * col_cost + kmax * stitch_cost;
        if (cost < best) { best = cost; best_seg = seg; }
        if (seg >= maxlen) break;
    }
    return best_seg;
}

int build_plan(const imc_obs *const *chunks, int n_chunks, int N, int S, int B, Plan **out)
{
    std::vector<uint64_t> key;
    key.reserve(n_chunks + 4);
    for (int f = 0; f < n_chunks; ++f) key.push_back(chunks[f]->id);
    key.push_back((uint64_t)N); key.push_back((uint64_t)S); key.push_back((uint64_t)B);
    key.push_back((uint64_t)g.seg_override);
    for (auto it = g_plans.begin(); it != g_plans.end(); ++it) {
        if ((*it)->key == key) {
            g_plans.splice(g_plans.begin(), g_plans, it);
            *out = g_plans.front().get();
            return IMC_OK;
        }
    }
    const KernelChoice *kc = choose_kernel(N);
    if (!kc) return fail(IMC_ERR_ARG, "N exceeds the largest built kernel (" + std::to_string(IMC_MAX_N) + ")");
    auto p = std::make_unique<Plan>();
    p->key = key; p->kc = kc; p->N = N; p->S = S; p->B = B; p->n_chunks = n_chunks;

    std::vector<size_t> lens(n_chunks);
    for (int f = 0; f < n_chunks; ++f) lens[f] = chunks[f]->L;
    size_t seg = g.seg_override ? round_up(std::max<size_t>(g.seg_override, 16), 16)
                                : choose_seglen(lens, N, B, kc, g.cus);
    p->seglen = seg;

    std::vector<SegDesc> segs;
    std::vector<VecDesc> vecs;
    std::vector<uint32_t> seg_vec0, chunk_seg(n_chunks + 1, 0);
    std::vector<uint8_t> seg_first;
    uint64_t vcols = 0;
    for (int f = 0; f < n_chunks; ++f) {
        chunk_seg[f] = (uint32_t)segs.size();
        const size_t L = lens[f];
        if (L) {
            const size_t K0 = (L + seg - 1) / seg;
            const size_t sl = round_up((L + K0 - 1) / K0, 16);   // equalised, multiple of 16
            for (size_t off = 0, k = 0; off < L; off += sl, ++k) {
                const size_t n = std::min(sl, L - off);
                SegDesc d{chunks[f]->d_sym + off, (uint32_t)n, k == 0 ? 1u : 0u};
                const uint32_t sid = (uint32_t)segs.size();
                segs.push_back(d);
                seg_first.push_back(k == 0);
                seg_vec0.push_back((uint32_t)vecs.size());
                const int nv = (k == 0) ? 1 : N;
                for (int c = 0; c < nv; ++c) vecs.push_back(VecDesc{sid, (uint32_t)c});
                vcols += (uint64_t)nv * n;
            }
        }
    }
    chunk_seg[n_chunks] = (uint32_t)segs.size();
    if (vecs.size() >= (size_t)UINT32_MAX / 2) return fail(IMC_ERR_ARG, "too many vectors in one call");
    p->n_segs = (uint32_t)segs.size();
    p->n_vecs = (uint32_t)vecs.size();
    p->vcols = vcols;
    p->pstride = round_up((size_t)kc->NP + (size_t)kc->NP * kc->NP + (size_t)S * kc->NP, 2);

    // keep at most MAX_PLANS plans alive
    while (g_plans.size() >= MAX_PLANS) { g_plans.back()->release(); g_plans.pop_back(); }

    Plan *q = p.get();
    auto up = [&](void **d, const void *h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(d, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = hipSuccess;
    const size_t nv = std::max<size_t>(q->n_vecs, 1), ns = std::max<size_t>(q->n_segs, 1);
    if (e == hipSuccess) e = up((void **)&q->d_segs, segs.data(), segs.size() * sizeof(SegDesc));
    if (e == hipSuccess) e = up((void **)&q->d_vecs, vecs.data(), vecs.size() * sizeof(VecDesc));
    if (e == hipSuccess) e = up((void **)&q->d_seg_vec0, seg_vec0.data(), seg_vec0.size() * 4);
    if (e == hipSuccess) e = up((void **)&q->d_seg_first, seg_first.data(), seg_first.size());
    if (e == hipSuccess) e = up((void **)&q->d_chunk_seg, chunk_seg.data(), chunk_seg.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&q->d_params, (size_t)B * q->pstride * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&q->d_P, (size_t)B * nv * kc->NP * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&q->d_EX, (size_t)B * nv * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&q->d_EMAX, (size_t)B * ns * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&q->d_out, (size_t)B * std::max(n_chunks, 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&q->d_partial, (size_t)B * 8);
    if (e == hipSuccess) e = hipHostMalloc((void **)&q->h_params, (size_t)B * q->pstride * 8, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&q->h_out, (size_t)B * std::max(n_chunks, 1) * 8, hipHostMallocDefault);
    if (e != hipSuccess) {
        q->release();
        return fail(e == hipErrorOutOfMemory ? IMC_ERR_OOM : IMC_ERR_HIP,
                    std::string("plan allocation: ") + hipGetErrorString(e));
    }
    g_plans.push_front(std::move(p));
    *out = q;
    return IMC_OK;
}

int check_args(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis,
               const double *Ts, const double *Es)
{
    if (n_chunks < 0 || (n_chunks > 0 && !chunks)) return fail(IMC_ERR_ARG, "chunks is null");
    if (B < 1) return fail(IMC_ERR_ARG, "B must be >= 1");
    if (N < 1) return fail(IMC_ERR_ARG, "N must be >= 1");
    if (S < 1 || S > 256) return fail(IMC_ERR_ARG, "S must be in [1,256]");
    if (!pis || !Ts || !Es) return fail(IMC_ERR_ARG, "null parameter pointer");
    for (int f = 0; f < n_chunks; ++f) {
        if (!chunks[f]) return fail(IMC_ERR_ARG, "null chunk handle");
        if (chunks[f]->pid != getpid()) return fail(IMC_ERR_ARG, "chunk handle was created in another process");
        if (chunks[f]->device != g.device && g.ready) return fail(IMC_ERR_ARG, "chunk lives on another device");
        if (chunks[f]->nsym > S) return fail(IMC_ERR_SYMBOL, "chunk alphabet larger than S");
    }
    return IMC_OK;
}

// Enqueue everything for one batch on `stream`.  Results land in plan->d_out ([B][n_chunks]).
int enqueue(Plan *p, const double *pis, const double *Ts, const double *Es, hipStream_t stream)
{
    const KernelChoice *kc = p->kc;
    const int N = p->N, S = p->S, NP = kc->NP, B = p->B;
    // pad parameters into the pinned staging buffer
    for (int b = 0; b < B; ++b) {
        double *pp = p->h_params + (size_t)b * p->pstride;
        std::memset(pp, 0, p->pstride * 8);
        const double *pi = pis + (size_t)b * N, *T = Ts + (size_t)b * N * N, *E = Es + (size_t)b * N * S;
        for (int i = 0; i < N; ++i) pp[i] = pi[i];
        double *Tp = pp + NP;
        for (int j = 0; j < N; ++j) std::memcpy(Tp + (size_t)j * NP, T + (size_t)j * N, (size_t)N * 8);
        double *Et = pp + NP + (size_t)NP * NP;
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < N; ++i) Et[(size_t)s * NP + i] = E[(size_t)i * S + s];
    }
    HIP_TRY(hipMemcpyAsync(p->d_params, p->h_params, (size_t)B * p->pstride * 8, hipMemcpyHostToDevice, stream));

    g.last_segments = p->n_segs; g.last_vectors = p->n_vecs; g.last_seglen = p->seglen;
    g.last_vcols = p->vcols * (uint64_t)B;
    if (p->n_vecs) {
        PropArgs a;
        a.segs = p->d_segs; a.vecs = p->d_vecs; a.n_vecs = p->n_vecs; a.N = N; a.S = S;
        a.params = p->d_params; a.pstride = p->pstride; a.P = p->d_P; a.EX = p->d_EX;
        const uint32_t vpb = (uint32_t)(WPB * kc->VPW);
        dim3 grid((p->n_vecs + vpb - 1) / vpb, (unsigned)B);
        const size_t lds = ((size_t)WPB * kc->VPW * NP + (size_t)S * NP) * 8;
        Ctx::Ev3 ev{nullptr, nullptr, nullptr};
        if (g.profile) {
            HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); HIP_TRY(hipEventCreate(&ev.c));
            HIP_TRY(hipEventRecord(ev.a, stream));
        }
        hipLaunchKernelGGL(kc->fn, grid, dim3(WPB * 64), lds, stream, a);
        HIP_TRY(hipGetLastError());
        if (g.profile) HIP_TRY(hipEventRecord(ev.b, stream));
        hipLaunchKernelGGL(k_emax, dim3((p->n_segs + 255) / 256, (unsigned)B), dim3(256), 0, stream,
                           p->d_seg_vec0, p->d_seg_first, p->n_segs, p->n_vecs, N, p->d_EX, p->d_EMAX);
        HIP_TRY(hipGetLastError());
        if (g.profile) g.events.push_back(ev);
    }
    if (p->n_chunks) {
        const int threads = (int)round_up((size_t)N, 64);
        const size_t lds = 2 * (size_t)((N + 1) & ~1) * 8;
        hipLaunchKernelGGL(k_stitch, dim3((unsigned)p->n_chunks, (unsigned)B), dim3(threads), lds, stream,
                           p->d_chunk_seg, p->d_seg_vec0, p->n_segs, p->n_vecs, N, NP, p->d_P, p->d_EX,
                           p->d_EMAX, p->d_out, p->n_chunks);
        HIP_TRY(hipGetLastError());
    }
    if (g.profile && p->n_vecs) HIP_TRY(hipEventRecord(g.events.back().c, stream));
    return IMC_OK;
}

int run_batch(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis, const double *Ts,
              const double *Es, double *out_sum, double *out_per_chunk)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = ensure_ctx()) return rc;
    if (int rc = check_args(chunks, n_chunks, B, N, S, pis, Ts, Es)) return rc;
    HIP_TRY(hipSetDevice(g.device));
    Plan *p = nullptr;
    if (int rc = build_plan(chunks, n_chunks, N, S, B, &p)) return rc;
    if (int rc = enqueue(p, pis, Ts, Es, g.stream)) return rc;
    if (n_chunks)
        HIP_TRY(hipMemcpyAsync(p->h_out, p->d_out, (size_t)B * n_chunks * 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    for (int b = 0; b < B; ++b) {
        double tot = 0.0;   // Python sum(): left to right from 0 (likelihood.py:33)
        for (int f = 0; f < n_chunks; ++f) {
            const double v = p->h_out[(size_t)b * n_chunks + f];
            if (out_per_chunk) out_per_chunk[(size_t)b * n_chunks + f] = v;
            tot += v;
        }
        if (out_sum) out_sum[b] = tot;
    }
    return IMC_OK;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================

extern "C" {

const char *imc_version(void) { return "imcoal_fwd 0.1 (gfx950)"; }
const char *imc_last_error(void) { return g_err.c_str(); }

int imc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int imc_set_device(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (device < 0) return fail(IMC_ERR_ARG, "negative device index");
    if (g.ready && g.pid == getpid() && g.device != device) {
        drop_plans();
        (void)hipStreamDestroy(g.stream);
        g.ready = false;
    }
    g.device = device;
    return ensure_ctx();
}

int imc_obs_create(const uint8_t *sym, size_t L, int nsym, imc_obs **out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out) return fail(IMC_ERR_ARG, "out is null");
    if (nsym < 1 || nsym > 256) return fail(IMC_ERR_ARG, "nsym must be in [1,256]");
    if (L && !sym) return fail(IMC_ERR_ARG, "sym is null");
    if (L >= (size_t)1 << 40) return fail(IMC_ERR_ARG, "chunk too long");
    for (size_t t = 0; t < L; ++t)
        if (sym[t] >= nsym) return fail(IMC_ERR_SYMBOL, "symbol " + std::to_string(sym[t]) + " at column " + std::to_string(t) + " >= nsym");
    return obs_upload(sym, L, nsym, out);
}

int imc_obs_create_i32(const int32_t *sym, size_t L, int nsym, imc_obs **out)
{
    if (!out) return fail(IMC_ERR_ARG, "out is null");
    if (nsym < 1 || nsym > 256) return fail(IMC_ERR_ARG, "nsym must be in [1,256]");
    if (L && !sym) return fail(IMC_ERR_ARG, "sym is null");
    std::vector<uint8_t> tmp(L);
    for (size_t t = 0; t < L; ++t) {
        if (sym[t] < 0 || sym[t] >= nsym)
            return fail(IMC_ERR_SYMBOL, "symbol " + std::to_string(sym[t]) + " at column " + std::to_string(t) + " outside [0,nsym)");
        tmp[t] = (uint8_t)sym[t];
    }
    std::lock_guard<std::mutex> lk(g_mu);
    return obs_upload(tmp.data(), L, nsym, out);
}

int imc_obs_create_from_text(const char *path, int nsym, imc_obs **out)
{
    if (!out || !path) return fail(IMC_ERR_ARG, "null argument");
    if (nsym < 1 || nsym > 256) return fail(IMC_ERR_ARG, "nsym must be in [1,256]");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(IMC_ERR_IO, std::string("cannot open ") + path + ": " + std::strerror(errno));
    std::vector<uint8_t> sym;
    std::vector<char> buf(1 << 22);
    long cur = -1;   // token being accumulated, -1 = none
    size_t n;
    int rc = IMC_OK;
    while (rc == IMC_OK && (n = std::fread(buf.data(), 1, buf.size(), fp)) > 0) {
        for (size_t i = 0; i < n; ++i) {
            const unsigned char ch = (unsigned char)buf[i];
            if (ch >= '0' && ch <= '9') {
                cur = (cur < 0 ? 0 : cur) * 10 + (ch - '0');
                if (cur > 1000000) cur = 1000000;
            } else if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v') {
                if (cur >= 0) {
                    if (cur >= nsym) { rc = fail(IMC_ERR_SYMBOL, "symbol " + std::to_string(cur) + " at column " + std::to_string(sym.size()) + " >= nsym"); break; }
                    sym.push_back((uint8_t)cur);
                    cur = -1;
                }
            } else {
                rc = fail(IMC_ERR_IO, std::string("unexpected character in ") + path);   // int() would raise ValueError (hmm.py:14)
                break;
            }
        }
    }
    std::fclose(fp);
    if (rc != IMC_OK) return rc;
    if (cur >= 0) {
        if (cur >= nsym) return fail(IMC_ERR_SYMBOL, "symbol " + std::to_string(cur) + " >= nsym");
        sym.push_back((uint8_t)cur);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    return obs_upload(sym.data(), sym.size(), nsym, out);
}

size_t imc_obs_length(const imc_obs *obs) { return obs ? obs->L : 0; }
int imc_obs_nsym(const imc_obs *obs) { return obs ? obs->nsym : 0; }

int imc_obs_free(imc_obs *obs)
{
    if (!obs) return IMC_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    if (obs->pid == getpid() && g.ready) {
        // plans hold raw pointers into this chunk's device buffer
        for (auto it = g_plans.begin(); it != g_plans.end();) {
            bool uses = false;
            for (size_t k = 0; k + 4 <= (*it)->key.size() && k < (size_t)(*it)->n_chunks; ++k)
                if ((*it)->key[k] == obs->id) uses = true;
            if (uses) { (*it)->release(); it = g_plans.erase(it); } else ++it;
        }
        (void)hipSetDevice(obs->device);
        (void)hipFree(obs->d_sym);
    }
    delete obs;
    return IMC_OK;
}

int imc_forward(const imc_obs *const *chunks, int n_chunks, int N, int S, const double *pi, const double *T,
                const double *E, double *out_loglik)
{
    if (!out_loglik) return fail(IMC_ERR_ARG, "out_loglik is null");
    return run_batch(chunks, n_chunks, 1, N, S, pi, T, E, out_loglik, nullptr);
}

int imc_forward_batch(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis,
                      const double *Ts, const double *Es, double *out_logliks)
{
    if (!out_logliks) return fail(IMC_ERR_ARG, "out_logliks is null");
    return run_batch(chunks, n_chunks, B, N, S, pis, Ts, Es, out_logliks, nullptr);
}

int imc_forward_batch_per_chunk(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis,
                                const double *Ts, const double *Es, double *out_per_chunk)
{
    if (!out_per_chunk) return fail(IMC_ERR_ARG, "out_per_chunk is null");
    return run_batch(chunks, n_chunks, B, N, S, pis, Ts, Es, nullptr, out_per_chunk);
}

int imc_forward_batch_device(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis,
                             const double *Ts, const double *Es, double *d_out_partial, void *hip_stream)
{
    if (!d_out_partial) return fail(IMC_ERR_ARG, "d_out_partial is null");
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = ensure_ctx()) return rc;
    if (int rc = check_args(chunks, n_chunks, B, N, S, pis, Ts, Es)) return rc;
    HIP_TRY(hipSetDevice(g.device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : g.stream;
    Plan *p = nullptr;
    if (int rc = build_plan(chunks, n_chunks, N, S, B, &p)) return rc;
    // the pinned staging buffer is reused by the next call: wait for the previous upload first
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = enqueue(p, pis, Ts, Es, st)) return rc;
    hipLaunchKernelGGL(k_sum_chunks, dim3((B + 63) / 64), dim3(64), 0, st, p->d_out, n_chunks, B, d_out_partial);
    HIP_TRY(hipGetLastError());
    return IMC_OK;
}

int imc_set_segment_length(size_t columns)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g.seg_override = columns;
    return IMC_OK;
}

int imc_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g.profile = on != 0;
    return IMC_OK;
}

int imc_profile_read(double *ms_propagate, double *ms_stitch, uint64_t *n_propagate, uint64_t *n_stitch)
{
    std::lock_guard<std::mutex> lk(g_mu);
    double mp = 0.0, ms = 0.0;
    for (auto &ev : g.events) {
        HIP_TRY(hipEventSynchronize(ev.c));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ev.a, ev.b));
        mp += t;
        HIP_TRY(hipEventElapsedTime(&t, ev.b, ev.c));
        ms += t;
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); (void)hipEventDestroy(ev.c);
    }
    if (ms_propagate) *ms_propagate = mp;
    if (ms_stitch) *ms_stitch = ms;
    if (n_propagate) *n_propagate = g.events.size();
    if (n_stitch) *n_stitch = g.events.size();
    g.events.clear();
    return IMC_OK;
}

int imc_last_plan(uint64_t *n_segments, uint64_t *n_vectors, uint64_t *segment_len, uint64_t *vector_columns)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (n_segments) *n_segments = g.last_segments;
    if (n_vectors) *n_vectors = g.last_vectors;
    if (segment_len) *segment_len = g.last_seglen;
    if (vector_columns) *vector_columns = g.last_vcols;
    return IMC_OK;
}

}  // extern "C"
