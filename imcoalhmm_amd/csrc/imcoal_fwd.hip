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
// Two propagate kernels share that frame:
//   k_propagate  (kernels_plain.hpp)  one step per alignment COLUMN; the R x N block of T' a lane needs
//                lives in VGPRs for the whole launch, the vector is exchanged through LDS broadcasts.
//   k_zpropagate (kernels_zip.hpp)    one step per TOKEN of the pair-compressed stream (zipHMM idea,
//                pair_dict.hpp); the per-token operators are built per evaluation in each CU's LDS.
// MFMA is deliberately not used (north_star): the matrices are tiny and fp64 MFMA has the VALU's rate.
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
#include <atomic>
#include <chrono>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../include/imcoal_fwd.h"
#include "kernels_plain.hpp"
#include "kernels_stitch.hpp"
#include "kernels_zip.hpp"
#include "kernels_big.hpp"
#include "kernels_zip2.hpp"
#include "kernels_zip3.hpp"
#include "kernels_zip4.hpp"
#include "pair_dict.hpp"
#include "model_host.hpp"
#include "../../include/imcoal_model.h"
#include "obs_io.hpp"

namespace {

thread_local std::string g_err;
std::mutex g_mu;
std::condition_variable g_cv;   // signalled when a plan stops being busy (see Plan::busy)

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

// k_zpropagate4 caches nothing in LDS while the operator tables of one launch (all parameter sets) fit this many bytes -
// they then stay in L2 / the Infinity Cache and a step's operands arrive from there a step ahead (IMC_Z4_STREAM=0/1
// forces the hybrid / the streamed form).  With more than one parameter set and the XCD-affine grid the streamed form is
// used whatever the total (PlanBuilder::z4_streamed).
constexpr double Z4_STREAM_MAX_BYTES = 32.0e6;
constexpr size_t STAGE_KERNEL_MAX_BYTES = 1u << 20;   // parameter sets up to this size are fetched by k_stage_params (enqueue)
constexpr size_t LDS_BUDGET = 160 * 1024 - 1024;   // bytes of LDS a compressed-path workgroup may use
constexpr size_t ZIP_MIN_COLUMNS = 4096;           // shorter chunks are not worth compressing
constexpr size_t DICT_TRAIN_MIN = 32768;           // first chunk at least this long trains the dictionary
constexpr size_t DICT_TRAIN_MAX = 8u << 20;        // byte tokens (< 256): train on at most this prefix
constexpr size_t DICT_WIDE_TRAIN_TOKENS = 1u << 19; // 16-bit tokens: train on at most this many byte-level tokens (~4e7 columns)
constexpr int DICT_MAX_DEPTH = 0;                  // 0: no limit on a token's depth in the dictionary.  (IMC_DICT_MAX_DEPTH, experiments:
                                                   // the table is built one depth per launch, but the bench alignment's 4096 tokens USE
                                                   // their 11 depths - a cap of 10 saves one launch at 149.5 instead of 150.2 columns per
                                                   // token, 9 already costs 13 % of the compression, 8 leaves 208 tokens)
constexpr size_t DICT_WIDE_MIN_COUNT = 6;          // ... in which a pair must occur this often
constexpr size_t CTAB_BUDGET = (size_t)24 << 30;   // largest operator table (all parameter sets) a plan may allocate
#ifndef IMC_HYBRID_MAX_ALPHABET
#define IMC_HYBRID_MAX_ALPHABET 4096
#endif
// hybrid table (k_zpropagate4): largest dictionary level considered.  Measured at N = 20 on the bench alignment: 1024 tokens
// (3.3 MB of operators, resident in every XCD's 4 MB L2) 0.360 ms per evaluation, 1536: 0.347, 4096 (13 MB, Infinity
// Cache): 0.342 - the columns per token saturate (133 / 139 / 150) while a cold step gets dearer
constexpr int HYBRID_MAX_ALPHABET = IMC_HYBRID_MAX_ALPHABET;
constexpr size_t Z2GRAN = 4;                       // blocked kernels: segment lengths are multiples of this many tokens
constexpr size_t R1_MIN_SEGLEN = 1024;             // rank-one hand-off: segments at least this long (stream elements) ...
constexpr size_t R1_MIN_HEAD = 256;                // ... run at least this many on the GEMM chain before the test
constexpr double R1_HEAD_COLUMNS = 65536.0;        // planner's estimate of the columns a head needs before it collapses
constexpr double R1_HEAD_MIN_COLUMNS = 8192.0;     // first checkpoint of the hand-off test, in alignment columns
constexpr int R1_MAX_ROUNDS = 24;                  // checkpoints per evaluation
constexpr double R1_DENSE_COLUMNS = 1.0e5;         // ... spaced 1.125x up to this many alignment columns, 1.5x beyond

hipError_t dev_alloc(void **p, size_t bytes);
void dev_free(void *p);

struct DictDev {                                   // one trained dictionary + its device copy
    imc::PairDict dict;
    uint16_t *d_left = nullptr, *d_right = nullptr;
    uint16_t *d_order = nullptr;                   // merged tokens sorted by (depth, id)
    std::vector<uint16_t> order;                   // host copy of d_order
    std::vector<int> depth;                        // per token; raw symbols have depth 0
    uint64_t trained_on = 0;                       // columns of the chunk(s) the training sample was drawn from
    pid_t pid = 0;
    ~DictDev()
    {
        if (pid == getpid()) { dev_free(d_left); dev_free(d_right); dev_free(d_order); }
    }
};

struct Ev3 { hipEvent_t a, b, c; };   // a: before propagate, b: after propagate, c: after stitch

struct Ctx {
    pid_t pid = 0;
    int device = -1;          // -1: use the thread's current device at first use
    bool ready = false;
    hipStream_t stream = nullptr;
    int cus = 256;
    uint64_t next_obs_id = 1, next_dict_id = 1;
    size_t seg_override = 0;
    int compression = 1;      // 0 = raw symbol stream, 1 = pair-compressed token stream where possible
    int kernel_pref = 0;      // 0 = automatic, 1 = vector kernels (k_propagate / k_zpropagate), 2 = blocked (k_zpropagate2)
    bool profile = false;
    bool rank1_handoff = true; // IMC_RANK1=0 switches the rank-one hand-off of the GEMM chain off (A/B measurements)
    int blocked_variant = 4;  // register-blocked kernel: 4 = fp64 MFMA 4x4x4 with the hybrid LDS/L2 table where it pays
                              // (k_zpropagate4) and the LDS table otherwise (k_zpropagate3); 3 = LDS table only;
                              // 5 = hybrid wherever it is possible (tests); 2 = k_zpropagate2 (DPP, VALU).
                              // IMC_BLOCKED=2|3|4|5 at start-up
    int z4_stream = -1;       // k_zpropagate4's table: -1 = streamed (nothing cached in LDS) while the launch's tables are
                              // cache resident, 0 = always the hybrid LDS cache, 1 = always streamed (IMC_Z4_STREAM)
    bool table_pairs = true;  // IMC_TABLE_PAIRS=0: k_zpropagate4's table one dictionary depth per launch (A/B measurements)
    int table_triples = -1;   // three dictionary depths per table launch, one wavefront per token (k_z4_level3): -1 = up to 12 states
                              // (measured at 10 states: 106 vs 108 us and 113.5 vs 116 us per evaluation; at 20 states 54 vs 52.5 us for
                              // the table - 2197 wavefronts of 125 MFMAs in the depth 7-9 launch - so pairs stay), 0 = never, 1 = always
                              // (IMC_TABLE_TRIPLES)
    bool xcd_affine = true;   // k_zpropagate4: a parameter set's workgroups all on one XCD when B is 2, 4 or a multiple of 8 (IMC_XCD_AFFINE=0: off)
    int fuse_tail = 1;        // the chunk's last workgroup finishes the chunk (zip3_tail) instead of k_chain launches: 1 = where a chunk is
                              // at most four workgroups (one in-wavefront fold; measured: 100 x 1e6 columns -1.5 us, and +6 us at 13
                              // workgroups of 20 states, where the chain's twenty parallel wavefronts win), 2 = wherever possible, 0 = never (IMC_FUSE_TAIL)
    bool fuse_head = true;    // IMC_FUSE_HEAD=0: k_stage_params + k_z4_raw as launches of their own (A/B measurements)
    bool pack_table = true;   // IMC_PACK_TABLE=0: the mat-vec chain reads the padded table (A/B measurements)
    bool guard = false;       // IMC_GUARD=1: every device buffer ends flush against an unmapped guard range (dev_alloc)
    bool use_graphs = false;  // IMC_GRAPH=1: replay each plan's launch sequence as a hipGraph (measured: no gain, the
                              // per-evaluation latency is kernel time + kernel boundaries, not host launch cost)
    std::vector<Ev3> events;
    std::map<int, std::shared_ptr<DictDev>> dicts;   // by raw alphabet size
    uint64_t last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t r1_checked = 0, r1_collapsed = 0;   // last call: operator segments tested / certified rank one
    std::string last_kernels;   // propagate kernels of the last enqueue, e.g. "k_zpropagate2<5>[tokens]"
} g;

void drop_plans();
void reset_kernel_attributes();

int ensure_ctx()
{
    const pid_t me = getpid();
    if (g.ready && g.pid == me) return IMC_OK;
    if (g.ready && g.pid != me)
        // A forked child of a process that had already used the GPU: the parent's HIP state (context, streams, every
        // device pointer held by cached plans and dictionaries) is not usable here and HIP cannot be re-initialised
        // after fork().  Refuse instead of guessing; nothing of the parent's is freed or touched.  Children that
        // build their own Forwarders (mcmc.py:112-121) must be forked BEFORE the parent's first library call.
        return fail(IMC_ERR_HIP, "libimcoal_fwd was initialised in the parent process before fork(): fork the chain "
                                 "processes before the first imc_* call of the parent, or use the spawn start method");
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
    g.use_graphs = std::getenv("IMC_GRAPH") != nullptr;
    if (const char *gd = std::getenv("IMC_GUARD")) g.guard = std::atoi(gd) != 0;
    if (const char *bv = std::getenv("IMC_BLOCKED")) { const int v = std::atoi(bv); if (v >= 2 && v <= 5) g.blocked_variant = v; }
    if (const char *r1 = std::getenv("IMC_RANK1")) g.rank1_handoff = std::atoi(r1) != 0;
    if (const char *pt = std::getenv("IMC_PACK_TABLE")) g.pack_table = std::atoi(pt) != 0;
    if (const char *tp = std::getenv("IMC_TABLE_PAIRS")) g.table_pairs = std::atoi(tp) != 0;
    if (const char *fh = std::getenv("IMC_FUSE_HEAD")) g.fuse_head = std::atoi(fh) != 0;
    if (const char *tt = std::getenv("IMC_TABLE_TRIPLES")) g.table_triples = std::atoi(tt) != 0 ? 1 : 0;
    if (const char *ft = std::getenv("IMC_FUSE_TAIL")) g.fuse_tail = std::max(0, std::min(2, std::atoi(ft)));
    if (const char *xa = std::getenv("IMC_XCD_AFFINE")) g.xcd_affine = std::atoi(xa) != 0;
    if (const char *zs = std::getenv("IMC_Z4_STREAM")) { const int v = std::atoi(zs); if (v >= -1 && v <= 1) g.z4_stream = v; }
    g.pid = me;
    g.ready = true;
    return IMC_OK;
}

// ---- device memory -----------------------------------------------------------------------------------
// Every device buffer of the library comes from here.  Normal mode: hipMalloc / hipFree.  Guard mode (IMC_GUARD=1, a
// test facility): each buffer is its own virtual-memory reservation with NO mapping behind its last 16-byte unit, so
// a kernel that reads or writes even one vector load past a buffer's declared size faults deterministically instead of
// silently touching whatever the allocator happened to place next (tests/test_gpu_guard.py runs the create / free /
// compression-flip / create sequence of round 1's unexplained fault this way).
struct GuardRec { hipDeviceptr_t va; size_t va_size; hipMemGenericAllocationHandle_t handle; hipDeviceptr_t map; size_t map_size; };
// (never destroyed: the context `g` - whose dictionaries free device buffers in their destructors - outlives every
// other static at process exit, and dev_free must still find the guard records then; a plain static map is destroyed
// first and the lookup walked freed nodes: "malloc_consolidate(): invalid chunk size" at exit under IMC_GUARD=1)
std::map<void *, GuardRec> &g_guard = *new std::map<void *, GuardRec>();

hipError_t dev_alloc(void **p, size_t bytes)
{
    bytes = std::max<size_t>(bytes, 16);
    if (!g.guard) return hipMalloc(p, bytes);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = g.device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorUnknown;
    GuardRec r{};
    r.map_size = (bytes + gran - 1) / gran * gran;
    r.va_size = r.map_size + 2 * gran;              // one unmapped granule before, one after
    e = hipMemAddressReserve(&r.va, r.va_size, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&r.handle, r.map_size, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(r.va, r.va_size); return e; }
    r.map = (hipDeviceptr_t)((char *)r.va + gran);
    e = hipMemMap(r.map, r.map_size, 0, r.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc ad{};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(r.map, r.map_size, &ad, 1);
        if (e != hipSuccess) (void)hipMemUnmap(r.map, r.map_size);
    }
    if (e != hipSuccess) { (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.va, r.va_size); return e; }
    void *user = (char *)r.map + (r.map_size - (bytes + 15) / 16 * 16);   // the buffer's end is the mapping's end
    g_guard[user] = r;
    *p = user;
    return hipSuccess;
}

void dev_free(void *p)
{
    if (!p) return;
    auto it = g_guard.find(p);
    if (it == g_guard.end()) { (void)hipFree(p); return; }
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(it->second.map, it->second.map_size);
    (void)hipMemRelease(it->second.handle);
    // The address range is NOT given back (hipMemAddressFree): a freed guard buffer stays unmapped for the rest of the
    // process, so a use after free faults too - and no later buffer can be mapped at an address some CU may still hold a
    // translation for.  (Round 3: with the ranges recycled, a Forwarder created right after another had been freed
    // sometimes evaluated to a wrong value in every plan - 3 of 4 processes, under IMC_GUARD=1 only, never with hipMalloc -
    // and a later launch faulted at an address nobody had computed; with the ranges kept, neither was seen again.)
    g_guard.erase(it);
}

}  // namespace

struct imc_obs {
    uint64_t id;
    pid_t pid;
    int device;
    int nsym;
    size_t L;
    uint8_t *d_sym;                        // L raw symbols + zero padding (bytes, or 16-bit values when wide_raw)
    bool wide_raw;                         // raw alphabet beyond 256 symbols
    std::shared_ptr<DictDev> dict;         // null: not compressed
    uint8_t *d_tok[imc::kNumLevels];       // token streams per level (aliases allowed), null if none
    bool wide[imc::kNumLevels];            // ... holding 16-bit ids (alphabets beyond 256)
    size_t ntok[imc::kNumLevels];
    int alphabet[imc::kNumLevels];
    std::vector<uint32_t> tok_count[imc::kNumLevels];   // byte levels: occurrences of every token id (hot-set choice of the hybrid table)
};

namespace {

constexpr size_t OBS_PAD = 256;

hipError_t upload_padded(const uint8_t *host, size_t n, uint8_t **dptr)
{
    const size_t bytes = ((n + OBS_PAD - 1) / OBS_PAD) * OBS_PAD + OBS_PAD;
    hipError_t e = dev_alloc((void **)dptr, bytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(*dptr, 0, bytes, g.stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(*dptr, host, n, hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    return e;
}

void release_tokens(imc_obs *o)          // the chunk falls back to its raw symbols
{
    for (int l = 0; l < imc::kNumLevels; ++l) {
        bool alias = false;
        for (int m = 0; m < l; ++m) alias |= (o->d_tok[m] == o->d_tok[l]);
        if (o->d_tok[l] && !alias) dev_free(o->d_tok[l]);
    }
    for (int l = 0; l < imc::kNumLevels; ++l) {
        o->d_tok[l] = nullptr; o->wide[l] = false; o->ntok[l] = 0; o->alphabet[l] = o->nsym; o->tok_count[l].clear();
    }
    o->dict.reset();
}

void obs_release(imc_obs *o)
{
    dev_free(o->d_sym);
    release_tokens(o);
}

// Build a chunk from validated host symbols.  Called WITHOUT g_mu: the O(L) host work (dictionary training, the
// multi-level encoding: ~1.4 s at 1e8 columns) runs unlocked so that other threads keep evaluating; the lock is
// taken only to read / publish the shared dictionary and for the HIP calls on the library's stream.
// A pair becomes a 16-bit token when it occurs this often in the training sample: DICT_WIDE_MIN_COUNT in a full
// sample, fewer in a short one - the dictionary of the FIRST chunk serves every later chunk of the alphabet, and a
// workload of many 1e7-column chunks (BASELINE config[3]) trains on one of them: with the full-sample threshold it got
// 1553 tokens (141 columns per token), with 3 occurrences 3473 (155) and a 3 % shorter evaluation; a token that turns
// out to be rare costs one table entry.  (IMC_DICT_MIN_COUNT overrides: experiments.)
size_t wide_min_count(size_t sample_tokens)
{
    if (const char *e = std::getenv("IMC_DICT_MIN_COUNT")) return (size_t)std::max(1, std::atoi(e));
    return sample_tokens >= 400000 ? DICT_WIDE_MIN_COUNT : sample_tokens >= 200000 ? 4 : 3;
}

// Train a pair dictionary on host symbols (exactly one of host / host16 is non-null); host work only.
std::shared_ptr<DictDev> make_dictionary(const uint8_t *host, const imc::tok_t *host16, size_t L, int nsym)
{
    auto nd = std::make_shared<DictDev>();
    const int max_depth = std::getenv("IMC_DICT_MAX_DEPTH") ? std::atoi(std::getenv("IMC_DICT_MAX_DEPTH")) : DICT_MAX_DEPTH;
    if (host16) {                                    // symbols are not bytes: straight to the 16-bit rounds on the raw stream
        imc::init_dict(nd->dict, nsym);
        const size_t nt = std::min(L - 1, DICT_WIDE_TRAIN_TOKENS * 8);
        imc::train_dict_wide(nd->dict, std::vector<imc::tok_t>(host16 + 1, host16 + 1 + nt), DICT_WIDE_MIN_COUNT, max_depth);
    } else {
        const size_t n = std::min(L - 1, DICT_TRAIN_MAX);
        imc::train_dict(nd->dict, nsym, std::vector<uint8_t>(host + 1, host + 1 + n), 64, max_depth);
        // The 16-bit rounds only start from a FULL byte dictionary.  On a large sample the byte phase can stop a few entries
        // short (the best remaining pair under the depth cap has 63 occurrences) - and the chunk then has no level beyond
        // ~250 tokens at all (seen once in a bench run: 96 instead of 150 columns per token, every evaluation 25-40 %
        // slower).  Retrain with a lower threshold rather than lose the wide levels.
        for (size_t min_count : {(size_t)8, (size_t)2}) {
            if (nd->dict.alphabet >= imc::kByteAlphabet || nd->dict.alphabet < 3 * imc::kByteAlphabet / 4 || n < DICT_TRAIN_MIN) break;
            imc::train_dict(nd->dict, nsym, std::vector<uint8_t>(host + 1, host + 1 + n), min_count, max_depth);
        }
        if (nd->dict.alphabet >= imc::kByteAlphabet) {   // 16-bit tokens: rounds over the whole chunk's byte-level stream
            const size_t cols = std::min(L, DICT_WIDE_TRAIN_TOKENS * 96);   // a byte-level token covers ~60-100 columns
            const std::vector<uint8_t> lvl256 = imc::encode_bytes(nd->dict, host, cols, nullptr);
            const size_t nt = std::min(lvl256.size() - 1, DICT_WIDE_TRAIN_TOKENS);
            imc::train_dict_wide(nd->dict, std::vector<imc::tok_t>(lvl256.begin() + 1, lvl256.begin() + 1 + nt), wide_min_count(nt), max_depth);
        }
    }
    nd->depth.assign(nd->dict.alphabet, 0);
    for (int z = nsym; z < nd->dict.alphabet; ++z)
        nd->depth[z] = 1 + std::max(nd->depth[nd->dict.left[z]], nd->depth[nd->dict.right[z]]);
    for (int z = nsym; z < nd->dict.alphabet; ++z) nd->order.push_back((uint16_t)z);
    std::stable_sort(nd->order.begin(), nd->order.end(),
                     [&](uint16_t x, uint16_t y) { return nd->depth[x] < nd->depth[y]; });
    return nd;
}

// Device copy of a freshly trained dictionary (g_mu held, context ready).
int upload_dictionary(const std::shared_ptr<DictDev> &nd)
{
    HIP_TRY(hipSetDevice(g.device));
    nd->pid = g.pid;
    nd->dict.id = g.next_dict_id++;
    const size_t abytes = (size_t)nd->dict.alphabet * sizeof(uint16_t);
    hipError_t e2 = dev_alloc((void **)&nd->d_left, abytes);
    if (e2 == hipSuccess) e2 = dev_alloc((void **)&nd->d_right, abytes);
    if (e2 == hipSuccess) e2 = hipMemcpy(nd->d_left, nd->dict.left.data(), abytes, hipMemcpyHostToDevice);
    if (e2 == hipSuccess) e2 = hipMemcpy(nd->d_right, nd->dict.right.data(), abytes, hipMemcpyHostToDevice);
    if (e2 == hipSuccess) e2 = dev_alloc((void **)&nd->d_order, std::max<size_t>(abytes, 16));
    if (e2 == hipSuccess && !nd->order.empty())
        e2 = hipMemcpy(nd->d_order, nd->order.data(), nd->order.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e2 != hipSuccess) return fail(IMC_ERR_HIP, std::string("dictionary upload: ") + hipGetErrorString(e2));
    return IMC_OK;
}

// The token streams of `enc` (encoded with dd's dictionary) become chunk o's (g_mu held).  On failure the chunk has no
// token streams left (it evaluates on its raw symbols).
int install_encoding(imc_obs *o, const std::shared_ptr<DictDev> &dd, const imc::EncodedLevels &enc)
{
    const int nsym = o->nsym;
    o->dict = dd;
    for (int l = 0; l < imc::kNumLevels; ++l) {
        o->alphabet[l] = enc.alphabet[l];
        o->ntok[l] = enc.length[l];
        o->wide[l] = enc.is_wide[l];
        o->tok_count[l].clear();
        if (l > 0 && enc.alphabet[l] == enc.alphabet[l - 1]) { o->d_tok[l] = o->d_tok[l - 1]; o->wide[l] = o->wide[l - 1]; o->tok_count[l] = o->tok_count[l - 1]; continue; }
        if (enc.alphabet[l] <= nsym) { o->d_tok[l] = nullptr; continue; }   // raw stream: use d_sym
        if (!enc.is_wide[l]) {
            o->tok_count[l].assign((size_t)enc.alphabet[l], 0u);
            for (size_t t = 1; t < enc.length[l]; ++t) o->tok_count[l][enc.bytes[l][t]]++;   // (position 0 is a raw symbol)
        } else if (enc.alphabet[l] <= HYBRID_MAX_ALPHABET) {
            o->tok_count[l].assign((size_t)enc.alphabet[l], 0u);
            for (size_t t = 1; t < enc.length[l]; ++t) o->tok_count[l][enc.wide[l][t]]++;
        }
        hipError_t e3 = enc.is_wide[l]
            ? upload_padded(reinterpret_cast<const uint8_t *>(enc.wide[l].data()), enc.length[l] * sizeof(imc::tok_t), &o->d_tok[l])
            : upload_padded(enc.bytes[l].data(), enc.length[l], &o->d_tok[l]);
        if (e3 != hipSuccess) {
            release_tokens(o);
            return fail(e3 == hipErrorOutOfMemory ? IMC_ERR_OOM : IMC_ERR_HIP,
                        std::string("token stream upload: ") + hipGetErrorString(e3));
        }
    }
    return IMC_OK;
}

int obs_upload(const uint8_t *host, const imc::tok_t *host16, size_t L, int nsym, imc_obs **out)
{
    // exactly one of host (alphabets up to 256: bytes) and host16 (larger alphabets) is non-null when L > 0
    const bool wide_raw = nsym > imc::kByteAlphabet;
    std::shared_ptr<DictDev> dd;
    bool want_zip = false, train = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ctx()) return rc;
        // ---- compression (the preprocess_raw_observations analogue, hmm.py:16) ----
        want_zip = g.compression && L >= ZIP_MIN_COLUMNS && nsym < imc::kMaxAlphabet / 2;
        if (want_zip) {
            auto it = g.dicts.find(nsym);
            if (it != g.dicts.end()) dd = it->second;
            else train = L >= DICT_TRAIN_MIN;
        }
    }
    if (train) {                                        // host only, unlocked
        auto nd = make_dictionary(host, host16, L, nsym);
        nd->trained_on = L;
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g.dicts.find(nsym);
        if (it != g.dicts.end()) dd = it->second;       // another thread published one meanwhile: use that
        else {
            if (int rc = upload_dictionary(nd)) return rc;
            g.dicts[nsym] = nd;
            dd = nd;
        }
    }
    imc::EncodedLevels enc;
    const bool zipped = dd && dd->dict.alphabet > nsym;
    if (zipped) imc::encode_levels(dd->dict, host, host16, L, enc);   // host only, unlocked (a published dictionary is immutable)

    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = ensure_ctx()) return rc;
    HIP_TRY(hipSetDevice(g.device));
    auto *o = new (std::nothrow) imc_obs();
    if (!o) return fail(IMC_ERR_OOM, "host allocation failed");
    o->id = g.next_obs_id++;
    o->pid = g.pid;
    o->device = g.device;
    o->nsym = nsym;
    o->L = L;
    o->d_sym = nullptr;
    o->wide_raw = wide_raw;
    for (int l = 0; l < imc::kNumLevels; ++l) { o->d_tok[l] = nullptr; o->wide[l] = false; o->ntok[l] = 0; o->alphabet[l] = nsym; }
    hipError_t e = wide_raw ? upload_padded(reinterpret_cast<const uint8_t *>(host16), L * sizeof(imc::tok_t), &o->d_sym)
                            : upload_padded(host, L, &o->d_sym);
    if (e != hipSuccess) {
        delete o;
        return fail(e == hipErrorOutOfMemory ? IMC_ERR_OOM : IMC_ERR_HIP,
                    std::string("observation upload: ") + hipGetErrorString(e));
    }
    if (zipped) {
        if (int rc = install_encoding(o, dd, enc)) {
            obs_release(o);
            delete o;
            return rc;
        }
    }
    *out = o;
    return IMC_OK;
}

// ---- kernel table ---------------------------------------------------------------------------------

using ChainFn = void (*)(const ChainDesc *, int, const uint32_t *, uint32_t, uint32_t, const double *, const int *,
                         const int *, uint32_t, double *, int *, double *, int);

struct KernelChoice {
    int R, G, NP, VPW, minw;       // R == 0: large-N GEMM-chain path (kernels_big.hpp), NP = 32 * TR
    void (*plain)(PropArgs);
    void (*zip)(PropArgs);
    size_t (*zip_lds)(int);
    ChainFn chain;
    bool zip_attr_set;
    void (*big_table_raw)(BigArgs);
    void (*big_table_level)(BigArgs, const uint16_t *, int);
    void (*big_prop)(BigArgs, const BigBlock *);
    int big_prop_waves;            // wavefronts per workgroup of big_prop (NT, or 8 for the dealt-tiles variant)
    void (*big_vec)(BigArgs, const BigBlock *, int, int);
    void (*big_vec_tail)(BigArgs, const BigBlock *, int, int);   // rank-one hand-off: mat-vec chain over segment tails
    int big_vec_waves;
    int big_nslab;
    size_t big_lds;
    void (*zip2)(BigArgs);         // register-blocked token kernel (NP = 4 RB <= 24), else null
    size_t (*zip2_lds)(int);
    bool zip2_attr_set;
    bool plain_attr_set = false;
    void (*zip3)(BigArgs) = nullptr;   // ... its fp64-MFMA form (same launch geometry and block list)
    size_t (*zip3_lds)(int) = nullptr;
    bool zip3_attr_set = false;
    void (*zip4)(BigArgs) = nullptr;   // ... with the hybrid LDS / L2 operator table, and the kernels that build that table
    void (*zip4w)(BigArgs) = nullptr;  // ... on 16-bit token streams (dictionary levels beyond 256 tokens)
    void (*zip4s)(BigArgs) = nullptr, (*zip4sw)(BigArgs) = nullptr;   // ... with the STREAMED table (nothing cached in LDS)
    void (*zip4_raw)(BigArgs) = nullptr;
    void (*zip4_level)(BigArgs, int, int) = nullptr;
    void (*zip4_level2)(BigArgs, const int4 *, int, int, const double *) = nullptr;   // two dictionary depths per launch
    void (*zip4_level2_first)(BigArgs, const int4 *, int, int, const double *) = nullptr;   // ... the first one: parameters + raw operators too
    void (*zip4_level3)(BigArgs, const int4 *, int, int, const double *) = nullptr;        // three dictionary depths per launch
    void (*zip4_level3_first)(BigArgs, const int4 *, int, int, const double *) = nullptr;
    size_t (*zip4_level3_lds)(size_t) = nullptr;
    size_t (*zip4_lds)(int, int) = nullptr;
    int (*zip4_max_hot)(int, size_t) = nullptr;
    int tok_doubles = 0;               // doubles per table entry of the MFMA kernels
    bool chain_self_emax = false;      // the stitch kernel finds the units' largest exponents itself (no k_emax launch)
    bool zip4_attr_set = false, zip4w_attr_set = false, zip4s_attr_set = false, zip4sw_attr_set = false, zip4_attr_l3 = false;
    // the blocked kernel in use (g.blocked_variant) and its LDS need for an alphabet of A tokens
    bool use3() const;
    size_t blocked_lds(int A) const { return use3() ? zip3_lds(A) : zip2_lds(A); }
};

template <int R, int G, int MW>
KernelChoice make_kc()
{
    constexpr int NP = R * G;
    KernelChoice k{R, G, NP, 64 / G, MW, k_propagate<R, G, MW>, k_zpropagate<R, G>, &ZipGeom<R, G>::lds_bytes,
                   k_chain<NP, (NP <= 12 ? 6 : NP <= 24 ? 4 : NP <= 32 ? 3 : NP <= 64 ? 2 : 1)>, false, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, false};
    k.chain_self_emax = true;          // (k_chain<NP, D > 0>: one wavefront, prefetch ring)
    if constexpr (NP % 4 == 0 && NP <= 24) {
        k.zip2 = k_zpropagate2<NP / 4>;
        k.zip2_lds = &Zip2Geom<NP / 4>::lds_bytes;
        k.zip3 = k_zpropagate3<NP / 4>;
        k.zip3_lds = &Zip3Geom<NP / 4>::lds_bytes;
        k.tok_doubles = Zip3Geom<NP / 4>::TOK;
        if constexpr (NP <= 24) {   // (NP = 24: 27-29 registers of the streamed form spill, 35-72 of the hybrid one)
            k.zip4 = k_zpropagate4<NP / 4, false, true>;
            k.zip4w = k_zpropagate4<NP / 4, true, true>;
            k.zip4s = k_zpropagate4<NP / 4, false, false>;
            k.zip4sw = k_zpropagate4<NP / 4, true, false>;
            k.zip4_raw = k_z4_raw<NP / 4>;
            k.zip4_level = k_z4_level<NP / 4>;
            k.zip4_level2 = k_z4_level2<NP / 4, false>;
            k.zip4_level2_first = k_z4_level2<NP / 4, true>;
            k.zip4_level3 = k_z4_level3<NP / 4, false>;
            k.zip4_level3_first = k_z4_level3<NP / 4, true>;
            k.zip4_level3_lds = &Z4L3Geom<NP / 4>::lds_bytes;
            k.zip4_lds = &Zip4Geom<NP / 4>::lds_bytes;
            k.zip4_max_hot = &Zip4Geom<NP / 4>::max_hot;
        }
    }
    return k;
}

template <int NT, int NSLAB>
KernelChoice make_big()
{
    constexpr int NP = 16 * NT;   // NT wavefronts per workgroup
    // NT = 6, 10, 14: one tile-row per wavefront would load the SIMDs unevenly -> tiles dealt over 8 wavefronts
    constexpr bool dealt = NT > 4 && NT % 4 != 0;
    void (*prop)(BigArgs, const BigBlock *) = k_big_propagate<NT, NSLAB>;
    if constexpr (dealt) prop = k_big_propagate_s<NT, NSLAB>;
    return KernelChoice{0, NT, NP, 0, 1, nullptr, nullptr, nullptr, k_chain<NP, 0>, false, k_big_table_raw<NT>,
                        k_big_table_level<NT>, prop, dealt ? BS_WAVES : NT, k_big_vector<NT, false>, k_big_vector<NT, true>, BigVec<NT>::WAVES, NSLAB,
                        BigSlab<NT, NSLAB>::bytes, nullptr, nullptr, false};
}

bool KernelChoice::use3() const { return zip3 && g.blocked_variant >= 3; }

KernelChoice kChoices[] = {
    make_kc<4, 1, 2>(), make_kc<4, 2, 2>(), make_kc<4, 3, 2>(), make_kc<4, 4, 2>(), make_kc<4, 5, 2>(),
    make_kc<3, 8, 2>(), make_kc<2, 14, 2>(), make_kc<2, 16, 2>(), make_kc<2, 20, 1>(), make_kc<1, 48, 1>(),
    make_kc<1, 56, 1>(), make_kc<1, 64, 1>(),
    make_big<6, 1>(), make_big<7, 1>(), make_big<8, 2>(), make_big<9, 3>(), make_big<10, 2>(), make_big<12, 3>(), make_big<14, 7>(),
    make_big<16, 8>(),
};
constexpr int IMC_MAX_N = 256;

// MFMA GEMM-chain variants for 24 < N <= 64 (small workgroups, several per CU): used instead of the vector
// kernels when a launch is dominated by operator segments (long chunks).
KernelChoice kMidChoices[] = {make_big<2, 1>(), make_big<3, 1>(), make_big<4, 1>()};   // NP = 32, 48, 64

KernelChoice *choose_kernel(int N, bool prefer_gemm)
{
    if (prefer_gemm && N > 24 && N <= 64)
        for (auto &k : kMidChoices)
            if (k.NP >= N) return &k;
    for (auto &k : kChoices)
        if (k.NP >= N) return &k;
    return nullptr;
}

void reset_kernel_attributes()
{
    for (auto &k : kChoices) k.zip4_attr_l3 = k.zip_attr_set = k.zip2_attr_set = k.zip3_attr_set = k.zip4_attr_set = k.zip4w_attr_set = k.zip4s_attr_set = k.zip4sw_attr_set = k.plain_attr_set = false;
    for (auto &k : kMidChoices) k.zip4_attr_l3 = k.zip_attr_set = k.zip2_attr_set = k.zip3_attr_set = k.zip4_attr_set = k.zip4w_attr_set = k.zip4s_attr_set = k.zip4sw_attr_set = k.plain_attr_set = false;
}

// ---- launch plan ----------------------------------------------------------------------------------

struct Group {             // one propagate launch
    bool big = false;      // large-N GEMM-chain kernel (one workgroup per segment)
    bool bigvec = false;   // ... every chunk is one segment: mat-vec chain kernel (k_big_vector), no operators
    bool zip2 = false;     // register-blocked token kernel (one 16-lane row per segment)
    std::vector<uint32_t> seg_ids, seg_out;   // big: segment ids and their level-0 vector index
    std::vector<Z2Block> blocks;              // zip2: one entry per workgroup
    std::vector<BigBlock> big_blocks;         // big: one entry per (segment, column slab) workgroup
    BigBlock *d_big_blocks = nullptr;
    uint32_t *d_seg_ids = nullptr, *d_seg_out = nullptr;
    Z2Block *d_blocks = nullptr;
    bool zip4 = false;                        // blocked MFMA kernel with the hybrid table: alphabet beyond LDS, n_hot operators cached
    bool wide_tokens = false;                 // ... its token stream holds 16-bit ids
    bool stream_table = false;                // ... nothing cached in LDS: the launch's tables are small enough to stay cache resident
    int n_hot = 0;
    std::vector<uint16_t> hot;
    uint16_t *d_hot = nullptr;
    uint16_t *d_tab_order = nullptr;          // blocked MFMA kernel: merged tokens of the alphabet by dictionary depth
    int *d_tab_lvl = nullptr;
    int4 *d_tab_desc = nullptr;               // hybrid table: {token, left, right, 0} per entry of the depth order
    int4 *d_tab_desc2 = nullptr;              // ... two int4 per entry of the two-depths-per-launch schedule (k_z4_level2)
    std::vector<std::pair<int, int>> tab2;    // ... (first entry, entries) per launch
    // fused tail (zip3_tail): per workgroup {chunk, unit, units of the chunk}; published operators, exponents, arrival counters
    std::vector<Z2Tail> tails;
    Z2Tail *d_tails = nullptr;
    double *d_tailX = nullptr;
    int *d_tailE = nullptr, *d_tail_arrive = nullptr;
    int tail_stride = 0;
    int4 *d_tab_desc3 = nullptr;              // ... three int4 per token of the three-depths-per-launch schedule (k_z4_level3): {token, leaves 0-2}, {leaves 3-6}, {leaf 7}
    std::vector<std::pair<int, int>> tab3;
    std::vector<int> tab_lvl;                 // host copy of the depth offsets
    int tab_nlvl = 0;
    double *d_Ctab = nullptr;
    double *d_Cpack = nullptr;                // big groups that run the mat-vec chain: packed copy of the table ([B][A][N][TS])
    int *d_cex = nullptr;
    // rank-one hand-off (GEMM chain only): operator segments run on the GEMM chain in rounds that end at the
    // checkpoints (token counts); after each round k_rank1_check tests the segments still on the chain, and those
    // that collapsed to u alpha^T finish on the mat-vec chain from that checkpoint.  The schedule is fixed per plan,
    // the decision per segment comes from the data of the evaluation: nothing is carried from call to call.
    bool rank1 = false;
    int head_len = 0;                         // the planner's estimate (cost model only)
    std::vector<int> checkpoints;
    std::vector<BigBlock> tail_blocks;        // one entry per segment (operator tails and first segments)
    std::vector<std::pair<uint32_t, uint32_t>> r1_segs;   // (plan-wide id, length) of the operator segments
    BigBlock *d_tail_blocks = nullptr;
    int *d_r1flag = nullptr, *d_r1at = nullptr;
    double *d_r1u = nullptr, *d_r1alpha = nullptr;
    bool zip = false;
    int level = -1, A = 0;
    std::shared_ptr<DictDev> dict;
    std::vector<int> chunks;
    uint32_t vec_begin = 0, n_vecs = 0;
    size_t seglen = 0;
    uint64_t vsteps = 0;   // executed vector-steps per parameter set (columns or tokens)
    uint64_t stream_len = 0;
};

struct Level {             // one level of the stitch hierarchy (level 0 = propagate output)
    uint32_t n_segs = 0, n_vecs = 0, n_chains = 0;
    uint32_t *d_vec0 = nullptr;
    uint8_t *d_first = nullptr;
    ChainDesc *d_chains = nullptr;   // chains that produce THIS level from the previous one
    double *d_P = nullptr;
    int *d_EX = nullptr, *d_EMAX = nullptr;
    void release()
    {
        dev_free(d_vec0); dev_free(d_first); dev_free(d_chains);
        dev_free(d_P); dev_free(d_EX); dev_free(d_EMAX);
    }
};

struct Plan {
    std::vector<uint64_t> key;
    KernelChoice *kc = nullptr;
    int N = 0, S = 0, B = 0, n_chunks = 0;
    uint32_t n_segs = 0, n_vecs = 0;
    std::vector<Group> groups;
    size_t pstride = 0;
    SegDesc *d_segs = nullptr;
    VecDesc *d_vecs = nullptr;
    std::vector<Level> levels;       // levels[0] holds the propagate output
    int32_t *d_final_vec = nullptr;  // per chunk: vector index in the last level, -1 for an empty chunk
    uint64_t chain_steps = 0;        // serial depth of the stitch (sum over levels of the longest chain)
    bool finish_fused = false;       // the last chain level writes the log-likelihoods itself (no k_finish launch)
    double *d_params = nullptr, *d_out = nullptr;
    // pinned staging of the caller's parameters: two slots used alternately, each guarded by an event recorded behind
    // the call that read it, so a call only waits when the call issued two calls earlier is still running
    double *h_params[2] = {nullptr, nullptr};
    hipEvent_t ev_params[2] = {nullptr, nullptr};
    bool ev_used[2] = {false, false};
    int slot = 0;
    double *h_out = nullptr;                        // pinned, mapped: the synchronous path's k_finish writes results straight to the host
    double *h_out_dev = nullptr;                    // its device-visible alias
    double *h_params_dev[2] = {nullptr, nullptr};   // device-visible aliases of the staging slots (k_stage_params)
    // The plan's device buffers are shared by every call on these chunks.  Calls may arrive on different streams (the
    // library's own for the synchronous entry points, the caller's for imc_forward_batch_device): ev_params[slot] is
    // recorded behind each call, and a call on another stream than its predecessor's first waits for it.
    hipStream_t last_stream = nullptr;
    int last_slot = 0;
    bool have_last = false;
    // A synchronous call (run_batch) enqueues under g_mu, then waits for its results WITHOUT it, so that other threads can
    // enqueue their own evaluations behind it (round 2 held the one mutex across the wait: two Python threads with
    // distinct chunk sets serialised completely).  While it waits the plan is `busy`: a second synchronous call on the
    // SAME plan (they share the mapped result slots) waits on g_cv, and nothing releases a busy plan.
    bool busy = false;
    hipGraphExec_t graph = nullptr;                 // captured enqueue(), replayed by run_batch
    uint64_t calls = 0;
    uint64_t lp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::string kernels;
    void release()
    {
        if (graph) (void)hipGraphExecDestroy(graph);
        dev_free(d_segs); dev_free(d_vecs); dev_free(d_final_vec);
        for (auto &l : levels) l.release();
        for (auto &gr : groups) { dev_free(gr.d_seg_ids); dev_free(gr.d_seg_out); dev_free(gr.d_blocks); dev_free(gr.d_hot); dev_free(gr.d_tab_desc); dev_free(gr.d_tab_desc2); dev_free(gr.d_tab_desc3); dev_free(gr.d_tails); dev_free(gr.d_tailX); dev_free(gr.d_tailE); dev_free(gr.d_tail_arrive); dev_free(gr.d_tab_order); dev_free(gr.d_tab_lvl); dev_free(gr.d_big_blocks); dev_free(gr.d_Ctab); dev_free(gr.d_Cpack); dev_free(gr.d_cex); dev_free(gr.d_tail_blocks); dev_free(gr.d_r1flag); dev_free(gr.d_r1at); dev_free(gr.d_r1u); dev_free(gr.d_r1alpha); }
        dev_free(d_params); dev_free(d_out);
        for (int k = 0; k < 2; ++k) { (void)hipHostFree(h_params[k]); if (ev_params[k]) (void)hipEventDestroy(ev_params[k]); }
        (void)hipHostFree(h_out);
    }
};

std::list<std::unique_ptr<Plan>> g_plans;   // most recent first
constexpr size_t MAX_PLANS = 4;

// Wait (g_mu held through `lk`) until no synchronous call is waiting on any plan: callers that release plans.
void wait_all_idle(std::unique_lock<std::mutex> &lk)
{
    g_cv.wait(lk, [] {
        for (auto &p : g_plans)
            if (p->busy) return false;
        return true;
    });
}

void drop_plans()
{
    for (auto &p : g_plans) p->release();
    g_plans.clear();
}

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// Pick the segment length (in stream elements) that minimises a simple machine model: equal-length
// wavefront tasks run in rounds of `resident` wavefronts at `step_cost` cycles per step; the serial
// stitch adds ~stitch_cost per segment of the longest chunk.
size_t choose_seglen(const std::vector<size_t> &lens, int N, int B, int VPW, double resident, double step_cost,
                     double *cost_out = nullptr)
{
    size_t maxlen = 0;
    for (size_t L : lens) maxlen = std::max(maxlen, L);
    if (cost_out) *cost_out = (double)maxlen * step_cost;
    if (maxlen <= 256) return std::max<size_t>(round_up(maxlen, 16), 16);
    const double stitch_cost = 12.0 * N + 300.0;   // cycles per stitched segment
    double best = 1e300;
    size_t best_seg = maxlen;
    for (double s = 128.0;; s *= 1.189207115) {
        const size_t seg = std::min(round_up((size_t)s, 16), round_up(maxlen, 16));
        double vecs = 0.0, kmax = 0.0;
        for (size_t L : lens) {
            if (!L) continue;
            const double K = std::ceil((double)L / (double)seg);
            vecs += 1.0 + (K - 1.0) * N;
            kmax = std::max(kmax, K);
        }
        const double waves = std::ceil(vecs * B / VPW);
        const double rounds = std::ceil(waves / resident);
        const double cost = rounds * (double)seg * step_cost + kmax * stitch_cost;
        if (cost < best) { best = cost; best_seg = seg; }
        if (seg >= maxlen) break;
    }
    if (cost_out) *cost_out = best;
    return best_seg;
}

// Cycles per chain step and CU of the mat-vec chain kernel (k_big_vector): NP^2 doubles streamed per step.
// Measured at N=150, 2048 chains: ~NP^2/3 cycles while one parameter set's table stays within ~6 MB (its share of
// an XCD's L2), growing by ~3 % per further MB as reads fall through to the Infinity Cache, up to the HBM rate.
// GEMM chain: cycles (at ~2.2 GHz) a CU needs to advance ONE segment by one step - all column slabs of the segment, chip
// full.  Measured round 3 on every built tile count below (profiles/r03_e_calib_big.txt: one 1e7-column chunk, all CUs
// busy): 0.55 us at NP = 32, 1.52 at 48, 2.9 at 64, 9.4 at 96, 15.0 at 112, 33 at 160.  Rounds 1-2 used
// 0.027 NP^3 + 20000 cycles, fitted at NP = 160 on round 1's kernel: 1.8x too high there and 16x too high at NP = 32, so
// that 24 < N <= 64 with a handful of chunks x parameter sets took the mat-vec chain (one workgroup per chain, 10-40
// workgroups on 256 CUs) at 3-9x the GEMM chain's time.
static double gemm_cu_step_cycles(int np)
{
    const int nt = np / 16;
    double us;
    switch (nt) {
    case 2: us = 0.55; break;
    case 3: us = 1.52; break;
    case 4: us = 2.9; break;
    case 6: us = 9.4; break;
    case 7: us = 15.0; break;
    case 8: us = 22.4; break;
    case 9: us = 24.1; break;
    default: us = 33.0 * (nt / 10.0) * (nt / 10.0) * (nt / 10.0); break;
    }
    return us * 2200.0;
}

static double matvec_step_cycles(double np2, int alphabet)
{
    const double table_mb = (double)alphabet * np2 * 8.0 / 1.0e6;
    return np2 / 3.0 * std::min(2.2, 1.0 + 0.03 * std::max(0.0, table_mb - 6.0));
}

// Rank-one hand-off estimate for a GEMM-chain group whose base plan has `nseg` segments of `seglen` elements (one
// round of workgroups): m times the segments = m rounds of `head`-long GEMM heads, tails 1/m as long on the mat-vec
// chain.  A tail step streams NP^2 doubles; with many chains that is the memory system's bandwidth (measured: 256
// chains of a 1.6 GB table read 7 TB/s), with few it is one workgroup's latency.  Returns the best m (0: not
// worth it) and its cycles.
struct HandoffEstimate { int m; double cycles; };
static HandoffEstimate estimate_handoff(double seglen, double nseg, int B, double head, double np, int nslab, int alphabet, int cus)
{
    // (a segment's GEMM step is shared by its nslab column-slab workgroups)
    const double np2 = np * np, t_gemm = 1.15 * gemm_cu_step_cycles((int)np) / nslab;   // (a head step of one slab's workgroup)
    const double table_mb = (double)alphabet * np2 * 8.0 / 1.0e6;
    const double bw = table_mb > 128.0 ? 7.0e12 : table_mb > 16.0 ? 8.6e12 : 15.0e12;   // bytes/s: HBM, Infinity Cache, L2
    HandoffEstimate best{0, 1e300};
    for (int m = 1; m <= 6; ++m) {
        const double sl = seglen / m;
        if (sl < 2.0 * head) break;
        const double chains = m * nseg * B, active = std::min(chains, (double)cus);
        const double t_vec = std::max(np2 / 4.5 + 1500.0, active * np2 * 8.0 * 2.1e9 / bw) * std::max(1.0, chains / cus);
        const double est = m * head * t_gemm + (sl - head) * t_vec;
        if (est < best.cycles) best = HandoffEstimate{m, est};
    }
    return best;
}

// Builds one launch plan in phases; every phase reads what the earlier ones left in the members.
struct PlanBuilder {
    const imc_obs *const *chunks = nullptr;
    int n_chunks = 0, N = 0, S = 0, B = 0;
    bool op_mode = false;               // imc_forward_state(as_operator): no segment is a chunk's "first"
    KernelChoice *kc = nullptr;
    std::unique_ptr<Plan> p;
    bool big = false;                   // GEMM-chain / mat-vec chain kernels (global-memory operator table)
    std::vector<int> chunk_group;       // chunk -> index into p->groups
    std::vector<SegDesc> segs;          // all segments, chunk order
    std::vector<uint8_t> seg_first;
    std::vector<uint32_t> chunk_seg;    // chunk -> first segment (n_chunks + 1 entries)
    std::vector<VecDesc> vecs;          // level-0 vectors
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> chunk_units;
    std::vector<uint32_t> chunk_unit, unit_vec0;
    std::vector<uint8_t> unit_first;
    struct HostLevel { std::vector<uint32_t> chunk_seg, vec0; std::vector<uint8_t> first; std::vector<ChainDesc> chains; uint32_t n_vecs; };
    std::vector<HostLevel> hl;          // stitch hierarchy, level 0 = propagate output
    std::vector<int32_t> final_vec;     // chunk -> its single remaining unit at the last level (-1: empty chunk)

    // k_zpropagate4 on the streamed table (nothing cached in LDS)?  With the XCD-affine grid (enqueue) an XCD reads one or
    // two parameter sets' tables at a time whatever B is, and the streamed form then beat the hybrid one at every
    // dictionary level and every B measured (profiles/r03_d_affine_levels.txt); without it, while all tables of the launch
    // fit the L2s.
    static bool z4_streamed(double tables_bytes, int n_sets)
    {
        if (g.z4_stream >= 0) return g.z4_stream == 1;
        return (g.xcd_affine && n_sets > 1) || tables_bytes <= Z4_STREAM_MAX_BYTES;
    }

    // hand-off head of a group, in stream elements: ~R1_HEAD_COLUMNS alignment columns
    size_t handoff_head(const Group &gr) const
    {
        double cols = 0.0, toks = 0.0;
        for (int f : gr.chunks) { cols += (double)chunks[f]->L; toks += (double)(gr.zip ? chunks[f]->ntok[gr.level] : chunks[f]->L); }
        const double span = toks > 0.0 ? cols / toks : 1.0;
        return round_up(std::max<size_t>(R1_MIN_HEAD, (size_t)(R1_HEAD_COLUMNS / span)), 16);
    }

    void assign_groups()
    {
        // ---- assign chunks to launch groups: plain, or (dictionary, level) ----
        big = kc->R == 0;
        int a_max = 0;   // largest alphabet whose operator table fits LDS for this N (no limit on the large-N path)
        if (big) a_max = imc::kMaxAlphabet;
        else
            for (int A = 1; A <= imc::kByteAlphabet; ++A)   // (the LDS-table kernels read byte streams only)
                if (kc->zip_lds(A) <= LDS_BUDGET || (kc->zip2 && g.kernel_pref != 1 && kc->blocked_lds(A) <= LDS_BUDGET)) a_max = A;
        // One dictionary level per dictionary: the deepest level is not always the best - every workgroup rebuilds
        // the operator table per evaluation ((A - S) dependent small products), which dominates on short inputs.
        // Estimate: table build + main loop with all 16-lane rows of the machine busy.
        std::map<const DictDev *, int> dict_level;
        const bool mfma_blocked = !big && kc->use3() && g.kernel_pref != 1;
        if (g.compression) {
            std::map<const DictDev *, std::vector<int>> by_dict;
            for (int f = 0; f < n_chunks; ++f)
                if (chunks[f]->dict && chunks[f]->nsym == S) by_dict[chunks[f]->dict.get()].push_back(f);
            for (auto &kv : by_dict) {
                double best = 1e300;
                int best_l = -1;
                for (int l = 0; l < imc::kNumLevels; ++l) {
                    const imc_obs *o0 = chunks[kv.second[0]];
                    if (mfma_blocked) {
                        // Blocked MFMA kernels, estimated in microseconds: the table is built one dictionary depth at
                        // a time (32 tokens per pass), the scan advances cus * 32 segments per wavefront-step.  Levels
                        // that fit LDS run k_zpropagate3; byte levels beyond that can run k_zpropagate4 (hybrid table:
                        // one workgroup builds it in L2, steps on tokens outside the LDS-cached hot set cost ~12 % more).
                        const int A = o0->alphabet[l];
                        if (!(A > o0->nsym && A <= HYBRID_MAX_ALPHABET && o0->d_tok[l]) || o0->wide_raw) continue;
                        const bool fits = !o0->wide[l] && kc->blocked_lds(A) <= LDS_BUDGET;   // (k_zpropagate3 reads byte streams)
                        const int max_hot = kc->zip4 ? kc->zip4_max_hot(A, LDS_BUDGET) : 0;
                        const double table_bytes = (double)B * (A + 1) * kc->tok_doubles * 8.0;
                        const bool hybrid_ok = !fits && kc->zip4 && g.blocked_variant >= 4 && max_hot >= 8 && !o0->tok_count[l].empty() &&
                                               table_bytes <= 2.0e9;
                        if (!fits && !hybrid_ok) continue;
                        // (the scan's LDS: the fold's exchange area + the exponents and slot map of all A + 1 entries - at 24
                        // states the exchange area alone is 152 KB and levels beyond 768 tokens do not fit)
                        if (!fits && kc->zip4_lds(A, 0) > LDS_BUDGET) continue;
                        int passes = 0;
                        {
                            std::map<int, int> per_depth;
                            for (int z = o0->nsym; z < A; ++z) per_depth[kv.first->depth[z]]++;
                            for (auto &pd : per_depth) passes += (pd.second + (int)Z2SLOTS - 1) / (int)Z2SLOTS;
                        }
                        // (up to 8 states a step is not MFMA time: measured round 3, 4 and 8 states, 64 sets - 0.08 / 0.14 us from
                        // the LDS table, 0.19 / 0.22 from the global one; from 12 states on both follow the instruction count)
                        const double nt = kc->NP / 4.0;
                        const double t_step = nt < 1.5 ? 0.08 : nt < 2.5 ? 0.14 : std::max(0.25, 1.88 * nt * nt * nt / 125.0);
                        const double t_step_g = nt < 1.5 ? 0.19 : nt < 2.5 ? 0.22 : t_step;
                        // Time of the scan itself: workgroups of 32 segments are dealt PER CHUNK (a chunk of u workgroups is cut
                        // into 32 u segments), and a launch of more workgroups than CUs runs in rounds, each paying the
                        // workgroup's fixed part again.  rounds x (fixed + segment length x step) for the best number of rounds -
                        // not tokens / slots: 100 chunks x 1e6 columns fill 78 % of one round or 98 % of two.
                        double maxtok = 0.0;
                        for (int f : kv.second) maxtok = std::max(maxtok, (double)chunks[f]->ntok[l]);
                        // (up to 12 states two workgroups of the global-table kernel share a CU: twice the slots, 1.85x the step)
                        const bool two_per_cu = !fits && kc->NP <= 12 && z4_streamed(table_bytes, B);
                        const double n_ch = (double)kv.second.size(), slots = (double)g.cus * Z2SLOTS * (two_per_cu ? 2.0 : 1.0);
                        auto scan_time = [&](double fixed_us, double step_us) {
                            if (two_per_cu) step_us *= 1.85;
                            double best_t = 1e300;
                            for (int r = 1; r <= 6; ++r) {
                                const double units = std::max(1.0, std::floor(slots * r / (n_ch * B) / Z2SLOTS));
                                const double seg = std::max(16.0, std::ceil(maxtok / (Z2SLOTS * units)));
                                const double rounds = std::ceil(n_ch * B * units * Z2SLOTS / slots);
                                best_t = std::min(best_t, rounds * (fixed_us + seg * step_us));
                            }
                            return best_t;
                        };
                        double cost;
                        // LDS table: EVERY workgroup rebuilds it, one barrier-separated pass per 32 tokens of a depth
                        // (measured round 3: ~1.9 us per pass at 10 states, ~2.8 at 20), plus launch and the in-kernel
                        // fold (~14 us).  Round 2 priced a pass at 1.1 us and no fixed part: at 10 states and 100 x 1e6
                        // columns it then preferred the 128-token LDS table (135 us) to the global table (102 us).
                        if (fits) cost = 4.0 + scan_time(10.0 + passes * (1.6 + 1.2 * nt * nt * nt / 125.0), t_step);
                        else {
                            std::vector<uint64_t> cnt((size_t)A, 0);
                            for (int f : kv.second)
                                for (size_t z = 0; z < chunks[f]->tok_count[l].size() && z < cnt.size(); ++z) cnt[z] += chunks[f]->tok_count[l][z];
                            std::vector<uint64_t> sorted(cnt);
                            std::sort(sorted.begin(), sorted.end(), std::greater<uint64_t>());
                            uint64_t all = 0, top = 0;
                            for (size_t z = 0; z < sorted.size(); ++z) { all += sorted[z]; if ((int)z < std::min(max_hot, A)) top += sorted[z]; }
                            const double cold = all ? 1.0 - (double)top / (double)all : 0.0;
                            const double depths = (double)std::max<int>(1, (int)std::set<int>(kv.first->depth.begin() + o0->nsym, kv.first->depth.begin() + A).size());
                            // one ~4 us launch per depth; a cold step costs ~11 % more while one parameter set's table
                            // stays in an XCD's L2, ~17 % from the Infinity Cache (scratch microbenchmark)
                            // (hybrid form; the streamed form - tables of the launch cache resident - runs every step
                            // from the global table at ~6 % over the LDS-table kernel's step: measured at config[1])
                            const double cold_pen = table_bytes / B <= 3.6e6 ? 0.11 : 0.17;
                            const bool streamed = z4_streamed(table_bytes, B);
                            // (table: one ~4.5 us launch per depth, or one ~5.7 us launch per pair of depths)
                            // (measured round 3: the first launch - it fetches the parameters - ~13 us, the others ~7.5)
                            double t_table = g.table_pairs ? 13.0 + (std::ceil(depths / 2.0) - 1.0) * 7.5 : 6.0 + depths * 4.9;
                            // ... which is latency; B tables of A operators are also two reads and a write of an operator per
                            // token and parameter set, at ~2.8 TB/s (measured round 3, 64 sets: 4096- against 512-token tables)
                            t_table += std::max(0.0, table_bytes * 3.0 / 2.8e6 - 0.5 * t_table);
                            cost = t_table + scan_time(8.0, t_step_g * (streamed ? 1.06 : 1.0 + cold_pen * cold));
                        }
                        if (g.blocked_variant == 5 && !fits) cost *= 1e-3;      // tests: the hybrid table wherever it is possible
                        if (std::getenv("IMC_DEBUG_LEVELS"))
                            std::fprintf(stderr, "[imc] level %d alphabet %d %s: model %.1f us\n", l, A,
                                         fits ? "LDS table" : two_per_cu ? "global table, 2 per CU" : "global table", cost);
                        if (cost < best) { best = cost; best_l = l; }
                        continue;
                    }
                    if (!(o0->alphabet[l] <= a_max && o0->alphabet[l] > o0->nsym && o0->d_tok[l])) continue;
                    if (big && (size_t)B * o0->alphabet[l] * kc->NP * kc->NP * 8 > CTAB_BUDGET) continue;   // table too large
                    double toks = 0.0;
                    for (int f : kv.second) toks += (double)chunks[f]->ntok[l];
                    const double n3 = (double)kc->NP * kc->NP * kc->NP;
                    double c_tab, c_main;   // cycles
                    if (big) {   // one GEMM per step per workgroup, table built by depth
                        const double gemm = gemm_cu_step_cycles(kc->NP);
                        c_tab = ((o0->alphabet[l] - S) / (double)g.cus + 12.0) * gemm;   // one workgroup per token, ~12 depth launches
                        c_main = std::max(16.0, toks * B / (double)g.cus) * gemm;
                        double lmax = 0.0;   // or the mat-vec chain kernel, when no chunk needs splitting
                        for (int f : kv.second) lmax = std::max(lmax, (double)chunks[f]->ntok[l]);
                        const double np2 = (double)kc->NP * kc->NP;
                        const double c_vec = std::max(lmax * (np2 / 4.5 + 1500.0), toks * B / (double)g.cus * matvec_step_cycles(np2, o0->alphabet[l]));
                        if (!op_mode && (g.kernel_pref == 1 || (g.kernel_pref == 0 && c_vec < c_main))) c_main = c_vec;
                    } else {     // ~5200 cycles per row-step at N=20; every workgroup rebuilds the table
                        c_tab = (o0->alphabet[l] - S) * (400.0 + n3 / 64.0);
                        c_main = std::max(16.0, toks * B / ((double)g.cus * 32.0)) * 0.65 * n3;
                    }
                    if (c_tab + c_main < best) { best = c_tab + c_main; best_l = l; }
                }
                if (const char *fl = std::getenv("IMC_FORCE_LEVEL")) {   // experiments only: pin the dictionary level index
                    const int l = std::atoi(fl);
                    const imc_obs *o0 = chunks[kv.second[0]];
                    const bool hybrid = mfma_blocked && kc->zip4 && g.blocked_variant >= 4 && !o0->wide_raw;
                    const int amax_forced = hybrid ? HYBRID_MAX_ALPHABET : a_max;
                    if (l >= 0 && l < imc::kNumLevels && o0->alphabet[l] <= amax_forced && !(hybrid && kc->blocked_lds(o0->alphabet[l]) > LDS_BUDGET && kc->zip4_lds(o0->alphabet[l], 0) > LDS_BUDGET) && o0->alphabet[l] > o0->nsym && o0->d_tok[l] &&
                        (hybrid ? !o0->tok_count[l].empty() : !o0->wide[l])) best_l = l;
                }
                dict_level[kv.first] = best_l;
            }
        }
        chunk_group.assign(n_chunks, -1);
        for (int f = 0; f < n_chunks; ++f) {
            const imc_obs *o = chunks[f];
            int level = -1;
            if (g.compression && o->dict && o->nsym == S) {
                auto it = dict_level.find(o->dict.get());
                if (it != dict_level.end()) level = it->second;
            }
            int gi = -1;
            for (size_t q = 0; q < p->groups.size(); ++q) {
                Group &gr = p->groups[q];
                if (level < 0 ? !gr.zip : (gr.zip && gr.level == level && gr.dict == o->dict)) gi = (int)q;
            }
            if (gi < 0) {
                Group gr;
                gr.zip = level >= 0;
                gr.level = level;
                gr.big = big;
                gr.A = S;
                if (gr.zip) { gr.dict = o->dict; gr.A = o->alphabet[level]; }
                // an alphabet beyond LDS can only have been chosen for the hybrid-table kernel
                gr.zip4 = gr.zip && mfma_blocked && (kc->blocked_lds(gr.A) > LDS_BUDGET || o->wide[level]);
                gr.wide_tokens = gr.zip && o->wide[level];
                p->groups.push_back(gr);
                gi = (int)p->groups.size() - 1;
            }
            p->groups[gi].chunks.push_back(f);
            chunk_group[f] = gi;
        }
    }

    void choose_segment_lengths()
    {
        // ---- segment length per group ----
        for (Group &gr : p->groups) {
            std::vector<size_t> lens;
            for (int f : gr.chunks) lens.push_back(gr.zip ? chunks[f]->ntok[gr.level] : chunks[f]->L);
            if (gr.big) {
                // every segment costs N^3 per step whatever the split (one workgroup = one CU's worth of LDS), so
                // one equal-length segment per CU is both balanced and the fewest operators for the stitch
                size_t total = 0;
                for (size_t L : lens) total += L;
                const size_t per_cu = std::max<size_t>(1, std::min<size_t>(LDS_BUDGET / kc->big_lds, (size_t)32 / (size_t)kc->G));
                const size_t target = std::max<size_t>(1, (size_t)g.cus * per_cu / ((size_t)B * kc->big_nslab));
                gr.seglen = std::max<size_t>(16, round_up((total + target - 1) / target, 16));
                // Chunks are cut one by one (ceil(L / seglen) segments each): with several chunks the sum can exceed the
                // target by a few segments - and one segment more than the machine holds is a second ROUND of workgroups,
                // twice the time (measured round 3 at 150 states, 3 x 3e6 columns: 129 segments for 128 slots, 201 ms
                // against 103).  Lengthen the segments until the launch fits.
                {
                    auto count = [&](size_t sl) { size_t c = 0; for (size_t L : lens) c += (L + sl - 1) / sl; return c; };
                    size_t nonempty = 0;
                    for (size_t L : lens) nonempty += L > 0;
                    if (nonempty <= target)            // (more chunks than slots: rounds are unavoidable)
                        for (int it = 0; it < 65536 && count(gr.seglen) > target; ++it) gr.seglen += 16;
                }
                // ... unless there are so many (chunk, parameter set) chains that no chunk needs splitting: then
                // the mat-vec chain kernel streams NP^2 doubles per step instead of a GEMM (measured at N=150, 64 x 32
                // chains: 8900 cycles per step per CU, i.e. ~12 TB/s of operator reads over the whole chip)
                size_t lmax = 0;
                for (size_t L : lens) lmax = std::max(lmax, L);
                const double np2 = (double)kc->NP * kc->NP, per_cu_steps = (double)total * B / (double)g.cus;
                const double cost_vec = std::max((double)lmax * (np2 / 4.5 + 1500.0), per_cu_steps * matvec_step_cycles(np2, gr.A));
                double cost_gemm = std::max(16.0, per_cu_steps) * gemm_cu_step_cycles(kc->NP);
                if (g.rank1_handoff && !g.seg_override && gr.seglen >= R1_MIN_SEGLEN) {   // GEMM heads + mat-vec tails
                    const HandoffEstimate he = estimate_handoff((double)gr.seglen, std::max(1.0, (double)total / gr.seglen), B,
                                                                (double)handoff_head(gr), kc->NP, kc->big_nslab, gr.A, g.cus);
                    if (he.m) cost_gemm = std::min(cost_gemm, he.cycles);
                }
                if (std::getenv("IMC_DEBUG"))
                    std::fprintf(stderr, "[imc] plan: GEMM chain seg %zu cost %.3g cycles; mat-vec chain cost %.3g cycles\n",
                                 gr.seglen, cost_gemm, cost_vec);
                // (no chunk longer than a segment: there would be no operator segments anyway)
                if (!op_mode && (g.kernel_pref == 1 || (g.kernel_pref == 0 && (cost_vec < cost_gemm || lmax <= gr.seglen)))) {
                    gr.bigvec = true;
                    gr.seglen = std::max<size_t>(16, round_up(lmax, 16));
                }
            } else {
                // vector kernel (one vector per lane group) ...
                double cost_vec = 0.0;
                size_t seg_vec;
                if (gr.zip) {
                    // LDS-bound: a workgroup of ZWAVES wavefronts serialises on one CU's LDS
                    const double lds_cycles = ((double)kc->R * kc->NP / 2 + kc->NP / 2.0) * 4.0 + kc->R * 6.0 + 40.0;
                    seg_vec = choose_seglen(lens, N, B, kc->VPW, (double)g.cus * ZWAVES, lds_cycles * ZWAVES, &cost_vec);   // measured: 4600 cycles per wavefront-step at N=20
                } else {
                    // VALU-bound, minw wavefronts share a SIMD: measured 555 cycles per wavefront-column per SIMD at N=20
                    seg_vec = choose_seglen(lens, N, B, kc->VPW, (double)g.cus * 4.0 * kc->minw,
                                            kc->minw * 1.7 * (4.0 * kc->R * kc->NP + 16.0), &cost_vec);
                }
                gr.seglen = seg_vec;
                // ... or the register-blocked kernel (one operator per 16-lane row): fill every row of the machine
                // once; first segments waste 1 - 1/N of their row, which the cost comparison accounts for
                if (kc->zip2 && g.kernel_pref != 1 && (kc->blocked_lds(gr.A) <= LDS_BUDGET || gr.zip4) && (gr.zip || (S == gr.A && S <= imc::kByteAlphabet))) {
                    size_t total = 0;
                    for (size_t L : lens) total += L;
                    // Up to 12 states the global-table kernel needs 124 registers and no LDS to speak of: TWO workgroups share a
                    // CU (four wavefronts per SIMD), i.e. the machine has twice the rows, each step taking ~1.85x as long
                    // (measured at 10 states, 100 x 1e6 columns: 500 workgroups of 44-token segments 102 us, 200 workgroups
                    // of 104-token segments 113 us).
                    const bool two_per_cu = gr.zip4 && kc->use3() && kc->NP <= 12 &&
                                            z4_streamed((double)B * (gr.A + 1) * kc->tok_doubles * 8.0, B);
                    const double rows = (double)g.cus * Z2WAVES * 4 * (two_per_cu ? 2.0 : 1.0);
                    const double rb = kc->NP / 4.0;
                    // per wavefront-step (4 segments): DPP/VALU form measured ~2900 at N=20; the MFMA form issues
                    // (NP/4)^3 v_mfma_f64_4x4x4 at ~25.5 cycles each with two wavefronts per SIMD and nothing else
                    const double step_cycles = (kc->use3() ? 25.5 * rb * rb * rb * 0.55 : 5.8 * rb * rb * kc->NP) * (two_per_cu ? 1.85 : 1.0);
                    size_t seg_blk = 16;
                    double slots = 0.0, cost_blk = 1e300;
                    // a workgroup takes 32 consecutive segments of ONE chunk: rows are allocated per chunk in 32s
                    // (... except chunks that are ONE segment: those are packed, a row each - make_units)
                    const bool can_pack = kc->use3() && !op_mode;
                    auto rows_used = [&](size_t sg) {
                        double r = 0.0, singles = 0.0;
                        for (size_t L : lens) {
                            if (!L) continue;
                            const double nseg = std::ceil((double)L / (double)sg);
                            if (nseg <= 1.0 && can_pack) singles += 1.0;
                            else r += std::ceil(nseg / Z2SLOTS) * Z2SLOTS;
                        }
                        return r + std::ceil(singles / Z2SLOTS) * Z2SLOTS;
                    };
                    // Candidates: fill the machine's rows `rounds` times - or, when chunks x parameter sets alone need more
                    // rounds than that (every chunk takes at least one workgroup of 32 rows per parameter set: 100 chunks x 64
                    // sets = 25 rounds), cut every chunk into u workgroups' worth of segments.  (Round 2 only had the first
                    // family, up to 16 rounds; beyond that its search ran into its iteration limit and returned segments of
                    // thousands of tokens with two of a workgroup's 32 rows in use: 100 x 1e6 columns x 64 proposals took
                    // 446 us per proposal at 10 states against 64 us at 8 proposals.)
                    size_t lmax = 0;
                    for (size_t L : lens) lmax = std::max(lmax, L);
                    std::vector<size_t> cands;
                    for (int rounds = 1; rounds <= 16; ++rounds) {
                        const double target = std::max((double)Z2SLOTS, std::floor(rows * rounds / B));
                        // (the blocked kernels take segments of any multiple of 4 tokens - dword-aligned 16-byte
                        // loads - so the machine's rows can be filled to within a per cent, not to within 16 tokens)
                        size_t sg = std::max<size_t>(16, round_up((size_t)std::ceil((double)total / target), Z2GRAN));
                        int it = 0;
                        for (; it < 1024 && rows_used(sg) > target && sg < lmax + 16; ++it) sg += Z2GRAN;
                        if (rows_used(sg) > target) continue;                    // this many rounds cannot hold the launch
                        cands.push_back(sg);
                    }
                    for (size_t u = 1; u <= 8; ++u)
                        cands.push_back(std::max<size_t>(16, round_up((lmax + Z2SLOTS * u - 1) / (Z2SLOTS * u), Z2GRAN)));
                    // ... and, for mixes of one long chunk with many short ones, the same for the MEDIAN chunk plus a few fixed
                    // short lengths: both families above follow the longest chunk, and when the short chunks alone need more
                    // than 16 rounds (each takes a workgroup per parameter set whatever the segment length) every candidate
                    // was hundreds of tokens long - 500 chunks of 1e5 columns beside one of 2.5e7, 8 sets, 20 states: 432-token
                    // segments, two of a workgroup's 32 rows in use on 4000 workgroups, 5.7 ms against 2.5.
                    {
                        std::vector<size_t> sorted(lens);
                        std::sort(sorted.begin(), sorted.end());
                        const size_t med = sorted[sorted.size() / 2];
                        for (size_t u = 1; u <= 4; ++u)
                            cands.push_back(std::max<size_t>(16, round_up((med + Z2SLOTS * u - 1) / (Z2SLOTS * u), Z2GRAN)));
                        for (size_t fixed_len : {16, 32, 48, 64, 96, 128, 192}) cands.push_back(fixed_len);
                        // whole chunks as single segments (packed 32 to a workgroup): the length of the median, the 90 % and the
                        // longest chunk
                        if (can_pack)
                            for (size_t qi : {sorted.size() / 2, sorted.size() * 9 / 10, sorted.size() - 1})
                                cands.push_back(std::max<size_t>(16, round_up(sorted[qi], Z2GRAN)));
                    }
                    for (size_t sg : cands) {
                        // per round of workgroups: the main loop, plus the table rebuild and the 5-level in-kernel fold
                        // (table: the VALU form builds token by token; the MFMA form one dictionary depth per pass,
                        // ~10 depths; with the hybrid table a workgroup only copies its hot set from L2)
                        const double table = !kc->use3() ? (double)(gr.A - S) * (400.0 + (double)kc->NP * kc->NP * kc->NP / 64.0)
                                             : gr.zip4 ? 9000.0 : 2600.0 * std::min(12.0, (double)(gr.A - S));
                        const double fixed = table + 5.0 * (Z2WAVES / 4.0) * step_cycles;
                        // A segment length that is not a multiple of 16 ends in a MASKED block (the pipeline is re-primed per
                        // run of it: ~3.5 us; measured at 10 states, 100 x 1e6 columns: 48-token segments 101.9 us, 44-token
                        // ones 109.6) - so the next multiple of 16 is priced beside the exact fit.
                        for (size_t cand : {sg, round_up(sg, 16)}) {
                            const double u = rows_used(cand);
                            double c = std::ceil(u * B / rows) * ((double)cand * (Z2WAVES / 4.0) * step_cycles + fixed + (cand % 16 ? 7700.0 : 0.0));
                            // (every chunk a single packed segment: measured 10 % behind the best split where the model has a tie)
                            if (can_pack && cand >= lmax) c *= 1.15;
                            if (c < cost_blk) { cost_blk = c; seg_blk = cand; slots = u; }
                        }
                    }
                    if (std::getenv("IMC_DEBUG"))
                        std::fprintf(stderr, "[imc] plan: vector kernel seg %zu cost %.3g cycles; blocked kernel seg %zu slots %.0f cost %.3g cycles\n",
                                     seg_vec, cost_vec, seg_blk, slots, cost_blk);
                    const bool vec_fits = !gr.zip || kc->zip_lds(gr.A) <= LDS_BUDGET;   // the table may only fit the blocked kernel
                    if (g.kernel_pref == 2 || cost_blk < cost_vec || !vec_fits || gr.zip4) { gr.zip2 = true; gr.seglen = seg_blk; }
                }
            }
            if (g.seg_override) gr.seglen = round_up(std::max<size_t>(g.seg_override, 16), 16);   // tests: force stitching
            if (gr.bigvec)
                for (size_t L : lens) gr.bigvec = gr.bigvec && L <= gr.seglen && !op_mode;
            // rank-one hand-off: worth a test once segments are much longer than the HMM's memory (tens of thousands
            // of columns); a quarter of the segment on the GEMM chain, then the certified test
            if (gr.big && !gr.bigvec && g.rank1_handoff && gr.seglen >= R1_MIN_SEGLEN) {
                // head: ~R1_HEAD_COLUMNS alignment columns on the GEMM chain (the HMM's memory is a property of the
                // model, not of the segmentation).  The tails then cost a mat-vec per step, so MORE, shorter
                // segments pay: m times the segments = m rounds of heads, tails 1/m as long.  Pick m by the model;
                // if the test fails at run time the GEMM chain does the same total work as with m = 1.
                double cols = 0.0, toks = 0.0;
                for (int f : gr.chunks) { cols += (double)chunks[f]->L; toks += (double)(gr.zip ? chunks[f]->ntok[gr.level] : chunks[f]->L); }
                const double span = toks > 0.0 ? cols / toks : 1.0;
                const size_t head = handoff_head(gr);
                const double nseg = std::max(1.0, toks / (double)gr.seglen);
                if (g.seg_override) {                 // tests: one checkpoint at a quarter of the forced segment length
                    if (gr.seglen >= 4 * R1_MIN_HEAD) {
                        gr.rank1 = true;
                        gr.head_len = (int)round_up(gr.seglen / 4, 16);
                        gr.checkpoints.assign(1, gr.head_len);
                    }
                } else {
                    const int best_m = estimate_handoff((double)gr.seglen, nseg, B, (double)head, kc->NP, kc->big_nslab, gr.A, g.cus).m;
                    if (best_m) {
                        gr.rank1 = true;
                        gr.head_len = (int)head;
                        gr.seglen = std::max<size_t>(16, round_up(gr.seglen / best_m, 16));
                        // checkpoints: from ~8k alignment columns, each ~1.125x the previous (multiples of 16 tokens)
                        // up to ~100k columns - where the operators of the measured models collapse; a failing check
                        // is a quick reject and costs two near-empty launches - then 1.5x, while at least an eighth
                        // of the segment would still be left for the mat-vec chain
                        size_t c = round_up(std::max<size_t>(64, (size_t)(R1_HEAD_MIN_COLUMNS / span)), 16);
                        while ((int)gr.checkpoints.size() < R1_MAX_ROUNDS && c + gr.seglen / 8 < gr.seglen) {
                            gr.checkpoints.push_back((int)c);
                            const size_t step = (double)c * span < R1_DENSE_COLUMNS ? c / 8 : c / 2;
                            c = round_up(c + std::max<size_t>(16, step), 16);
                        }
                        if (gr.checkpoints.empty()) gr.rank1 = false;
                    }
                }
            }
        }
    }

    void cut_segments()
    {
        // ---- segments in chunk order ----
        chunk_seg.assign(n_chunks + 1, 0);
        for (int f = 0; f < n_chunks; ++f) {
            chunk_seg[f] = (uint32_t)segs.size();
            const Group &gr = p->groups[chunk_group[f]];
            const size_t L = gr.zip ? chunks[f]->ntok[gr.level] : chunks[f]->L;
            const uint8_t *base = gr.zip ? chunks[f]->d_tok[gr.level] : chunks[f]->d_sym;
            if (!L) continue;
            const size_t K0 = (L + gr.seglen - 1) / gr.seglen;
            const size_t sl = round_up((L + K0 - 1) / K0, gr.zip2 ? Z2GRAN : 16);   // equalised; a multiple of 16 (16-byte aligned
                                                                                   // loads) except for the blocked kernels
            for (size_t off = 0, k = 0; off < L; off += sl, ++k) {
                const bool wide = gr.zip ? chunks[f]->wide[gr.level] : chunks[f]->wide_raw;
                const bool fst = k == 0 && !op_mode;   // operator mode: the chunk's own first segment is an operator too
                segs.push_back(SegDesc{base + off * (wide ? 2 : 1), (uint32_t)std::min(sl, L - off),
                                       (fst ? SEG_FIRST : 0u) | (wide ? SEG_WIDE : 0u)});
                seg_first.push_back(fst);
            }
        }
        chunk_seg[n_chunks] = (uint32_t)segs.size();
    }

    int make_units()
    {
        // ---- stitch units and their vectors, contiguous per group ----
        // A unit is what the stitch hierarchy sees at level 0: a segment, or (blocked kernel) a workgroup's
        // run of up to 32 consecutive segments of one chunk, already folded inside the kernel.
        chunk_units.assign(n_chunks, {});   // per chunk: (seg0, nsegs)
        for (int f = 0; f < n_chunks; ++f) {
            const Group &gr = p->groups[chunk_group[f]];
            const uint32_t step = gr.zip2 ? (uint32_t)Z2SLOTS : 1u;
            for (uint32_t sid = chunk_seg[f]; sid < chunk_seg[f + 1]; sid += step)
                chunk_units[f].push_back({sid, std::min(step, chunk_seg[f + 1] - sid)});
        }
        chunk_unit.assign(n_chunks + 1, 0);
        for (int f = 0; f < n_chunks; ++f) chunk_unit[f + 1] = chunk_unit[f] + (uint32_t)chunk_units[f].size();
        unit_vec0.assign(chunk_unit[n_chunks], 0);
        unit_first.assign(chunk_unit[n_chunks], 0);
        for (Group &gr : p->groups) {
            gr.vec_begin = (uint32_t)vecs.size();
            for (int f : gr.chunks) {
                gr.stream_len += gr.zip ? chunks[f]->ntok[gr.level] : chunks[f]->L;
                for (size_t u = 0; u < chunk_units[f].size(); ++u) {
                    const uint32_t sid = chunk_units[f][u].first, ns = chunk_units[f][u].second;
                    const uint32_t uid = chunk_unit[f] + (uint32_t)u;
                    const bool fst = seg_first[sid] != 0;
                    unit_first[uid] = fst;
                    unit_vec0[uid] = (uint32_t)vecs.size();
                    const int nv = fst ? 1 : N;
                    for (int c = 0; c < nv; ++c) vecs.push_back(VecDesc{sid, (uint32_t)c});
                    if (gr.big) { gr.seg_ids.push_back(sid); gr.seg_out.push_back(unit_vec0[uid]); }
                    if (gr.zip2) {
                        // A chunk that is ONE segment needs no fold: up to 32 consecutive such chunks share a workgroup, each
                        // in a slot of its own (Z2Block::first == 2).  One workgroup per chunk left 31 of 32 rows idle when
                        // the chunks are short (10000 chunks of 1e4 columns, 20 states: 39 rounds of workgroups, 1.97 ms).
                        const bool whole = kc->use3() && fst && ns == 1 && chunk_units[f].size() == 1;
                        if (whole && !gr.blocks.empty() && gr.blocks.back().first == 2 && gr.blocks.back().n < Z2SLOTS &&
                            gr.blocks.back().seg0 + gr.blocks.back().n == sid && gr.blocks.back().out_vec0 + gr.blocks.back().n == unit_vec0[uid])
                            ++gr.blocks.back().n;
                        else
                            gr.blocks.push_back(Z2Block{sid, ns, unit_vec0[uid], whole ? 2u : fst ? 1u : 0u});
                    }
                    for (uint32_t q2 = 0; q2 < ns; ++q2)
                        gr.vsteps += (uint64_t)((seg_first[sid + q2] && !gr.zip2) ? 1 : N) * segs[sid + q2].len;
                }
            }
            gr.n_vecs = (uint32_t)vecs.size() - gr.vec_begin;
        }
        // fused tail: one blocked-MFMA group holds every chunk, no chunk is empty or longer than Z2SLOTS workgroups
        if (p->groups.size() == 1 && p->groups[0].zip2 && kc->use3() && !op_mode && n_chunks > 0) {
            Group &gr = p->groups[0];
            bool ok = true;
            size_t most = 0;
            for (int f = 0; f < n_chunks; ++f) {
                ok = ok && !chunk_units[f].empty() && chunk_units[f].size() <= (size_t)Z2SLOTS;
                most = std::max(most, chunk_units[f].size());
            }
            if (ok) {
                for (int f : gr.chunks)
                    for (size_t u = 0; u < chunk_units[f].size(); ++u)
                        gr.tails.push_back(Z2Tail{(uint32_t)f, (uint32_t)u, (uint32_t)chunk_units[f].size(), 0u});
                gr.tail_stride = (int)most;
                if (gr.tails.size() != gr.blocks.size()) gr.tails.clear();      // packed blocks: several chunks share a workgroup
                else
                    for (Z2Block &bk : gr.blocks)                               // (a packed block of ONE chunk is an ordinary first block,
                        if (bk.first == 2) bk.first = 1;                        //  and the fused tail writes that chunk's result)
            }
        }
        if (vecs.size() >= (size_t)UINT32_MAX / 2) return fail(IMC_ERR_ARG, "too many vectors in one call");
        p->n_segs = (uint32_t)segs.size();
        p->n_vecs = (uint32_t)vecs.size();
        p->pstride = round_up((size_t)kc->NP + (size_t)kc->NP * kc->NP + (size_t)S * kc->NP, 2);

        return IMC_OK;
    }

    void make_hierarchy()
    {
        // ---- stitch hierarchy: fold runs of g ~ sqrt(K) consecutive segments until one vector per chunk ----
        hl.assign(1, HostLevel());
        hl[0].chunk_seg = chunk_unit; hl[0].vec0 = unit_vec0; hl[0].first = unit_first; hl[0].n_vecs = p->n_vecs;
        for (;;) {
            const HostLevel &cur = hl.back();
            uint32_t kmax = 0;
            for (int f = 0; f < n_chunks; ++f) kmax = std::max(kmax, cur.chunk_seg[f + 1] - cur.chunk_seg[f]);
            // a level whose chunks all hold one first-vector is final; level 0 operators always need one pass
            if (kmax <= 1 && hl.size() > 1) break;
            if (kmax == 0) break;
            // ~3 levels: per level a chain costs g serial steps (~0.3 us each) plus two launches (~6 us)
            const uint32_t gsz = kmax <= 16 ? kmax : std::max<uint32_t>(12, (uint32_t)std::ceil(std::cbrt((double)kmax)));
            p->chain_steps += gsz;
            HostLevel nx;
            nx.chunk_seg.assign(n_chunks + 1, 0);
            nx.n_vecs = 0;
            for (int f = 0; f < n_chunks; ++f) {
                nx.chunk_seg[f] = (uint32_t)nx.vec0.size();
                const uint32_t s0 = cur.chunk_seg[f], s1 = cur.chunk_seg[f + 1];
                for (uint32_t rb = s0; rb < s1; rb += gsz) {
                    const uint32_t re = std::min(rb + gsz, s1);
                    const bool fst = rb == s0 && !op_mode;
                    nx.vec0.push_back(nx.n_vecs);
                    nx.first.push_back(fst);
                    const int nvv = fst ? 1 : N;
                    for (int c = 0; c < nvv; ++c) nx.chains.push_back(ChainDesc{rb, re, (uint32_t)c, nx.n_vecs + c, fst ? 1u : 0u, 0u});
                    nx.n_vecs += nvv;
                }
            }
            nx.chunk_seg[n_chunks] = (uint32_t)nx.vec0.size();
            const bool done = kmax <= gsz;
            hl.push_back(std::move(nx));
            if (done) break;
        }
        final_vec.assign(std::max(n_chunks, 1), -1);
        bool all_from_chains = hl.size() > 1 && !op_mode && n_chunks > 0;
        for (int f = 0; f < n_chunks; ++f) {
            const HostLevel &last = hl.back();
            if (last.chunk_seg[f + 1] > last.chunk_seg[f]) final_vec[f] = (int32_t)last.vec0[last.chunk_seg[f]];
            else all_from_chains = false;               // an empty chunk: k_finish writes its 0.0
        }
        // every chunk's final vector comes out of a chain of the last level: those chains write the log-likelihoods
        // themselves and k_finish is not launched
        p->finish_fused = false;
        if (all_from_chains) {
            HostLevel &last = hl.back();
            std::vector<char> seen((size_t)n_chunks, 0);
            for (ChainDesc &cd : last.chains)
                for (int f = 0; f < n_chunks; ++f)
                    if (cd.first && final_vec[f] == (int32_t)cd.out_vec) { cd.chunk1 = (uint32_t)f + 1; seen[f] = 1; }
            p->finish_fused = std::all_of(seen.begin(), seen.end(), [](char c) { return c != 0; });
            if (!p->finish_fused)
                for (ChainDesc &cd : last.chains) cd.chunk1 = 0;
        }

    }

    int upload(Plan **out)
    {
        while (g_plans.size() >= MAX_PLANS) {            // least recently used first; a plan somebody waits on stays
            auto victim = g_plans.end();
            for (auto it = g_plans.begin(); it != g_plans.end(); ++it)
                if (!(*it)->busy) victim = it;
            if (victim == g_plans.end()) break;
            (*victim)->release();
            g_plans.erase(victim);
        }

        Plan *q = p.get();
        auto up = [&](void **d, const void *h, size_t bytes) -> hipError_t {
            hipError_t e = dev_alloc(d, std::max<size_t>(bytes, 16));
            if (e != hipSuccess) return e;
            if (bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
            return e;
        };
        auto zalloc = [&](void **d, size_t bytes) -> hipError_t {
            hipError_t e = dev_alloc(d, std::max<size_t>(bytes, 16));
            if (e == hipSuccess) e = hipMemset(*d, 0, std::max<size_t>(bytes, 16));   // padded operator columns stay 0
            return e;
        };
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = up((void **)&q->d_segs, segs.data(), segs.size() * sizeof(SegDesc));
        if (e == hipSuccess) e = up((void **)&q->d_vecs, vecs.data(), vecs.size() * sizeof(VecDesc));
        if (e == hipSuccess) e = up((void **)&q->d_final_vec, final_vec.data(), final_vec.size() * 4);
        q->levels.resize(hl.size());
        for (size_t l = 0; l < hl.size() && e == hipSuccess; ++l) {
            Level &lv = q->levels[l];
            lv.n_segs = (uint32_t)hl[l].vec0.size();
            lv.n_vecs = hl[l].n_vecs;
            lv.n_chains = (uint32_t)hl[l].chains.size();
            const size_t nv = std::max<size_t>(lv.n_vecs, 1), ns = std::max<size_t>(lv.n_segs, 1);
            e = up((void **)&lv.d_vec0, hl[l].vec0.data(), hl[l].vec0.size() * 4);
            if (e == hipSuccess) e = up((void **)&lv.d_first, hl[l].first.data(), hl[l].first.size());
            if (e == hipSuccess) e = up((void **)&lv.d_chains, hl[l].chains.data(), hl[l].chains.size() * sizeof(ChainDesc));
            if (e == hipSuccess) e = zalloc((void **)&lv.d_P, (size_t)B * nv * kc->NP * 8);
            if (e == hipSuccess) e = zalloc((void **)&lv.d_EX, (size_t)B * nv * 4);
            if (e == hipSuccess) e = zalloc((void **)&lv.d_EMAX, (size_t)B * ns * 4);
        }
        for (Group &gr : q->groups) {
            if (gr.zip2 && e == hipSuccess) e = up((void **)&gr.d_blocks, gr.blocks.data(), gr.blocks.size() * sizeof(Z2Block));
            if (!gr.tails.empty() && e == hipSuccess) {
                const size_t slots = (size_t)B * n_chunks * gr.tail_stride;
                e = up((void **)&gr.d_tails, gr.tails.data(), gr.tails.size() * sizeof(Z2Tail));
                if (e == hipSuccess) e = dev_alloc((void **)&gr.d_tailX, slots * kc->tok_doubles * 8);
                if (e == hipSuccess) e = dev_alloc((void **)&gr.d_tailE, slots * 128);
                if (e == hipSuccess) e = zalloc((void **)&gr.d_tail_arrive, (size_t)B * n_chunks * 4);
            }
            if (gr.zip4 && e == hipSuccess) {
                // hot set: the most frequent tokens of this group's chunks, as many as LDS holds beside the identity
                std::vector<uint64_t> cnt((size_t)gr.A, 0);
                for (int f : gr.chunks)
                    for (size_t z = 0; z < chunks[f]->tok_count[gr.level].size() && z < cnt.size(); ++z) cnt[z] += chunks[f]->tok_count[gr.level][z];
                std::vector<uint16_t> ids((size_t)gr.A);
                for (int z = 0; z < gr.A; ++z) ids[z] = (uint16_t)z;
                std::stable_sort(ids.begin(), ids.end(), [&](uint16_t x, uint16_t y) { return cnt[x] > cnt[y]; });
                const double tables = (double)B * (gr.A + 1) * kc->tok_doubles * 8.0;
                gr.stream_table = z4_streamed(tables, B);
                gr.n_hot = gr.stream_table ? 0 : std::min(gr.A, kc->zip4_max_hot(gr.A, LDS_BUDGET));
                gr.hot.assign(ids.begin(), ids.begin() + gr.n_hot);
                e = up((void **)&gr.d_hot, gr.hot.data(), gr.hot.size() * sizeof(uint16_t));
                if (e == hipSuccess) e = dev_alloc((void **)&gr.d_Ctab, (size_t)B * (gr.A + 1) * kc->tok_doubles * 8);
                if (e == hipSuccess) e = dev_alloc((void **)&gr.d_cex, (size_t)B * (gr.A + 1) * 4 + 16);
            }
            if (gr.zip2 && gr.zip && e == hipSuccess) {
                // merged tokens of this level's alphabet (ids S .. A-1) grouped by dictionary depth, for the table build
                const DictDev &dd = *gr.dict;
                std::vector<uint16_t> order;
                std::vector<int> lvl(1, 0);
                int cur = -1;
                for (uint16_t z : dd.order) {                       // (sorted by depth, then id)
                    if ((int)z >= gr.A) continue;
                    if (dd.depth[z] != cur) {
                        if (cur >= 0) lvl.push_back((int)order.size());
                        cur = dd.depth[z];
                    }
                    order.push_back(z);
                }
                lvl.push_back((int)order.size());
                gr.tab_nlvl = order.empty() ? 0 : (int)lvl.size() - 1;
                gr.tab_lvl = lvl;
                e = up((void **)&gr.d_tab_order, order.data(), order.size() * sizeof(uint16_t));
                if (e == hipSuccess) e = up((void **)&gr.d_tab_lvl, lvl.data(), lvl.size() * sizeof(int));
                if (e == hipSuccess && gr.zip4) {
                    std::vector<int4> desc;
                    for (uint16_t z : order) desc.push_back(make_int4((int)z, (int)dd.dict.left[z], (int)dd.dict.right[z], 0));
                    e = up((void **)&gr.d_tab_desc, desc.data(), desc.size() * sizeof(int4));
                    // two depths per launch (k_z4_level2): launch k builds depths 2k+1 and 2k+2; a second-depth token
                    // whose child sits in the first depth recomputes it from the grandchildren.  Entries of a launch:
                    // the first depth's, then the second depth's grouped by which children they recompute, every group
                    // padded to whole wavefronts (four entries) with idle entries (token -1).
                    std::vector<int4> d2;
                    gr.tab2.clear();
                    auto pad4 = [&]() { while ((d2.size() / 2) % 4) { d2.push_back(make_int4(-1, 0, 0, 0)); d2.push_back(make_int4(0, 0, 0, 0)); } };
                    for (int d = 0; d < gr.tab_nlvl; d += 2) {
                        const int first = (int)(d2.size() / 2);
                        for (int k = lvl[d]; k < lvl[d + 1]; ++k) {
                            const int z = order[k];
                            d2.push_back(make_int4(z, (int)dd.dict.left[z], (int)dd.dict.right[z], 0));
                            d2.push_back(make_int4(0, 0, 0, 0));
                        }
                        pad4();
                        if (d + 1 < gr.tab_nlvl) {
                            const int d_first = dd.depth[order[lvl[d]]];
                            for (int flags = 1; flags <= 3; ++flags) {
                                for (int k = lvl[d + 1]; k < lvl[d + 2]; ++k) {
                                    const int z = order[k], zl = dd.dict.left[z], zr = dd.dict.right[z];
                                    const bool nl = zl >= S && dd.depth[zl] == d_first, nr = zr >= S && dd.depth[zr] == d_first;
                                    if ((nl ? 1 : 0) + (nr ? 2 : 0) != flags) continue;
                                    d2.push_back(make_int4(z, zl, zr, flags));
                                    d2.push_back(make_int4(nl ? (int)dd.dict.left[zl] : 0, nl ? (int)dd.dict.right[zl] : 0,
                                                           nr ? (int)dd.dict.left[zr] : 0, nr ? (int)dd.dict.right[zr] : 0));
                                }
                                pad4();
                            }
                        }
                        gr.tab2.push_back({first, (int)(d2.size() / 2) - first});
                    }
                    if (e == hipSuccess && !d2.empty()) e = up((void **)&gr.d_tab_desc2, d2.data(), d2.size() * sizeof(int4));
                    // three depths per launch (k_z4_level3): launch k builds depths 3k+1 .. 3k+3, one wavefront per token;
                    // a token's eight leaves are the nodes of its dictionary tree that lie at depth <= 3k (already in the
                    // table), a ready node in the first leaf of its range and the identity in the rest of it
                    std::vector<int4> d3;
                    gr.tab3.clear();
                    for (int d = 0; d < gr.tab_nlvl; d += 3) {
                        const int first = (int)(d3.size() / 3);
                        const int d_ready = dd.depth[order[lvl[d]]] - 1;            // entries up to this depth exist
                        for (int k = lvl[d]; k < lvl[std::min(d + 3, gr.tab_nlvl)]; ++k) {
                            int leaves[8];
                            for (int &x : leaves) x = gr.A;                       // the identity entry
                            struct Fill {
                                const DictDev &dd; int S, d_ready; int *leaves;
                                void operator()(int t, int lo, int hi) const
                                {
                                    if (t < S || dd.depth[t] <= d_ready || hi - lo == 1) { leaves[lo] = t; return; }
                                    const int mid = (lo + hi) / 2;
                                    (*this)((int)dd.dict.left[t], lo, mid);
                                    (*this)((int)dd.dict.right[t], mid, hi);
                                }
                            } fill{dd, S, d_ready, leaves};
                            fill((int)order[k], 0, 8);
                            d3.push_back(make_int4((int)order[k], leaves[0], leaves[1], leaves[2]));
                            d3.push_back(make_int4(leaves[3], leaves[4], leaves[5], leaves[6]));
                            d3.push_back(make_int4(leaves[7], 0, 0, 0));
                        }
                        gr.tab3.push_back({first, (int)(d3.size() / 3) - first});
                    }
                    if (e == hipSuccess && !d3.empty()) e = up((void **)&gr.d_tab_desc3, d3.data(), d3.size() * sizeof(int4));
                }
            }
            if (!gr.big || e != hipSuccess) continue;
            const size_t np2 = (size_t)kc->NP * kc->NP;
            // workgroup list: slabs of one segment 8 ids apart (same XCD -> they share the operator rows in L2)
            {
                std::vector<BigBlock> lin;
                for (size_t i2 = 0; i2 < gr.seg_ids.size(); ++i2) {
                    const bool fst = seg_first[gr.seg_ids[i2]] != 0;
                    for (int sl = 0; sl < (fst ? 1 : kc->big_nslab); ++sl)
                        lin.push_back(BigBlock{gr.seg_ids[i2], (uint32_t)sl, gr.seg_out[i2], 0u});
                }
                // lin is segment-major; re-deal non-first segments in tiles of 8 segments x nslab
                gr.big_blocks.clear();
                std::vector<BigBlock> firsts, rest;
                for (const BigBlock &bb : lin) (seg_first[bb.seg] ? firsts : rest).push_back(bb);
                const size_t ns = (size_t)kc->big_nslab;
                for (size_t base = 0; base < rest.size(); base += 8 * ns) {
                    const size_t nseg = std::min<size_t>(8, (rest.size() - base) / ns);
                    for (size_t sl = 0; sl < ns; ++sl)
                        for (size_t k2 = 0; k2 < nseg; ++k2) gr.big_blocks.push_back(rest[base + k2 * ns + sl]);
                }
                for (const BigBlock &bb : firsts) gr.big_blocks.push_back(bb);
            }
            e = up((void **)&gr.d_big_blocks, gr.big_blocks.data(), gr.big_blocks.size() * sizeof(BigBlock));
            if (e == hipSuccess) e = dev_alloc((void **)&gr.d_Ctab, (size_t)B * gr.A * np2 * 8);
            if (e == hipSuccess && (gr.bigvec || gr.rank1) && N < kc->NP && g.pack_table)   // the mat-vec chain reads a packed copy
                e = dev_alloc((void **)&gr.d_Cpack, (size_t)B * gr.A * N * (size_t)(N + (N & 1)) * 8);
            if (e == hipSuccess) e = dev_alloc((void **)&gr.d_cex, (size_t)B * gr.A * 4 + 16);
            if (gr.rank1) {
                for (size_t i2 = 0; i2 < gr.seg_ids.size(); ++i2)
                {
                    gr.tail_blocks.push_back(BigBlock{gr.seg_ids[i2], 0u, gr.seg_out[i2], 0u});   // (first segments included)
                    if (!seg_first[gr.seg_ids[i2]]) gr.r1_segs.push_back({gr.seg_ids[i2], segs[gr.seg_ids[i2]].len});
                }
                const size_t nrec = (size_t)B * segs.size();
                if (e == hipSuccess) e = up((void **)&gr.d_tail_blocks, gr.tail_blocks.data(), gr.tail_blocks.size() * sizeof(BigBlock));
                if (e == hipSuccess) e = zalloc((void **)&gr.d_r1flag, nrec * 4);
                if (e == hipSuccess) e = zalloc((void **)&gr.d_r1at, nrec * 4);
                if (e == hipSuccess) e = dev_alloc((void **)&gr.d_r1u, std::max<size_t>(nrec * kc->NP * 8, 16));
                if (e == hipSuccess) e = dev_alloc((void **)&gr.d_r1alpha, std::max<size_t>(nrec * kc->NP * 8, 16));
            }
        }
        if (e == hipSuccess) e = dev_alloc((void **)&q->d_params, (size_t)B * q->pstride * 8);
        if (e == hipSuccess) e = dev_alloc((void **)&q->d_out, (size_t)B * std::max(n_chunks, 1) * 8);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipHostMalloc((void **)&q->h_params[k], (size_t)B * q->pstride * 8, hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&q->h_params_dev[k], q->h_params[k], 0);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&q->ev_params[k], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipHostMalloc((void **)&q->h_out, (size_t)B * std::max(n_chunks, 1) * 8, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&q->h_out_dev, q->h_out, 0);
        if (e != hipSuccess) {
            q->release();
            return fail(e == hipErrorOutOfMemory ? IMC_ERR_OOM : IMC_ERR_HIP,
                        std::string("plan allocation: ") + hipGetErrorString(e));
        }
        g_plans.push_front(std::move(p));
        *out = q;
        return IMC_OK;
    }
};

int build_plan(const imc_obs *const *chunks, int n_chunks, int N, int S, int B, bool op_mode, Plan **out)
{
    std::vector<uint64_t> key;
    key.reserve(n_chunks + 5);
    for (int f = 0; f < n_chunks; ++f) key.push_back(chunks[f]->id);
    key.push_back((uint64_t)N); key.push_back((uint64_t)S); key.push_back((uint64_t)B);
    key.push_back((uint64_t)g.seg_override); key.push_back((uint64_t)(g.compression * 4 + g.kernel_pref + (op_mode ? 64 : 0) + g.blocked_variant * 128 + (g.z4_stream + 1) * 1024));
    { const char *fl = std::getenv("IMC_FORCE_LEVEL"); key.push_back(fl ? (uint64_t)(std::atoi(fl) + 1) : 0u); }   // (experiments / tests)
    for (auto it = g_plans.begin(); it != g_plans.end(); ++it) {
        if ((*it)->key == key) {
            g_plans.splice(g_plans.begin(), g_plans, it);
            *out = g_plans.front().get();
            return IMC_OK;
        }
    }
    // 24 < N <= 64: the GEMM-chain kernels beat the vector kernels by 5-100x when chunks are long enough to be
    // cut into many operator segments; short-chunk / many-theta launches keep the vector kernels
    bool prefer_gemm = g.kernel_pref == 2;
    if (g.kernel_pref == 0 && N > 24 && N <= 64 && n_chunks > 0) {
        double total = 0.0;
        for (int f = 0; f < n_chunks; ++f) total += (double)chunks[f]->L;
        bool tokens = g.compression != 0;   // on the raw column stream the vector kernel still wins up to N=40 (no padding to 16s)
        for (int f = 0; f < n_chunks; ++f) tokens = tokens && chunks[f]->dict && chunks[f]->nsym == S;
        // (round 3: 1e6 columns in all and 5e3 per chunk on average - it used to be 2e5 per chunk, and 100 chunks of 1e5
        // columns then ran on the LDS-table vector kernels at 4-9x the GEMM chain's time, profiles/r03_e_calib_big.txt)
        prefer_gemm = total >= 1.0e6 && total / n_chunks >= 5.0e3 && (tokens || N > 40);
    }

    KernelChoice *kc = choose_kernel(N, prefer_gemm);
    if (!kc) return fail(IMC_ERR_ARG, "N exceeds the largest built kernel (" + std::to_string(IMC_MAX_N) + ")");
    if (kc->R == 0 && (kc->G == 7 || kc->G == 9) && kc + 1 < kChoices + sizeof(kChoices) / sizeof(kChoices[0])) {
        // The odd tile counts (NP = 112, 144) pad less, but run one workgroup per segment: their base segments are
        // half as long as the next shape's (two column slabs per segment), which on mid-sized inputs can fall below
        // what the rank-one hand-off needs.  Take the next shape when it can hand off and this one cannot.
        double toks = 0.0, cols = 0.0;
        for (int f = 0; f < n_chunks; ++f) {
            size_t nt = chunks[f]->L;
            if (g.compression && chunks[f]->dict && chunks[f]->nsym == S)
                for (int l = 0; l < imc::kNumLevels; ++l)
                    if (chunks[f]->d_tok[l]) nt = std::min(nt, chunks[f]->ntok[l]);
            toks += (double)nt;
            cols += (double)chunks[f]->L;
        }
        const double head = std::max<double>((double)R1_MIN_HEAD, R1_HEAD_COLUMNS / std::max(1.0, cols / std::max(1.0, toks)));
        auto eligible = [&](const KernelChoice *k) {
            const double per_cu = (double)std::max<size_t>(1, std::min<size_t>(LDS_BUDGET / k->big_lds, (size_t)32 / (size_t)k->G));
            const double target = std::max(1.0, (double)g.cus * per_cu / ((double)B * k->big_nslab));
            const double seglen = toks / target;
            return seglen >= (double)R1_MIN_SEGLEN && seglen >= 2.0 * head;
        };
        if (g.rank1_handoff && !g.seg_override && !eligible(kc) && eligible(kc + 1)) kc = kc + 1;
    }
    auto p = std::make_unique<Plan>();
    p->key = key; p->kc = kc; p->N = N; p->S = S; p->B = B; p->n_chunks = n_chunks;
    PlanBuilder pb;
    pb.chunks = chunks; pb.n_chunks = n_chunks; pb.N = N; pb.S = S; pb.B = B; pb.op_mode = op_mode; pb.kc = kc;
    pb.p = std::move(p);
    pb.assign_groups();
    pb.choose_segment_lengths();
    pb.cut_segments();
    if (int rc = pb.make_units()) return rc;
    pb.make_hierarchy();
    return pb.upload(out);
}

int check_args(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis,
               const double *Ts, const double *Es)
{
    if (n_chunks < 0 || (n_chunks > 0 && !chunks)) return fail(IMC_ERR_ARG, "chunks is null");
    if (B < 1) return fail(IMC_ERR_ARG, "B must be >= 1");
    if (N < 1) return fail(IMC_ERR_ARG, "N must be >= 1");
    if (S < 1 || S > imc::kMaxRawAlphabet) return fail(IMC_ERR_ARG, "S must be in [1," + std::to_string(imc::kMaxRawAlphabet) + "]");
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
// Pad the caller's parameters into the plan's pinned staging buffer (host work only).
int stage_params(Plan *p, const double *pis, const double *Ts, const double *Es, bool fixed_slot = false)
{
    KernelChoice *kc = p->kc;
    const int N = p->N, S = p->S, NP = kc->NP, B = p->B;
    p->slot = fixed_slot ? 0 : p->slot ^ 1;          // (a captured graph always copies from slot 0)
    if (p->ev_used[p->slot]) HIP_TRY(hipEventSynchronize(p->ev_params[p->slot]));
    for (int b = 0; b < B; ++b) {
        double *pp = p->h_params[p->slot] + (size_t)b * p->pstride;
        std::memset(pp, 0, p->pstride * 8);
        const double *pi = pis + (size_t)b * N, *T = Ts + (size_t)b * N * N, *E = Es + (size_t)b * N * S;
        for (int i = 0; i < N; ++i) pp[i] = pi[i];
        double *Tp = pp + NP;
        for (int j = 0; j < N; ++j) std::memcpy(Tp + (size_t)j * NP, T + (size_t)j * N, (size_t)N * 8);
        double *Et = pp + NP + (size_t)NP * NP;
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < N; ++i) Et[(size_t)s * NP + i] = E[(size_t)i * S + s];
    }
    return IMC_OK;
}

// Enqueue one batch evaluation on `stream`: parameter upload, propagate, stitch.  Per-chunk results are written
// to `out` ([B][n_chunks]; device memory or mapped pinned host memory).  Contains no synchronisation, so the
// same call sequence can be captured into a hipGraph.
int enqueue(Plan *p, hipStream_t stream, double *out, bool allow_tail = true)
{
    KernelChoice *kc = p->kc;
    const int N = p->N, S = p->S, NP = kc->NP, B = p->B;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (cap == hipStreamCaptureStatusNone && p->have_last && p->last_stream != stream)
        HIP_TRY(hipStreamWaitEvent(stream, p->ev_params[p->last_slot], 0));     // the plan's previous call ran on another stream
    // Parameter upload.  Small sets (every BASELINE shape but the 64-proposal batch at N = 150) are fetched by a kernel
    // from the mapped staging slot: a copy command costs its own ~3 us plus a ~10 us hand-over between the copy and the
    // first kernel (rocprofv3 kernel trace), a kernel in the same queue costs one launch.
    const size_t pbytes = (size_t)B * p->pstride * 8;
    // One streamed / hybrid-table group and nothing else (BASELINE config[1], the config[3] slice): the first table launch
    // fetches the parameters itself (k_z4_level2<., true>) - no k_stage_params, no k_z4_raw.
    bool fuse_head = false, direct_params = false;
    {
        int active = 0;
        const Group *only = nullptr;
        for (const Group &gr : p->groups)
            if (gr.n_vecs) { ++active; only = &gr; }
        const size_t head_lds = p->pstride * 8;
        fuse_head = g.fuse_head && active == 1 && only->zip4 && g.table_pairs && only->d_tab_desc2 && !only->tab2.empty() &&
                    kc->zip4_level2_first && pbytes <= STAGE_KERNEL_MAX_BYTES && head_lds <= 16 * 1024;
        // ... and a SMALL launch of the LDS-table MFMA kernel (the reference's own data sizes: one alignment of 1e5..1e6
        // columns): each of its few workgroups fetches the parameter set itself
        direct_params = g.fuse_head && active == 1 && only->zip2 && !only->zip4 && kc->use3() && pbytes <= STAGE_KERNEL_MAX_BYTES &&
                        only->blocks.size() * (size_t)B <= 32 && p->pstride * 8 <= 8192 &&
                        ((kc->blocked_lds(only->A) + 15) & ~(size_t)15) + p->pstride * 8 <= LDS_BUDGET;
    }
    if (fuse_head || direct_params) {
        // (nothing to enqueue here)
    } else if (pbytes <= STAGE_KERNEL_MAX_BYTES) {
        const unsigned n2 = (unsigned)(pbytes / 16);
        hipLaunchKernelGGL(k_stage_params, dim3((n2 + 255) / 256), dim3(256), 0, stream,
                           reinterpret_cast<const double2 *>(p->h_params_dev[p->slot]), reinterpret_cast<double2 *>(p->d_params), n2);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(p->d_params, p->h_params[p->slot], pbytes, hipMemcpyHostToDevice, stream));
    }

    uint64_t *lp = p->lp;
    lp[0] = p->n_segs; lp[1] = p->n_vecs; lp[2] = lp[3] = lp[4] = lp[5] = lp[6] = lp[7] = 0;
    Ev3 ev{nullptr, nullptr, nullptr};
    const bool prof = g.profile && p->n_vecs;
    // Profiling: ev.a .. ev.b brackets the propagate kernel(s).  For k_zpropagate4, whose operator table is built by
    // launches of its own, ev.a goes between the table launches and the scan: "kernel_ms" is then the duration of the
    // scan launch itself - the number rocprofv3's per-kernel average has to agree with.
    bool a_recorded = false;
    if (prof) { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); HIP_TRY(hipEventCreate(&ev.c)); }
#define IMC_MARK_A() do { if (prof && !a_recorded) { HIP_TRY(hipEventRecord(ev.a, stream)); a_recorded = true; } } while (0)
    bool tail_used = false;             // the propagate launch finishes the chunks itself (zip3_tail): no stitch launches
    p->kernels.clear();
    auto note = [&](const std::string &k) { p->kernels += (p->kernels.empty() ? "" : "+") + k; };
    for (const Group &gr : p->groups) {
        if (!gr.n_vecs) continue;
        if (!gr.zip4) IMC_MARK_A();
        const std::string strm = gr.zip ? "[tokens]" : "[columns]";
        PropArgs a;
        a.segs = p->d_segs; a.vecs = p->d_vecs + gr.vec_begin; a.n_vecs = gr.n_vecs; a.vec_base = gr.vec_begin;
        a.n_vecs_total = p->n_vecs; a.N = N; a.S = S;
        a.params = p->d_params; a.pstride = p->pstride; a.P = p->levels[0].d_P; a.EX = p->levels[0].d_EX;
        a.A = gr.A; a.tok_left = gr.zip ? gr.dict->d_left : nullptr; a.tok_right = gr.zip ? gr.dict->d_right : nullptr;
        if (gr.big) {
            BigArgs ba;
            ba.n_phases = 0;
            ba.tail = nullptr; ba.tailX = nullptr; ba.tailE = nullptr; ba.tail_arrive = nullptr; ba.tail_out = nullptr; ba.tail_stride = 0; ba.n_chunks = p->n_chunks;
            ba.segs = p->d_segs; ba.seg_ids = gr.d_seg_ids; ba.seg_vec0 = gr.d_seg_out; ba.blocks = nullptr;
            ba.n_group_segs = (uint32_t)gr.seg_ids.size(); ba.n_vecs_total = p->n_vecs;
            ba.N = N; ba.S = S; ba.A = gr.A; ba.params = p->d_params; ba.params_src = nullptr; ba.pstride = p->pstride; ba.PP = NP;
            ba.tok_left = gr.zip ? gr.dict->d_left : nullptr; ba.tok_right = gr.zip ? gr.dict->d_right : nullptr;
            ba.Ctab = gr.d_Ctab; ba.cex = gr.d_cex;
            ba.Cpack = gr.d_Cpack; ba.TS = N + (N & 1);
            ba.P = p->levels[0].d_P; ba.EX = p->levels[0].d_EX;
            ba.phase = gr.rank1 ? 1 : 0; ba.t_from = 0; ba.t_to = gr.rank1 ? gr.checkpoints[0] : INT_MAX;
            ba.r1flag = gr.d_r1flag; ba.r1at = gr.d_r1at; ba.r1u = gr.d_r1u;
            ba.r1alpha = gr.d_r1alpha; ba.n_segs = p->n_segs;
            ba.tab_order = nullptr; ba.tab_lvl = nullptr; ba.tab_nlvl = 0; ba.hot = nullptr; ba.n_hot = 0; ba.tab_desc = nullptr;
            if (gr.rank1) HIP_TRY(hipMemsetAsync(gr.d_r1flag, 0, (size_t)B * p->n_segs * 4, stream));   // nothing certified yet
            hipLaunchKernelGGL(kc->big_table_raw, dim3((unsigned)S, (unsigned)B), dim3(kc->G * 64), 0, stream, ba);
            HIP_TRY(hipGetLastError());
            if (gr.zip) {   // merged tokens, one launch per dictionary depth (tokens of a depth are independent)
                const DictDev &dd = *gr.dict;
                size_t i0 = 0;
                while (i0 < dd.order.size()) {   // tokens >= A are not part of this level's alphabet: filtered below
                    size_t i1 = i0;
                    while (i1 < dd.order.size() && dd.depth[dd.order[i1]] == dd.depth[dd.order[i0]]) ++i1;
                    // launch the contiguous sub-runs of [i0,i1) whose token id < A
                    size_t r0 = i0;
                    while (r0 < i1) {
                        while (r0 < i1 && (int)dd.order[r0] >= gr.A) ++r0;
                        size_t r1 = r0;
                        while (r1 < i1 && (int)dd.order[r1] < gr.A) ++r1;
                        if (r1 > r0) {
                            hipLaunchKernelGGL(kc->big_table_level, dim3((unsigned)(r1 - r0), (unsigned)B), dim3(kc->G * 64), 0,
                                               stream, ba, (const uint16_t *)dd.d_order, (int)r0);
                            HIP_TRY(hipGetLastError());
                        }
                        r0 = r1;
                    }
                    i0 = i1;
                }
            }
            if (gr.bigvec) {
                const unsigned grid = B >= 8 ? 8u * (unsigned)gr.big_blocks.size() * (unsigned)((B + 7) / 8)
                                             : (unsigned)gr.big_blocks.size() * (unsigned)B;
                hipLaunchKernelGGL(kc->big_vec, dim3(grid), dim3(kc->big_vec_waves * 64), 0, stream, ba,
                                   (const BigBlock *)gr.d_big_blocks, (int)gr.big_blocks.size(), B);
                note("k_big_vector<" + std::to_string(kc->G) + ">" + strm);
                if (gr.zip) { lp[4] = gr.seglen; lp[5] += gr.vsteps * (uint64_t)B; lp[6] += gr.stream_len; lp[7] = std::max(lp[7], (uint64_t)gr.A); }
                else { lp[2] = gr.seglen; lp[3] += gr.vsteps * (uint64_t)B; }
                HIP_TRY(hipGetLastError());
                continue;
            }
            if (!kc->zip_attr_set) {   // (flag reused: dynamic LDS size of the large-N propagate kernel)
                HIP_TRY(hipFuncSetAttribute((const void *)kc->big_prop, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)LDS_BUDGET));
                kc->zip_attr_set = true;
            }
            hipLaunchKernelGGL(kc->big_prop, dim3((unsigned)gr.big_blocks.size(), (unsigned)B), dim3(kc->big_prop_waves * 64), kc->big_lds,
                               stream, ba, (const BigBlock *)gr.d_big_blocks);
            note(std::string(kc->big_prop_waves != kc->G ? "k_big_propagate_s<" : "k_big_propagate<") + std::to_string(kc->G) + ">" + strm);
            if (gr.rank1 && !gr.tail_blocks.empty()) {
                // The first round of heads is done.  Per checkpoint: certify which of the operators still on the GEMM
                // chain collapsed to rank one, then run the others up to the next checkpoint; after the last one the
                // certified segments finish on the mat-vec chain and the rest on the GEMM chain.  Every launch covers
                // all segments and a workgroup with nothing to do exits at once, so there is no host round trip.
                for (size_t r = 0; r < gr.checkpoints.size(); ++r) {
                    HIP_TRY(hipGetLastError());
                    ba.t_to = gr.checkpoints[r];
                    hipLaunchKernelGGL(k_rank1_check, dim3((unsigned)gr.tail_blocks.size(), (unsigned)B), dim3(1024), 0, stream, ba,
                                       (const BigBlock *)gr.d_tail_blocks, NP);
                    HIP_TRY(hipGetLastError());
                    ba.t_from = gr.checkpoints[r];
                    ba.t_to = r + 1 < gr.checkpoints.size() ? gr.checkpoints[r + 1] : INT_MAX;
                    if (r + 1 == gr.checkpoints.size()) {   // the mat-vec tails first: they are the long launch
                        const unsigned grid = B >= 8 ? 8u * (unsigned)gr.tail_blocks.size() * (unsigned)((B + 7) / 8)
                                                     : (unsigned)gr.tail_blocks.size() * (unsigned)B;
                        hipLaunchKernelGGL(kc->big_vec_tail, dim3(grid), dim3(kc->big_vec_waves * 64), 0, stream, ba,
                                           (const BigBlock *)gr.d_tail_blocks, (int)gr.tail_blocks.size(), B);
                        HIP_TRY(hipGetLastError());
                    }
                    hipLaunchKernelGGL(kc->big_prop, dim3((unsigned)gr.big_blocks.size(), (unsigned)B), dim3(kc->big_prop_waves * 64), kc->big_lds,
                                       stream, ba, (const BigBlock *)gr.d_big_blocks);
                }
                note("rank1-handoff");
            }
            if (gr.zip) { lp[4] = gr.seglen; lp[5] += gr.vsteps * (uint64_t)B; lp[6] += gr.stream_len; lp[7] = std::max(lp[7], (uint64_t)gr.A); }
            else { lp[2] = gr.seglen; lp[3] += gr.vsteps * (uint64_t)B; }
        } else if (gr.zip2) {
            BigArgs ba;
            ba.n_phases = 0;
            ba.tail = nullptr; ba.tailX = nullptr; ba.tailE = nullptr; ba.tail_arrive = nullptr; ba.tail_out = nullptr; ba.tail_stride = 0; ba.n_chunks = p->n_chunks;
            ba.segs = p->d_segs; ba.seg_ids = nullptr; ba.seg_vec0 = nullptr; ba.blocks = gr.d_blocks;
            ba.n_group_segs = (uint32_t)gr.blocks.size(); ba.n_vecs_total = p->n_vecs;
            ba.N = N; ba.S = S; ba.A = gr.A; ba.params = p->d_params; ba.params_src = nullptr; ba.pstride = p->pstride; ba.PP = NP;
            ba.tok_left = gr.zip ? gr.dict->d_left : nullptr; ba.tok_right = gr.zip ? gr.dict->d_right : nullptr;
            ba.Ctab = nullptr; ba.cex = nullptr; ba.Cpack = nullptr; ba.TS = 0;
            ba.P = p->levels[0].d_P; ba.EX = p->levels[0].d_EX;
            ba.tab_order = gr.d_tab_order; ba.tab_lvl = gr.d_tab_lvl; ba.tab_nlvl = gr.tab_nlvl;
            ba.hot = gr.d_hot; ba.n_hot = gr.n_hot; ba.tab_desc = gr.d_tab_desc;
            if (gr.d_tails && (g.fuse_tail == 2 || (g.fuse_tail == 1 && gr.tail_stride <= 4)) && allow_tail && kc->use3()) {
                ba.tail = gr.d_tails; ba.tailX = gr.d_tailX; ba.tailE = gr.d_tailE; ba.tail_arrive = gr.d_tail_arrive;
                ba.tail_out = out; ba.tail_stride = gr.tail_stride; ba.n_chunks = p->n_chunks;
                tail_used = true;
            }
            if (gr.zip4) {
                // hybrid table: one workgroup per parameter set builds the operators in global memory (they stay in
                // L2), then the scan caches the hot ones in LDS and streams the rest a step ahead
                ba.Ctab = gr.d_Ctab; ba.cex = gr.d_cex;
                if (!fuse_head) {
                    hipLaunchKernelGGL(kc->zip4_raw, dim3((unsigned)S + 1, (unsigned)B), dim3(256), 0, stream, ba);
                    HIP_TRY(hipGetLastError());
                }
                if (g.table_pairs && (g.table_triples == 1 || (g.table_triples < 0 && NP <= 12)) && gr.d_tab_desc3) {
                    bool head = fuse_head;
                    if (!kc->zip4_attr_l3) {
                        HIP_TRY(hipFuncSetAttribute((const void *)kc->zip4_level3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
                        HIP_TRY(hipFuncSetAttribute((const void *)kc->zip4_level3_first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
                        kc->zip4_attr_l3 = true;
                    }
                    for (const auto &lc : gr.tab3) {          // three dictionary depths per launch, one wavefront per token
                        const dim3 grid((unsigned)(lc.second + Z4L3_WAVES - 1) / Z4L3_WAVES, (unsigned)B);
                        if (head)
                            hipLaunchKernelGGL(kc->zip4_level3_first, grid, dim3(Z4L3_WAVES * 64), kc->zip4_level3_lds(p->pstride), stream, ba,
                                               (const int4 *)gr.d_tab_desc3, lc.first, lc.second, (const double *)p->h_params_dev[p->slot]);
                        else
                            hipLaunchKernelGGL(kc->zip4_level3, grid, dim3(Z4L3_WAVES * 64), kc->zip4_level3_lds(0), stream, ba,
                                               (const int4 *)gr.d_tab_desc3, lc.first, lc.second, (const double *)nullptr);
                        head = false;
                        HIP_TRY(hipGetLastError());
                    }
                } else if (g.table_pairs && gr.d_tab_desc2) {
                    bool head = fuse_head;
                    for (const auto &lc : gr.tab2) {          // two dictionary depths per launch
                        if (head)
                            hipLaunchKernelGGL(kc->zip4_level2_first, dim3((unsigned)(lc.second + 3) / 4, (unsigned)B), dim3(64),
                                               p->pstride * 8, stream, ba,
                                               (const int4 *)gr.d_tab_desc2, lc.first, lc.second, (const double *)p->h_params_dev[p->slot]);
                        else
                            hipLaunchKernelGGL(kc->zip4_level2, dim3((unsigned)(lc.second + 3) / 4, (unsigned)B), dim3(64), 0, stream, ba,
                                               (const int4 *)gr.d_tab_desc2, lc.first, lc.second, (const double *)nullptr);
                        head = false;
                        HIP_TRY(hipGetLastError());
                    }
                } else
                for (int d = 0; d < gr.tab_nlvl; ++d) {   // one launch per dictionary depth: kernel boundaries order the depths
                    const int first = gr.tab_lvl[d], count = gr.tab_lvl[d + 1] - first;
                    hipLaunchKernelGGL(kc->zip4_level, dim3((unsigned)(count + 3) / 4, (unsigned)B), dim3(64), 0, stream, ba, first, count);
                    HIP_TRY(hipGetLastError());
                }
                void (*scan)(BigArgs) = gr.stream_table ? (gr.wide_tokens ? kc->zip4sw : kc->zip4s) : (gr.wide_tokens ? kc->zip4w : kc->zip4);
                bool &attr4 = gr.stream_table ? (gr.wide_tokens ? kc->zip4sw_attr_set : kc->zip4s_attr_set)
                                              : (gr.wide_tokens ? kc->zip4w_attr_set : kc->zip4_attr_set);
                if (!attr4) {
                    HIP_TRY(hipFuncSetAttribute((const void *)scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
                    attr4 = true;
                }
                IMC_MARK_A();
                // XCD-affine grid (BigArgs::n_phases): with the two-dimensional grid every XCD's L2 sees the tables of all
                // B parameter sets
                dim3 scan_grid(ba.n_group_segs, (unsigned)B);
                if (g.xcd_affine && B > 1) {
                    int first = 0, wg = 0;
                    auto phase = [&](int sets, int n_sets_total) {
                        ba.ph_begin[ba.n_phases] = wg; ba.ph_first[ba.n_phases] = first; ba.ph_sets[ba.n_phases] = sets;
                        ++ba.n_phases;
                        const int nb = (int)ba.n_group_segs;
                        wg += sets >= 8 ? 8 * nb * (n_sets_total / 8) : 8 * ((nb + 8 / sets - 1) / (8 / sets));
                        first += n_sets_total;
                    };
                    if (B >= 8) phase(8, B / 8 * 8);
                    for (int sets : {4, 2, 1})
                        if ((B % 8) & sets) phase(sets, sets);
                    scan_grid = dim3((unsigned)wg);
                }
                hipLaunchKernelGGL(scan, scan_grid, dim3(Z2WAVES * 64), kc->zip4_lds(gr.A, gr.n_hot), stream, ba);
                note(std::string("k_zpropagate4<") + std::to_string(NP / 4) + (gr.wide_tokens ? ",16" : "") + (gr.stream_table ? ",streamed>" : ">") + strm);
                lp[4] = gr.seglen; lp[5] += gr.vsteps * (uint64_t)B; lp[6] += gr.stream_len; lp[7] = std::max(lp[7], (uint64_t)gr.A);
                HIP_TRY(hipGetLastError());
                continue;
            }
            const bool v3 = kc->use3();
            bool &attr_set = v3 ? kc->zip3_attr_set : kc->zip2_attr_set;
            if (!attr_set) {
                HIP_TRY(hipFuncSetAttribute((const void *)(v3 ? kc->zip3 : kc->zip2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)LDS_BUDGET));
                attr_set = true;
            }
            size_t lds3 = kc->blocked_lds(gr.A);
            if (direct_params) { ba.params_src = p->h_params_dev[p->slot]; lds3 = ((lds3 + 15) & ~(size_t)15) + p->pstride * 8; }
            hipLaunchKernelGGL(v3 ? kc->zip3 : kc->zip2, dim3(ba.n_group_segs, (unsigned)B), dim3(Z2WAVES * 64), lds3, stream, ba);
            note(std::string(v3 ? "k_zpropagate3<" : "k_zpropagate2<") + std::to_string(NP / 4) + ">" + strm);
            if (gr.zip) { lp[4] = gr.seglen; lp[5] += gr.vsteps * (uint64_t)B; lp[6] += gr.stream_len; lp[7] = std::max(lp[7], (uint64_t)gr.A); }
            else { lp[2] = gr.seglen; lp[3] += gr.vsteps * (uint64_t)B; }
        } else if (gr.zip) {
            const size_t lds = kc->zip_lds(gr.A);
            if (!kc->zip_attr_set) {
                HIP_TRY(hipFuncSetAttribute((const void *)kc->zip, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)LDS_BUDGET));
                kc->zip_attr_set = true;
            }
            const uint32_t vpb = (uint32_t)(ZWAVES * kc->VPW);
            dim3 grid((gr.n_vecs + vpb - 1) / vpb, (unsigned)B);
            hipLaunchKernelGGL(kc->zip, grid, dim3(ZWAVES * 64), lds, stream, a);
            note("k_zpropagate<" + std::to_string(kc->R) + "," + std::to_string(kc->G) + ">" + strm);
            lp[4] = gr.seglen; lp[5] += gr.vsteps * (uint64_t)B; lp[6] += gr.stream_len; lp[7] = std::max(lp[7], (uint64_t)gr.A);
        } else {
            const uint32_t vpb = (uint32_t)(WPB * kc->VPW);
            dim3 grid((gr.n_vecs + vpb - 1) / vpb, (unsigned)B);
            const size_t lds = ((size_t)WPB * kc->VPW * NP + (size_t)S * NP) * 8;
            if (lds > LDS_BUDGET) return fail(IMC_ERR_ARG, "emission table (S x N) too large for LDS");
            if (lds > 48 * 1024 && !kc->plain_attr_set) {   // large alphabets: opt in to > 64 KB of dynamic LDS
                HIP_TRY(hipFuncSetAttribute((const void *)kc->plain, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)LDS_BUDGET));
                kc->plain_attr_set = true;
            }
            hipLaunchKernelGGL(kc->plain, grid, dim3(WPB * 64), lds, stream, a);
            note("k_propagate<" + std::to_string(kc->R) + "," + std::to_string(kc->G) + ">" + strm);
            lp[2] = gr.seglen; lp[3] += gr.vsteps * (uint64_t)B;
        }
        HIP_TRY(hipGetLastError());
    }
    IMC_MARK_A();
#undef IMC_MARK_A
    if (prof) HIP_TRY(hipEventRecord(ev.b, stream));
    if (tail_used) note("fused-tail");
    for (size_t l = 0; l + 1 < p->levels.size() && !tail_used; ++l) {
        const Level &in = p->levels[l], &ot = p->levels[l + 1];
        if (!in.n_segs || !ot.n_chains) continue;
        if (!kc->chain_self_emax) {   // (the single-wavefront chain kernels find the units' largest exponents themselves)
            hipLaunchKernelGGL(k_emax, dim3((in.n_segs + 255) / 256, (unsigned)B), dim3(256), 0, stream,
                               in.d_vec0, in.d_first, in.n_segs, in.n_vecs, N, in.d_EX, in.d_EMAX);
            HIP_TRY(hipGetLastError());
        }
        const int threads = (int)round_up((size_t)NP, 64);
        const bool last_level = l + 2 == p->levels.size();
        hipLaunchKernelGGL(kc->chain, dim3(ot.n_chains, (unsigned)B), dim3(threads), 0, stream,
                           ot.d_chains, N, in.d_vec0, in.n_segs, in.n_vecs, in.d_P, in.d_EX, in.d_EMAX,
                           ot.n_vecs, ot.d_P, ot.d_EX, (last_level && p->finish_fused) ? out : (double *)nullptr, p->n_chunks);
        HIP_TRY(hipGetLastError());
    }
    if (p->n_chunks && !p->finish_fused && !tail_used) {
        const Level &last = p->levels.back();
        hipLaunchKernelGGL(k_finish, dim3((p->n_chunks + 63) / 64, (unsigned)B), dim3(64), 0, stream,
                           p->d_final_vec, p->n_chunks, N, NP, last.n_vecs, last.d_P, last.d_EX, out);
        HIP_TRY(hipGetLastError());
    }
    if (prof) {
        HIP_TRY(hipEventRecord(ev.c, stream));
        g.events.push_back(ev);
    }
    if (cap == hipStreamCaptureStatusNone) {
        // one event per call, behind its last kernel (an event record holds the queue for ~5 us: not in front of the
        // first kernel): it marks the staging slot free again and orders the plan's next call if that comes on
        // another stream
        HIP_TRY(hipEventRecord(p->ev_params[p->slot], stream));
        p->ev_used[p->slot] = true;
        p->last_stream = stream;
        p->last_slot = p->slot;
        p->have_last = true;
    }
    return IMC_OK;
}

// After a call has been synchronised: how many operator segments the rank-one test saw and how many it certified
// (at any checkpoint).  Statistics only: nothing here feeds back into later calls.
void collect_rank1_stats(Plan *p)
{
    g.r1_checked = g.r1_collapsed = 0;
    for (Group &gr : p->groups) {
        if (!gr.rank1 || gr.r1_segs.empty() || gr.checkpoints.empty()) continue;
        std::vector<int> flags((size_t)p->B * p->n_segs);
        if (hipMemcpy(flags.data(), gr.d_r1flag, flags.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) continue;
        for (int b = 0; b < p->B; ++b)
            for (const auto &sl : gr.r1_segs) {
                if ((int)sl.second <= gr.checkpoints[0]) continue;
                ++g.r1_checked;
                g.r1_collapsed += flags[(size_t)b * p->n_segs + sl.first] ? 1 : 0;
            }
        if (getenv("IMC_DEBUG_R1")) {          // where each head was certified (diagnostics only)
            std::vector<int> at((size_t)p->B * p->n_segs);
            if (hipMemcpy(at.data(), gr.d_r1at, at.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) continue;
            std::map<int, int> hist;
            for (const auto &sl : gr.r1_segs) hist[flags[sl.first] ? at[sl.first] : -1]++;
            fprintf(stderr, "[imc] rank-one hand-off: segment length %zu tokens, checkpoints", (size_t)gr.seglen);
            for (int c : gr.checkpoints) fprintf(stderr, " %d", c);
            fprintf(stderr, "\n[imc]   certified at (tokens: segments)");
            for (auto &kv : hist) fprintf(stderr, " %d:%d", kv.first, kv.second);
            fprintf(stderr, "\n");
        }
    }
}

// The caller of a synchronous entry point is waiting for the value.  k_finish writes the per-chunk results straight
// into mapped host memory; the slots are set to a sentinel (a NaN payload no computation produces) before the launch
// and the host polls them for up to two milliseconds - an evaluation of BASELINE config[1] takes 0.3 ms, and a
// blocking hipStreamSynchronize returns some 15-20 us after the last kernel has finished - before it falls back to
// the blocking wait.  The stream itself is idle a few microseconds later; later calls are ordered behind it anyway.
constexpr uint64_t OUT_SENTINEL = 0x7ff8dead5e471e15ull;

hipError_t wait_results(hipStream_t st, const double *h_out, size_t n)
{
    const volatile uint64_t *v = reinterpret_cast<const volatile uint64_t *>(h_out);
    const auto t0 = std::chrono::steady_clock::now();
    size_t k = 0;
    for (unsigned spin = 0;; ++spin) {
        while (k < n && v[k] != OUT_SENTINEL) ++k;
        if (k == n) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
        if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(st);
    }
}

int run_batch(const imc_obs *const *chunks, int n_chunks, int B, int N, int S, const double *pis, const double *Ts,
              const double *Es, double *out_sum, double *out_per_chunk)
{
    std::unique_lock<std::mutex> lk(g_mu);
    static const bool dbg_host = std::getenv("IMC_DEBUG_HOST") != nullptr;     // diagnostics: host time per phase
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto h0 = now();
    if (int rc = ensure_ctx()) return rc;
    if (int rc = check_args(chunks, n_chunks, B, N, S, pis, Ts, Es)) return rc;
    HIP_TRY(hipSetDevice(g.device));
    Plan *p = nullptr;
    for (;;) {                                   // (the plan is looked up again after every wait: it may have been released)
        if (int rc = build_plan(chunks, n_chunks, N, S, B, false, &p)) return rc;
        if (!p->busy) break;
        g_cv.wait(lk);                           // another thread is waiting for this plan's results
    }
    struct Busy {                                // destroyed with g_mu held (declared after lk)
        Plan *p;
        explicit Busy(Plan *q) : p(q) { p->busy = true; }
        ~Busy() { p->busy = false; g_cv.notify_all(); }
    } busy(p);
    const auto h1 = now();
    if (int rc = stage_params(p, pis, Ts, Es, g.use_graphs)) return rc;
    const size_t n_out = (size_t)B * n_chunks;
    for (size_t k = 0; k < n_out; ++k) reinterpret_cast<volatile uint64_t *>(p->h_out)[k] = OUT_SENTINEL;
    std::atomic_thread_fence(std::memory_order_release);
    const auto h2 = now();
    // First call of a plan runs eagerly (kernel attributes get set); the second is captured into a hipGraph that
    // every later call replays: one graph launch instead of ~8 stream operations per evaluation.
    const bool use_graph = !g.profile && g.use_graphs && p->calls >= 1;
    if (use_graph && !p->graph) {
        hipGraph_t gr = nullptr;
        HIP_TRY(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(p, g.stream, p->h_out_dev);
        const hipError_t ec = hipStreamEndCapture(g.stream, &gr);
        if (rc) { if (gr) (void)hipGraphDestroy(gr); return rc; }
        if (ec != hipSuccess) return fail(IMC_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
        const hipError_t ei = hipGraphInstantiate(&p->graph, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        if (ei != hipSuccess) { p->graph = nullptr; return fail(IMC_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
    }
    if (use_graph) HIP_TRY(hipGraphLaunch(p->graph, g.stream));
    else if (int rc = enqueue(p, g.stream, p->h_out_dev)) return rc;
    ++p->calls;
    for (int k = 0; k < 8; ++k) g.last_plan[k] = p->lp[k];
    g.last_kernels = p->kernels;
    const auto h3 = now();
    const hipStream_t st = g.stream;
    lk.unlock();                                 // other threads may enqueue their evaluations while this one waits
    const hipError_t ew = n_out ? wait_results(st, p->h_out, n_out) : hipStreamSynchronize(st);
    lk.lock();
    HIP_TRY(ew);
    const auto h4 = now();
    collect_rank1_stats(p);
    if (dbg_host) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        static double acc[5] = {0, 0, 0, 0, 0};
        static int n_acc = 0;
        acc[0] += us(h0, h1); acc[1] += us(h1, h2); acc[2] += us(h2, h3); acc[3] += us(h3, h4); acc[4] += us(h4, now());
        if (++n_acc % 20 == 0) {
            fprintf(stderr, "[imc] host us per call: check+plan %.1f  stage %.1f  enqueue %.1f  sync wait %.1f  stats %.1f\n",
                    acc[0] / 20, acc[1] / 20, acc[2] / 20, acc[3] / 20, acc[4] / 20);
            for (double &a : acc) a = 0.0;
        }
    }
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

// State export (imc_forward_state): the plan runs as usual, then every chunk's final vector (from pi) or transfer
// operator is copied out with its power-of-two exponents instead of being reduced to a log-likelihood.
int run_state(const imc_obs *const *chunks, int n_chunks, bool op_mode, int B, int N, int S, const double *pis,
              const double *Ts, const double *Es, double *out_state, int *out_exp)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = ensure_ctx()) return rc;
    if (int rc = check_args(chunks, n_chunks, B, N, S, pis, Ts, Es)) return rc;
    if (!out_state || !out_exp) return fail(IMC_ERR_ARG, "output buffer is null");
    for (int f = 0; f < n_chunks; ++f)
        if (chunks[f]->L == 0) return fail(IMC_ERR_ARG, "imc_forward_state: empty chunk");
    HIP_TRY(hipSetDevice(g.device));
    Plan *p = nullptr;
    if (int rc = build_plan(chunks, n_chunks, N, S, B, op_mode, &p)) return rc;
    if (int rc = stage_params(p, pis, Ts, Es, g.use_graphs)) return rc;
    if (int rc = enqueue(p, g.stream, p->d_out, false)) return rc;      // (the log-likelihoods are not read here; the state comes from the stitch levels)
    ++p->calls;
    for (int k = 0; k < 8; ++k) g.last_plan[k] = p->lp[k];
    g.last_kernels = p->kernels;
    const size_t per = op_mode ? (size_t)N * N : (size_t)N, pere = op_mode ? (size_t)N : 1;
    const size_t n_state = (size_t)B * n_chunks * per, n_exp = (size_t)B * n_chunks * pere;
    double *d_state = nullptr;
    int *d_exp = nullptr;
    hipError_t e = dev_alloc((void **)&d_state, std::max<size_t>(n_state * 8, 16));
    if (e == hipSuccess) e = dev_alloc((void **)&d_exp, std::max<size_t>(n_exp * 4, 16));
    if (e == hipSuccess) {
        const Level &last = p->levels.back();
        hipLaunchKernelGGL(k_export, dim3((unsigned)n_chunks, (unsigned)B), dim3(256), 0, g.stream, p->d_final_vec, n_chunks, N,
                           p->kc->NP, last.n_vecs, last.d_P, last.d_EX, op_mode ? 1 : 0, d_state, d_exp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_state, d_state, n_state * 8, hipMemcpyDeviceToHost, g.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_exp, d_exp, n_exp * 4, hipMemcpyDeviceToHost, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    dev_free(d_state);
    dev_free(d_exp);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? IMC_ERR_OOM : IMC_ERR_HIP, std::string("state export: ") + hipGetErrorString(e));
    return IMC_OK;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================

extern "C" {

const char *imc_version(void) { return "imcoal_fwd 0.2 (gfx950)"; }
const char *imc_last_error(void) { return g_err.c_str(); }

int imc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int imc_set_device(int device)
{
    std::unique_lock<std::mutex> lk(g_mu);
    if (device < 0) return fail(IMC_ERR_ARG, "negative device index");
    wait_all_idle(lk);
    if (g.ready && g.pid == getpid() && g.device != device) {
        drop_plans();
        g.dicts.clear();
        (void)hipStreamDestroy(g.stream);
        g.ready = false;
        reset_kernel_attributes();   // the dynamic-LDS opt-in is per device
    }
    g.device = device;
    return ensure_ctx();
}

int imc_obs_create(const uint8_t *sym, size_t L, int nsym, imc_obs **out)
{
    if (!out) return fail(IMC_ERR_ARG, "out is null");
    if (nsym < 1 || nsym > 256) return fail(IMC_ERR_ARG, "nsym must be in [1,256] for byte symbols (larger alphabets: imc_obs_create_i32)");
    if (L && !sym) return fail(IMC_ERR_ARG, "sym is null");
    if (L >= (size_t)1 << 31) return fail(IMC_ERR_ARG, "chunk too long (limit 2^31-1 columns per chunk)");
    for (size_t t = 0; t < L; ++t)
        if (sym[t] >= nsym) return fail(IMC_ERR_SYMBOL, "symbol " + std::to_string(sym[t]) + " at column " + std::to_string(t) + " >= nsym");
    return obs_upload(sym, nullptr, L, nsym, out);
}

int imc_obs_create_i32(const int32_t *sym, size_t L, int nsym, imc_obs **out)
{
    if (!out) return fail(IMC_ERR_ARG, "out is null");
    if (nsym < 1 || nsym > imc::kMaxRawAlphabet) return fail(IMC_ERR_ARG, "nsym must be in [1," + std::to_string(imc::kMaxRawAlphabet) + "]");
    if (L && !sym) return fail(IMC_ERR_ARG, "sym is null");
    if (L >= (size_t)1 << 31) return fail(IMC_ERR_ARG, "chunk too long (limit 2^31-1 columns per chunk)");
    for (size_t t = 0; t < L; ++t)
        if (sym[t] < 0 || sym[t] >= nsym)
            return fail(IMC_ERR_SYMBOL, "symbol " + std::to_string(sym[t]) + " at column " + std::to_string(t) + " outside [0,nsym)");
    if (nsym > imc::kByteAlphabet) {                    // e.g. the 257-symbol quartet alphabet (prepare-alignments.py:186-190)
        std::vector<imc::tok_t> wide(sym, sym + L);
        return obs_upload(nullptr, wide.data(), L, nsym, out);
    }
    std::vector<uint8_t> tmp(sym, sym + L);
    return obs_upload(tmp.data(), nullptr, L, nsym, out);
}

int imc_obs_create_from_text(const char *path, int nsym, imc_obs **out)
{
    if (!out || !path) return fail(IMC_ERR_ARG, "null argument");
    if (nsym < 1 || nsym > imc::kMaxRawAlphabet) return fail(IMC_ERR_ARG, "nsym must be in [1," + std::to_string(imc::kMaxRawAlphabet) + "]");
    if (nsym > imc::kByteAlphabet) {
        std::vector<imc::tok_t> wide;
        const imc::IoResult rw = imc::read_observation_file(path, nsym, wide);
        if (rw.code) return fail(rw.code, rw.msg);
        if (wide.size() >= (size_t)1 << 31) return fail(IMC_ERR_ARG, "chunk too long (limit 2^31-1 columns per chunk)");
        return obs_upload(nullptr, wide.data(), wide.size(), nsym, out);
    }
    std::vector<uint8_t> sym;
    const imc::IoResult r = imc::read_observation_file(path, nsym, sym);
    if (r.code) return fail(r.code, r.msg);
    if (sym.size() >= (size_t)1 << 31) return fail(IMC_ERR_ARG, "chunk too long (limit 2^31-1 columns per chunk)");
    return obs_upload(sym.data(), nullptr, sym.size(), nsym, out);
}

int imc_read_observations(const char *path, int nsym, uint8_t *sym_out, size_t capacity, size_t *length)
{
    if (!path || !length) return fail(IMC_ERR_ARG, "null argument");
    if (nsym < 1 || nsym > 256) return fail(IMC_ERR_ARG, "nsym must be in [1,256]");
    std::vector<uint8_t> sym;
    const imc::IoResult r = imc::read_observation_file(path, nsym, sym);
    if (r.code) return fail(r.code, r.msg);
    *length = sym.size();
    if (sym_out && capacity >= sym.size() && !sym.empty()) std::memcpy(sym_out, sym.data(), sym.size());
    return IMC_OK;
}

int imc_write_cache(const char *path, const uint8_t *sym, size_t L, int nsym)
{
    if (!path || (L && !sym)) return fail(IMC_ERR_ARG, "null argument");
    if (nsym < 1 || nsym > 256) return fail(IMC_ERR_ARG, "nsym must be in [1,256]");
    const imc::IoResult r = imc::write_cache(path, sym, L, nsym);
    return r.code ? fail(r.code, r.msg) : IMC_OK;
}

int imc_encode_pairwise(const char *seq1, const char *seq2, size_t L, uint8_t *sym_out)
{
    if (L && (!seq1 || !seq2 || !sym_out)) return fail(IMC_ERR_ARG, "null argument");
    imc::encode_pairwise(seq1, seq2, L, sym_out);
    return IMC_OK;
}

size_t imc_obs_length(const imc_obs *obs) { return obs ? obs->L : 0; }
int imc_obs_nsym(const imc_obs *obs) { return obs ? obs->nsym : 0; }

size_t imc_obs_compressed_length(const imc_obs *obs, int alphabet_limit, int *alphabet_used)
{
    if (!obs) return 0;
    size_t n = obs->L;
    int used = obs->nsym;
    if (obs->dict)
        for (int l = 0; l < imc::kNumLevels; ++l)
            if (obs->alphabet[l] <= alphabet_limit && obs->alphabet[l] > obs->nsym && obs->d_tok[l]) { n = obs->ntok[l]; used = obs->alphabet[l]; }
    if (alphabet_used) *alphabet_used = used;
    return n;
}

int imc_obs_dictionary(const imc_obs *obs, int alphabet_limit, uint16_t *left, uint16_t *right, size_t capacity, int *alphabet_used)
{
    if (!obs || !alphabet_used) return fail(IMC_ERR_ARG, "null argument");
    int used = obs->nsym;
    if (obs->dict)
        for (int l = 0; l < imc::kNumLevels; ++l)
            if (obs->alphabet[l] <= alphabet_limit && obs->alphabet[l] > obs->nsym && obs->d_tok[l]) used = obs->alphabet[l];
    *alphabet_used = used;
    if ((left || right) && capacity < (size_t)(used - obs->nsym)) return fail(IMC_ERR_ARG, "capacity smaller than the number of merged tokens");
    for (int z = obs->nsym; z < used; ++z) {
        if (left) left[z - obs->nsym] = obs->dict->dict.left[z];
        if (right) right[z - obs->nsym] = obs->dict->dict.right[z];
    }
    return IMC_OK;
}

int imc_obs_tokens(const imc_obs *obs, int alphabet_limit, uint16_t *tokens, size_t capacity, size_t *length, int *alphabet_used)
{
    if (!obs || !length) return fail(IMC_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    if (obs->pid != getpid()) return fail(IMC_ERR_ARG, "chunk handle was created in another process");
    int level = -1, used = obs->nsym;
    if (obs->dict)
        for (int l = 0; l < imc::kNumLevels; ++l)
            if (obs->alphabet[l] <= alphabet_limit && obs->alphabet[l] > obs->nsym && obs->d_tok[l]) { level = l; used = obs->alphabet[l]; }
    const size_t n = level >= 0 ? obs->ntok[level] : obs->L;
    *length = n;
    if (alphabet_used) *alphabet_used = used;
    if (!tokens) return IMC_OK;
    if (capacity < n) return fail(IMC_ERR_ARG, "capacity smaller than the stream");
    if (int rc = ensure_ctx()) return rc;
    HIP_TRY(hipSetDevice(obs->device));
    const bool wide = level >= 0 ? obs->wide[level] : obs->wide_raw;
    const uint8_t *src = level >= 0 ? obs->d_tok[level] : obs->d_sym;
    if (wide) HIP_TRY(hipMemcpy(tokens, src, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    else {
        std::vector<uint8_t> tmp(n);
        if (n) HIP_TRY(hipMemcpy(tmp.data(), src, n, hipMemcpyDeviceToHost));
        for (size_t t = 0; t < n; ++t) tokens[t] = tmp[t];
    }
    return IMC_OK;
}

// Joint re-compression (see the header).  The first sufficiently long chunk of an alphabet trains the dictionary that
// every later chunk shares; when a data set arrives as many chunks, that sample is one chunk.  Here a new dictionary
// is trained on a sample drawn evenly from ALL the given chunks and they are re-encoded with it.
int imc_obs_recompress(imc_obs *const *chunks, int n_chunks)
{
    if (n_chunks < 0 || (n_chunks > 0 && !chunks)) return fail(IMC_ERR_ARG, "chunks is null");
    std::map<int, std::vector<imc_obs *>> by_alphabet;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ctx()) return rc;
        if (!g.compression) return IMC_OK;
        for (int f = 0; f < n_chunks; ++f) {
            imc_obs *o = chunks[f];
            if (!o) return fail(IMC_ERR_ARG, "chunk is null");
            if (o->pid != g.pid) return fail(IMC_ERR_HIP, "chunk was created in another process");
            if (o->L >= ZIP_MIN_COLUMNS && o->nsym < imc::kMaxAlphabet / 2) by_alphabet[o->nsym].push_back(o);
        }
    }
    for (auto &kv : by_alphabet) {
        const int nsym = kv.first;
        std::vector<imc_obs *> &obs = kv.second;
        // (in creation order, not in address order: the training sample is the concatenation of the chunks' heads, and with
        // pointer order the dictionary - alphabet, token counts, now and then whether the byte phase filled its 256 entries
        // at all - changed from run to run of the same program)
        std::sort(obs.begin(), obs.end(), [](const imc_obs *x, const imc_obs *y) { return x->id < y->id; });
        obs.erase(std::unique(obs.begin(), obs.end()), obs.end());
        const bool wide_raw = nsym > imc::kByteAlphabet;
        const size_t unit = wide_raw ? sizeof(imc::tok_t) : 1;
        size_t total = 0;
        for (imc_obs *o : obs) total += o->L;
        if (total < DICT_TRAIN_MIN) continue;
        {   // nothing to gain if these chunks already share a dictionary trained on at least this much data
            // (a second Likelihood over the same forwarders, a subset of a recompressed set)
            std::lock_guard<std::mutex> lk(g_mu);
            bool same = obs[0]->dict != nullptr;
            for (imc_obs *o : obs) same = same && o->dict == obs[0]->dict;
            if (same && obs[0]->dict->trained_on >= total) continue;
        }
        // ---- raw symbols back to the host (the chunks keep them on the device) ----
        std::vector<std::vector<uint8_t>> raw(obs.size());
        {
            std::lock_guard<std::mutex> lk(g_mu);
            HIP_TRY(hipSetDevice(g.device));
            for (size_t k = 0; k < obs.size(); ++k) {
                raw[k].resize(obs[k]->L * unit);
                HIP_TRY(hipMemcpy(raw[k].data(), obs[k]->d_sym, raw[k].size(), hipMemcpyDeviceToHost));
            }
        }
        // ---- training sample: an equal share of every chunk (its head), up to the training budget ----
        const size_t budget = wide_raw ? DICT_WIDE_TRAIN_TOKENS * 8 : std::max(DICT_TRAIN_MAX, DICT_WIDE_TRAIN_TOKENS * 96);
        const size_t share = std::max<size_t>(DICT_TRAIN_MIN, budget / obs.size());
        std::vector<uint8_t> sample;
        for (size_t k = 0; k < obs.size(); ++k) {
            const size_t n = std::min(obs[k]->L, share);
            sample.insert(sample.end(), raw[k].begin(), raw[k].begin() + n * unit);
        }
        const size_t ns = sample.size() / unit;
        auto nd = wide_raw ? make_dictionary(nullptr, reinterpret_cast<const imc::tok_t *>(sample.data()), ns, nsym)
                           : make_dictionary(sample.data(), nullptr, ns, nsym);
        std::vector<uint8_t>().swap(sample);
        nd->trained_on = total;
        // ---- re-encode (host, unlocked), then swap the streams in under the lock ----
        std::vector<imc::EncodedLevels> enc(obs.size());
        const bool zipped = nd->dict.alphabet > nsym;
        if (zipped)
            for (size_t k = 0; k < obs.size(); ++k) {
                if (wide_raw) imc::encode_levels(nd->dict, nullptr, reinterpret_cast<const imc::tok_t *>(raw[k].data()), obs[k]->L, enc[k]);
                else imc::encode_levels(nd->dict, raw[k].data(), nullptr, obs[k]->L, enc[k]);
                std::vector<uint8_t>().swap(raw[k]);
            }
        std::unique_lock<std::mutex> lk(g_mu);
        wait_all_idle(lk);
        HIP_TRY(hipSetDevice(g.device));
        HIP_TRY(hipDeviceSynchronize());                 // nothing of these chunks is in flight any more
        for (auto it = g_plans.begin(); it != g_plans.end();) {      // plans hold raw pointers into the old streams
            bool uses = false;
            for (size_t k = 0; k < (size_t)(*it)->n_chunks; ++k)
                for (imc_obs *o : obs) uses |= (*it)->key[k] == o->id;
            if (uses) { (*it)->release(); it = g_plans.erase(it); } else ++it;
        }
        if (!zipped) continue;                           // (incompressible sample: keep what the chunks have)
        if (int rc = upload_dictionary(nd)) return rc;
        g.dicts[nsym] = nd;                              // later chunks of this alphabet share it too
        for (size_t k = 0; k < obs.size(); ++k) {
            release_tokens(obs[k]);
            if (int rc = install_encoding(obs[k], nd, enc[k])) return rc;
        }
    }
    return IMC_OK;
}

int imc_obs_free(imc_obs *obs)
{
    if (!obs) return IMC_OK;
    std::unique_lock<std::mutex> lk(g_mu);
    wait_all_idle(lk);
    if (obs->pid == getpid() && g.ready) {
        // plans hold raw pointers into this chunk's device buffers
        for (auto it = g_plans.begin(); it != g_plans.end();) {
            bool uses = false;
            for (size_t k = 0; k < (size_t)(*it)->n_chunks; ++k)
                if ((*it)->key[k] == obs->id) uses = true;
            if (uses) { (*it)->release(); it = g_plans.erase(it); } else ++it;
        }
        (void)hipSetDevice(obs->device);
        obs_release(obs);
    }
    delete obs;
    return IMC_OK;
}

int imc_forward_state(const imc_obs *const *chunks, int n_chunks, int as_operator, int B, int N, int S, const double *pis,
                      const double *Ts, const double *Es, double *out_state, int *out_exp)
{
    return run_state(chunks, n_chunks, as_operator != 0, B, N, S, pis, Ts, Es, out_state, out_exp);
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
    // NULL is the default stream, as everywhere in HIP - NOT the library's own (non-blocking) stream: the caller goes on
    // to use d_out_partial on the stream it named (torch's default stream has the handle 0), and work queued on a
    // non-blocking stream would not be ordered before that use.
    hipStream_t st = (hipStream_t)hip_stream;
    Plan *p = nullptr;
    if (int rc = build_plan(chunks, n_chunks, N, S, B, false, &p)) return rc;
    // no stream synchronisation here: the parameters go through the two-slot pinned staging (stage_params waits only
    // for the upload issued two calls ago), and nothing is read back - the call returns once everything is enqueued
    if (int rc = stage_params(p, pis, Ts, Es, g.use_graphs)) return rc;
    if (int rc = enqueue(p, st, p->d_out)) return rc;
    ++p->calls;
    for (int k = 0; k < 8; ++k) g.last_plan[k] = p->lp[k];
    g.last_kernels = p->kernels;
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

int imc_set_compression(int mode)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (mode < 0 || mode > 5) return fail(IMC_ERR_ARG, "compression mode must be in [0,5]");
    g.compression = (mode == 1 || mode == 2 || mode == 3) ? 1 : 0;
    g.kernel_pref = (mode == 2 || mode == 4) ? 1 : (mode == 3 || mode == 5) ? 2 : 0;
    return IMC_OK;
}

int imc_dictionary_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g.dicts.clear();   // chunks already created keep (and share) the dictionary they were encoded with
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

const char *imc_last_kernels(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    static thread_local std::string copy;
    copy = g.last_kernels;
    return copy.c_str();
}

int imc_set_rank1_handoff(int on)
{
    std::unique_lock<std::mutex> lk(g_mu);
    wait_all_idle(lk);
    if (g.rank1_handoff != (on != 0)) drop_plans();   // cached plans were built for the other setting
    g.rank1_handoff = on != 0;
    return IMC_OK;
}

#ifdef IMC_Z4_TIMING
int imc_debug_z4_timing(long long *out, int n)     // diagnostics build only (see kernels_zip4.hpp)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_z4_dbg), (size_t)std::min(n, 3 * 1024 * 8) * 8) != hipSuccess) return IMC_ERR_HIP;
    return IMC_OK;
}
#endif

int imc_set_blocked_kernel(int variant)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (variant < 2 || variant > 5) return fail(IMC_ERR_ARG, "blocked kernel variant must be 2 (VALU/DPP), 3 (fp64 MFMA, LDS table), 4 (+ hybrid table, default) or 5 (hybrid wherever possible)");
    g.blocked_variant = variant;            // (part of the plan key: cached plans of the other variant stay valid)
    return IMC_OK;
}

int imc_set_table_streaming(int mode)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (mode < -1 || mode > 1) return fail(IMC_ERR_ARG, "table streaming mode must be -1 (automatic), 0 (hybrid LDS cache) or 1 (streamed)");
    g.z4_stream = mode;                     // (part of the plan key)
    return IMC_OK;
}

int imc_last_rank1(uint64_t *checked, uint64_t *collapsed)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (checked) *checked = g.r1_checked;
    if (collapsed) *collapsed = g.r1_collapsed;
    return IMC_OK;
}

int imc_last_plan(uint64_t *out8)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out8) return fail(IMC_ERR_ARG, "out8 is null");
    for (int k = 0; k < 8; ++k) out8[k] = g.last_plan[k];
    return IMC_OK;
}

// ---- host-side model construction (include/imcoal_model.h; no device, no context) ----
int imc_model_transitions(int n_systems, int n_intervals, const int32_t *space_size, const int32_t *cls_off,
                          const int32_t *cls_idx, const int32_t *piece_q, const int32_t *piece_proj, int n_q,
                          const int32_t *q_size, int n_proj, const int32_t *proj_off, const double *proj, const double *Q,
                          const double *dt, const double *start, double *pi, double *T, int n_threads)
{
    if (n_systems < 0 || n_intervals < 1 || n_q < 1 || !space_size || !cls_off || !cls_idx || !q_size || !Q || !start || !pi || !T ||
        (n_intervals > 1 && (!piece_q || !piece_proj || !dt)))
        return fail(IMC_ERR_ARG, "imc_model_transitions: bad arguments");
    std::vector<int32_t> q_off((size_t)n_q);
    int stride = 0;
    for (int k = 0; k < n_q; ++k) {
        if (q_size[k] < 1) return fail(IMC_ERR_ARG, "imc_model_transitions: rate matrix of order < 1");
        q_off[k] = stride;
        stride += q_size[k] * q_size[k];
    }
    for (int i = 0; i + 1 < n_intervals; ++i) {
        if (piece_q[i] < 0 || piece_q[i] >= n_q || piece_proj[i] >= n_proj || (piece_proj[i] >= 0 && (!proj || !proj_off)))
            return fail(IMC_ERR_ARG, "imc_model_transitions: piece index out of range");
    }
    for (int i = 0; i < n_intervals; ++i)
        for (int k = cls_off[3 * i]; k < cls_off[3 * i + 3]; ++k)
            if (cls_idx[k] < 0 || cls_idx[k] >= space_size[i]) return fail(IMC_ERR_ARG, "imc_model_transitions: class index outside its state space");
    imc_model::Structure st{n_intervals, space_size, cls_off, cls_idx, piece_q, piece_proj, n_q, q_size, q_off.data(), stride, proj_off, proj};
    const size_t n = (size_t)n_intervals, s0 = (size_t)space_size[0];
    const int threads = std::max(1, std::min(std::min(n_threads, n_systems), 64));
    std::vector<std::string> errs((size_t)threads);
    auto work = [&](int t) {
        for (int b = t; b < n_systems && errs[t].empty(); b += threads)
            errs[t] = imc_model::transitions_one(st, Q + (size_t)b * stride, dt ? dt + (size_t)b * (n - 1) : nullptr, start + (size_t)b * s0,
                                                 pi + (size_t)b * n, T + (size_t)b * n * n);
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
    }
    for (const std::string &e : errs)
        if (!e.empty()) return fail(IMC_ERR_ARG, e);
    return IMC_OK;
}

int imc_model_expm(int n, const double *A, double *out)
{
    if (n < 1 || !A || !out) return fail(IMC_ERR_ARG, "imc_model_expm: bad arguments");
    std::vector<double> work;
    if (!imc_model::expm(A, out, n, work)) return fail(IMC_ERR_ARG, "imc_model_expm: singular Pade denominator");
    return IMC_OK;
}

}  // extern "C"
