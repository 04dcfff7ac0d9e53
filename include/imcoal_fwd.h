/*
 * imcoal_fwd.h - C ABI of libimcoal_fwd.so, the MI355X (gfx950) HMM forward log-likelihood engine
 * that replaces the `ziphmm` calls behind IMCoalHMM's Forwarder / Likelihood.
 *
 * The reference has no FFI of its own for this path: it is pure Python calling the third-party
 * `ziphmm` extension module.  Each entry point below names the reference call it replaces
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes stub a maintainer would
 * put in src/IMCoalHMM/hmm.py.
 *
 * Conventions (identical to what the reference's model layer produces, model.py:44-49):
 *   all matrices row-major float64;  pi[N];  T[i*N+j] = P(next=j|cur=i) (transitions.py:241-248);
 *   E[j*S+s] = P(symbol s|state j) (emissions.py:89-100);  symbols are integers in [0,S).
 *   Every chunk (alignment file) restarts from pi and chunk log-likelihoods are summed
 *   left-to-right starting from 0.0 (likelihood.py:33).
 *   A zero-probability sequence yields -inf (not an error); NaN in -> NaN out.
 *   Limits: N <= 256 states, S <= 4096 symbols (alphabets beyond 256, e.g. the 257-symbol quartet alphabet of
 *   scripts/prepare-alignments.py:186-190, go through imc_obs_create_i32 / imc_obs_create_from_text and are kept as
 *   16-bit symbols), < 2^31 columns per chunk, and |log-likelihood| of one chunk
 *   below ~1.4e9 nats (the power-of-two exponents of a chunk are summed in int32).
 *
 * Ownership: input buffers are borrowed for the duration of the call only.  Observations are
 * copied to device memory at imc_obs_create* and owned by the library until imc_obs_free.
 * All functions return IMC_OK (0) or a negative error code; imc_last_error() gives the message
 * (thread-local).  There is NO CPU fallback: without a HIP device every compute entry point
 * fails with IMC_ERR_NODEVICE.
 *
 * Threads / processes: no HIP call is made before the first compute/create call, and the device
 * context is re-created lazily per PID, so Forwarders may be built inside multiprocessing
 * children as mcmc.py:112-121 does.  One mutex serialises the ENQUEUE of calls (plan lookup, parameter staging, kernel
 * launches); a synchronous call waits for its results without it, so other threads queue their evaluations behind it
 * (two synchronous calls on the SAME chunk list share a plan's result slots and run one after the other; a chunk is not
 * freed, and no plan released, while somebody waits on it).  The O(L) host work of imc_obs_create* (validation,
 * dictionary training, encoding) runs outside the mutex.
 *
 * Environment (diagnostics only, read when a plan is built / the context is created):
 *   IMC_DEBUG=1        print the planner's cost estimates to stderr
 *   IMC_FORCE_LEVEL=k  pin the pair-dictionary level index (experiments; ignored if the level does not fit)
 *   IMC_GRAPH=1        replay each plan as a hipGraph (measured: no gain)
 *   IMC_GUARD=1        test facility: every device buffer is its own virtual-memory mapping that ends flush against an
 *                      unmapped range, so any access past a buffer's end faults at once (tests/test_gpu_guard.py)
 *   IMC_DEBUG_R1=1     print where each operator segment was certified rank one (hand-off checkpoints) to stderr
 *   IMC_DEBUG_HOST=1   print the host time per phase of the synchronous entry points (plan, stage, enqueue, wait)
 *   A/B switches for measurements (the default is the faster setting): IMC_TABLE_PAIRS=0 (k_zpropagate4's table one
 *   dictionary depth per launch instead of two), IMC_PACK_TABLE=0 (the mat-vec chain reads the 16-padded operator
 *   table), IMC_Z4_STREAM / IMC_BLOCKED / IMC_RANK1 (see the setters below), IMC_DICT_MIN_COUNT=n and
 *   IMC_DICT_MAX_DEPTH=d (dictionary training: occurrences a pair needs; cap on a token's depth),
 *   IMC_FUSE_HEAD=0 (the parameters fetched by a k_stage_params launch and the raw operators built by k_z4_raw, instead
 *   of by the evaluation's first table launch / by each workgroup of a small launch), IMC_TABLE_TRIPLES=0|1 (three
 *   dictionary depths per table launch, one wavefront per token: never / always; default: up to 12 states, where it is 2 % faster), IMC_FUSE_TAIL=0|1|2 (the chunk's
 *   last workgroup finishes the chunk instead of stitch launches: never / chunks of at most four workgroups (default) /
 *   wherever a chunk is at most 32 workgroups; 2 changes results by re-association only),
 *   IMC_XCD_AFFINE=0 (k_zpropagate4 on a plain (blocks, parameter sets) grid: by default its one-dimensional grid puts all
 *   workgroups of a parameter set on one XCD - two or four for the last B % 8 sets - so that a set's operator table is read
 *   through one L2 instead of all eight; placement only, every bit of every result is the same)
 *
 * fork(): a child forked AFTER the parent's first imc_* call gets IMC_ERR_HIP from every call (HIP state does not
 * survive fork and nothing of the parent's is touched); fork chain processes first, or use the spawn start method.
 */
#ifndef IMCOAL_FWD_H
#define IMCOAL_FWD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    IMC_OK = 0,
    IMC_ERR_ARG = -1,      /* bad shape / null pointer / unsupported size */
    IMC_ERR_SYMBOL = -2,   /* observation symbol >= nsym */
    IMC_ERR_HIP = -3,      /* HIP runtime error */
    IMC_ERR_OOM = -4,      /* host or device allocation failed */
    IMC_ERR_IO = -5,       /* cannot read / parse the observation file */
    IMC_ERR_NODEVICE = -6  /* no usable gfx950 device */
};

typedef struct imc_obs imc_obs; /* one alignment file ("chunk"), resident in HBM */

/* Library / device management ------------------------------------------------------------ */
const char *imc_version(void);
const char *imc_last_error(void);
int imc_device_count(void);          /* number of HIP devices, 0 if none (never an error) */
int imc_set_device(int device);      /* device used by subsequent calls of this process (default:
                                        the calling thread's current HIP device) */

/* Observations: replaces Forwarder.__init__, src/IMCoalHMM/hmm.py:12-16
 * (read text -> np.int32[L] -> ziphmm.preprocess_raw_observations(obs, NSYM)). -------------- */
int imc_obs_create(const uint8_t *sym, size_t L, int nsym, imc_obs **out);
int imc_obs_create_i32(const int32_t *sym, size_t L, int nsym, imc_obs **out);   /* hmm.py:14 dtype */
/* File of whitespace-separated decimal tokens, the format written by
 * scripts/prepare-alignments.py:92-105 and read at hmm.py:13-14. */
int imc_obs_create_from_text(const char *path, int nsym, imc_obs **out);   /* also accepts imc_write_cache files */
size_t imc_obs_length(const imc_obs *obs);
int imc_obs_nsym(const imc_obs *obs);
/* Length of the chunk's pair-compressed token stream (the `new_obs` of hmm.py:16) at the deepest
 * dictionary level whose alphabet is <= alphabet_limit; *alphabet_used gets that alphabet
 * (`new_nsyms`).  Returns the raw length when the chunk is not compressed. */
size_t imc_obs_compressed_length(const imc_obs *obs, int alphabet_limit, int *alphabet_used);
/* Joint re-compression of a data set that arrived as several chunks.  The pair dictionary of an alphabet is trained
 * by the first sufficiently long chunk created and shared by every later one (one operator table per evaluation for
 * all of them); with many chunks - 256 x 1e6 columns in BASELINE config[4] - that training sample is a single chunk
 * and the dictionary stays small (62 columns per token there, against 110+ from a 3e7-column sample).  This call
 * trains a new dictionary on a sample drawn evenly from all the given chunks, re-encodes them with it and makes it
 * the alphabet's dictionary for chunks created later.  Results change by re-association only.  Cached plans of the
 * chunks are dropped; the call synchronises the device.  Chunks too short to compress are left alone.  The
 * counterpart in the reference is ziphmm's per-Forwarder preprocessing (hmm.py:16), which compresses every file on
 * its own; imcoalhmm_amd.Likelihood calls this once for its forwarders.  The chunks must stay alive for the duration
 * of the call (do not free them from another thread meanwhile); evaluations on other threads simply wait. */
int imc_obs_recompress(imc_obs *const *chunks, int n_chunks);

/* The other two results of ziphmm.preprocess_raw_observations (hmm.py:16), at the deepest dictionary level whose
 * alphabet is <= alphabet_limit:  imc_obs_dictionary = `sym2pair` (token nsym + k is left[k] followed by right[k];
 * call with left = right = NULL to get *alphabet_used first), imc_obs_tokens = `new_obs` (copied back from the
 * device as 16-bit ids; call with tokens = NULL to get *length first).  Nothing in the reference reads them outside
 * hmm.py; they exist for inspection and tests. */
int imc_obs_dictionary(const imc_obs *obs, int alphabet_limit, uint16_t *left, uint16_t *right, size_t capacity,
                       int *alphabet_used);
int imc_obs_tokens(const imc_obs *obs, int alphabet_limit, uint16_t *tokens, size_t capacity, size_t *length,
                   int *alphabet_used);
int imc_obs_free(imc_obs *obs);

/* Host-side ingestion helpers (no GPU needed) -------------------------------------------- */
/* Parse a text or cache file into caller memory.  Call with sym_out = NULL to get *length first. */
int imc_read_observations(const char *path, int nsym, uint8_t *sym_out, size_t capacity, size_t *length);
/* Packed on-disk cache of a chunk: 2 bits per column when nsym <= 4 (8x smaller than the text format). */
int imc_write_cache(const char *path, const uint8_t *sym, size_t L, int nsym);
/* The pairwise symbol rule of scripts/prepare-alignments.py:99-105 on two aligned sequences:
 * 2 = either base not in ACGT (case-insensitive), 0 = equal, 1 = different. */
int imc_encode_pairwise(const char *seq1, const char *seq2, size_t L, uint8_t *sym_out);

/* Forward log-likelihood: replaces Forwarder.forward -> ziphmm.zip_forward,
 * src/IMCoalHMM/hmm.py:19-21, and the sum over forwarders at likelihood.py:33.
 * out_loglik[0] = sum_f log P(chunk_f | pi,T,E). */
int imc_forward(const imc_obs *const *chunks, int n_chunks, int N, int S,
                const double *pi, const double *T, const double *E, double *out_loglik);

/* B parameter sets against the same chunks in one pass (the evaluation streams of
 * genetic_algorithm.py:818-831, mcmc.py:165-174).  pis[B][N], Ts[B][N][N], Es[B][N][S];
 * out_logliks[B] (summed over chunks). */
int imc_forward_batch(const imc_obs *const *chunks, int n_chunks, int B, int N, int S,
                      const double *pis, const double *Ts, const double *Es, double *out_logliks);

/* Same, but also returns every chunk's own value: out_per_chunk[B][n_chunks]. */
int imc_forward_batch_per_chunk(const imc_obs *const *chunks, int n_chunks, int B, int N, int S,
                                const double *pis, const double *Ts, const double *Es,
                                double *out_per_chunk);

/* Multi-GPU building block: partial sums stay on the device.  d_out_partial is a DEVICE pointer
 * to B doubles, written on `hip_stream` (a hipStream_t; NULL is the DEFAULT stream, as in every HIP call - torch's
 * default stream has the handle 0 - and NOT the library's own non-blocking stream: the caller's next use of
 * d_out_partial on the stream it named is ordered behind this call, and only that).  The call returns after
 * enqueueing (no stream synchronisation: the parameters are staged through two pinned slots, so only a third call in
 * flight waits for the first one's upload).  The caller then all-reduces d_out_partial over ranks (RCCL sum).
 * Parameters are still host pointers.  Calls on one set of chunks share the plan's device buffers: a call on another
 * stream than its predecessor's (including the synchronous entry points, which use the library's stream) first waits
 * for the predecessor, so calls may move between streams; they run one after the other. */
int imc_forward_batch_device(const imc_obs *const *chunks, int n_chunks, int B, int N, int S,
                             const double *pis, const double *Ts, const double *Es,
                             double *d_out_partial, void *hip_stream);

/* One long alignment split over GPUs ("a single long file across GPUs"): every rank holds a
 * contiguous slice as its own chunk and exports the slice's STATE instead of a log-likelihood.
 *   as_operator == 0: the forward vector after the chunk, started from pi (the first slice):
 *       out_state[B][n_chunks][N], out_exp[B][n_chunks];         a_i = out_state * 2^out_exp
 *   as_operator != 0: the chunk's exact transfer operator, every column of the chunk (the first included)
 *       acting as C_o = diag(E[:,o]) T':   out_state[B][n_chunks][N][N] row-major P[i][c], out_exp[B][n_chunks][N]
 *       (one exponent per column c);                              P_ic = out_state * 2^out_exp[c]
 * The caller gathers the ranks' states (N^2 + N numbers each) and applies them in order; the log-likelihood of
 * the whole alignment is log sum_i (P_last ... P_1 a)_i.  There is no counterpart in the reference (its
 * Likelihood only sums independent files, likelihood.py:33); empty chunks are refused. */
int imc_forward_state(const imc_obs *const *chunks, int n_chunks, int as_operator, int B, int N, int S,
                      const double *pis, const double *Ts, const double *Es, double *out_state, int *out_exp);

/* Tuning / measurement ------------------------------------------------------------------ */
/* Target segment length, in stream elements (columns, or tokens on the compressed path), for the
 * parallel-in-time split (0 = automatic). */
int imc_set_segment_length(size_t columns);
/* Stream and kernel selection.
 *   1 (default): chunks are pair-compressed at creation and evaluated from the token stream whenever the
 *      operator table fits LDS for the model's N; the kernel variant is chosen from the segment/chunk ratio.
 *   0: raw symbol stream only (also skips the compression of chunks created while it is 0); kernel chosen
 *      automatically.
 *   2 / 3: as 1 but pin the kernel: 2 = one vector per lane group (k_zpropagate; N > 64: the mat-vec chain kernel
 *      k_big_vector, one segment per chunk), 3 = register-blocked operator per 16-lane row (k_zpropagate2, N <= 24;
 *      24 < N: the MFMA GEMM-chain kernels).
 *   4 / 5: as 0 but pin the kernel: 4 = k_propagate (T' in registers, LDS broadcast), 5 = k_zpropagate2. */
int imc_set_compression(int mode);
/* Forget the per-process pair dictionaries: the next sufficiently long chunk trains a new one.
 * Chunks that already exist keep the dictionary they were encoded with. */
int imc_dictionary_reset(void);
/* Kernel timing with HIP events on the launch stream.  After imc_profile_enable(1) every
 * propagate/stitch launch is bracketed by events; imc_profile_read synchronises, returns the
 * accumulated device milliseconds and launch counts since the last reset, and resets.  ms_propagate is the
 * propagate kernel(s) of a call; k_zpropagate4's separate table launches (k_z4_raw, k_z4_level) lie in front of
 * the bracket, so the figure is the scan launch's own duration. */
int imc_profile_enable(int on);
int imc_profile_read(double *ms_propagate, double *ms_stitch, uint64_t *n_propagate, uint64_t *n_stitch);
/* Rank-one hand-off of the GEMM-chain kernels (long segments, N > 24): a segment's transfer operator is tested at a
 * fixed schedule of checkpoints (token counts, from ~8k alignment columns, each ~1.25x the previous); at the first one
 * where every column has collapsed onto one direction (component-wise, within 2^-42) the rest of the segment
 * propagates that vector instead of the N x N operator.  The decision is taken per segment from the data of the
 * evaluation alone - no state is carried between calls, so identical calls return identical bits.  For the last
 * synchronous imc_forward* call: how many operator segments were tested and how many were certified.
 * IMC_RANK1=0 in the environment switches the hand-off off. */
int imc_last_rank1(uint64_t *checked, uint64_t *collapsed);
/* Switch the hand-off on (default) or off for plans built from now on (A/B comparisons, tests). */
int imc_set_rank1_handoff(int on);
/* Register-blocked kernel for N <= 24: 4 (default) = the fp64-MFMA scan (v_mfma_f64_4x4x4: the DP units are the same,
 * the matrix form needs a quarter of the issue slots and no cross-lane moves) with its operator table in LDS
 * (k_zpropagate3) or, where the planner's estimate says it pays, with a hybrid table - a dictionary level of up to
 * 4096 tokens in global memory / L2, the hottest operators cached in LDS (k_zpropagate4); 3 = LDS table only;
 * 5 = hybrid wherever a level beyond LDS exists (tests); 2 = k_zpropagate2, the VALU / DPP form (A/B measurements).
 * IMC_BLOCKED=2..5 in the environment selects the variant at start-up. */
int imc_set_blocked_kernel(int variant);
/* Where k_zpropagate4 takes a step's operator from: -1 (default) = automatic - STREAMED from the global table (every
 * step's operands are requested one step ahead and arrive from L1 / L2 / the Infinity Cache; nothing is cached in LDS)
 * whenever a launch holds several parameter sets (XCD-affine grid, IMC_XCD_AFFINE) or one set's table of at most 32 MB
 * (always, up to 24 states), the HYBRID form (hottest operators cached in LDS, the others streamed) otherwise;
 * 0 = always hybrid; 1 = always streamed (A/B measurements, tests).
 * IMC_Z4_STREAM=-1|0|1 in the environment sets the mode at start-up. */
int imc_set_table_streaming(int mode);
/* Description of the last launch plan, out8[0..7] = segments, vectors, per-column segment length,
 * executed vector-columns (per-column kernel), token segment length, executed vector-tokens (token
 * kernel), tokens in the compressed streams, token alphabet. */
int imc_last_plan(uint64_t *out8);
/* Propagate kernels launched by the last forward call, '+'-joined, e.g. "k_zpropagate2<5>[tokens]". */
const char *imc_last_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* IMCOAL_FWD_H */
