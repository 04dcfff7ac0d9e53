/*
 * oracle/forward_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the HMM forward log-likelihood that IMCoalHMM obtains from the third-party
 * `ziphmm` module (reference call sites: src/IMCoalHMM/hmm.py:16 `preprocess_raw_observations`,
 * src/IMCoalHMM/hmm.py:20-21 `zip_forward`; summed over forwarders at
 * src/IMCoalHMM/likelihood.py:33).  `ziphmm` (PyPI "ziphmm", upstream birc-aeh/mini-ziphmm,
 * un-pinned in requirements.txt:4 / setup.py:27-30) is NOT vendored in /root/reference and is not
 * installable offline, and no reference test pins a forward value:
 *
 *        >>>  PARITY UNPINNED against ziphmm itself.  <<<
 *
 * What is restated here is (1) the mathematical definition the engine implements - the scaled
 * forward recursion of SURVEY.md section 3.5 - and (2) the published zipHMM algorithm
 * (Sand et al. 2013, "zipHMMlib"): byte-pair style compression of the observation sequence and a
 * forward pass over the compressed sequence with one N x N operator per (original or new) symbol.
 * (2) is an exact re-association of (1); tests/ check they agree to <= 1e-12 relative.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Layouts (all row-major float64, as produced by the reference's model layer):
 *   pi[N]                      transitions.py:244
 *   T[i*N + j] = P(next=j | cur=i), row-stochastic   transitions.py:241-248
 *   E[j*S + s] = P(symbol s | state j)               emissions.py:89-100 (s=2 "missing" column = 1.0)
 *   obs[t] in [0,S)            scripts/prepare-alignments.py:92-105
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * (1) Textbook scaled forward:
 *     a_0 = pi .* E[:,o_0];  c_0 = sum a_0;  a_0 /= c_0
 *     a_t = (T' a_{t-1}) .* E[:,o_t];  c_t = sum a_t;  a_t /= c_t
 *     loglik = sum_t log c_t
 * Returns -inf when some c_t == 0 (impossible sequence), NaN in -> NaN out.
 * ------------------------------------------------------------------------------------------ */
double orc_forward_scaled(int N, int S, const double *pi, const double *T, const double *E,
                          const uint8_t *obs, size_t L)
{
    if (L == 0) return 0.0;
    double *a = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    double *b = a + N;
    double ll = 0.0, c = 0.0;
    for (int j = 0; j < N; ++j) { a[j] = pi[j] * E[(size_t)j * S + obs[0]]; c += a[j]; }
    if (!(c > 0.0)) { free(a); return (c == 0.0) ? -INFINITY : NAN; }
    for (int j = 0; j < N; ++j) a[j] /= c;
    ll += log(c);
    for (size_t t = 1; t < L; ++t) {
        const int o = obs[t];
        for (int j = 0; j < N; ++j) b[j] = 0.0;
        for (int i = 0; i < N; ++i) {
            const double ai = a[i];
            const double *Ti = T + (size_t)i * N;
            for (int j = 0; j < N; ++j) b[j] += Ti[j] * ai;
        }
        c = 0.0;
        for (int j = 0; j < N; ++j) { b[j] *= E[(size_t)j * S + o]; c += b[j]; }
        if (!(c > 0.0)) { free(a); return (c == 0.0) ? -INFINITY : NAN; }
        const double r = 1.0 / c;
        for (int j = 0; j < N; ++j) a[j] = b[j] * r;
        ll += log(c);
    }
    free(a);
    return ll;
}

/* The same recursion for observations given as 16-bit symbols (alphabets beyond 256 symbols, e.g. the 257-symbol
 * quartet alphabet of scripts/prepare-alignments.py:186-190); statement for statement the function above. */
double orc_forward_scaled_u16(int N, int S, const double *pi, const double *T, const double *E,
                              const uint16_t *obs, size_t L)
{
    if (L == 0) return 0.0;
    double *a = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    double *b = a + N;
    double ll = 0.0, c = 0.0;
    for (int j = 0; j < N; ++j) { a[j] = pi[j] * E[(size_t)j * S + obs[0]]; c += a[j]; }
    if (!(c > 0.0)) { free(a); return (c == 0.0) ? -INFINITY : NAN; }
    for (int j = 0; j < N; ++j) a[j] /= c;
    ll += log(c);
    for (size_t t = 1; t < L; ++t) {
        const int o = obs[t];
        for (int j = 0; j < N; ++j) b[j] = 0.0;
        for (int i = 0; i < N; ++i) {
            const double ai = a[i];
            const double *Ti = T + (size_t)i * N;
            for (int j = 0; j < N; ++j) b[j] += Ti[j] * ai;
        }
        c = 0.0;
        for (int j = 0; j < N; ++j) { b[j] *= E[(size_t)j * S + o]; c += b[j]; }
        if (!(c > 0.0)) { free(a); return (c == 0.0) ? -INFINITY : NAN; }
        const double r = 1.0 / c;
        for (int j = 0; j < N; ++j) a[j] = b[j] * r;
        ll += log(c);
    }
    free(a);
    return ll;
}

/* Same recursion in long double with a compensated (Neumaier) log accumulator: the spot-check
 * arbiter for the fp64 implementations. */
double orc_forward_scaled_ld(int N, int S, const double *pi, const double *T, const double *E,
                             const uint8_t *obs, size_t L)
{
    if (L == 0) return 0.0;
    long double *a = (long double *)malloc(sizeof(long double) * 2 * (size_t)N);
    long double *b = a + N;
    long double sum = 0.0L, comp = 0.0L, c = 0.0L;
    for (int j = 0; j < N; ++j) { a[j] = (long double)pi[j] * E[(size_t)j * S + obs[0]]; c += a[j]; }
    if (!(c > 0.0L)) { free(a); return (c == 0.0L) ? -INFINITY : NAN; }
    for (int j = 0; j < N; ++j) a[j] /= c;
    sum = logl(c);
    for (size_t t = 1; t < L; ++t) {
        const int o = obs[t];
        for (int j = 0; j < N; ++j) b[j] = 0.0L;
        for (int i = 0; i < N; ++i) {
            const long double ai = a[i];
            const double *Ti = T + (size_t)i * N;
            for (int j = 0; j < N; ++j) b[j] += (long double)Ti[j] * ai;
        }
        c = 0.0L;
        for (int j = 0; j < N; ++j) { b[j] *= E[(size_t)j * S + o]; c += b[j]; }
        if (!(c > 0.0L)) { free(a); return (c == 0.0L) ? -INFINITY : NAN; }
        for (int j = 0; j < N; ++j) a[j] = b[j] / c;
        const long double x = logl(c);
        const long double t2 = sum + x;
        if (fabsl(sum) >= fabsl(x)) comp += (sum - t2) + x; else comp += (x - t2) + sum;
        sum = t2;
    }
    free(a);
    return (double)(sum + comp);
}

/* Sum over independent chunks, each restarted from pi (likelihood.py:33), left-to-right. */
double orc_forward_chunks(int N, int S, const double *pi, const double *T, const double *E,
                          const uint8_t *const *obs, const size_t *L, int n_chunks)
{
    double total = 0.0;
    for (int f = 0; f < n_chunks; ++f) total += orc_forward_scaled(N, S, pi, T, E, obs[f], L[f]);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * (2) zipHMM restatement.
 *
 * preprocess (the analogue of ziphmm.preprocess_raw_observations, hmm.py:16): position 0 is kept
 * raw (it is consumed by the pi .* E initialisation); on the rest, repeatedly take the most frequent
 * adjacent pair (a,b), give it a new symbol z and replace its non-overlapping occurrences
 * left-to-right, until the best pair occurs fewer than `min_count` times or `max_new` symbols were
 * added.  mini-ziphmm's exact stopping rule is unknown here (source absent): this follows the
 * published cost argument (a new symbol costs one N^3 product per evaluation and saves one N^2
 * mat-vec per occurrence).
 * ------------------------------------------------------------------------------------------ */
typedef struct orc_zip {
    int nsym;          /* original alphabet size */
    int new_nsyms;     /* enlarged alphabet size */
    int32_t *pair;     /* [2*new_nsyms]; for z >= nsym: (left,right); for z < nsym: (-1,-1) */
    int32_t *seq;      /* compressed sequence, seq[0] is the raw first symbol */
    size_t len;        /* compressed length */
    size_t raw_len;
} orc_zip;

orc_zip *orc_zip_preprocess(const uint8_t *obs, size_t L, int nsym, long min_count, int max_new)
{
    orc_zip *z = (orc_zip *)calloc(1, sizeof(orc_zip));
    z->nsym = nsym; z->raw_len = L;
    int cap = nsym + (max_new > 0 ? max_new : 0);
    z->pair = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    for (int i = 0; i < 2 * nsym; ++i) z->pair[i] = -1;
    z->seq = (int32_t *)malloc(sizeof(int32_t) * (L ? L : 1));
    for (size_t t = 0; t < L; ++t) z->seq[t] = obs[t];
    z->len = L; z->new_nsyms = nsym;
    if (L < 3) return z;
    long *count = (long *)malloc(sizeof(long) * (size_t)cap * cap);
    while (z->new_nsyms < cap) {
        const int A = z->new_nsyms;
        memset(count, 0, sizeof(long) * (size_t)A * A);
        /* count non-overlapping-safe: plain adjacent counts over positions >= 1 */
        for (size_t t = 1; t + 1 < z->len; ++t) count[(size_t)z->seq[t] * A + z->seq[t + 1]]++;
        long best = 0; int ba = -1, bb = -1;
        for (int a = 0; a < A; ++a)
            for (int b = 0; b < A; ++b)
                if (count[(size_t)a * A + b] > best) { best = count[(size_t)a * A + b]; ba = a; bb = b; }
        if (best < min_count || ba < 0) break;
        const int nz = z->new_nsyms++;
        z->pair[2 * nz] = ba; z->pair[2 * nz + 1] = bb;
        size_t w = 1;
        for (size_t t = 1; t < z->len;) {
            if (t + 1 < z->len && z->seq[t] == ba && z->seq[t + 1] == bb) { z->seq[w++] = nz; t += 2; }
            else { z->seq[w++] = z->seq[t]; t += 1; }
        }
        z->len = w;
    }
    free(count);
    return z;
}

size_t orc_zip_length(const orc_zip *z) { return z->len; }
size_t orc_zip_raw_length(const orc_zip *z) { return z->raw_len; }
int orc_zip_new_nsyms(const orc_zip *z) { return z->new_nsyms; }
void orc_zip_free(orc_zip *z) { if (z) { free(z->pair); free(z->seq); free(z); } }

/* zip_forward (the analogue of ziphmm.zip_forward, hmm.py:20-21).
 * Per original symbol s:  C_s[i][j] = E[i][s] * T[j][i]      (alpha' = C_s alpha)
 * Per new symbol z=(l,r): C_z = C_r * C_l, rescaled by its largest entry (log kept in lscale[z]).
 * Then the scaled mat-vec scan over the compressed sequence. */
double orc_zip_forward(const orc_zip *z, int N, const double *pi, const double *T, const double *E)
{
    if (z->raw_len == 0) return 0.0;
    const int S = z->nsym, A = z->new_nsyms;
    const size_t NN = (size_t)N * N;
    double *C = (double *)malloc(sizeof(double) * NN * A);
    double *lscale = (double *)calloc((size_t)A, sizeof(double));
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j)
                C[s * NN + (size_t)i * N + j] = E[(size_t)i * S + s] * T[(size_t)j * N + i];
    for (int q = S; q < A; ++q) {
        const double *Cl = C + (size_t)z->pair[2 * q] * NN, *Cr = C + (size_t)z->pair[2 * q + 1] * NN;
        double *Cq = C + (size_t)q * NN, mx = 0.0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double acc = 0.0;
                for (int k = 0; k < N; ++k) acc += Cr[(size_t)i * N + k] * Cl[(size_t)k * N + j];
                Cq[(size_t)i * N + j] = acc;
                if (acc > mx) mx = acc;
            }
        lscale[q] = lscale[z->pair[2 * q]] + lscale[z->pair[2 * q + 1]];
        if (mx > 0.0) {
            const double r = 1.0 / mx;
            for (size_t k = 0; k < NN; ++k) Cq[k] *= r;
            lscale[q] += log(mx);
        }
    }
    double *a = (double *)malloc(sizeof(double) * 2 * (size_t)N), *b = a + N;
    double ll = 0.0, c = 0.0;
    const int o0 = z->seq[0];
    for (int j = 0; j < N; ++j) { a[j] = pi[j] * E[(size_t)j * S + o0]; c += a[j]; }
    if (!(c > 0.0)) { free(a); free(C); free(lscale); return (c == 0.0) ? -INFINITY : NAN; }
    for (int j = 0; j < N; ++j) a[j] /= c;
    ll = log(c);
    for (size_t t = 1; t < z->len; ++t) {
        const int o = z->seq[t];
        const double *Co = C + (size_t)o * NN;
        c = 0.0;
        for (int i = 0; i < N; ++i) {
            double acc = 0.0;
            const double *row = Co + (size_t)i * N;
            for (int j = 0; j < N; ++j) acc += row[j] * a[j];
            b[i] = acc; c += acc;
        }
        if (!(c > 0.0)) { free(a); free(C); free(lscale); return (c == 0.0) ? -INFINITY : NAN; }
        const double r = 1.0 / c;
        for (int i = 0; i < N; ++i) a[i] = b[i] * r;
        ll += log(c) + lscale[o];
    }
    free(a); free(C); free(lscale);
    return ll;
}

/* ------------------------------------------------------------------------------------------
 * Timing helper for bench.py's cpu_baseline leg: n_chunks independent chunks evaluated on
 * `threads` host threads (OpenMP over chunks), each chunk restarted from pi.  `use_zip` selects
 * orc_zip_forward on pre-compressed chunks (zips[] non-NULL) instead of the textbook recursion.
 * Returns the left-to-right sum of the per-chunk log-likelihoods.
 * ------------------------------------------------------------------------------------------ */
double orc_forward_chunks_mt(int N, int S, const double *pi, const double *T, const double *E,
                             const uint8_t *const *obs, const size_t *L, orc_zip *const *zips,
                             int n_chunks, int threads, double *per_chunk)
{
    double *res = per_chunk ? per_chunk : (double *)malloc(sizeof(double) * (size_t)n_chunks);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int f = 0; f < n_chunks; ++f)
        res[f] = zips ? orc_zip_forward(zips[f], N, pi, T, E)
                      : orc_forward_scaled(N, S, pi, T, E, obs[f], L[f]);
    double total = 0.0;
    for (int f = 0; f < n_chunks; ++f) total += res[f];
    if (!per_chunk) free(res);
    return total;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
