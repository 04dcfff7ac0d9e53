// model_host.hpp - host-side (pi, T) of the pairwise CoalHMMs: SURVEY.md section 8f rank 3 ("host-side (pi,T,E)
// construction throughput"), on the CPU as north_star asks.  Included by imcoal_fwd.hip; no device code, no HIP call.
//
// What it replaces: /root/reference/src/IMCoalHMM/transitions.py:204-248 (CTMCSystem -> joint matrix J -> pi, T) with the
// matrix exponentials of CTMC.py:39-51.  The ALGORITHM is the vector recursion of imcoalhmm_amd/models.py
// (hmm_transitions_batch: only the B -> B, B -> L, B -> E, L -> L and L -> E blocks of every `through` matrix matter) - this
// file restates it for state spaces small enough that numpy's per-call overhead, not arithmetic, is the cost (the 4- and
// 15-state spaces of the isolation models: ~40 small numpy calls per interval; a population of 64 parameter points took
// 10 ms to build for a 2.8 ms device pass).  models.py keeps the numpy path for the larger spaces (BLAS wins there) and as
// the cross-check: tests/test_models_cpu.py compares the two paths and both against outputs of the reference's own classes.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace imc_model {

// ---- dense helpers on row-major n x n matrices (n <= ~100) ----
static inline void matmul(const double *A, const double *B, double *C, int n)
{
    for (int i = 0; i < n; ++i) {
        double *c = C + (size_t)i * n;
        for (int j = 0; j < n; ++j) c[j] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double a = A[(size_t)i * n + k];
            if (a == 0.0) continue;
            const double *b = B + (size_t)k * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
    }
}

static inline double norm1(const double *A, int n)
{
    double best = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += std::fabs(A[(size_t)i * n + j]);
        best = std::max(best, s);
    }
    return best;
}

// Solve M X = R in place (X overwrites R) by LU with partial pivoting; M is destroyed.  false: singular.
static inline bool solve(double *M, double *R, int n)
{
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double big = std::fabs(M[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(M[(size_t)i * n + k]) > big) { big = std::fabs(M[(size_t)i * n + k]); piv = i; }
        if (big == 0.0) return false;
        if (piv != k)
            for (int j = 0; j < n; ++j) { std::swap(M[(size_t)k * n + j], M[(size_t)piv * n + j]); std::swap(R[(size_t)k * n + j], R[(size_t)piv * n + j]); }
        const double inv = 1.0 / M[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = M[(size_t)i * n + k] * inv;
            if (f == 0.0) continue;
            for (int j = k + 1; j < n; ++j) M[(size_t)i * n + j] -= f * M[(size_t)k * n + j];
            for (int j = 0; j < n; ++j) R[(size_t)i * n + j] -= f * R[(size_t)k * n + j];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        const double inv = 1.0 / M[(size_t)k * n + k];
        for (int j = 0; j < n; ++j) {
            double s = R[(size_t)k * n + j];
            for (int i = k + 1; i < n; ++i) s -= M[(size_t)k * n + i] * R[(size_t)i * n + j];
            R[(size_t)k * n + j] = s * inv;
        }
    }
    return true;
}

// exp(A) by scaling and squaring with Pade approximants (Higham, SIAM J. Matrix Anal. Appl. 26(4), 2005, Algorithm 2.3:
// degree 3, 5, 7 or 9 while the 1-norm is below theta_m, else [13/13] after scaling below theta_13 = 5.37;
// scipy.linalg.expm - what CTMC.py:39-51 and models.py call - chooses among the same approximants, so the two agree
// to rounding).  work: 7 n^2 doubles.
static inline bool expm(const double *A, double *out, int n, std::vector<double> &work)
{
    static const double b3[4] = {120.0, 60.0, 12.0, 1.0};
    static const double b5[6] = {30240.0, 15120.0, 3360.0, 420.0, 30.0, 1.0};
    static const double b7[8] = {17297280.0, 8648640.0, 1995840.0, 277200.0, 25200.0, 1512.0, 56.0, 1.0};
    static const double b9[10] = {17643225600.0, 8821612800.0, 2075673600.0, 302702400.0, 30270240.0, 2162160.0, 110880.0, 3960.0, 90.0, 1.0};
    static const double b[14] = {64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
                                 129060195264000.0, 10559470521600.0, 670442572800.0, 33522128640.0, 1323241920.0,
                                 40840800.0, 960960.0, 16380.0, 182.0, 1.0};
    const size_t nn = (size_t)n * n;
    work.resize(8 * nn);
    double *As = work.data(), *A2 = As + nn, *A4 = A2 + nn, *A6 = A4 + nn, *U = A6 + nn, *V = U + nn, *Tm = V + nn, *A8 = Tm + nn;
    const double nrm = norm1(A, n);
    auto finish = [&]() {                       // (V - U) R = V + U
        for (size_t k = 0; k < nn; ++k) { Tm[k] = V[k] - U[k]; out[k] = V[k] + U[k]; }
        return solve(Tm, out, n);
    };
    if (nrm <= 2.097847961257068) {             // low degrees, no scaling
        const int m = nrm <= 1.495585217958292e-2 ? 3 : nrm <= 2.539398330063230e-1 ? 5 : nrm <= 9.504178996162932e-1 ? 7 : 9;
        const double *c = m == 3 ? b3 : m == 5 ? b5 : m == 7 ? b7 : b9;
        matmul(A, A, A2, n);
        if (m >= 5) matmul(A2, A2, A4, n);
        if (m >= 7) matmul(A4, A2, A6, n);
        if (m >= 9) matmul(A6, A2, A8, n);
        // W = sum of odd coefficients times even powers (then U = A W); V = sum of even coefficients times even powers
        for (size_t k = 0; k < nn; ++k) {
            double w = c[3] * A2[k], v = c[2] * A2[k];
            if (m >= 5) { w += c[5] * A4[k]; v += c[4] * A4[k]; }
            if (m >= 7) { w += c[7] * A6[k]; v += c[6] * A6[k]; }
            if (m >= 9) { w += c[9] * A8[k]; v += c[8] * A8[k]; }
            Tm[k] = w;
            V[k] = v;
        }
        for (int i = 0; i < n; ++i) { Tm[(size_t)i * n + i] += c[1]; V[(size_t)i * n + i] += c[0]; }
        matmul(A, Tm, U, n);
        return finish();
    }
    int s = 0;
    if (nrm > 5.371920351148152) s = std::max(0, (int)std::ceil(std::log2(nrm / 5.371920351148152)));
    const double scale = std::ldexp(1.0, -s);
    for (size_t k = 0; k < nn; ++k) As[k] = A[k] * scale;
    matmul(As, As, A2, n);
    matmul(A2, A2, A4, n);
    matmul(A4, A2, A6, n);
    // U = As (A6 (b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I)
    for (size_t k = 0; k < nn; ++k) Tm[k] = b[13] * A6[k] + b[11] * A4[k] + b[9] * A2[k];
    matmul(A6, Tm, V, n);                                   // (V as scratch)
    for (size_t k = 0; k < nn; ++k) V[k] += b[7] * A6[k] + b[5] * A4[k] + b[3] * A2[k];
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] += b[1];
    matmul(As, V, U, n);
    // V = A6 (b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I
    for (size_t k = 0; k < nn; ++k) Tm[k] = b[12] * A6[k] + b[10] * A4[k] + b[8] * A2[k];
    matmul(A6, Tm, V, n);
    for (size_t k = 0; k < nn; ++k) V[k] += b[6] * A6[k] + b[4] * A4[k] + b[2] * A2[k];
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] += b[0];
    if (!finish()) return false;
    for (int q = 0; q < s; ++q) {
        matmul(out, out, Tm, n);
        std::memcpy(out, Tm, nn * sizeof(double));
    }
    return true;
}

struct Structure {
    int n_intervals;
    const int32_t *space_size;   // [n_intervals]
    const int32_t *cls_off;      // [3 n_intervals + 1]: B, L, E index lists of interval i's space at 3 i, 3 i + 1, 3 i + 2
    const int32_t *cls_idx;
    const int32_t *piece_q;      // [n_intervals - 1] which rate matrix
    const int32_t *piece_proj;   // [n_intervals - 1] which projection, or -1
    int n_q;
    const int32_t *q_size, *q_off;       // [n_q] matrix order and offset (doubles) inside one system's Q block
    int q_stride;                        // doubles of Q per system
    const int32_t *proj_off;             // [n_proj] offset (doubles) of projection k: space_size[i] x space_size[i + 1], row-major
    const double *proj;
};

// One system: rate matrices Q, interval lengths dt, start vector over interval 0's space -> pi[n], T[n][n].
// Returns "" or an error text.
static inline std::string transitions_one(const Structure &st, const double *Q, const double *dt, const double *start,
                                          double *pi, double *T)
{
    const int n = st.n_intervals;
    auto cls = [&](int i, int k, int &count) { count = st.cls_off[3 * i + k + 1] - st.cls_off[3 * i + k]; return st.cls_idx + st.cls_off[3 * i + k]; };
    std::vector<double> work, scaled;
    // ---- through matrices: expm(Q dt) (de-duplicated on (matrix, dt) as CTMC.py:39-51 caches) times the projection ----
    std::vector<std::vector<double>> through((size_t)std::max(0, n - 1));
    std::vector<int> through_cols((size_t)std::max(0, n - 1));
    for (int i = 0; i + 1 < n; ++i) {
        const int rows = st.space_size[i], cols = st.space_size[i + 1], q = st.piece_q[i];
        if (st.q_size[q] != rows) return "a rate matrix does not match its interval's state space";
        int same = -1;
        for (int k = 0; k < i && same < 0; ++k)
            if (st.piece_q[k] == q && dt[k] == dt[i] && st.piece_proj[k] == st.piece_proj[i]) same = k;
        through_cols[i] = cols;
        if (same >= 0) { through[i] = through[same]; continue; }
        const size_t nn = (size_t)rows * rows;
        scaled.resize(nn);
        const double *Qm = Q + st.q_off[q];
        for (size_t k = 0; k < nn; ++k) scaled[k] = Qm[k] * dt[i];
        std::vector<double> ex(nn);
        if (!expm(scaled.data(), ex.data(), rows, work)) return "expm: singular Pade denominator";
        if (st.piece_proj[i] < 0) {
            if (cols != rows) return "state space changes without a projection";
            through[i].swap(ex);
        } else {                                   // 0/1 projection: columns of the next interval's space
            const double *P = st.proj + st.proj_off[st.piece_proj[i]];
            through[i].assign((size_t)rows * cols, 0.0);
            for (int r = 0; r < rows; ++r)
                for (int k = 0; k < rows; ++k) {
                    const double a = ex[(size_t)r * rows + k];
                    if (a == 0.0) continue;
                    const double *p = P + (size_t)k * cols;
                    double *o = through[i].data() + (size_t)r * cols;
                    for (int c = 0; c < cols; ++c) o[c] += a * p[c];
                }
        }
    }
    // ---- begin[i]: distribution over the B class at the start of interval i ----
    std::vector<std::vector<double>> begin((size_t)n);
    {
        int nb;
        const int32_t *B0 = cls(0, 0, nb);
        std::vector<char> inB((size_t)st.space_size[0], 0);
        for (int k = 0; k < nb; ++k) inB[B0[k]] = 1;
        for (int k = 0; k < st.space_size[0]; ++k)
            if (!inB[k] && start[k] != 0.0) return "the start distribution must be supported on the B class";
        begin[0].resize(nb);
        for (int k = 0; k < nb; ++k) begin[0][k] = start[B0[k]];
    }
    for (int i = 1; i < n; ++i) {
        int nb0, nb1;
        const int32_t *Bp = cls(i - 1, 0, nb0), *Bn = cls(i, 0, nb1);
        const double *M = through[i - 1].data();
        const int cols = through_cols[i - 1];
        begin[i].assign(nb1, 0.0);
        for (int a = 0; a < nb0; ++a) {
            const double w = begin[i - 1][a];
            if (w == 0.0) continue;
            const double *row = M + (size_t)Bp[a] * cols;
            for (int c = 0; c < nb1; ++c) begin[i][c] += w * row[Bn[c]];
        }
    }
    std::vector<double> J((size_t)n * n, 0.0);
    // diagonal (transitions.py:217-224); the i = 0 term applies interval 0's E indices to the columns of through_0,
    // exactly as the reference does
    for (int i = 0; i + 1 < n; ++i) {
        int nb, ne;
        const int32_t *Bi = cls(i, 0, nb), *Ee = cls(i == 0 ? 0 : i + 1, 2, ne);
        const double *M = through[i].data();
        const int cols = through_cols[i];
        double s = 0.0;
        for (int a = 0; a < nb; ++a) {
            const double *row = M + (size_t)Bi[a] * cols;
            double r = 0.0;
            for (int e = 0; e < ne; ++e) {
                if (Ee[e] >= cols) return "an end-state index exceeds the next interval's state space";
                r += row[Ee[e]];
            }
            s += begin[i][a] * r;
        }
        J[(size_t)i * n + i] = s;
    }
    {
        double s = 0.0;
        for (double v : begin[n - 1]) s += v;
        J[(size_t)(n - 1) * n + (n - 1)] = s;
    }
    // i < j (transitions.py:226-235): row i of V is interval i's vector over the L class of interval j
    std::vector<double> V, Vn, closing;
    for (int j = 1; j < n; ++j) {
        int nl, nbp;
        const int32_t *Lj = cls(j, 1, nl), *Bp = cls(j - 1, 0, nbp);
        if (j == 1) V.assign((size_t)(n - 1) * nl, 0.0);
        {   // V[j - 1] = begin[j - 1] through_{j-1}[B, L_j]
            const double *M = through[j - 1].data();
            const int cols = through_cols[j - 1];
            double *v = V.data() + (size_t)(j - 1) * nl;
            for (int c = 0; c < nl; ++c) v[c] = 0.0;
            for (int a = 0; a < nbp; ++a) {
                const double w = begin[j - 1][a];
                if (w == 0.0) continue;
                const double *row = M + (size_t)Bp[a] * cols;
                for (int c = 0; c < nl; ++c) v[c] += w * row[Lj[c]];
            }
        }
        if (j == n - 1) {                     // pseudo through matrix: every L state ends in E
            for (int i = 0; i < j; ++i) {
                double s = 0.0;
                for (int c = 0; c < nl; ++c) s += V[(size_t)i * nl + c];
                J[(size_t)i * n + j] = s;
            }
            break;
        }
        int ne, nln;
        const int32_t *En = cls(j + 1, 2, ne), *Ln = cls(j + 1, 1, nln);
        const double *M = through[j].data();
        const int cols = through_cols[j];
        closing.assign(nl, 0.0);
        for (int a = 0; a < nl; ++a) {
            const double *row = M + (size_t)Lj[a] * cols;
            double s = 0.0;
            for (int e = 0; e < ne; ++e) s += row[En[e]];
            closing[a] = s;
        }
        for (int i = 0; i < j; ++i) {
            double s = 0.0;
            for (int a = 0; a < nl; ++a) s += V[(size_t)i * nl + a] * closing[a];
            J[(size_t)i * n + j] = s;
        }
        Vn.assign((size_t)(n - 1) * nln, 0.0);
        for (int i = 0; i < j; ++i)
            for (int a = 0; a < nl; ++a) {
                const double w = V[(size_t)i * nl + a];
                if (w == 0.0) continue;
                const double *row = M + (size_t)Lj[a] * cols;
                double *o = Vn.data() + (size_t)i * nln;
                for (int c = 0; c < nln; ++c) o[c] += w * row[Ln[c]];
            }
        V.swap(Vn);
    }
    double total = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) J[(size_t)j * n + i] = J[(size_t)i * n + j];
    for (double v : J) total += v;
    if (!(std::fabs(total - 1.0) < 1.5e-7))            // numpy.testing.assert_almost_equal, 7 decimals (transitions.py:237)
        return "joint probabilities sum to " + std::to_string(total) + ", not 1";
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += J[(size_t)i * n + j];
        pi[i] = s;
        for (int j = 0; j < n; ++j) T[(size_t)i * n + j] = J[(size_t)i * n + j] / s;
    }
    return "";
}

}   // namespace imc_model
