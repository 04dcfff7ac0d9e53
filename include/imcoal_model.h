/* imcoal_model.h - host-side helper of libimcoal_fwd.so for SURVEY.md section 8f rank 3 ("host-side (pi,T,E)
 * construction throughput").  NOT part of the forward boundary (include/imcoal_fwd.h is): no device is touched, the
 * call works on a machine without a GPU, and imcoalhmm_amd/models.py computes the same numbers with numpy when the
 * library is absent or the state spaces are large (more than 32 states), or when IMC_MODEL_NATIVE=0 is set.
 *
 * Replaces, for one or many parameter points at once, /root/reference/src/IMCoalHMM/transitions.py:204-248
 * (CTMCSystem -> joint matrix J -> initial distribution pi and transition matrix T) including the matrix exponentials of
 * CTMC.py:39-51.  All systems of a call share one STRUCTURE (the demographic model) and differ in rates, interval lengths
 * and start vector (the parameter point):
 *   n_intervals              HMM states = time intervals
 *   space_size[i]            size of the CTMC state space of interval i
 *   cls_off / cls_idx        for interval i the state indices of its classes B (neither locus coalesced), L (left only) and
 *                            E (both): cls_idx[cls_off[3 i + k] .. cls_off[3 i + k + 1]) for k = 0, 1, 2
 *   piece_q[i], piece_proj[i]  for i < n_intervals - 1: through_i = expm(Q[piece_q[i]] * dt[i]) (@ projection piece_proj[i]
 *                            unless it is -1); projection k is a row-major space_size[i] x space_size[i + 1] matrix at
 *                            proj + proj_off[k]
 *   q_size[k]                order of rate matrix k; one system's matrices lie back to back in Q (stride = sum of squares)
 *   Q, dt, start             [n_systems][...]: rate matrices, interval lengths (n_intervals - 1 each), start vector over
 *                            interval 0's space (supported on its B class)
 *   pi [n_systems][n], T [n_systems][n][n]   outputs, row-major
 *   n_threads                systems are dealt over this many threads (<= 1: the calling thread)
 * Returns IMC_OK or IMC_ERR_ARG (imc_last_error(): e.g. "joint probabilities sum to ..., not 1", the reference's assertion at
 * transitions.py:237). */
#ifndef IMCOAL_MODEL_H
#define IMCOAL_MODEL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int imc_model_transitions(int n_systems, int n_intervals, const int32_t *space_size, const int32_t *cls_off,
                          const int32_t *cls_idx, const int32_t *piece_q, const int32_t *piece_proj, int n_q,
                          const int32_t *q_size, int n_proj, const int32_t *proj_off, const double *proj, const double *Q,
                          const double *dt, const double *start, double *pi, double *T, int n_threads);
/* exp(A) of one row-major n x n matrix (scaling and squaring, [13/13] Pade): the start vectors of the models
 * (expm of the first epoch's rate matrix, isolation_model.py:112-115) without scipy's per-call overhead. */
int imc_model_expm(int n, const double *A, double *out);
#ifdef __cplusplus
}
#endif
#endif /* IMCOAL_MODEL_H */
