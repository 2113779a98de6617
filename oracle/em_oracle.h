/* em_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libemsar_hip.so, emsar-hip) never links or calls it.
 *
 * It restates, in plain C on flat CSR arrays, the reference's abundance path:
 *   - the segment Poisson log-likelihood  Fp/lambdap      /root/reference/src/emsar_functions.c:2946-2975
 *   - the per-set coordinate pattern search  MLE           emsar_functions.c:3033-3126
 *   - its static pthread partition  run_MLE_threads        emsar_functions.c:2977-3026
 *   - connected sets  build_TC_from_CT_2 / propagate_2     emsar_functions.c:2201-2259, emsar_main.c:411-425
 *   - compute_iEUMA, print_FPKMfinal arithmetic            emsar_functions.c:3163-3232
 * and the EM for the same likelihood, which is what the HIP kernels run.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against the files the compiled
 * reference (oracle/_ref/emsar, built from /root/reference/src by oracle/Makefile) wrote for the
 * fixtures in tests/golden/.
 *
 * Matrix convention: row c = segment (or read), columns = transcript ids, a tid may
 * repeat inside a row (multiplicity m_ct, A2); R[c] observed count (NULL = all 1); E[c] = EUMAps[c]
 * (emsar_functions.c:3152); rows with E[c]==0 are outside the likelihood (emsar_functions.c:2952).
 */
#ifndef EM_ORACLE_H
#define EM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t max_iter;      /* cap on EM passes (a SQUAREM cycle costs 3) */
    int32_t accel;         /* 0 = plain EM, 1 = SQUAREM (S3 steplength, likelihood-safeguarded) */
    double  tol;           /* stop when max_t |dtheta_t| / (theta_t + abs_floor) < tol */
    double  abs_floor;     /* see tol; 1e-6 = the reference's %lf print quantum */
    int32_t n_threads;     /* OpenMP threads for the row loop (<=0: omp default) */
} oracle_em_params;

typedef struct {
    int32_t iters;         /* EM passes executed */
    int32_t converged;
    double  final_delta;
    double  loglik;        /* F at the returned theta (reference definition incl. the E terms) */
    double  seconds;
} oracle_em_stats;

/* F(theta) exactly as Fp/lambdap compute it over ALL rows with E != 0 (sets are block-diagonal, so the sum
 * over sets equals the sum over rows).  Returns -9.9e307 on an infeasible point like the reference. */
double oracle_loglik(int64_t n_rows, const uint64_t *row_ptr, const int32_t *col_idx,
                     const int32_t *R, const double *E, const double *theta);

/* den[t] = sum over rows with E != 0 of m_ct * E_c */
void oracle_den(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                const double *E, double *den);

/* one plain EM pass: theta_out = theta_in * acc / den ; returns sum_c R_c log S_c over rows with E!=0,R>0 */
double oracle_em_step(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                      const int32_t *R, const double *E, const double *den,
                      const double *theta_in, double *theta_out, int n_threads);

int oracle_em_solve(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                    const int32_t *R, const double *E, const oracle_em_params *p,
                    double *theta_out, oracle_em_stats *st);

/* Connected sets with the reference's numbering (set ids in order of first cid, emsar_main.c:414-416).
 * L = adjEUMA.  Multi-tid rows with L < *eumacut are left out (CS = -1).  If a set exceeds max_ntid tids
 * *eumacut is raised by 2 and everything is redone (emsar_main.c:417-423).  Returns number of sets. */
int32_t oracle_components(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                          const double *L, double *eumacut, int32_t max_ntid, int32_t *CS, int32_t *TS);

/* The reference's solver: per set, coordinate pattern search from a rand() start.  One "round".
 * CS/TS/n_sets from oracle_components.  seed feeds srand(); n_threads = the -p static partition.
 * Returns total number of sweeps executed (sum over sets), -1 on allocation failure. */
int64_t oracle_mle_pattern_search(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                  const int32_t *R, const double *E, const int32_t *CS, int32_t n_sets,
                                  double eps, double eps_step, int32_t max_niter, int32_t max_nloop,
                                  uint32_t seed, int32_t n_threads, double *fpkm_out);

/* iEUMA[t] = sum over ALL rows of m_ct * L_c  (emsar_functions.c:3218-3232) */
void oracle_ieuma(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                  const double *L, double *ieuma);

/* .fpkm columns 2..7 (emsar_functions.c:3176-3207): rounds is [n_round][n_tx] row-major. */
void oracle_fpkm_table(int32_t n_tx, int32_t n_round, const double *rounds, const double *ieuma,
                       int64_t total_read_count, double *mean, double *sd, double *ireadcount,
                       int32_t *ireadcount_int, double *tpm);

/* read -> segment collapse, the reference's update_ReadCounts (emsar_functions.c:838-943) on a flat read-level matrix:
 * every row's ids are sorted (insertion with >=: repeats are kept, emsar_functions.c:889); rows with the same sorted
 * tuple are one segment whose count is the sum of the members' weights (NULL = 1 each; weight 0 and empty rows are
 * skipped); segments are numbered in the order in which they are first met.  Arrays at worst-case size; returns the
 * number of segments, -1 if out of memory. */
int64_t oracle_collapse_rows(int64_t n_rows, const uint64_t *row_ptr, const int32_t *col_idx, const int32_t *row_weight,
                             uint64_t *row_ptr_out, int32_t *col_idx_out, int64_t *weight_out, int32_t *row_map_out);

#ifdef __cplusplus
}
#endif
#endif
