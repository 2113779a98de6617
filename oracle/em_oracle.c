/* em_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see em_oracle.h).
 *
 * Plain-C restatement of the reference's abundance path on flat CSR arrays.  Every function cites the
 * reference lines it follows (/root/reference/src/...).  Written from the behaviour, not copied: the
 * reference works on ragged inta/inta2 arrays and globals, this file on CSR and explicit arguments.
 */
#define _POSIX_C_SOURCE 200809L
#include "em_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOWEST      (-1E308)    /* emsar.h:20 */
#define NEAR_LOWEST (-9.9E307)  /* emsar.h:21 */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* lambda_c = E_c * sum_t theta_t over occurrences; -1 if any theta < 0      (emsar_functions.c:2966-2975) */
static inline double row_lambda(const uint64_t *row_ptr, const int32_t *col_idx, const double *E,
                                const double *theta, int64_t c) {
    double s = 0;
    for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) {
        double v = theta[col_idx[k]];
        if (v < 0) return -1;
        s += v;
    }
    return E[c] * s;
}

/* one row's term of Fp; *bad set when the reference would return NEAR_LOWEST  (emsar_functions.c:2950-2960) */
static inline double row_logp(const uint64_t *row_ptr, const int32_t *col_idx, const int32_t *R,
                              const double *E, const double *theta, int64_t c, int *bad) {
    if (E[c] == 0) return 0;
    double lamb = row_lambda(row_ptr, col_idx, E, theta, c);
    int32_t r = R ? R[c] : 1;
    if (lamb == 0) {
        if (r == 0) return 0;
        *bad = 1;
        return 0;
    }
    if (lamb < 0) { *bad = 1; return 0; }
    return (double)r * log(lamb) - lamb;
}

double oracle_loglik(int64_t n_rows, const uint64_t *row_ptr, const int32_t *col_idx,
                     const int32_t *R, const double *E, const double *theta) {
    double sum = 0;
    int bad = 0;
    for (int64_t c = 0; c < n_rows; c++) {
        sum += row_logp(row_ptr, col_idx, R, E, theta, c, &bad);
        if (bad) return NEAR_LOWEST;
    }
    if (sum < NEAR_LOWEST) sum = NEAR_LOWEST;
    return sum;
}

void oracle_den(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                const double *E, double *den) {
    memset(den, 0, sizeof(double) * (size_t)n_tx);
    for (int64_t c = 0; c < n_rows; c++) {
        if (E[c] == 0) continue;
        for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) den[col_idx[k]] += E[c];
    }
}

void oracle_ieuma(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                  const double *L, double *ieuma) {
    /* compute_iEUMA: every cid, also those left out of the sets  (emsar_functions.c:3224-3231) */
    memset(ieuma, 0, sizeof(double) * (size_t)n_tx);
    for (int64_t c = 0; c < n_rows; c++)
        for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) ieuma[col_idx[k]] += L[c];
}

/* ------------------------------------------------------------------------------------------------
 * EM for the same objective:
 *   E-step  w_c = R_c / S_c,  S_c = sum_t m_ct theta_t         (rows with E_c == 0 dropped)
 *   M-step  theta_t <- theta_t * (sum_c m_ct w_c) / den_t
 * Stationary points of this map are the stationary points of F: dF/dtheta_t = sum_c m_ct (R_c/S_c - E_c).
 * ---------------------------------------------------------------------------------------------- */
double oracle_em_step(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                      const int32_t *R, const double *E, const double *den,
                      const double *theta_in, double *theta_out, int n_threads) {
    double *acc = (double *)calloc((size_t)n_tx, sizeof(double));
    double ll = 0;
#ifdef _OPENMP
    int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    int nt = 1;
    (void)n_threads;
#endif
    if (nt <= 1) {
        for (int64_t c = 0; c < n_rows; c++) {
            if (E[c] == 0) continue;
            int32_t r = R ? R[c] : 1;
            if (r == 0) continue;
            double s = 0;
            for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) s += theta_in[col_idx[k]];
            if (!(s > 0)) continue;
            double w = (double)r / s;
            ll += (double)r * log(s);
            for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) acc[col_idx[k]] += w;
        }
    } else {
#ifdef _OPENMP
        double *priv = (double *)calloc((size_t)n_tx * (size_t)nt, sizeof(double));
#pragma omp parallel num_threads(nt) reduction(+ : ll)
        {
            double *a = priv + (size_t)omp_get_thread_num() * (size_t)n_tx;
#pragma omp for schedule(static)
            for (int64_t c = 0; c < n_rows; c++) {
                if (E[c] == 0) continue;
                int32_t r = R ? R[c] : 1;
                if (r == 0) continue;
                double s = 0;
                for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) s += theta_in[col_idx[k]];
                if (!(s > 0)) continue;
                double w = (double)r / s;
                ll += (double)r * log(s);
                for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) a[col_idx[k]] += w;
            }
#pragma omp for schedule(static)
            for (int32_t t = 0; t < n_tx; t++) {
                double s = 0;
                for (int i = 0; i < nt; i++) s += priv[(size_t)i * (size_t)n_tx + (size_t)t];
                acc[t] = s;
            }
        }
        free(priv);
#endif
    }
    for (int32_t t = 0; t < n_tx; t++) theta_out[t] = den[t] > 0 ? theta_in[t] * acc[t] / den[t] : 0.0;
    free(acc);
    return ll;
}

static double max_rel_delta(int32_t n_tx, const double *a, const double *b, double abs_floor) {
    double d = 0;
    for (int32_t t = 0; t < n_tx; t++) {
        double x = fabs(a[t] - b[t]) / (fabs(b[t]) + abs_floor);
        if (x > d) d = x;
    }
    return d;
}

int oracle_em_solve(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                    const int32_t *R, const double *E, const oracle_em_params *p,
                    double *theta_out, oracle_em_stats *st) {
    double t0 = now_s();
    size_t T = (size_t)n_tx;
    double *den = (double *)malloc(T * sizeof(double));
    double *th0 = (double *)malloc(T * sizeof(double));
    double *th1 = (double *)malloc(T * sizeof(double));
    double *th2 = (double *)malloc(T * sizeof(double));
    double *thx = (double *)malloc(T * sizeof(double));
    if (!den || !th0 || !th1 || !th2 || !thx) return -1;
    oracle_den(n_rows, n_tx, row_ptr, col_idx, E, den);
    /* uniform interior start on every transcript that the likelihood sees; 0 elsewhere
     * (the reference leaves such tids at their random start; SURVEY.md section 9 defines them as 0) */
    for (size_t t = 0; t < T; t++) th0[t] = den[t] > 0 ? 1.0 : 0.0;
    int iters = 0, conv = 0;
    double delta = INFINITY;
    double stepmax = 1.0, stepmin = 1.0; /* SQUAREM works with alpha = -steplength; we keep s = -alpha >= 1 */
    const double mstep = 4.0;
    while (iters < p->max_iter) {
        if (!p->accel) {
            oracle_em_step(n_rows, n_tx, row_ptr, col_idx, R, E, den, th0, th1, p->n_threads);
            iters++;
            delta = max_rel_delta(n_tx, th0, th1, p->abs_floor);
            memcpy(th0, th1, T * sizeof(double));
            if (delta < p->tol) { conv = 1; break; }
            continue;
        }
        /* SQUAREM cycle (Varadhan & Roland 2008, scheme S3) with the likelihood safeguard:
         * the by-product of pass k is sum R log S at its INPUT, so we know ll(th0), ll(th1), ll(thx). */
        double ll0 = oracle_em_step(n_rows, n_tx, row_ptr, col_idx, R, E, den, th0, th1, p->n_threads);
        (void)ll0;
        iters++;
        delta = max_rel_delta(n_tx, th0, th1, p->abs_floor);
        if (delta < p->tol) { memcpy(th0, th1, T * sizeof(double)); conv = 1; break; }
        double ll1 = oracle_em_step(n_rows, n_tx, row_ptr, col_idx, R, E, den, th1, th2, p->n_threads);
        iters++;
        double sr2 = 0, sv2 = 0;
        for (size_t t = 0; t < T; t++) {
            double r = th1[t] - th0[t], v = (th2[t] - th1[t]) - r;
            sr2 += r * r;
            sv2 += v * v;
        }
        double s = sv2 > 0 ? sqrt(sr2 / sv2) : 1.0;
        if (s < stepmin) s = stepmin;
        if (s > stepmax) s = stepmax;
        int extrap = s > 1.0 + 1e-2;
        if (extrap) {
            for (size_t t = 0; t < T; t++) {
                double r = th1[t] - th0[t], v = (th2[t] - th1[t]) - r;
                double x = th0[t] + 2.0 * s * r + s * s * v;
                thx[t] = (x > 0 && th2[t] > 0) ? x : th2[t]; /* stay in the interior; zeros stay zero */
            }
            double llx = oracle_em_step(n_rows, n_tx, row_ptr, col_idx, R, E, den, thx, th0, p->n_threads);
            iters++;
            /* sum_c E_c S_c is not conserved by the extrapolated point, so compare full F: F = ll - sum theta*den */
            double pen_x = 0, pen_1 = 0;
            for (size_t t = 0; t < T; t++) { pen_x += thx[t] * den[t]; pen_1 += th1[t] * den[t]; }
            if (!(llx - pen_x >= ll1 - pen_1)) { /* reject: fall back to the plain EM point */
                memcpy(th0, th2, T * sizeof(double));
                if (s >= stepmax) stepmax = fmax(1.0, stepmax / mstep);
                s = 1.0;
            }
        } else {
            memcpy(th0, th2, T * sizeof(double));
        }
        if (s >= stepmax) stepmax *= mstep;
    }
    memcpy(theta_out, th0, T * sizeof(double));
    if (st) {
        /* F with the reference's definition, E terms included */
        double *Ed = (double *)E;
        st->loglik = oracle_loglik(n_rows, row_ptr, col_idx, R, Ed, theta_out);
        st->iters = iters;
        st->converged = conv;
        st->final_delta = delta;
        st->seconds = now_s() - t0;
    }
    free(den); free(th0); free(th1); free(th2); free(thx);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Connected sets.  The reference does a recursive DFS from each cid in order (emsar_main.c:414-416,
 * emsar_functions.c:2234-2259); set ids are therefore numbered by the smallest member cid that is
 * eligible to seed.  A multi-tid row with L < EUMAcut never joins (CS=-1) and does not connect tids.
 * Union-find gives the same partition; numbering is recovered by scanning cids in order.
 * ---------------------------------------------------------------------------------------------- */
static int32_t uf_find(int32_t *p, int32_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

int32_t oracle_components(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                          const double *L, double *eumacut, int32_t max_ntid, int32_t *CS, int32_t *TS) {
    int32_t *par = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tx);
    int32_t *root_sid = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tx);
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tx);
    int32_t n_sets;
    for (;;) {
        for (int32_t t = 0; t < n_tx; t++) par[t] = t;
        for (int64_t c = 0; c < n_rows; c++) {
            uint64_t b = row_ptr[c], e = row_ptr[c + 1];
            if (e - b > 1 && L[c] < *eumacut) continue;                        /* emsar_functions.c:2242 */
            for (uint64_t k = b + 1; k < e; k++) {
                int32_t a = uf_find(par, col_idx[b]), d = uf_find(par, col_idx[k]);
                if (a != d) par[d] = a;
            }
        }
        for (int32_t t = 0; t < n_tx; t++) { root_sid[t] = -1; cnt[t] = 0; TS[t] = -1; }
        n_sets = 0;
        for (int64_t c = 0; c < n_rows; c++) {
            uint64_t b = row_ptr[c], e = row_ptr[c + 1];
            if (e == b || (e - b > 1 && L[c] < *eumacut)) { CS[c] = -1; continue; }
            int32_t r = uf_find(par, col_idx[b]);
            if (root_sid[r] < 0) root_sid[r] = n_sets++;
            CS[c] = root_sid[r];
        }
        int too_big = 0;
        for (int32_t t = 0; t < n_tx; t++) {
            int32_t r = uf_find(par, t);
            if (root_sid[r] >= 0) { TS[t] = root_sid[r]; if (++cnt[r] > max_ntid) too_big = 1; }
        }
        if (!too_big) break;
        *eumacut += 2.0;                                                       /* emsar.h:18, emsar_main.c:418 */
    }
    free(par); free(root_sid); free(cnt);
    return n_sets;
}

/* ------------------------------------------------------------------------------------------------
 * The reference's solver, one set at a time  (emsar_functions.c:3033-3126).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const uint64_t *row_ptr; const int32_t *col_idx; const int32_t *R; const double *E;
    const int64_t *sc_ptr; const int64_t *sc;      /* rows of each set, in cid order        (SC) */
    const int64_t *st_ptr; const int32_t *st;      /* unique tids of each set, first-seen order (ST) */
    double eps, eps_step; int32_t max_niter, max_nloop;
    double *fpkm;
    int32_t n_sets, n_threads;
    int64_t sweeps;
    pthread_mutex_t mu;
} mle_ctx;

static double set_F(const mle_ctx *m, int32_t sid) {
    /* Fp over SC[sid]  (emsar_functions.c:2946-2964) */
    double sum = 0;
    for (int64_t i = m->sc_ptr[sid]; i < m->sc_ptr[sid + 1]; i++) {
        int bad = 0;
        sum += row_logp(m->row_ptr, m->col_idx, m->R, m->E, m->fpkm, m->sc[i], &bad);
        if (bad) return NEAR_LOWEST;
    }
    if (sum < NEAR_LOWEST) sum = NEAR_LOWEST;
    return sum;
}

static int64_t mle_one_set(mle_ctx *m, int32_t sid) {
    int64_t nc = m->sc_ptr[sid + 1] - m->sc_ptr[sid];
    int64_t nt = m->st_ptr[sid + 1] - m->st_ptr[sid];
    const int64_t *C = m->sc + m->sc_ptr[sid];
    const int32_t *Tt = m->st + m->st_ptr[sid];
    double *fpkm = m->fpkm;
    int64_t sweeps = 0;
    /* no reads anywhere in the set -> all zero                                 (3053-3059) */
    long sumr = 0;
    for (int64_t i = 0; i < nc; i++) sumr += m->R ? m->R[C[i]] : 1;
    if (sumr == 0) { for (int64_t i = 0; i < nt; i++) fpkm[Tt[i]] = 0; return 0; }
    /* one segment, one transcript -> closed form                              (3061-3066) */
    if (nc == 1 && nt == 1) {
        fpkm[Tt[0]] = (double)(m->R ? m->R[C[0]] : 1) / m->E[C[0]];
        return 0;
    }
    double *step = (double *)malloc(sizeof(double) * (size_t)nt);
    const double acc = 2.0;
    int nloop = 0, notconv;
    do {
        int32_t niter = 0;
        notconv = 0;
        for (int64_t i = 0; i < nt; i++) fpkm[Tt[i]] = rand() / (RAND_MAX + 1.0) * (100.0 - 0.0) + 0.0; /* 3079 */
        for (int64_t i = 0; i < nt; i++) step[i] = 100.0;
        double pre, maxstep;
        do {
            niter++; sweeps++;
            pre = set_F(m, sid);
            maxstep = 0;
            for (int64_t i = 0; i < nt; i++) {
                int32_t tid = Tt[i];
                if (step[i] < m->eps_step) continue;
                double cand[5] = {0, -step[i] * acc * 2, -step[i], step[i], step[i] * acc * 2};   /* 3094 */
                int best = -1;
                double maxF = LOWEST;
                for (int j = 0; j < 5; j++) {
                    fpkm[tid] = fpkm[tid] + cand[j];
                    double F = set_F(m, sid);
                    fpkm[tid] = fpkm[tid] - cand[j];
                    if (F > maxF) { maxF = F; best = j; }
                }
                if (best == 0) step[i] = step[i] / acc;
                else if (best == 2 || best == 3) fpkm[tid] = fpkm[tid] + cand[best];
                else { fpkm[tid] = fpkm[tid] + cand[best]; step[i] = step[i] * acc; }
                if (step[i] > maxstep) maxstep = step[i];
            }
            if (niter > m->max_niter) { notconv = 1; break; }                                      /* 3118 */
        } while (set_F(m, sid) - pre >= m->eps || maxstep > m->eps_step);                           /* 3119 */
        nloop++;
    } while (notconv == 1 && nloop < m->max_nloop);
    free(step);
    return sweeps;
}

typedef struct { mle_ctx *m; int32_t start, end; } mle_range_arg;

static void *mle_range(void *a) {
    mle_range_arg *r = (mle_range_arg *)a;
    int64_t s = 0;
    for (int32_t sid = r->start; sid <= r->end; sid++) s += mle_one_set(r->m, sid);
    pthread_mutex_lock(&r->m->mu);
    r->m->sweeps += s;
    pthread_mutex_unlock(&r->m->mu);
    return NULL;
}

int64_t oracle_mle_pattern_search(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                  const int32_t *R, const double *E, const int32_t *CS, int32_t n_sets,
                                  double eps, double eps_step, int32_t max_niter, int32_t max_nloop,
                                  uint32_t seed, int32_t n_threads, double *fpkm_out) {
    if (n_sets <= 0) return 0;
    /* SC / ST  (generate_SC_ST, emsar_functions.c:2303-2356): rows in cid order; tids de-duplicated in
     * order of first appearance while walking the set's rows in cid order. */
    int64_t *sc_ptr = (int64_t *)calloc((size_t)n_sets + 1, sizeof(int64_t));
    int64_t *st_ptr = (int64_t *)calloc((size_t)n_sets + 1, sizeof(int64_t));
    int64_t n_in = 0;
    for (int64_t c = 0; c < n_rows; c++) if (CS[c] >= 0) { sc_ptr[CS[c] + 1]++; n_in++; }
    for (int32_t s = 0; s < n_sets; s++) sc_ptr[s + 1] += sc_ptr[s];
    int64_t *sc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_in > 0 ? n_in : 1));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_sets);
    memcpy(fill, sc_ptr, sizeof(int64_t) * (size_t)n_sets);
    for (int64_t c = 0; c < n_rows; c++) if (CS[c] >= 0) sc[fill[CS[c]]++] = c;
    int32_t *seen = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tx);
    for (int32_t t = 0; t < n_tx; t++) seen[t] = -1;
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tx);
    int64_t nst = 0;
    for (int32_t s = 0; s < n_sets; s++) {
        st_ptr[s] = nst;
        for (int64_t i = sc_ptr[s]; i < sc_ptr[s + 1]; i++) {
            int64_t c = sc[i];
            for (uint64_t k = row_ptr[c]; k < row_ptr[c + 1]; k++) {
                int32_t t = col_idx[k];
                if (seen[t] != s) { seen[t] = s; st[nst++] = t; }
            }
        }
    }
    st_ptr[n_sets] = nst;

    mle_ctx m = {row_ptr, col_idx, R, E, sc_ptr, sc, st_ptr, st, eps, eps_step, max_niter, max_nloop,
                 fpkm_out, n_sets, n_threads, 0, PTHREAD_MUTEX_INITIALIZER};
    if (seed) srand(seed);                                  /* emsar_main.c:441 (0 = keep the stream going) */
    /* static contiguous partition of [0,max_sid]           (emsar_functions.c:2984-3000) */
    int32_t max_sid = n_sets - 1, P = n_threads < 1 ? 1 : n_threads;
    int32_t inc = max_sid / P + 1;
    int32_t nrange = 0;
    mle_range_arg *args = (mle_range_arg *)malloc(sizeof(mle_range_arg) * (size_t)P);
    for (int32_t i = 0; i < P; i++) {
        int32_t s = i * inc;
        if (s > max_sid) break;
        int32_t e = s + max_sid / P;
        if (e > max_sid) e = max_sid;
        args[nrange++] = (mle_range_arg){&m, s, e};
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)P);
    for (int32_t i = 0; i < nrange - 1; i++) pthread_create(&th[i], NULL, mle_range, &args[i]);
    mle_range(&args[nrange - 1]);                            /* the caller runs the last range itself */
    for (int32_t i = 0; i < nrange - 1; i++) pthread_join(th[i], NULL);
    free(th); free(args); free(seen); free(st); free(fill); free(sc); free(sc_ptr); free(st_ptr);
    return m.sweeps;
}

void oracle_srand(uint32_t seed) { srand(seed); }

/* print_FPKMfinal arithmetic  (emsar_functions.c:3176-3207) */
void oracle_fpkm_table(int32_t n_tx, int32_t n_round, const double *rounds, const double *ieuma,
                       int64_t total_read_count, double *mean, double *sd, double *ireadcount,
                       int32_t *ireadcount_int, double *tpm) {
    double total = 0;
    for (int32_t t = 0; t < n_tx; t++)
        for (int32_t r = 0; r < n_round; r++) total += rounds[(size_t)r * (size_t)n_tx + (size_t)t] / n_round;
    for (int32_t t = 0; t < n_tx; t++) {
        double sum = 0;
        for (int32_t r = 0; r < n_round; r++) sum += rounds[(size_t)r * (size_t)n_tx + (size_t)t];
        double mu = sum / n_round, sq = 0;
        for (int32_t r = 0; r < n_round; r++) sq += pow(rounds[(size_t)r * (size_t)n_tx + (size_t)t] - mu, 2);
        mean[t] = mu;
        sd[t] = sqrt(sq / (n_round - 1)) / n_round;          /* divides by n, not sqrt(n): kept (3200) */
        double ir = (ieuma[t] / 1E3) * mu * ((double)total_read_count / 1E6);
        ireadcount[t] = ir;
        ireadcount_int[t] = (ir - (int)ir >= 0.5) ? (int)ir + 1 : (int)ir;   /* Round_off (3215-3217) */
        tpm[t] = mu * 1E6 / total;
    }
}


/* ---- read -> segment collapse (update_ReadCounts, emsar_functions.c:838-943) ---- */
typedef struct { const uint64_t *rp; const int32_t *sorted; } collapse_ctx;
static const collapse_ctx *g_cc;                      /* qsort has no context argument in C11 */
static int collapse_cmp(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    uint64_t lx = g_cc->rp[x + 1] - g_cc->rp[x], ly = g_cc->rp[y + 1] - g_cc->rp[y];
    if (lx != ly) return lx < ly ? -1 : 1;
    const int32_t *p = g_cc->sorted + g_cc->rp[x], *q = g_cc->sorted + g_cc->rp[y];
    for (uint64_t k = 0; k < lx; k++) if (p[k] != q[k]) return p[k] < q[k] ? -1 : 1;
    return x < y ? -1 : x > y;                         /* equal tuples: by row id, so the first occurrence leads */
}
static int cmp_i64(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? -1 : x > y; }

int64_t oracle_collapse_rows(int64_t n_rows, const uint64_t *row_ptr, const int32_t *col_idx, const int32_t *row_weight,
                             uint64_t *row_ptr_out, int32_t *col_idx_out, int64_t *weight_out, int32_t *row_map_out) {
    const uint64_t nnz = row_ptr[n_rows];
    int32_t *sorted = (int32_t *)malloc(sizeof(int32_t) * (nnz ? nnz : 1));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_rows ? n_rows : 1));
    int64_t *leader = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_rows ? n_rows : 1));   /* first occurrence of each row's tuple */
    int64_t *firsts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_rows ? n_rows : 1));
    if (!sorted || !order || !leader || !firsts) { free(sorted); free(order); free(leader); free(firsts); return -1; }
    int64_t n_act = 0;
    for (int64_t r = 0; r < n_rows; r++) {
        uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        for (uint64_t k = b; k < e; k++) {              /* insertion with >= : the reference's own order of equal ids */
            int32_t v = col_idx[k];
            uint64_t j = k;
            while (j > b && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; j--; }
            sorted[j] = v;
        }
        leader[r] = -1;
        if (e > b && (!row_weight || row_weight[r] > 0)) order[n_act++] = r;
    }
    collapse_ctx cc = {row_ptr, sorted};
    g_cc = &cc;
    qsort(order, (size_t)n_act, sizeof(int64_t), collapse_cmp);
    int64_t n_seg = 0;
    for (int64_t i = 0; i < n_act;) {
        int64_t j = i + 1;
        while (j < n_act) {
            int64_t x = order[i], y = order[j];
            uint64_t lx = row_ptr[x + 1] - row_ptr[x];
            if (lx != row_ptr[y + 1] - row_ptr[y] || memcmp(sorted + row_ptr[x], sorted + row_ptr[y], lx * sizeof(int32_t)) != 0) break;
            j++;
        }
        for (int64_t k = i; k < j; k++) leader[order[k]] = order[i];
        firsts[n_seg++] = order[i];
        i = j;
    }
    qsort(firsts, (size_t)n_seg, sizeof(int64_t), cmp_i64);             /* segments in order of first occurrence */
    int64_t *seg_of_first = order;                                        /* reuse: row id -> segment id for leaders */
    for (int64_t r = 0; r < n_rows; r++) seg_of_first[r] = -1;
    uint64_t o = 0;
    for (int64_t s = 0; s < n_seg; s++) {
        int64_t r = firsts[s];
        seg_of_first[r] = s;
        row_ptr_out[s] = o;
        uint64_t len = row_ptr[r + 1] - row_ptr[r];
        memcpy(col_idx_out + o, sorted + row_ptr[r], len * sizeof(int32_t));
        o += len;
        weight_out[s] = 0;
    }
    row_ptr_out[n_seg] = o;
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t s = leader[r] < 0 ? -1 : seg_of_first[leader[r]];
        if (s >= 0) weight_out[s] += row_weight ? row_weight[r] : 1;
        if (row_map_out) row_map_out[r] = (int32_t)s;
    }
    free(sorted); free(order); free(leader); free(firsts);
    return n_seg;
}
