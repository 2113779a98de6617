/* model.c -- per-sample model preparation on the host: the O(segments * nFraglen) and O(nnz) glue that turns
 * read counts into the vectors the solver consumes.  Not accelerated by design (once per
 * sample, negligible next to parsing); it DEFINES E_c, so it follows the reference's arithmetic operation by
 * operation:
 *   transfer_fraglendist_to_Wf   emsar_functions.c:2503-2513    Wf[i] = count[i+min] / sum
 *   compute_adjEUMA              emsar_functions.c:2517-2523    L_c = sum_i Wf[i] * (double)EUMA_c[i]   (in order)
 *   construct_EUMAps             emsar_functions.c:3148-3154    E_c = L_c / 1E3 * (N / 1E6) * pow(10, DELTA)
 *   connected sets               emsar_functions.c:2201-2259 + the EUMAcut retry loop emsar_main.c:411-425
 */
#include "emsar_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int32_t uf_find(int32_t *p, int32_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

int emsar_model_wf(const emsar_rsh *r, const emsar_counts *c, double *wf) {
    /* the window [frag_min, frag_max] of the observed fragment-length histogram, normalised */
    double sum = 0;
    for (int i = 0; i < r->nfl; i++) {
        int fl = i + r->frag_min;
        wf[i] = (fl < c->n_frag) ? (double)c->frag_counts[fl] : 0.0;
        sum += wf[i];
    }
    if (!(sum > 0)) return EMSAR_HOST_ERR_FORMAT;
    for (int i = 0; i < r->nfl; i++) wf[i] /= sum;
    return EMSAR_HOST_OK;
}

int emsar_model_build(const emsar_rsh *r, const emsar_counts *c, int delta, double *eumacut_io, emsar_model **out,
                      char *err, size_t errlen) {
    return emsar_model_build_L(r, c, delta, eumacut_io, NULL, out, err, errlen);
}

int emsar_model_build_L(const emsar_rsh *r, const emsar_counts *c, int delta, double *eumacut_io, const double *L_pre,
                        emsar_model **out, char *err, size_t errlen) {
    *out = NULL;
    emsar_model *m = (emsar_model *)calloc(1, sizeof(*m));
    if (!m) return EMSAR_HOST_ERR_OOM;
    const int64_t C = r->n_rows;
    const int32_t T = r->n_tx, nfl = r->nfl;
    m->n_rows = C; m->n_tx = T; m->nfl = nfl;
    m->Wf = (double *)malloc(sizeof(double) * (size_t)nfl);
    m->L = (double *)malloc(sizeof(double) * (size_t)C);
    m->E = (double *)malloc(sizeof(double) * (size_t)C);
    m->E_solver = (double *)malloc(sizeof(double) * (size_t)C);
    m->CS = (int32_t *)malloc(sizeof(int32_t) * (size_t)C);
    m->TS = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    int32_t *par = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    int32_t *sid_of_root = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    int rc = EMSAR_HOST_OK;
    if (!m->Wf || !m->L || !m->E || !m->E_solver || !m->CS || !m->TS || !par || !sid_of_root || !cnt) { rc = EMSAR_HOST_ERR_OOM; goto done; }

    if (emsar_model_wf(r, c, m->Wf) != EMSAR_HOST_OK) {
        if (err) snprintf(err, errlen, "no read inside the fragment-length range [%d,%d]", r->frag_min, r->frag_max);
        rc = EMSAR_HOST_ERR_FORMAT; goto done;
    }

    const double scale_n = (double)c->total_reads / 1E6, scale_d = pow(10, delta);
    for (int64_t cid = 0; cid < C; cid++) {
        double a = 0;
        if (L_pre) a = L_pre[cid];                     /* computed on the device (emsar_hip_adj_euma), same arithmetic */
        else if (r->has_node[cid]) {
            const int32_t *e = r->euma + (size_t)cid * (size_t)nfl;
            for (int i = 0; i < nfl; i++) a += m->Wf[i] * (double)e[i];
        }
        m->L[cid] = a;
        m->E[cid] = a / 1E3 * scale_n * scale_d;
    }

    /* connected sets; multi-tid rows with L < EUMAcut join nothing; retry with a larger cut while a set is too big */
    for (;;) {
        for (int32_t t = 0; t < T; t++) par[t] = t;
        for (int64_t cid = T; cid < C; cid++) {
            uint64_t b = r->row_ptr[cid], e = r->row_ptr[cid + 1];
            if (e - b > 1 && m->L[cid] < *eumacut_io) continue;
            for (uint64_t k = b + 1; k < e; k++) {
                int32_t x = uf_find(par, r->col_idx[b]), y = uf_find(par, r->col_idx[k]);
                if (x != y) par[y] = x;
            }
        }
        for (int32_t t = 0; t < T; t++) { sid_of_root[t] = -1; cnt[t] = 0; m->TS[t] = -1; }
        m->n_sets = 0;
        for (int64_t cid = 0; cid < C; cid++) {                /* set ids in order of first cid (emsar_main.c:414-416) */
            uint64_t b = r->row_ptr[cid], e = r->row_ptr[cid + 1];
            if (e - b > 1 && m->L[cid] < *eumacut_io) { m->CS[cid] = -1; continue; }
            int32_t root = uf_find(par, r->col_idx[b]);
            if (sid_of_root[root] < 0) sid_of_root[root] = m->n_sets++;
            m->CS[cid] = sid_of_root[root];
        }
        int too_big = 0;
        for (int32_t t = 0; t < T; t++) {
            int32_t root = uf_find(par, t);
            m->TS[t] = sid_of_root[root];
            if (++cnt[root] > EMSAR_MAX_NTID_PER_SID) too_big = 1;
        }
        if (!too_big) break;
        *eumacut_io += EMSAR_EUMACUT_INCREMENT;
    }
    m->eumacut = *eumacut_io;
    for (int64_t cid = 0; cid < C; cid++) m->E_solver[cid] = m->CS[cid] >= 0 ? m->E[cid] : 0.0;
done:
    free(par); free(sid_of_root); free(cnt);
    if (rc != EMSAR_HOST_OK) { emsar_model_free(m); return rc; }
    *out = m;
    return EMSAR_HOST_OK;
}

void emsar_model_free(emsar_model *m) {
    if (!m) return;
    free(m->Wf); free(m->L); free(m->E); free(m->E_solver); free(m->CS); free(m->TS); free(m);
}

void emsar_mean_sd(int32_t n_tx, int32_t n_round, const double *rounds, double *mean, double *sd) {
    for (int32_t t = 0; t < n_tx; t++) {                         /* print_FPKMfinal, 3188-3200 */
        double s = 0;
        for (int32_t k = 0; k < n_round; k++) s += rounds[(size_t)k * (size_t)n_tx + (size_t)t];
        double mu = s / n_round, sq = 0;
        for (int32_t k = 0; k < n_round; k++) sq += pow(rounds[(size_t)k * (size_t)n_tx + (size_t)t] - mu, 2);
        mean[t] = mu;
        sd[t] = sqrt(sq / (n_round - 1)) / n_round;
    }
}
