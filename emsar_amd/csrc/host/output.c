/* output.c -- the reference's three output files, byte for byte in layout:
 *   .fpkm               print_FPKMfinal      emsar_functions.c:3184,3207   "%s\t%lf\t%lf\t%lf\t%lf\t%d\t%lf\n"
 *   .fraglength_effect  print_FraglengthDist emsar_functions.c:2489-2490   "%d\t%d\t%lg\n"
 *   .segments           print_aEUMA_3        emsar_functions.c:2274-2297
 * Column order of .fpkm is a contract: the reference's Perl utilities read columns 0,1,4,6 (util/FPKM2gFPKM.pl:19).
 */
#include "emsar_host.h"

#include <stdio.h>

int emsar_write_fpkm(const char *path, const emsar_rsh *r, const double *mean, const double *sd, const double *ieuma,
                     const double *ireadcount, const int32_t *ireadcount_int, const double *tpm, int64_t *total_ir) {
    FILE *f = fopen(path, "w");
    if (!f) return EMSAR_HOST_ERR_IO;
    int64_t tot = 0;
    fprintf(f, "transcriptID\tFPKM\tsd.of.FPKM\teff.length\tiReadcount\tiReadcount.int\tTPM\n");
    for (int32_t t = 0; t < r->n_tx; t++) {
        tot += ireadcount_int[t];
        fprintf(f, "%s\t%lf\t%lf\t%lf\t%lf\t%d\t%lf\n", r->names[t], mean[t], sd[t], ieuma[t], ireadcount[t],
                ireadcount_int[t], tpm[t]);
    }
    if (total_ir) *total_ir = tot;
    return fclose(f) == 0 ? EMSAR_HOST_OK : EMSAR_HOST_ERR_IO;
}

int emsar_write_fraglength(const char *path, const emsar_rsh *r, const emsar_counts *c, const emsar_model *m) {
    FILE *f = fopen(path, "w");
    if (!f) return EMSAR_HOST_ERR_IO;
    fprintf(f, "Fragment.length\tObs.Counts\tnormalized.Fragment.length.sampling.prob\n");
    for (int i = 0; i < r->nfl; i++) {
        int fl = i + r->frag_min;
        fprintf(f, "%d\t%d\t%lg\n", fl, fl < c->n_frag ? c->frag_counts[fl] : 0, m->Wf[i]);
    }
    return fclose(f) == 0 ? EMSAR_HOST_OK : EMSAR_HOST_ERR_IO;
}

int emsar_write_segments(const char *path, const emsar_rsh *r, const emsar_counts *c, const emsar_model *m,
                         const double *mean_fpkm) {
    FILE *f = fopen(path, "w");
    if (!f) return EMSAR_HOST_ERR_IO;
    fprintf(f, "segment_id\tsequence_sharing_set_id\ttranscript_id\ttranscript_names\teff.length\tReadcount\texpected_Readcount\n");
    const double nm = (double)c->total_reads / 1E6;
    for (int64_t cid = 0; cid < r->n_rows; cid++) {
        uint64_t b = r->row_ptr[cid], e = r->row_ptr[cid + 1];
        fprintf(f, "c%lld\ts%d\t", (long long)cid, m->CS[cid]);
        for (uint64_t k = b; k < e; k++) fprintf(f, "%st%d", k > b ? "," : "", r->col_idx[k]);
        fprintf(f, "\t");
        for (uint64_t k = b; k < e; k++) fprintf(f, "%s%s", k > b ? "+" : "", r->names[r->col_idx[k]]);
        fprintf(f, "\t%lf", m->L[cid]);
        double expc = 0;
        for (uint64_t k = b; k < e; k++) expc += mean_fpkm[r->col_idx[k]] * (m->L[cid] / 1E3) * nm;   /* 2295 */
        fprintf(f, "\t%d\t%f\n", c->R[cid], expc);
    }
    return fclose(f) == 0 ? EMSAR_HOST_OK : EMSAR_HOST_ERR_IO;
}
