/* emsar_hip_main.c -- command-line driver: the per-sample loop of the reference's main()
 * (/root/reference/src/emsar_main.c:380-488) with run_MLE_threads() replaced by the HIP library.
 *
 *   emsar-hip [options] -I index.rsh outdir outprefix alignmentfile            (single sample)
 *   emsar-hip [options] -M -I index.rsh outdir outprefix alignmentfilelist     (one alignment file per line)
 *
 * Same option letters as the reference for everything that reaches this path (emsar_main.c:100-214):
 *   -I rsh  -P  -s strand  -k max_repeat  -n rounds  -e tol  -i max_passes  -d delta  -g  -M  -S  -B  -q  -v  -p threads(ignored)
 * plus  --gpus N (devices used by -M, default all), --devices a,b,.. (one -M worker per entry; an id may repeat, so that
 *       several workers share one card), --device D (single sample), --plain (no SQUAREM), --stats-json FILE.
 * Not taken over: -x fasta (index build, emsar-build's job), -m positional bias
 * (unfinished in the reference, emsar_main.c:371), -F/-f (the reference overwrites both from the rsh header,
 * emsar_functions.c:1419-1420, so they have no effect with -I).
 *
 * Outputs: <outdir>/<prefix>.<i>.fpkm, .fraglength_effect and, with -g, .segments -- i = 0-based index of the
 * alignment file (emsar_main.c:459-469).  Column 3 (sd.of.FPKM) is 0: the EM is deterministic, the reference's
 * spread comes from its random restarts (documented deviation).
 *
 * -M: samples are independent (emsar_main.c:380-488 resets every count per file), so sample i runs on GPU
 * i mod G with one host thread per GPU; no collective.  The only cross-sample state of the reference, EUMAcut
 * (never reset, emsar_main.c:95,418), is carried in sample order.
 */
#include <getopt.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "../../../include/emsar_hip.h"
#include "emsar_host.h"

typedef struct {
    const char *rsh_path, *outdir, *prefix;
    char **aln; int n_aln;
    emsar_aln_opts ao;
    int n_round, delta, print_segments, verbose, accel, set_mode, device_collapse, no_deterministic;
    double tol, count_floor, zero_cut, abs_step; int max_iter;
    const char *stats_json;
    const char *rsh_cache;      /* NULL = off, "" = <rsh>.bin, else the path */
} config;

typedef struct {
    config *cfg; const emsar_rsh *rsh;
    int device, n_workers, worker;
    /* ordered hand-over of EUMAcut between samples */
    pthread_mutex_t *mu; pthread_cond_t *cv; int *next_model; double *eumacut;
    int *status;          /* per sample */
    emsar_em_stats *stats; /* per sample */
    double *parse_s;
    double *model_s, *host_s;   /* per sample: model preparation, and all host work of run_sample outside the library calls */
    int *go;              /* start gate: the workers wait until main() knows how many of them exist */
    emsar_aln_opts ao;    /* the job's alignment options, plus this worker's collapse device when --device-collapse */
    struct { emsar_hip_ctx *ctx; pthread_mutex_t mu; } cdev;
} worker_arg;

/* emsar_aln_opts.collapse on the GPU (--device-collapse): the parse threads of a worker hand their read-level rows to a context
 * of their own on the worker's device (the solve context is busy with the sample before); one call at a time */
static int cli_collapse(void *user, int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                        int64_t *n_unique, uint64_t *row_ptr_out, int32_t *col_idx_out, int32_t *weight_out) {
    worker_arg *w = (worker_arg *)user;
    pthread_mutex_lock(&w->cdev.mu);
    const int rc = emsar_hip_collapse_rows(w->cdev.ctx, n_rows, n_tx, row_ptr, col_idx, NULL, n_unique, row_ptr_out, col_idx_out, weight_out, NULL, NULL);
    pthread_mutex_unlock(&w->cdev.mu);
    return rc;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* The alignments of a sample are counted on a thread of their own: the first sample of a worker while its GPU context, the
 * device layout and the EUMA table are being set up, every later one while the sample before it is being solved. */
typedef struct { worker_arg *w; int i; emsar_counts *cnt; int rc; char err[512]; double secs; pthread_t th; int started; } parse_job;
static void *parse_main(void *a) {
    parse_job *j = (parse_job *)a;
    double t0 = now_s();
    j->rc = emsar_count_alignments(j->w->rsh, j->w->cfg->aln[j->i], &j->w->ao, &j->cnt, j->err, sizeof j->err);
    j->secs = now_s() - t0;
    return NULL;
}
static void parse_start(parse_job *j, worker_arg *w, int i) {
    memset(j, 0, sizeof *j);
    j->w = w; j->i = i;
    if (pthread_create(&j->th, NULL, parse_main, j) == 0) j->started = 1;
}
static void parse_wait(parse_job *j) {
    if (j->started) { pthread_join(j->th, NULL); j->started = 0; }
    else if (j->w) parse_main(j);                                    /* no thread to be had: count here */
    j->w = NULL;
}

static int run_sample(worker_arg *w, emsar_hip_ctx *ctx, int i, parse_job *parsed) {
    config *cfg = w->cfg; const emsar_rsh *r = w->rsh;
    char err[512] = "", path[4096];
    emsar_counts *cnt = parsed->cnt; emsar_model *m = NULL;
    double *theta = NULL, *rounds = NULL, *mean = NULL, *sd = NULL, *ieuma = NULL, *tpm = NULL, *ir = NULL, *den = NULL; int32_t *iri = NULL;
    int rc = parsed->rc;
    parsed->cnt = NULL;
    snprintf(err, sizeof err, "%s", parsed->err);
    w->parse_s[i] = parsed->secs;
    /* compute_adjEUMA (emsar_main.c:403): L = EUMA . Wf on the device, bit-identical to the host loop */
    double *Ldev = NULL;
    if (rc == 0) {
        double *wf = (double *)malloc(sizeof(double) * (size_t)r->nfl);
        Ldev = (double *)malloc(sizeof(double) * (size_t)(r->n_rows > 0 ? r->n_rows : 1));
        if (!wf || !Ldev) rc = EMSAR_HOST_ERR_OOM;
        else if (emsar_model_wf(r, cnt, wf) == EMSAR_HOST_OK) {
            int hrc = emsar_hip_adj_euma(ctx, wf, Ldev);
            if (hrc) { snprintf(err, sizeof err, "adj_euma: %s (%s)", emsar_hip_strerror(hrc), emsar_hip_last_error(ctx)); rc = EMSAR_HOST_ERR_IO; }
        } else { free(Ldev); Ldev = NULL; }            /* model_build reports the empty fragment-length window */
        free(wf);
    }
    /* Model preparation.  EUMAcut carries over in sample order (emsar_main.c:95,418: never reset), but it only ever changes
     * when a set exceeds 5000 transcripts (emsar_main.c:417-423).  So every worker builds its model at once, without the
     * lock, from the value it sees now; at its turn in the sample order it checks that the samples before it have left
     * that value in place and publishes its own (possibly raised) one.  Only if an earlier sample did raise the cut in
     * the meantime is the model rebuilt, at the turn, from the value that sample left. */
    double t_model = now_s();
    pthread_mutex_lock(w->mu);
    const double cut_seen = *w->eumacut;
    pthread_mutex_unlock(w->mu);
    double cut_mine = cut_seen;
    if (rc == 0) rc = emsar_model_build_L(r, cnt, cfg->delta, &cut_mine, Ldev, &m, err, sizeof err);
    pthread_mutex_lock(w->mu);
    while (*w->next_model != i) pthread_cond_wait(w->cv, w->mu);
    if (rc == 0 && *w->eumacut != cut_seen) {
        emsar_model_free(m); m = NULL;
        cut_mine = *w->eumacut;
        rc = emsar_model_build_L(r, cnt, cfg->delta, &cut_mine, Ldev, &m, err, sizeof err);
    }
    if (rc == 0) *w->eumacut = cut_mine;
    (*w->next_model)++;
    pthread_cond_broadcast(w->cv);
    pthread_mutex_unlock(w->mu);
    free(Ldev);
    w->model_s[i] = now_s() - t_model;
    if (rc) { fprintf(stderr, "alnfile[%d]=%s: %s\n", i, cfg->aln[i], err); goto done; }
    if (cfg->verbose > 0)
        fprintf(stdout, "alnfile[%d]=%s  reads=%lld (seen %lld, >k %lld, bad fraglen %lld, discrepant %lld, no segment %lld)  sets=%d  gpu=%d\n",
                i, cfg->aln[i], (long long)cnt->total_reads, (long long)cnt->reads_seen, (long long)cnt->reads_over_k,
                (long long)cnt->reads_bad_fraglen, (long long)cnt->reads_discrepant, (long long)cnt->reads_no_segment, m->n_sets, w->device);

    const size_t T = (size_t)r->n_tx;
    theta = (double *)malloc(T * 8); mean = (double *)malloc(T * 8); sd = (double *)malloc(T * 8);
    ieuma = (double *)malloc(T * 8); tpm = (double *)malloc(T * 8); ir = (double *)malloc(T * 8); iri = (int32_t *)malloc(T * 4);
    rounds = (double *)malloc(T * 8 * (size_t)cfg->n_round);
    if (!theta || !mean || !sd || !ieuma || !tpm || !ir || !iri || !rounds) { rc = EMSAR_HOST_ERR_OOM; goto done; }

    /* ---- the replaced call: run_MLE_threads() + construct_FPKMfinal, emsar_main.c:444-450 ---- */
    emsar_em_params p = {cfg->max_iter, cfg->accel, cfg->tol, 0.0, 0, cfg->set_mode, cfg->count_floor, cfg->zero_cut, cfg->abs_step};
    /* den_t = sum_c m_ct E_c in row order on the host: the device's own scatter adds with atomics, whose order (and with it
     * the last bits of den, of theta and now and then the sixth printed decimal) changes from run to run */
    double t_host = now_s();
    den = (double *)calloc(T, sizeof(double));
    if (!den) { rc = EMSAR_HOST_ERR_OOM; goto done; }
    /* ... and compute_iEUMA (emsar_functions.c:3218: iEUMA_t = sum of adjEUMA over the segments that hold t) in the same sweep, for the
     * same reason: the eff.length and iReadcount columns are then the same bytes in every run */
    memset(ieuma, 0, T * 8);
    for (int64_t c = 0; c < r->n_rows; c++) {
        const double e = m->E_solver[c], l = m->L[c];
        if (e != 0.0) for (uint64_t k = r->row_ptr[c]; k < r->row_ptr[c + 1]; k++) den[r->col_idx[k]] += e;
        if (l != 0.0) for (uint64_t k = r->row_ptr[c]; k < r->row_ptr[c + 1]; k++) ieuma[r->col_idx[k]] += l;
    }
    w->host_s[i] = w->model_s[i] + (now_s() - t_host);
    if ((rc = emsar_hip_upload_sample(ctx, cnt->R, m->E_solver, den)) ||
        (rc = emsar_hip_solve(ctx, &p, theta, &w->stats[i]))) {
        fprintf(stderr, "alnfile[%d]: %s (%s)\n", i, emsar_hip_strerror(rc), emsar_hip_last_error(ctx));
        goto done;
    }
    for (int k = 0; k < cfg->n_round; k++) memcpy(rounds + (size_t)k * T, theta, T * 8);   /* deterministic solver: rounds coincide */
    emsar_mean_sd(r->n_tx, cfg->n_round, rounds, mean, sd);
    /* ---- compute_iEUMA + print_FPKMfinal, emsar_main.c:454-460 ---- */
    if ((rc = emsar_hip_normalise(ctx, mean, ieuma, cnt->total_reads, tpm, ir, iri))) {
        fprintf(stderr, "alnfile[%d]: %s (%s)\n", i, emsar_hip_strerror(rc), emsar_hip_last_error(ctx));
        goto done;
    }
    int64_t tot = 0;
    t_host = now_s();
    snprintf(path, sizeof path, "%s/%s.%d.fpkm", cfg->outdir, cfg->prefix, i);
    if ((rc = emsar_write_fpkm(path, r, mean, sd, ieuma, ir, iri, tpm, &tot))) { fprintf(stderr, "can't write %s\n", path); goto done; }
    if (cfg->verbose > 0) fprintf(stdout, "Total inferred readcount=%lld\n", (long long)tot);
    snprintf(path, sizeof path, "%s/%s.%d.fraglength_effect", cfg->outdir, cfg->prefix, i);
    if ((rc = emsar_write_fraglength(path, r, cnt, m))) { fprintf(stderr, "can't write %s\n", path); goto done; }
    if (cfg->print_segments) {
        snprintf(path, sizeof path, "%s/%s.%d.segments", cfg->outdir, cfg->prefix, i);
        if ((rc = emsar_write_segments(path, r, cnt, m, mean))) { fprintf(stderr, "can't write %s\n", path); goto done; }
    }
    w->host_s[i] += now_s() - t_host;
    if (cfg->verbose > 0)
        fprintf(stdout, "Complete: %s/%s.%d.fpkm  (EM passes %d, converged %d, solve %.1f ms, logL %.6f)\n", cfg->outdir, cfg->prefix, i,
                w->stats[i].iters, w->stats[i].converged, w->stats[i].solve_ms, w->stats[i].loglik);
done:
    free(theta); free(rounds); free(mean); free(sd); free(ieuma); free(tpm); free(ir); free(iri); free(den);
    emsar_counts_free(cnt); emsar_model_free(m);
    return rc;
}

static void *worker_main(void *a) {
    worker_arg *w = (worker_arg *)a;
    emsar_hip_ctx *ctx = NULL;
    parse_job slot[2];                                               /* the sample in hand and the one being counted ahead */
    int c = 0;
    memset(slot, 0, sizeof slot);
    pthread_mutex_lock(w->mu);
    while (!*w->go) pthread_cond_wait(w->cv, w->mu);                 /* n_workers is final from here on */
    pthread_mutex_unlock(w->mu);
    w->ao = w->cfg->ao;
    int crc = 0;
    if (w->cfg->device_collapse) {                                    /* before the first parse thread starts: it needs the device */
        pthread_mutex_init(&w->cdev.mu, NULL);
        crc = emsar_hip_create(&w->cdev.ctx, w->device);
        if (crc == 0) { w->ao.collapse = cli_collapse; w->ao.collapse_user = w; }
        else fprintf(stderr, "GPU %d: %s (--device-collapse)\n", w->device, emsar_hip_strerror(crc));
    }
    if (w->worker < w->cfg->n_aln) parse_start(&slot[c], w, w->worker);
    int rc = crc ? crc : emsar_hip_create(&ctx, w->device);
    if (rc == 0) rc = emsar_hip_set_deterministic(ctx, !w->cfg->no_deterministic);     /* the streamed part of a solve: same bytes every run */
    if (rc == 0) rc = emsar_hip_upload_structure(ctx, w->rsh->n_rows, w->rsh->n_tx, w->rsh->row_ptr, w->rsh->col_idx, EMSAR_LAYOUT_AUTO);
    if (rc == 0) rc = emsar_hip_upload_euma(ctx, w->rsh->euma, w->rsh->nfl);     /* once per rsh: compute_adjEUMA runs on the device */
    if (rc) fprintf(stderr, "GPU %d: %s\n", w->device, emsar_hip_strerror(rc));
    for (int i = w->worker; i < w->cfg->n_aln; i += w->n_workers) {
        parse_job *cur = &slot[c];
        parse_wait(cur);
        if (i + w->n_workers < w->cfg->n_aln) parse_start(&slot[c ^ 1], w, i + w->n_workers);
        if (rc) {   /* keep the ordered hand-over alive so the other workers are not stuck */
            emsar_counts_free(cur->cnt); cur->cnt = NULL;
            pthread_mutex_lock(w->mu);
            while (*w->next_model != i) pthread_cond_wait(w->cv, w->mu);
            (*w->next_model)++;
            pthread_cond_broadcast(w->cv);
            pthread_mutex_unlock(w->mu);
            w->status[i] = rc;
        } else {
            w->status[i] = run_sample(w, ctx, i, cur);
        }
        c ^= 1;
    }
    emsar_hip_destroy(ctx);
    if (w->cfg->device_collapse) { emsar_hip_destroy(w->cdev.ctx); pthread_mutex_destroy(&w->cdev.mu); }
    return NULL;
}

static void usage(const char *a0) {
    fprintf(stderr,
            "Usage : %s <options> -I rshfile outdir outprefix alignmentfile|alignmentfilelist\n"
            "  -I, --rsh <file>        rsh index (from emsar-build)            [required]\n"
            "  -M, --multisample       last argument lists one alignment file per line; samples are spread over GPUs\n"
            "  -P, --PE                paired-end          -s, --strand_type ns|ssf|ssr|ssfr|ssrf (default ns)\n"
            "  -S, --SAM / -B, --BAM   SAM text / BAM input (default: default bowtie output)\n"
            "  -k, --max_repeat <n>    reads with more alignments are discarded (default 100)\n"
            "  -n, --nround <n>        rounds reported in the mean/sd columns (default 4; the EM is deterministic)\n"
            "  -e, --epsilon <tol>     EM stops when max |dtheta|/(theta+1e-6) < tol (default 1e-10)\n"
            "  -i, --max_niter_mle <n> cap on EM passes (default 200000)\n"
            "  -d, --delta <d>         10^d scaling of the effective lengths (default 0)\n"
            "  -g, --print_segments    also write .segments\n"
            "      --count-floor <reads> stopping-rule floor in inferred reads (default 0 = off; e.g. 1e-3 for large samples)\n"
            "      --zero-cut <x>        components below x and still falling do not hold the solve up (default 2.5e-7: they print as\n"
            "                            0.000000 either way; 0 = every component must meet -e)\n"
            "      --abs-step <x>        components that move by less than x FPKM per pass count as converged (default 1e-13; 0 = off)\n"
            "      --rsh-cache[=file]    read the parsed index from a binary cache (default <rshfile>.bin), write it after a text parse\n"
            "      --streaming-only      do not split the problem into connected sets (every pass streams the whole matrix)\n"
            "      --no-deterministic    streaming passes with floating atomics (run-to-run differences in the last digits) instead of\n"
            "                            the fixed-point sums that make two runs of the same input print the same bytes\n"
            "      --device-collapse     reads with two or more transcripts are merged into weighted segments on the GPU\n"
            "                            (emsar_hip_collapse_rows) instead of one index lookup per read on the host; same counts\n"
            "      --gpus <n> / --devices <a,b,..> (-M: one worker per entry, ids may repeat) / --device <d> / --plain /\n"
            "      --stats-json <file> / -q / -v\n", a0);
}

int main(int argc, char **argv) {
    config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.ao.max_repeat = 100; cfg.n_round = 4; cfg.verbose = 1; cfg.accel = 1; cfg.tol = 1e-10; cfg.max_iter = 200000;
    cfg.zero_cut = 2.5e-7;      /* a quarter of the "%lf" print quantum of the .fpkm file */
    cfg.abs_step = 1e-13;       /* see emsar_em_params.abs_step */
    const char *strand = "ns"; int multisample = 0, gpus = 0, device = 0;
    int dev_map[64], n_dev_map = 0;
    static struct option lo[] = {
        {"rsh", required_argument, 0, 'I'}, {"PE", no_argument, 0, 'P'}, {"strand_type", required_argument, 0, 's'},
        {"maxthread", required_argument, 0, 'p'}, {"max_repeat", required_argument, 0, 'k'}, {"nround", required_argument, 0, 'n'},
        {"epsilon", required_argument, 0, 'e'}, {"max_niter_mle", required_argument, 0, 'i'}, {"delta", required_argument, 0, 'd'},
        {"print_segments", no_argument, 0, 'g'}, {"multisample", no_argument, 0, 'M'}, {"SAM", no_argument, 0, 'S'}, {"BAM", no_argument, 0, 'B'},
        {"verbose", no_argument, 0, 'v'}, {"no_verbose", no_argument, 0, 'q'}, {"gpus", required_argument, 0, 1000},
        {"device", required_argument, 0, 1001}, {"plain", no_argument, 0, 1002}, {"stats-json", required_argument, 0, 1003},
        {"count-floor", required_argument, 0, 1004}, {"streaming-only", no_argument, 0, 1005}, {"rsh-cache", optional_argument, 0, 1006}, {"zero-cut", required_argument, 0, 1007}, {"abs-step", required_argument, 0, 1008}, {"devices", required_argument, 0, 1009}, {"device-collapse", no_argument, 0, 1010}, {"no-deterministic", no_argument, 0, 1011},
        {"maxfraglen", required_argument, 0, 'F'}, {"minfraglen", required_argument, 0, 'f'}, {0, 0, 0, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "vqPs:p:F:f:n:e:d:gMSBk:i:I:", lo, NULL)) != -1) {
        switch (c) {
            case 'I': cfg.rsh_path = optarg; break;
            case 'P': cfg.ao.pe = 1; break;
            case 's': strand = optarg; break;
            case 'p': break;                       /* CPU threads of the reference's solver: no meaning here */
            case 'F': case 'f': break;             /* overwritten by the rsh header in the reference too */
            case 'k': cfg.ao.max_repeat = atoi(optarg); break;
            case 'n': cfg.n_round = atoi(optarg); if (cfg.n_round <= 0) { fprintf(stderr, "option -n must be a natural number.\n"); return 1; } break;
            case 'e': cfg.tol = atof(optarg); if (cfg.tol <= 0) { fprintf(stderr, "option -e must be positive.\n"); return 1; } break;
            case 'i': cfg.max_iter = atoi(optarg); if (cfg.max_iter <= 0) { fprintf(stderr, "option -i must be positive.\n"); return 1; } break;
            case 'd': cfg.delta = atoi(optarg); break;
            case 'g': cfg.print_segments = 1; break;
            case 'M': multisample = 1; break;
            case 'S': cfg.ao.format = 1; break;
            case 'B': cfg.ao.format = 2; break;
            case 'v': cfg.verbose = 2; break;
            case 'q': cfg.verbose = 0; break;
            case 1000: gpus = atoi(optarg); break;
            case 1001: device = atoi(optarg); break;
            case 1002: cfg.accel = 0; break;
            case 1003: cfg.stats_json = optarg; break;
            case 1004: cfg.count_floor = atof(optarg); if (cfg.count_floor < 0) { fprintf(stderr, "--count-floor must be >= 0.\n"); return 1; } break;
            case 1005: cfg.set_mode = 1; break;
            case 1006: cfg.rsh_cache = optarg ? optarg : ""; break;
            case 1007: cfg.zero_cut = atof(optarg); break;
            case 1008: cfg.abs_step = atof(optarg); break;
            case 1010: cfg.device_collapse = 1; break;
            case 1011: cfg.no_deterministic = 1; break;
            case 1009: {
                const char *q = optarg;
                while (*q && n_dev_map < 64) {
                    char *end; long d = strtol(q, &end, 10);
                    if (end == q || d < 0 || d > 1023) { fprintf(stderr, "--devices wants a comma-separated list of device ids.\n"); return 1; }
                    dev_map[n_dev_map++] = (int)d;
                    q = *end == ',' ? end + 1 : end;
                    if (*end && *end != ',') { fprintf(stderr, "--devices wants a comma-separated list of device ids.\n"); return 1; }
                }
                if (!n_dev_map) { fprintf(stderr, "--devices wants a comma-separated list of device ids.\n"); return 1; }
                break;
            }
            default: usage(argv[0]); return 1;
        }
    }
    if (!cfg.rsh_path || optind + 2 >= argc) { usage(argv[0]); return 1; }
    if (emsar_set_strand(strand, cfg.ao.pe, &cfg.ao.strand)) { fprintf(stderr, "error: invalid strand type.\n"); return 1; }
    cfg.outdir = argv[optind]; cfg.prefix = argv[optind + 1];
    const char *last = argv[optind + 2];
    char **list = NULL; int n_list = 0;
    if (!multisample) { list = (char **)malloc(sizeof(char *)); list[0] = strdup(last); n_list = 1; }
    else {
        FILE *f = fopen(last, "r");
        if (!f) { fprintf(stderr, "Can't open alignment list file.\n"); return 1; }
        char line[4096];
        while (fgets(line, sizeof line, f)) {
            size_t n = strlen(line);
            while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
            if (!n) continue;
            list = (char **)realloc(list, sizeof(char *) * (size_t)(n_list + 1));
            list[n_list++] = strdup(line);
        }
        fclose(f);
        if (!n_list) { fprintf(stderr, "No alignment files in the alignment list\n"); return 1; }
    }
    cfg.aln = list; cfg.n_aln = n_list;
    /* the reference runs `mkdir -p outdir` (emsar_main.c:284-285); find out now, not after the solve, that it cannot be written */
    {
        char tmp[4096];
        size_t n = strlen(cfg.outdir);
        if (n == 0 || n >= sizeof tmp) { fprintf(stderr, "can't create output directory %s\n", cfg.outdir); return 1; }
        memcpy(tmp, cfg.outdir, n + 1);
        for (size_t k = 1; k <= n; k++)
            if (tmp[k] == '/' || tmp[k] == 0) {
                const char keep = tmp[k];
                tmp[k] = 0;
                if (mkdir(tmp, 0777) != 0 && errno != EEXIST) { fprintf(stderr, "can't create output directory %s\n", tmp); return 1; }
                tmp[k] = keep;
            }
        if (access(cfg.outdir, W_OK | X_OK) != 0) { fprintf(stderr, "can't write to output directory %s\n", cfg.outdir); return 1; }
    }

    char err[512];
    emsar_rsh *rsh = NULL;
    double t0 = now_s();
    int rc = -1, from_cache = 0;
    char *cache_path = NULL;
    if (cfg.rsh_cache) {                       /* <rsh>.bin next to the text unless a path was given */
        size_t n = strlen(cfg.rsh_cache[0] ? cfg.rsh_cache : cfg.rsh_path) + 8;
        cache_path = (char *)malloc(n);
        if (!cache_path) { fprintf(stderr, "out of memory\n"); return 1; }
        if (cfg.rsh_cache[0]) snprintf(cache_path, n, "%s", cfg.rsh_cache); else snprintf(cache_path, n, "%s.bin", cfg.rsh_path);
        rc = emsar_rsh_read_cache(cfg.rsh_path, cache_path, &rsh, err, sizeof err);
        if (rc == 0) from_cache = 1;
        else if (cfg.verbose > 1) fprintf(stdout, "rsh cache not used (%s)\n", err);
    }
    if (rc) rc = emsar_rsh_read(cfg.rsh_path, &rsh, err, sizeof err);
    if (rc) { fprintf(stderr, "%s\n", err); return 1; }
    if (cache_path && !from_cache && emsar_rsh_write_cache(rsh, cfg.rsh_path, cache_path) != 0)
        fprintf(stderr, "warning: can't write the rsh cache %s\n", cache_path);
    if (cfg.verbose > 0 && from_cache) fprintf(stdout, "rsh: binary cache %s\n", cache_path);
    free(cache_path);
    if (cfg.verbose > 0) fprintf(stdout, "rsh: %d transcripts, %lld segments, fragment lengths %d-%d (%.2fs)\n", rsh->n_tx,
                                 (long long)rsh->n_rows, rsh->frag_min, rsh->frag_max, now_s() - t0);

    int n_workers = 1;
    if (multisample) {
        /* one worker per GPU; a probe context tells us how many devices exist without touching HIP here */
        int avail = 0;
        for (int d = 0; d < 64; d++) { emsar_hip_ctx *probe = NULL; if (emsar_hip_create(&probe, d) != 0) break; emsar_hip_destroy(probe); avail++; }
        if (avail == 0) { fprintf(stderr, "%s\n", emsar_hip_strerror(EMSAR_HIP_ERR_NO_DEVICE)); return 1; }
        if (n_dev_map) {
            for (int g = 0; g < n_dev_map; g++)
                if (dev_map[g] >= avail) { fprintf(stderr, "--devices: device %d does not exist (%d found)\n", dev_map[g], avail); return 1; }
            n_workers = n_dev_map;
        } else {
            n_workers = gpus > 0 && gpus < avail ? gpus : avail;
            for (int g = 0; g < n_workers && g < 64; g++) dev_map[g] = g;
            if (n_workers > 64) n_workers = 64;
        }
        if (n_workers > n_list) n_workers = n_list;
    }
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER; pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    int next_model = 0; double eumacut = 0.0;
    int *status = (int *)calloc((size_t)n_list, sizeof(int));
    emsar_em_stats *stats = (emsar_em_stats *)calloc((size_t)n_list, sizeof(emsar_em_stats));
    double *parse_s = (double *)calloc((size_t)n_list, sizeof(double));
    double *model_s = (double *)calloc((size_t)n_list, sizeof(double)), *host_s = (double *)calloc((size_t)n_list, sizeof(double));
    worker_arg *wa = (worker_arg *)calloc((size_t)n_workers, sizeof(worker_arg));
    pthread_t *th = (pthread_t *)calloc((size_t)n_workers, sizeof(pthread_t));
    if (!status || !stats || !parse_s || !model_s || !host_s || !wa || !th) { fprintf(stderr, "out of memory\n"); return 1; }
    t0 = now_s();
    /* Workers wait at a gate until their number is final: a thread that cannot be started must not leave the others
     * waiting for samples nobody will take (the EUMAcut hand-over is in sample order). */
    int go = 0, n_started = 1;
    for (int g = 0; g < n_workers; g++)
        wa[g] = (worker_arg){&cfg, rsh, multisample ? dev_map[g] : device, n_workers, g, &mu, &cv, &next_model, &eumacut, status, stats, parse_s,
                             model_s, host_s, &go, cfg.ao, {NULL, PTHREAD_MUTEX_INITIALIZER}};
    for (int g = 1; g < n_workers; g++) {
        if (pthread_create(&th[g], NULL, worker_main, &wa[g]) != 0) { fprintf(stderr, "warning: worker %d could not be started, using %d\n", g, g); break; }
        n_started++;
    }
    /* every parallel host step of a worker gets its share of the cores: G workers x 16 reader threads each would
     * oversubscribe the host (parse and inflate pools, emsar_host_threads) */
    {
        long nc = sysconf(_SC_NPROCESSORS_ONLN);
        int share = (int)((nc < 1 ? 1 : nc) / n_started);
        emsar_host_set_thread_budget(share < 1 ? 1 : share > 16 ? 16 : share);
    }
    pthread_mutex_lock(&mu);
    n_workers = n_started;
    for (int g = 0; g < n_workers; g++) wa[g].n_workers = n_workers;
    go = 1;
    pthread_cond_broadcast(&cv);
    pthread_mutex_unlock(&mu);
    worker_main(&wa[0]);
    for (int g = 1; g < n_workers; g++) pthread_join(th[g], NULL);
    double wall = now_s() - t0;
    int bad = 0;
    for (int i = 0; i < n_list; i++) if (status[i]) bad++;
    if (cfg.stats_json) {
        FILE *f = fopen(cfg.stats_json, "w");
        if (f) {
            fprintf(f, "{\"samples\": %d, \"gpus\": %d, \"wall_s\": %.6f, \"failed\": %d, \"per_sample\": [", n_list, n_workers, wall, bad);
            for (int i = 0; i < n_list; i++)
                fprintf(f, "%s{\"status\": %d, \"parse_s\": %.6f, \"model_s\": %.6f, \"host_s\": %.6f, \"em_passes\": %d, \"converged\": %d, \"solve_ms\": %.4f, \"kernel_ms\": %.4f, \"loglik\": %.9g, \"bytes_per_pass\": %lld, "
                           "\"sets_resident\": %d, \"sets_streamed\": %d, \"set_passes_max\": %d, \"set_passes_sum\": %lld, \"sets_build_ms\": %.4f, \"sets_kernel_ms\": %.4f}",
                        i ? ", " : "", status[i], parse_s[i], model_s[i], host_s[i], stats[i].iters, stats[i].converged, stats[i].solve_ms, stats[i].kernel_ms, stats[i].loglik,
                        (long long)stats[i].bytes_per_pass, stats[i].sets_resident, stats[i].sets_streamed, stats[i].set_passes_max,
                        (long long)stats[i].set_passes_sum, stats[i].sets_build_ms, stats[i].sets_kernel_ms);
            fprintf(f, "]}\n");
            fclose(f);
        }
    }
    emsar_rsh_free(rsh);
    for (int i = 0; i < n_list; i++) free(list[i]);
    free(list); free(status); free(stats); free(parse_s); free(model_s); free(host_s); free(wa); free(th);
    return bad ? 1 : 0;
}
