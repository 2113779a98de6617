/* emsar_host.h -- C host side of the MI355X EMSAR core (stays on the CPU by design: BASELINE north_star
 * "alignment parsing and rsh I/O untouched").  It re-states, on flat arrays, what the reference does between
 * reading its inputs and calling the solver, and what it prints afterwards:
 *
 *   rsh text reader           construct_rsh_from_rshfile      /root/reference/src/emsar_functions.c:1351-1510
 *   alignment readers         read_bowtie_SE/PE, read_BAM_* (SAM text and BAM)  emsar_functions.c:323-836
 *   per-read collapse         add_alignment_to_list + update_ReadCounts       alignment.c:29-95, emsar_functions.c:838-943
 *   fragment-length weights   transfer_fraglendist_to_Wf, compute_adjEUMA     emsar_functions.c:2503-2523
 *   row order (cid)           scan_rshbucket                                  emsar_functions.c:2135-2192
 *   connected sets            build_TC_from_CT_2, propagate_2, EUMAcut loop   emsar_functions.c:2201-2259, emsar_main.c:411-425
 *   EUMAps                    construct_EUMAps                                emsar_functions.c:3148-3154
 *   outputs                   print_FPKMfinal, print_FraglengthDist, print_aEUMA_3   emsar_functions.c:3163-3212, 2477-2493, 2262-2300
 *
 * Nothing here exits the process: functions return 0 or a negative code and fill a message buffer.
 */
#ifndef EMSAR_HOST_H
#define EMSAR_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMSAR_HOST_OK 0
#define EMSAR_HOST_ERR_IO (-1)
#define EMSAR_HOST_ERR_FORMAT (-2)
#define EMSAR_HOST_ERR_OOM (-3)
#define EMSAR_HOST_ERR_ARG (-4)

#define EMSAR_MAX_NTID_PER_SID 5000   /* emsar.h:17 */
#define EMSAR_EUMACUT_INCREMENT 2.0   /* emsar.h:18 */

typedef struct emsar_rsh {
    int32_t n_tx;            /* max_tid + 1 */
    char **names;            /* IndexTable: tid -> transcript name */
    int32_t hdr_minfrag, hdr_maxfrag, hdr_readlength, max_t_size;   /* header line "#max_tid,max_t_size,minfrag,maxfrag,readlength" */
    int32_t frag_min, frag_max, nfl;    /* Fraglengths after determine_fraglength_range (emsar_functions.c:2471-2475) */
    int64_t n_rows;          /* max_cid + 1 = n_tx single-tid rows + multi-tid rows, in cid order */
    uint64_t *row_ptr;       /* n_rows + 1 */
    int32_t *col_idx;        /* CT: tids of each row; a tid may repeat (internal repeats) */
    int32_t *euma;           /* n_rows * nfl effective position counts per fragment length (0 where absent) */
    uint8_t *has_node;       /* 0 for a single-tid row whose transcript has no unique region (no rsh node) */
    void *name_index;        /* opaque: name -> tid */
    void *set_index;         /* opaque: sorted tid multiset -> row */
} emsar_rsh;

int  emsar_rsh_read(const char *path, emsar_rsh **out, char *err, size_t errlen);
void emsar_rsh_free(emsar_rsh *r);
/* binary cache of the parsed arrays (a human PE rsh is gigabytes of text): write after a text parse, read instead of
 * one.  read_cache refuses a file that is not ours, is truncated or inconsistent, or -- when src_path is given -- was
 * made from a text of another size / mtime; the caller then parses the text. */
int  emsar_rsh_write_cache(const emsar_rsh *r, const char *src_path, const char *cache_path);
int  emsar_rsh_read_cache(const char *src_path /* may be NULL: no staleness check */, const char *cache_path, emsar_rsh **out,
                          char *err, size_t errlen);
int32_t emsar_rsh_tid_of(const emsar_rsh *r, const char *name);                 /* -1 if unknown */
int64_t emsar_rsh_row_of(const emsar_rsh *r, const int32_t *sorted_tids, int n); /* -1 if no such segment */

/* host thread budget of the parallel readers (hostutil.c): min(online cores, 16) by default; a caller that runs several
 * readers at once (emsar-hip -M: one worker per GPU) divides it; EMSAR_HOST_THREADS overrides both */
void emsar_host_set_thread_budget(int n);   /* 0 = default */
int  emsar_host_threads(void);

/* parallel BGZF inflate (pbgzf.c): NULL from open = not a seekable BGZF file, use zlib's gzread instead.
 * read returns the bytes delivered (< n only at the end of the file), -1 on a damaged block. */
struct emsar_pbgzf;
struct emsar_pbgzf *emsar_pbgzf_open(const char *path);
long emsar_pbgzf_read(struct emsar_pbgzf *p, void *dst, size_t n);
void emsar_pbgzf_close(struct emsar_pbgzf *p);
const char *emsar_pbgzf_engine(void);   /* "libdeflate" when libdeflate.so.0 could be loaded (EMSAR_HOST_INFLATE=zlib: never), else "zlib" */

/* Optional collapse of read-level rows OUTSIDE this library (emsar_hip_collapse_rows has exactly this signature behind a
 * context): rows with the same multiset of ids become one row with the number of members as its weight; ids sorted in the
 * output; the caller provides the three output arrays at worst-case size.  0 = ok. */
typedef int (*emsar_collapse_fn)(void *user, int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                 int64_t *n_unique_out, uint64_t *row_ptr_out, int32_t *col_idx_out, int32_t *weight_out);

typedef struct {
    int pe;              /* -P */
    char strand;         /* library_strand_type: 0, '+', '-'  (set_library_strand_type, emsar_functions.c:16-22) */
    int max_repeat;      /* -k, default 100 */
    int format;          /* 0 default-bowtie text, 1 SAM text, 2 BAM */
    /* update_ReadCounts (emsar_functions.c:838-943) in two halves: the filters stay here, read by read; with `collapse` set, the
     * kept reads with two or more transcripts are not looked up one by one but gathered as read-level rows (sorted ids) and handed
     * to `collapse` in batches of `collapse_batch_rows`; only the UNIQUE rows that come back are looked up in the rsh, their
     * weights added to ReadCount.  Same counts as the per-read path, by construction.  NULL = per-read lookup.
     * Threading: the ranged / batched parsers flush from their worker threads; the library serialises the calls (one at a time,
     * process-wide), so the function need not be re-entrant -- but it may be entered from a thread other than the caller's. */
    emsar_collapse_fn collapse;
    void *collapse_user;
    int64_t collapse_batch_rows;     /* <= 0: 4M rows */
} emsar_aln_opts;

typedef struct {
    int64_t n_rows;
    int32_t *R;              /* ReadCount per row */
    int32_t n_frag;          /* hdr_maxfrag + 1 */
    int32_t *frag_counts;    /* FraglengthCounts[0..hdr_maxfrag] */
    int64_t total_reads;     /* TotalReadCount */
    int64_t reads_seen, reads_over_k, reads_bad_fraglen, reads_discrepant, reads_no_segment;   /* bookkeeping only */
    int32_t readlength;      /* PE: learnt from the data when the header says -1 */
    void *batch;             /* private: read-level rows waiting for emsar_aln_opts.collapse */
} emsar_counts;

int  emsar_set_strand(const char *strand_type, int pe, char *out);   /* "ns","ssf","ssr","ssfr","ssrf" */
int  emsar_count_alignments(const emsar_rsh *r, const char *path, const emsar_aln_opts *o, emsar_counts **out,
                            char *err, size_t errlen);
void emsar_counts_free(emsar_counts *c);

typedef struct {
    int64_t n_rows;
    int32_t n_tx, nfl;
    double *Wf;              /* nfl */
    double *L;               /* adjEUMA per row */
    double *E;               /* EUMAps per row */
    double *E_solver;        /* E with the rows left out of every set (CS == -1) zeroed: what the solver sees */
    int32_t *CS, *TS;        /* set id per row / per transcript (-1 = in no set) */
    int32_t n_sets;
    double eumacut;
} emsar_model;

/* eumacut_io persists across samples like the reference's global (never reset between -M samples). */
int  emsar_model_build(const emsar_rsh *r, const emsar_counts *c, int delta, double *eumacut_io, emsar_model **out,
                       char *err, size_t errlen);
/* the two halves separately, for callers that compute L_c = sum_i Wf[i] * EUMA_c[i] elsewhere (emsar_hip_adj_euma):
 * model_wf fills wf[nfl] (error if no read falls inside the fragment-length range); build_L takes L (NULL = host loop) */
int  emsar_model_wf(const emsar_rsh *r, const emsar_counts *c, double *wf);
int  emsar_model_build_L(const emsar_rsh *r, const emsar_counts *c, int delta, double *eumacut_io, const double *L,
                         emsar_model **out, char *err, size_t errlen);
void emsar_model_free(emsar_model *m);

/* mean/sd over rounds exactly as print_FPKMfinal does (sd = sqrt(sum sq/(n-1))/n; n = 1 gives NaN like the reference) */
void emsar_mean_sd(int32_t n_tx, int32_t n_round, const double *rounds, double *mean, double *sd);

int emsar_write_fpkm(const char *path, const emsar_rsh *r, const double *mean, const double *sd, const double *ieuma,
                     const double *ireadcount, const int32_t *ireadcount_int, const double *tpm, int64_t *total_ireadcount);
int emsar_write_fraglength(const char *path, const emsar_rsh *r, const emsar_counts *c, const emsar_model *m);
int emsar_write_segments(const char *path, const emsar_rsh *r, const emsar_counts *c, const emsar_model *m,
                         const double *mean_fpkm);

#ifdef __cplusplus
}
#endif
#endif
