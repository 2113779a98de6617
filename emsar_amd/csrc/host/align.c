/* align.c -- alignment text readers and the per-read collapse into segment read counts.
 *
 * Re-states, without the linked lists, what the reference does per read group:
 *   parse_bowtieline / parse_bowtieline_PE / read_bowtie_SE / read_bowtie_PE    emsar_functions.c:552-836
 *   convert_bam_alignment_2_alignment(_PE), read_BAM_SE/PE (SAM TEXT ONLY here)  emsar_functions.c:323-548
 *   add_alignment_to_list, check_fraglen_discrepancy, parse_mmstr                alignment.c:29-108
 *   update_ReadCounts                                                            emsar_functions.c:838-943
 * Order of the filters for one read group (SURVEY.md 8a "input-side semantics"):
 *   1. an alignment equal in (tid,pos,fraglen) to one already held is dropped         alignment.c:37-41
 *   2. only alignments with the minimum mismatch count are kept                       alignment.c:43-47
 *   3. the read is discarded if more than -k alignments remain                        emsar_functions.c:372,752
 *   4. paired-end: discarded if its alignments disagree in fragment length           alignment.c:85-95
 *   5. discarded if the fragment length is outside [min,max] of the rsh header       emsar_functions.c:849
 *   6. tid multiset sorted ascending (duplicates kept) and looked up; a read whose set has no rsh node still
 *      counts in TotalReadCount and FraglengthCounts                                  emsar_functions.c:880-941
 * BAM is not read here (the reference vendors samtools 0.1.19 for it); SAM text covers the same records.
 */
#include "emsar_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void *emsar_lr_open(const char *path);
char *emsar_lr_next(void *h);
void emsar_lr_close(void *h);

typedef struct { int32_t tid, mm, fraglen, pos; } aln;
typedef struct { aln *a; int n, cap; int min_mm; } alist;

int emsar_set_strand(const char *s, int pe, char *out) {            /* set_library_strand_type, 16-22 */
    if (strcmp(s, "ns") == 0) { *out = 0; return 0; }
    if (strcmp(s, "ssf") == 0 && !pe) { *out = '+'; return 0; }
    if (strcmp(s, "ssr") == 0 && !pe) { *out = '-'; return 0; }
    if (strcmp(s, "ssfr") == 0 && pe) { *out = '+'; return 0; }
    if (strcmp(s, "ssrf") == 0 && pe) { *out = '-'; return 0; }
    return EMSAR_HOST_ERR_ARG;   /* the reference falls off the end here; we refuse */
}

static int parse_mmstr(const char *s) {                               /* alignment.c:99-108: commas + 1 */
    int mm = 0;
    if (s[0]) mm++;
    for (; *s; s++) if (*s == ',') mm++;
    return mm;
}
static int parse_sam_md(const char *s) {                              /* parse_SAM_mmstr, 418-424: non-digits */
    int mm = 0;
    for (; *s; s++) if (*s < '0' || *s > '9') mm++;
    return mm;
}

/* add_alignment_to_list (alignment.c:29-61) */
static int alist_add(alist *l, aln x) {
    for (int i = 0; i < l->n; i++)
        if (l->a[i].tid == x.tid && l->a[i].pos == x.pos && l->a[i].fraglen == x.fraglen) return 0;
    if (x.mm > l->min_mm) return 0;
    if (x.mm < l->min_mm) { l->n = 0; l->min_mm = x.mm; }
    if (l->n == l->cap) {
        int nc = l->cap ? l->cap * 2 : 64;
        aln *na = (aln *)realloc(l->a, sizeof(aln) * (size_t)nc);
        if (!na) return -1;
        l->a = na; l->cap = nc;
    }
    l->a[l->n++] = x;
    return 1;
}

static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return x < y ? -1 : x > y; }

/* update_ReadCounts (838-943), positional-bias bookkeeping left out (posmodel is 0 by default and unfinished) */
static int flush_group(const emsar_rsh *r, const emsar_aln_opts *o, alist *l, emsar_counts *c, int32_t **tmp, int *tmpcap) {
    if (l->n == 0) return 0;
    c->reads_seen++;
    if (l->n > o->max_repeat) { c->reads_over_k++; return 0; }
    if (o->pe) {                                                      /* check_fraglen_discrepancy */
        for (int i = 1; i < l->n; i++) if (l->a[i].fraglen != l->a[0].fraglen) { c->reads_discrepant++; return 0; }
    }
    int fl = l->a[0].fraglen;
    if (fl > r->hdr_maxfrag || fl < r->hdr_minfrag) { c->reads_bad_fraglen++; return 0; }
    if (l->n > *tmpcap) {
        int32_t *nt = (int32_t *)realloc(*tmp, sizeof(int32_t) * (size_t)l->n);
        if (!nt) return -1;
        *tmp = nt; *tmpcap = l->n;
    }
    for (int i = 0; i < l->n; i++) (*tmp)[i] = l->a[i].tid;
    qsort(*tmp, (size_t)l->n, sizeof(int32_t), cmp_i32);             /* the reference insertion-sorts with >= (889) */
    int64_t row = emsar_rsh_row_of(r, *tmp, l->n);
    if (row >= 0) c->R[row]++; else c->reads_no_segment++;
    c->frag_counts[fl]++;
    c->total_reads++;
    return 0;
}

/* split on tabs in place; returns number of fields */
static int split_tabs(char *line, char **f, int max) {
    int n = 0;
    f[n++] = line;
    for (char *p = line; *p; p++) if (*p == '\t') { *p = 0; if (n < max) f[n++] = p + 1; else break; }
    return n;
}

/* check_mate_readid_matching (alignment.c:113-127), precedence quirk kept: the "/1,/2" branch requires the prefix
 * compare only for the (2,1) order because && binds tighter than ||. Returns adjusted id length, 0 = no match. */
static int mate_id_len(const char *a, const char *b) {
    size_t la = strlen(a);
    if (la != strlen(b)) return 0;
    if (la >= 2 && a[la - 2] == '/' && b[la - 2] == '/' &&
        ((a[la - 1] == '1' && b[la - 1] == '2') || (a[la - 1] == '2' && b[la - 1] == '1' && strncmp(a, b, la - 2) == 0)))
        return (int)la - 2;
    for (size_t i = 0; i < la; i++) {
        if (a[i] == ' ' && b[i] == ' ') return (int)i;
        if (a[i] != b[i]) return 0;
    }
    return (int)la;
}

#define FAIL(code, ...) do { if (err) snprintf(err, errlen, __VA_ARGS__); rc = (code); goto done; } while (0)

int emsar_count_alignments(const emsar_rsh *r, const char *path, const emsar_aln_opts *o, emsar_counts **out,
                           char *err, size_t errlen) {
    int rc = EMSAR_HOST_OK;
    *out = NULL;
    emsar_counts *c = (emsar_counts *)calloc(1, sizeof(*c));
    void *lr = NULL;
    alist l = {NULL, 0, 0, 10000};
    int32_t *tmp = NULL; int tmpcap = 0;
    char *prev = NULL; size_t prevcap = 0;
    char *line2 = NULL;
    if (!c) return EMSAR_HOST_ERR_OOM;
    c->n_rows = r->n_rows;
    c->n_frag = r->hdr_maxfrag + 1;
    c->R = (int32_t *)calloc((size_t)r->n_rows, sizeof(int32_t));
    c->frag_counts = (int32_t *)calloc((size_t)c->n_frag, sizeof(int32_t));
    c->readlength = r->hdr_readlength;
    if (!c->R || !c->frag_counts) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    lr = emsar_lr_open(path);
    if (!lr) FAIL(EMSAR_HOST_ERR_IO, "can't open alignment file %s", path);

    char *line;
    int have_prev = 0;
    while ((line = emsar_lr_next(lr))) {
        aln x; char *rid = NULL; int keep = 0;
        char idbuf[1024];
        if (o->format == 1) {                                        /* ---------- SAM text ---------- */
            if (line[0] == '@') continue;
            char *f[64];
            int nf = split_tabs(line, f, 64);
            if (nf < 11) FAIL(EMSAR_HOST_ERR_FORMAT, "SAM record with %d fields", nf);
            if (strcmp(f[2], "*") == 0) continue;                    /* unaligned (core.tid == -1) */
            int flag = atoi(f[1]);
            const char *md = "";
            for (int i = 11; i < nf; i++) if (strncmp(f[i], "MD:Z:", 5) == 0) md = f[i] + 5;
            if (!o->pe) {
                int32_t tid = emsar_rsh_tid_of(r, f[2]);
                if (tid < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "unknown transcript %s in the alignment file", f[2]);
                char strand = (flag & 0x10) ? '-' : '+';
                rid = f[0];
                if (!(o->strand != 0 && o->strand != strand)) {
                    x.tid = tid; x.mm = parse_sam_md(md); x.fraglen = (int32_t)strlen(f[9]); x.pos = atoi(f[3]) - 1; keep = 1;
                }
            } else {                                                  /* mate on the next record (514-520) */
                size_t L1 = strlen(line);  (void)L1;
                /* copy what we need from record 1 before the reader reuses its buffer */
                char name1[1024], ref1[1024]; int flag1 = flag, pos1 = atoi(f[3]) - 1, len1 = (int)strlen(f[9]), mm1 = parse_sam_md(md);
                snprintf(name1, sizeof name1, "%s", f[0]); snprintf(ref1, sizeof ref1, "%s", f[2]);
                char *l2 = emsar_lr_next(lr);
                if (!l2) break;
                char *g[64];
                int ng = split_tabs(l2, g, 64);
                if (ng < 11) FAIL(EMSAR_HOST_ERR_FORMAT, "SAM record with %d fields", ng);
                int flag2 = atoi(g[1]), pos2 = atoi(g[3]) - 1, len2 = (int)strlen(g[9]);
                const char *md2 = "";
                for (int i = 11; i < ng; i++) if (strncmp(g[i], "MD:Z:", 5) == 0) md2 = g[i] + 5;
                int32_t tid = emsar_rsh_tid_of(r, ref1);
                if (tid < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "unknown transcript %s in the alignment file", ref1);
                if (c->readlength == -1) c->readlength = len1;
                if (c->readlength != len1 || c->readlength != len2) FAIL(EMSAR_HOST_ERR_FORMAT, "paired-end data with variable read length is not supported");
                int p1, p2; char s1, s2;
                if ((flag1 & 0x40) && (flag2 & 0x80)) { p1 = pos1; p2 = pos2; s1 = (flag1 & 0x10) ? '-' : '+'; s2 = (flag2 & 0x10) ? '-' : '+'; }
                else if ((flag2 & 0x40) && (flag1 & 0x80)) { p1 = pos2; p2 = pos1; s1 = (flag2 & 0x10) ? '-' : '+'; s2 = (flag1 & 0x10) ? '-' : '+'; }
                else FAIL(EMSAR_HOST_ERR_FORMAT, "mates are not grouped in the SAM file");
                snprintf(idbuf, sizeof idbuf, "%s", name1); rid = idbuf;
                x.tid = tid; x.mm = mm1 + parse_sam_md(md2);
                if (p2 > p1) { x.fraglen = p2 - p1 + c->readlength; x.pos = p1; keep = !(o->strand == '-') && (s1 == '+' && s2 == '-'); }
                else { x.fraglen = p1 - p2 + c->readlength; x.pos = p2; keep = !(o->strand == '+') && (s1 == '-' && s2 == '+'); }
            }
        } else if (!o->pe) {                                          /* ---------- default bowtie, single-end (552-587) ---------- */
            char *f[9];
            int nf = split_tabs(line, f, 9);
            if (nf < 7) FAIL(EMSAR_HOST_ERR_FORMAT, "input alignment file doesn't look like a bowtie output file");
            rid = f[0];
            char strand = f[1][0];
            if (!(o->strand != 0 && o->strand != strand)) {
                int32_t tid = emsar_rsh_tid_of(r, f[2]);
                if (tid < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "unknown transcript %s in the alignment file", f[2]);
                x.tid = tid; x.pos = atoi(f[3]); x.fraglen = (int32_t)strlen(f[4]); x.mm = parse_mmstr(nf > 7 ? f[7] : ""); keep = 1;
            }
        } else {                                                      /* ---------- default bowtie, paired-end (612-703) ---------- */
            size_t n1 = strlen(line) + 1;
            char *keep1 = (char *)realloc(line2, n1);
            if (!keep1) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            line2 = keep1; memcpy(line2, line, n1);                   /* record 1 survives the next read */
            char *l2 = emsar_lr_next(lr);
            if (!l2) break;
            char *f[9], *g[9];
            int nf = split_tabs(line2, f, 9), ng = split_tabs(l2, g, 9);
            if (nf < 7 || ng < 7) FAIL(EMSAR_HOST_ERR_FORMAT, "input alignment file doesn't look like a bowtie output file");
            int idlen = mate_id_len(f[0], g[0]);
            if (idlen == 0) FAIL(EMSAR_HOST_ERR_FORMAT, "mate read IDs don't match: %s / %s", f[0], g[0]);
            snprintf(idbuf, sizeof idbuf, "%.*s", idlen, f[0]); rid = idbuf;
            if (strcmp(f[2], g[2]) == 0) {                             /* mates on different transcripts: no alignment (667) */
                int len1 = (int)strlen(f[4]), len2 = (int)strlen(g[4]);
                if (c->readlength == -1) c->readlength = len1;
                if (c->readlength != len1 || c->readlength != len2) FAIL(EMSAR_HOST_ERR_FORMAT, "paired-end data with variable read length is not supported");
                int32_t tid = emsar_rsh_tid_of(r, f[2]);
                if (tid < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "unknown transcript %s in the alignment file", f[2]);
                /* The reference decides "order_reversed" by comparing the id's last char with the INTEGER 1
                 * (emsar_functions.c:652), which is never true for text: the two records are always swapped, i.e.
                 * record 2 is treated as mate 1.  Kept, because it decides which orientation passes the filter. */
                int p1 = atoi(g[3]), p2 = atoi(f[3]);
                char s1 = g[1][0], s2 = f[1][0];
                x.tid = tid; x.mm = parse_mmstr(nf > 7 ? f[7] : "") + parse_mmstr(ng > 7 ? g[7] : "");
                if (p2 > p1) { x.fraglen = p2 - p1 + c->readlength; x.pos = p1; keep = !(o->strand == '-') && (s1 == '+' && s2 == '-'); }
                else { x.fraglen = p1 - p2 + c->readlength; x.pos = p2; keep = !(o->strand == '+') && (s1 == '-' && s2 == '+'); }
            }
        }
        if (!keep) continue;                                          /* prev_read_id is untouched by filtered records (748,815) */
        if (have_prev && strcmp(prev, rid) == 0) {
            if (alist_add(&l, x) < 0) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
        } else {
            if (have_prev && flush_group(r, o, &l, c, &tmp, &tmpcap) < 0) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            l.n = 0; l.min_mm = 10000;
            if (alist_add(&l, x) < 0) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
        }
        size_t need = strlen(rid) + 1;
        if (need > prevcap) { char *np = (char *)realloc(prev, need * 2); if (!np) FAIL(EMSAR_HOST_ERR_OOM, "out of memory"); prev = np; prevcap = need * 2; }
        memcpy(prev, rid, need);
        have_prev = 1;
    }
    if (l.n == 0) FAIL(EMSAR_HOST_ERR_FORMAT, "no usable alignment in %s (the reference stops with 'NULL alignment list')", path);
    if (flush_group(r, o, &l, c, &tmp, &tmpcap) < 0) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
done:
    emsar_lr_close(lr);
    free(l.a); free(tmp); free(prev); free(line2);
    if (rc != EMSAR_HOST_OK) { emsar_counts_free(c); return rc; }
    *out = c;
    return EMSAR_HOST_OK;
}

void emsar_counts_free(emsar_counts *c) {
    if (!c) return;
    free(c->R); free(c->frag_counts); free(c);
}
