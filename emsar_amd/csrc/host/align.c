/* align.c -- alignment text readers and the per-read collapse into segment read counts.
 *
 * Re-states, without the linked lists, what the reference does per read group:
 *   parse_bowtieline / parse_bowtieline_PE / read_bowtie_SE / read_bowtie_PE    emsar_functions.c:552-836
 *   convert_bam_alignment_2_alignment(_PE), read_BAM_SE/PE (SAM text and BAM)   emsar_functions.c:323-548
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
 * BAM is read through zlib (BGZF = concatenated gzip members); the reference vendors samtools 0.1.19 for it.
 */
#include "emsar_host.h"

#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <zlib.h>

void *emsar_lr_open(const char *path);
char *emsar_lr_next(void *h);
void emsar_lr_close(void *h);
int64_t emsar_lr_offset(void *h);
int emsar_lr_is_plain(void *h);
void *emsar_lr_open_at(const char *path, int64_t offset);

/* One SAM/BAM alignment record reduced to what the reference reads from bam1_t (emsar_functions.c:391-469):
 * qname, flag, reference name, 0-based position, l_qseq and the MD:Z string. */
typedef struct { char qname[1024]; char rname[1024]; char md[4096]; int flag, pos, l_seq, unaligned; int32_t tid; /* >= 0: already resolved (BAM: per reference, once), -1: look rname up */ } samrec;

/* ---- BAM: BGZF is a series of gzip members, which zlib's gzread() inflates transparently (SAM/BAM spec section 4).
 * The reference reads BAM through its vendored samtools 0.1.19 (bam.c, bgzf.c); only the fields above are used. ---- */
typedef struct {
    gzFile f; struct emsar_pbgzf *pf; int n_ref; char **ref; int32_t *ref_tid; const emsar_rsh *rsh; unsigned char *buf; size_t cap;
    const unsigned char *mem; size_t mem_n, mem_pos;   /* a worker's view: whole records already in memory (count_bam_parallel) */
} bamreader;

/* bytes delivered (short only at the end of the stream), -1 on error: the BGZF blocks of a regular file are inflated
 * by a pool of threads (pbgzf.c), anything else (stdin, plain gzip) by zlib on this thread */
static long bam_get(bamreader *b, void *dst, size_t n) {
    if (b->pf) return emsar_pbgzf_read(b->pf, dst, n);
    int got = gzread(b->f, dst, (unsigned)n);
    return got < 0 ? -1 : (long)got;
}
static int bam_rd(bamreader *b, void *dst, size_t n) { return bam_get(b, dst, n) == (long)n ? 0 : -1; }
static int32_t le32(const unsigned char *p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }

static void bam_close(bamreader *b) {
    if (!b) return;
    if (b->f) gzclose(b->f);
    emsar_pbgzf_close(b->pf);
    if (b->ref) { for (int i = 0; i < b->n_ref; i++) free(b->ref[i]); free(b->ref); }
    free(b->ref_tid); free(b->buf); free(b);
}

static bamreader *bam_open(const char *path, const emsar_rsh *rsh) {
    bamreader *b = (bamreader *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->rsh = rsh;
    b->pf = emsar_pbgzf_open(path);
    if (!b->pf) b->f = (path && path[0] && strcmp(path, "-") != 0) ? gzopen(path, "rb") : gzdopen(0, "rb");
    unsigned char h[8];
    if ((!b->f && !b->pf) || bam_rd(b, h, 8) || memcmp(h, "BAM\1", 4) != 0) { bam_close(b); return NULL; }
    int32_t l_text = le32(h + 4);
    if (l_text < 0) { bam_close(b); return NULL; }
    for (int32_t i = 0; i < l_text; i++) { unsigned char c; if (bam_rd(b, &c, 1)) { bam_close(b); return NULL; } }
    if (bam_rd(b, h, 4)) { bam_close(b); return NULL; }
    b->n_ref = le32(h);
    if (b->n_ref < 0 || b->n_ref > (1 << 28)) { bam_close(b); return NULL; }
    b->ref = (char **)calloc((size_t)b->n_ref + 1, sizeof(char *));
    b->ref_tid = (int32_t *)malloc(sizeof(int32_t) * ((size_t)b->n_ref + 1));
    if (!b->ref || !b->ref_tid) { bam_close(b); return NULL; }
    for (int i = 0; i < b->n_ref; i++) b->ref_tid[i] = -2;          /* not looked up yet */
    for (int i = 0; i < b->n_ref; i++) {
        if (bam_rd(b, h, 4)) { bam_close(b); return NULL; }
        int32_t ln = le32(h);
        if (ln <= 0 || ln > 1 << 20) { bam_close(b); return NULL; }
        b->ref[i] = (char *)malloc((size_t)ln);
        if (!b->ref[i] || bam_rd(b, b->ref[i], (size_t)ln) || bam_rd(b, h, 4)) { bam_close(b); return NULL; }
        b->ref[i][ln - 1] = 0;
    }
    return b;
}

/* returns 1 record read, 0 end of file, -1 malformed */
static int bam_next(bamreader *b, samrec *r) {
    const unsigned char *p;
    int32_t bs;
    if (b->mem) {                                                     /* records of a batch, in place */
        if (b->mem_pos == b->mem_n) return 0;
        if (b->mem_n - b->mem_pos < 4) return -1;
        bs = le32(b->mem + b->mem_pos);
        if (bs < 32 || (size_t)bs > b->mem_n - b->mem_pos - 4) return -1;
        p = b->mem + b->mem_pos + 4;
        b->mem_pos += 4 + (size_t)bs;
    } else {
        unsigned char h[4];
        long got = bam_get(b, h, 4);
        if (got == 0) return 0;
        if (got != 4) return -1;
        bs = le32(h);
        if (bs < 32 || bs > (1 << 28)) return -1;
        if ((size_t)bs > b->cap) { unsigned char *nb = (unsigned char *)realloc(b->buf, (size_t)bs); if (!nb) return -1; b->buf = nb; b->cap = (size_t)bs; }
        if (bam_rd(b, b->buf, (size_t)bs)) return -1;
        p = b->buf;
    }
    int32_t refid = le32(p), pos = le32(p + 4), l_seq = le32(p + 16);
    int l_name = p[8], n_cig = p[12] | (p[13] << 8), flag = p[14] | (p[15] << 8);
    size_t off = 32;
    if (off + (size_t)l_name > (size_t)bs || l_name < 1 || l_seq < 0) return -1;
    {
        size_t nl = strnlen((const char *)(p + off), (size_t)l_name);
        if (nl >= sizeof r->qname) nl = sizeof r->qname - 1;
        memcpy(r->qname, p + off, nl); r->qname[nl] = 0;
    }
    off += (size_t)l_name + 4u * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (off > (size_t)bs) return -1;
    r->flag = flag; r->pos = pos; r->l_seq = l_seq; r->md[0] = 0;
    r->unaligned = refid < 0 || refid >= b->n_ref;
    r->tid = -1;
    if (r->unaligned) { r->rname[0] = '*'; r->rname[1] = 0; }
    else {
        /* reference id -> transcript id once per reference instead of a name lookup per record */
        if (b->rsh && b->ref_tid[refid] == -2) b->ref_tid[refid] = emsar_rsh_tid_of(b->rsh, b->ref[refid]);
        r->tid = b->rsh ? b->ref_tid[refid] : -1;
        if (r->tid < 0) snprintf(r->rname, sizeof r->rname, "%s", b->ref[refid]);      /* unknown: the caller reports the name */
        else r->rname[0] = 0;
    }
    while (off + 3 <= (size_t)bs) {                               /* auxiliary fields: find MD:Z */
        const unsigned char *a = p + off;
        char ty = (char)a[2];
        off += 3;
        size_t len;
        switch (ty) {
            case 'A': case 'c': case 'C': len = 1; break;
            case 's': case 'S': len = 2; break;
            case 'i': case 'I': case 'f': len = 4; break;
            case 'Z': case 'H': {
                size_t n = 0;
                while (off + n < (size_t)bs && p[off + n]) n++;
                if (off + n >= (size_t)bs) return -1;
                if (a[0] == 'M' && a[1] == 'D' && ty == 'Z') {        /* (snprintf("%s") here was a fifth of the per-record cost) */
                    const size_t k = n < sizeof r->md - 1 ? n : sizeof r->md - 1;
                    memcpy(r->md, p + off, k); r->md[k] = 0;
                }
                len = n + 1; break;
            }
            case 'B': {
                if (off + 5 > (size_t)bs) return -1;
                char sub = (char)p[off]; int32_t cnt = le32(p + off + 1);
                size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if (cnt < 0) return -1;
                len = 5 + es * (size_t)cnt; break;
            }
            default: return -1;
        }
        off += len;
    }
    return 1;
}

/* SAM text line -> samrec; returns 1 record, 0 header line, -1 malformed */
static int sam_parse(char *line, samrec *r) {
    if (line[0] == '@') return 0;
    char *f[64];
    int nf = 0;
    f[nf++] = line;
    for (char *p = line; *p; p++) if (*p == '\t') { *p = 0; if (nf < 64) f[nf++] = p + 1; else break; }
    if (nf < 11) return -1;
    snprintf(r->qname, sizeof r->qname, "%s", f[0]);
    snprintf(r->rname, sizeof r->rname, "%s", f[2]);
    r->tid = -1;
    r->flag = atoi(f[1]); r->pos = atoi(f[3]) - 1; r->l_seq = (int)strlen(f[9]); r->md[0] = 0;
    r->unaligned = strcmp(f[2], "*") == 0;
    for (int i = 11; i < nf; i++) if (strncmp(f[i], "MD:Z:", 5) == 0) snprintf(r->md, sizeof r->md, "%s", f[i] + 5);
    return 1;
}

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

/* ---- read-level rows gathered for emsar_aln_opts.collapse ---------------------------------------------------------- */
typedef struct { uint64_t *rp; int32_t *ci; int64_t n_rows, cap_rows; uint64_t nnz, cap_nnz; } row_batch;

static pthread_mutex_t g_collapse_mu = PTHREAD_MUTEX_INITIALIZER;      /* serialises emsar_aln_opts.collapse (see batch_flush) */

static void batch_free(row_batch *b) { if (b) { free(b->rp); free(b->ci); free(b); } }

/* hand the gathered rows to the collapse function, look the unique rows up, add their weights (update_rshbucket 'r',
 * emsar_functions.c:1597-1624: a tid set without an rsh node counts in TotalReadCount but in no ReadCount) */
static int batch_flush(const emsar_rsh *r, const emsar_aln_opts *o, emsar_counts *c) {
    row_batch *b = (row_batch *)c->batch;
    if (!b || b->n_rows == 0) return 0;
    int rc = 0;
    int64_t nu = 0;
    uint64_t *rp_o = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(b->n_rows + 1));
    int32_t *ci_o = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b->nnz ? b->nnz : 1));
    int32_t *w_o = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->n_rows);
    if (!rp_o || !ci_o || !w_o) rc = -1;
    b->rp[b->n_rows] = b->nnz;
    if (!rc) {
        /* the parse workers flush their batches from their own threads: the callback (one device context, one stream, one set of
         * buffers behind it) is entered by one thread at a time, whatever the caller passed */
        pthread_mutex_lock(&g_collapse_mu);
        if (o->collapse(o->collapse_user, b->n_rows, r->n_tx, b->rp, b->ci, &nu, rp_o, ci_o, w_o) != 0) rc = -2;
        pthread_mutex_unlock(&g_collapse_mu);
    }
    for (int64_t u = 0; !rc && u < nu; u++) {
        const int64_t row = emsar_rsh_row_of(r, ci_o + rp_o[u], (int)(rp_o[u + 1] - rp_o[u]));
        if (row >= 0) c->R[row] += w_o[u]; else c->reads_no_segment += w_o[u];
    }
    free(rp_o); free(ci_o); free(w_o);
    b->n_rows = 0; b->nnz = 0;
    return rc;
}

static int batch_append(const emsar_rsh *r, const emsar_aln_opts *o, emsar_counts *c, const int32_t *sorted, int n) {
    row_batch *b = (row_batch *)c->batch;
    if (!b) {
        b = (row_batch *)calloc(1, sizeof(*b));
        if (!b) return -1;
        c->batch = b;
    }
    const int64_t limit = o->collapse_batch_rows > 0 ? o->collapse_batch_rows : ((int64_t)4 << 20);
    if (b->n_rows + 1 >= b->cap_rows) {
        const int64_t nc = b->cap_rows ? b->cap_rows * 2 : 1 << 16;
        uint64_t *np = (uint64_t *)realloc(b->rp, sizeof(uint64_t) * (size_t)(nc + 1));
        if (!np) return -1;
        b->rp = np; b->cap_rows = nc;
    }
    if (b->nnz + (uint64_t)n > b->cap_nnz) {
        const uint64_t nc = (b->cap_nnz ? b->cap_nnz * 2 : 1 << 18) + (uint64_t)n;
        int32_t *np = (int32_t *)realloc(b->ci, sizeof(int32_t) * (size_t)nc);
        if (!np) return -1;
        b->ci = np; b->cap_nnz = nc;
    }
    b->rp[b->n_rows++] = b->nnz;
    memcpy(b->ci + b->nnz, sorted, sizeof(int32_t) * (size_t)n);
    b->nnz += (uint64_t)n;
    return b->n_rows >= limit ? batch_flush(r, o, c) : 0;
}

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
    if (o->collapse && l->n >= 2) {                                   /* counted later, one lookup per distinct id set */
        if (batch_append(r, o, c, *tmp, l->n) != 0) return -1;
    } else {
        int64_t row = emsar_rsh_row_of(r, *tmp, l->n);
        if (row >= 0) c->R[row]++; else c->reads_no_segment++;
    }
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

/* a worker's view of a batch of whole text lines already in memory (count_text_parallel): the lines are NUL-terminated in place,
 * with the line reader's rules (trailing newline and a CR before it stripped; a last line without a newline is a line) */
typedef struct { char *p; size_t n, pos; } memtext;
static char *mt_next(memtext *m) {
    if (m->pos >= m->n) return NULL;
    char *start = m->p + m->pos;
    char *nl = (char *)memchr(start, '\n', m->n - m->pos);
    size_t len;
    if (nl) { len = (size_t)(nl - start); m->pos += len + 1; }
    else { len = m->n - m->pos; m->pos = m->n; }                     /* (the batch buffer has one spare byte behind n) */
    start[len] = 0;
    if (len && start[len - 1] == '\r') start[len - 1] = 0;
    return start;
}

/* One worker of emsar_count_alignments: the read groups of a byte range [begin, end) of a plain text file (end < 0: to
 * the end of the file; begin = 0 and end < 0: the whole input, any format).  Groups are runs of KEPT records with one
 * read id (filtered records do not break a run, emsar_functions.c:748,815), so the ranges are stitched on kept
 * records: a worker with begin > 0 skips the group of its first kept record (the worker on its left finishes that
 * group, wherever it started), and every worker stops at the first kept record that follows the group of the first
 * kept record at or after `end`.  *got_group = 0 when the range held no group at all. */
static emsar_counts *counts_new(const emsar_rsh *r) {
    emsar_counts *c = (emsar_counts *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->n_rows = r->n_rows;
    c->n_frag = r->hdr_maxfrag + 1;
    c->R = (int32_t *)calloc((size_t)r->n_rows, sizeof(int32_t));
    c->frag_counts = (int32_t *)calloc((size_t)c->n_frag, sizeof(int32_t));
    c->readlength = r->hdr_readlength;
    if (!c->R || !c->frag_counts) { emsar_counts_free(c); return NULL; }
    return c;
}

/* mem_bam / acc (both or neither): the records of one batch of a BAM file, already in memory and starting on a read
 * group, counted into the caller's accumulator (count_bam_parallel) */
static int count_range(const emsar_rsh *r, const char *path, const emsar_aln_opts *o, int64_t begin, int64_t end,
                       emsar_counts **out, int *got_group, char *err, size_t errlen, bamreader *mem_bam, emsar_counts *acc, memtext *mem_txt) {
#define NEXT_LINE() (mem_txt ? mt_next(mem_txt) : emsar_lr_next(lr))
#define LINE_OFFSET() (mem_txt ? (int64_t)0 : emsar_lr_offset(lr))
    int rc = EMSAR_HOST_OK;
    *out = NULL; *got_group = 0;
    emsar_counts *c = acc ? acc : counts_new(r);
    int skipping = begin > 0, end_seen = 0, pe_align = begin > 0;
    char *skip_id = NULL, *end_id = NULL;
    int64_t rec_off = 0;
    void *lr = NULL;
    bamreader *bam = NULL;
    alist l = {NULL, 0, 0, 10000};
    int32_t *tmp = NULL; int tmpcap = 0;
    char *prev = NULL; size_t prevcap = 0;
    char *line2 = NULL;
    if (!c) { if (err) snprintf(err, errlen, "out of memory"); return EMSAR_HOST_ERR_OOM; }
    if (mem_bam) bam = mem_bam;
    else if (mem_txt) { /* lines come from the batch */ }
    else if (o->format == 2) {
        bam = bam_open(path, r);
        if (!bam) FAIL(EMSAR_HOST_ERR_IO, "can't open BAM file %s", path);
    } else {
        lr = begin > 0 ? emsar_lr_open_at(path, begin) : emsar_lr_open(path);
        if (!lr) FAIL(EMSAR_HOST_ERR_IO, "can't open alignment file %s", path);
    }

    char *line = NULL;
    int have_prev = 0;
    samrec r1, r2;
    for (;;) {
        aln x; char *rid = NULL; int keep = 0;
        char idbuf[1024];
        if (o->format != 0) {                                        /* ---------- SAM text / BAM ---------- */
            int st;
            if (bam) st = bam_next(bam, &r1);
            else { line = NEXT_LINE(); if (line) rec_off = LINE_OFFSET(); st = line ? sam_parse(line, &r1) : -2; if (st == 0) continue; if (st == -2) st = 0; }
            if (st == 0) break;
            if (st < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "malformed %s record", bam ? "BAM" : "SAM");
            if (r1.unaligned) continue;                               /* core.tid == -1 (359, 515) */
            if (!o->pe) {
                int32_t tid = r1.tid >= 0 ? r1.tid : emsar_rsh_tid_of(r, r1.rname);
                if (tid < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "unknown transcript %s in the alignment file", r1.rname);
                char strand = (r1.flag & 0x10) ? '-' : '+';
                rid = r1.qname;
                if (!(o->strand != 0 && o->strand != strand)) {
                    x.tid = tid; x.mm = parse_sam_md(r1.md); x.fraglen = r1.l_seq; x.pos = r1.pos; keep = 1;
                }
            } else {                                                  /* mate on the next record (514-520) */
                if (bam) st = bam_next(bam, &r2);
                else { char *l2 = NEXT_LINE(); st = l2 ? sam_parse(l2, &r2) : 0; }
                if (st == 0) break;
                if (st < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "malformed %s record", bam ? "BAM" : "SAM");
                int32_t tid = r1.tid >= 0 ? r1.tid : emsar_rsh_tid_of(r, r1.rname);
                if (tid < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "unknown transcript %s in the alignment file", r1.rname);
                if (c->readlength == -1) c->readlength = r1.l_seq;
                if (c->readlength != r1.l_seq || c->readlength != r2.l_seq) FAIL(EMSAR_HOST_ERR_FORMAT, "paired-end data with variable read length is not supported");
                int p1, p2; char s1, s2;
                const char *md1, *md2;
                if ((r1.flag & 0x40) && (r2.flag & 0x80)) { p1 = r1.pos; p2 = r2.pos; s1 = (r1.flag & 0x10) ? '-' : '+'; s2 = (r2.flag & 0x10) ? '-' : '+'; md1 = r1.md; md2 = r2.md; }
                else if ((r2.flag & 0x40) && (r1.flag & 0x80)) { p1 = r2.pos; p2 = r1.pos; s1 = (r2.flag & 0x10) ? '-' : '+'; s2 = (r1.flag & 0x10) ? '-' : '+'; md1 = r2.md; md2 = r1.md; }
                else FAIL(EMSAR_HOST_ERR_FORMAT, "mates are not grouped in the SAM/BAM file");
                snprintf(idbuf, sizeof idbuf, "%s", r1.qname); rid = idbuf;
                x.tid = tid; x.mm = parse_sam_md(md1) + parse_sam_md(md2);
                if (p2 > p1) { x.fraglen = p2 - p1 + c->readlength; x.pos = p1; keep = !(o->strand == '-') && (s1 == '+' && s2 == '-'); }
                else { x.fraglen = p1 - p2 + c->readlength; x.pos = p2; keep = !(o->strand == '+') && (s1 == '-' && s2 == '+'); }
            }
        } else if (!(line = NEXT_LINE())) {
            break;
        } else if (!o->pe) {                                          /* ---------- default bowtie, single-end (552-587) ---------- */
            rec_off = LINE_OFFSET();
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
            if (pe_align) {                                           /* a range may begin on the second mate of a pair: it belongs to the left */
                pe_align = 0;
                const char *tab = strchr(line, '\t');
                if (tab && tab - line >= 2 && tab[-2] == '/' && tab[-1] == '2') continue;
            }
            rec_off = LINE_OFFSET();
            size_t n1 = strlen(line) + 1;
            char *keep1 = (char *)realloc(line2, n1);
            if (!keep1) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            line2 = keep1; memcpy(line2, line, n1);                   /* record 1 survives the next read */
            char *l2 = NEXT_LINE();
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
        if (end >= 0 && !end_seen && rec_off >= end) {                /* the first kept record at or after the end of the range */
            end_seen = 1;
            end_id = strdup(rid);
            if (!end_id) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
        }
        if (skipping) {                                               /* the group of the first kept record is the left neighbour's */
            if (!skip_id) { skip_id = strdup(rid); if (!skip_id) FAIL(EMSAR_HOST_ERR_OOM, "out of memory"); continue; }
            if (strcmp(skip_id, rid) == 0) continue;
            skipping = 0;
        }
        if (end_seen && strcmp(end_id, rid) != 0) break;              /* the next range starts here */
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
    if (l.n > 0) {
        *got_group = 1;
        if (flush_group(r, o, &l, c, &tmp, &tmpcap) < 0) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    }
done:
    emsar_lr_close(lr);
    if (!mem_bam) bam_close(bam);
    free(l.a); free(tmp); free(prev); free(line2); free(skip_id); free(end_id);
    if (rc == EMSAR_HOST_OK && !acc && o->collapse && batch_flush(r, o, c) != 0) {      /* a worker's accumulator is flushed by its owner */
        rc = EMSAR_HOST_ERR_IO;
        if (err) snprintf(err, errlen, "the collapse of read-level rows failed");
    }
    if (rc != EMSAR_HOST_OK) { if (!acc) emsar_counts_free(c); return rc; }
    *out = c;
    return EMSAR_HOST_OK;
#undef NEXT_LINE
#undef LINE_OFFSET
}

/* ---- the ranges of a plain text file counted side by side ---------------------------------------------------------- */
typedef struct {
    const emsar_rsh *r; const char *path; const emsar_aln_opts *o; int64_t begin, end;
    emsar_counts *c; int got, rc; char err[256];
} range_job;
static void *range_main(void *a) {
    range_job *j = (range_job *)a;
    j->rc = count_range(j->r, j->path, j->o, j->begin, j->end, &j->c, &j->got, j->err, sizeof j->err, NULL, NULL, NULL);
    return NULL;
}

static int host_threads(void);

/* how many ranges: plain seekable text only (not BAM, gzip, stdin), single-end, or default-bowtie paired-end whose
 * mates are named .../1 and .../2 (that is how a range finds the start of a pair); one range per 16 MiB at least */
static int plan_ranges(const char *path, const emsar_aln_opts *o, int64_t *size_out) {
    *size_out = 0;
    if (o->format == 2 || !path || !path[0] || strcmp(path, "-") == 0) return 1;
    if (o->pe && o->format != 0) return 1;
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return 1;
    void *lr = emsar_lr_open(path);
    if (!lr) return 1;
    int ok = emsar_lr_is_plain(lr);
    if (ok && o->pe) {                                   /* first two records must look like  name/1 \t ...  name/2 \t ... */
        for (int m = 1; m <= 2 && ok; m++) {
            char *ln = emsar_lr_next(lr);
            const char *tab = ln ? strchr(ln, '\t') : NULL;
            ok = tab && tab - ln >= 2 && tab[-2] == '/' && tab[-1] == (char)('0' + m);
        }
    }
    emsar_lr_close(lr);
    if (!ok) return 1;
    int nt = host_threads();
    const char *e;
    int64_t min_bytes = (int64_t)16 << 20;
    if ((e = getenv("EMSAR_HOST_RANGE_BYTES")) && atoll(e) > 0) min_bytes = atoll(e);       /* tests use small files */
    int64_t by_size = (int64_t)st.st_size / min_bytes;
    if (by_size < nt) nt = by_size < 1 ? 1 : (int)by_size;
    *size_out = (int64_t)st.st_size;
    return nt;
}

static int host_threads(void) { return emsar_host_threads(); }

static void counts_add(emsar_counts *c, const emsar_counts *q) {
    for (int64_t i = 0; i < c->n_rows; i++) c->R[i] += q->R[i];
    for (int32_t i = 0; i < c->n_frag; i++) c->frag_counts[i] += q->frag_counts[i];
    c->total_reads += q->total_reads; c->reads_seen += q->reads_seen; c->reads_over_k += q->reads_over_k;
    c->reads_bad_fraglen += q->reads_bad_fraglen; c->reads_discrepant += q->reads_discrepant; c->reads_no_segment += q->reads_no_segment;
}

/* ---- BAM counted by a pool of threads ------------------------------------------------------------------------
 * A BAM record carries its length but no sync mark, so a worker cannot start in the middle of the inflated stream.  The
 * calling thread therefore walks the stream (inflated ahead of it by pbgzf's own pool), hops from record to record and
 * cuts it into batches of whole records.  It reads just enough of each record -- refID, FLAG and the read name, all in
 * the fixed part -- to cut only where a KEPT record opens a new read group (kept = aligned and on the wanted strand:
 * the same test count_range applies, emsar_functions.c:359,748), so the batches are independent: each worker runs the
 * sequential loop on its batches and adds into its own counts, and the sums equal the one-thread result exactly.
 * Paired-end: the unit is the pair the sequential loop would form (an unaligned record is skipped singly, an aligned one
 * is read together with its successor), kept by the same orientation test (emsar_functions.c:514-548). */
typedef struct bam_batch { unsigned char *buf; size_t n, cap; int64_t index; struct bam_batch *next; } bam_batch;
typedef struct {
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_free;
    bam_batch *work_head, *work_tail, *free_list;
    int closing;
    const emsar_rsh *r; const emsar_aln_opts *o; const bamreader *master;     /* master == NULL: the batches are text lines (count_text_parallel) */
    int64_t err_index; int rc; char err[256];                        /* the failing batch that comes first in the file */
} bam_pool;
typedef struct { bam_pool *pool; emsar_counts *c; int got; pthread_t th; int started; } bam_worker;

static void *bam_worker_main(void *a) {
    bam_worker *w = (bam_worker *)a;
    bam_pool *P = w->pool;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        while (!P->work_head && !P->closing) pthread_cond_wait(&P->cv_work, &P->mu);
        bam_batch *b = P->work_head;
        if (b) { P->work_head = b->next; if (!P->work_head) P->work_tail = NULL; }
        pthread_mutex_unlock(&P->mu);
        if (!b) return NULL;
        emsar_counts *c = NULL; int got = 0; char err[256]; err[0] = 0;
        int rc;
        if (P->master) {
            bamreader view = *P->master;                              /* header and refID -> tid table shared read-only */
            view.f = NULL; view.pf = NULL; view.buf = NULL; view.cap = 0;
            view.mem = b->buf; view.mem_n = b->n; view.mem_pos = 0;
            rc = w->c ? count_range(P->r, NULL, P->o, 0, -1, &c, &got, err, sizeof err, &view, w->c, NULL) : EMSAR_HOST_ERR_OOM;
        } else {
            memtext view = {(char *)b->buf, b->n, 0};
            rc = w->c ? count_range(P->r, NULL, P->o, 0, -1, &c, &got, err, sizeof err, NULL, w->c, &view) : EMSAR_HOST_ERR_OOM;
        }
        w->got |= got;
        pthread_mutex_lock(&P->mu);
        if (rc != EMSAR_HOST_OK && (P->rc == EMSAR_HOST_OK || b->index < P->err_index)) {
            P->rc = rc; P->err_index = b->index; snprintf(P->err, sizeof P->err, "%s", err[0] ? err : "out of memory");
        }
        b->next = P->free_list; P->free_list = b;
        pthread_cond_signal(&P->cv_free);
        pthread_mutex_unlock(&P->mu);
    }
}

static bam_batch *bam_batch_get(bam_pool *P) {                        /* blocks until a worker returns one */
    pthread_mutex_lock(&P->mu);
    while (!P->free_list) pthread_cond_wait(&P->cv_free, &P->mu);
    bam_batch *b = P->free_list;
    P->free_list = b->next;
    pthread_mutex_unlock(&P->mu);
    b->n = 0; b->next = NULL;
    return b;
}
static void bam_batch_put(bam_pool *P, bam_batch *b) {
    pthread_mutex_lock(&P->mu);
    if (P->work_tail) P->work_tail->next = b; else P->work_head = b;
    P->work_tail = b;
    pthread_cond_signal(&P->cv_work);
    pthread_mutex_unlock(&P->mu);
}

/* returns -100 when the input is not for this path (the caller then runs the one-thread loop) */
static int count_bam_parallel(const emsar_rsh *r, const char *path, const emsar_aln_opts *o, emsar_counts **out,
                              char *err, size_t errlen) {
    const int nt = host_threads();
    if (o->format != 2 || nt <= 1 || !path || !path[0] || strcmp(path, "-") == 0) return -100;
    size_t batch_bytes = (size_t)8 << 20;
    const char *e = getenv("EMSAR_HOST_RANGE_BYTES");                 /* tests: many batches on small files */
    if (e && atoll(e) > 0) batch_bytes = (size_t)atoll(e);
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode) || (!e && st.st_size < (1 << 20))) return -100;
    bamreader *bam = bam_open(path, r);
    if (!bam) { if (err) snprintf(err, errlen, "can't open BAM file %s", path); return EMSAR_HOST_ERR_IO; }
    if (!bam->pf) { bam_close(bam); return -100; }                    /* plain gzip: zlib on one thread */
    for (int i = 0; i < bam->n_ref; i++) bam->ref_tid[i] = emsar_rsh_tid_of(r, bam->ref[i]);   /* workers only read the table */

    int rc = EMSAR_HOST_OK;
    const int nq = 2 * nt;
    bam_pool P;
    memset(&P, 0, sizeof P);
    pthread_mutex_init(&P.mu, NULL); pthread_cond_init(&P.cv_work, NULL); pthread_cond_init(&P.cv_free, NULL);
    P.r = r; P.o = o; P.master = bam; P.rc = EMSAR_HOST_OK;
    bam_batch *all = (bam_batch *)calloc((size_t)nq, sizeof(*all));
    bam_worker *w = (bam_worker *)calloc((size_t)nt, sizeof(*w));
    int n_started = 0;
    if (!all || !w) rc = EMSAR_HOST_ERR_OOM;
    for (int i = 0; i < nq && rc == EMSAR_HOST_OK; i++) {
        all[i].cap = batch_bytes + (1 << 16);
        all[i].buf = (unsigned char *)malloc(all[i].cap);
        if (!all[i].buf) rc = EMSAR_HOST_ERR_OOM;
        all[i].next = P.free_list; P.free_list = &all[i];
    }
    for (int t = 0; t < nt && rc == EMSAR_HOST_OK; t++) {
        w[t].pool = &P;
        w[t].c = counts_new(r);
        if (!w[t].c) { rc = EMSAR_HOST_ERR_OOM; break; }
        if (pthread_create(&w[t].th, NULL, bam_worker_main, &w[t]) == 0) { w[t].started = 1; n_started++; }
    }
    if (rc == EMSAR_HOST_OK && n_started == 0) rc = EMSAR_HOST_ERR_OOM;

    int64_t n_batches = 0;
    int malformed = 0;
    if (rc == EMSAR_HOST_OK) {
        bam_batch *cur = bam_batch_get(&P);
        /* The stream is read into the batch buffer in large pieces (one memcpy per BGZF block instead of two calls per record) and
         * hopped in place: per record the walk reads its length, refID, FLAG and -- only where a cut is possible, i.e. beyond
         * batch_bytes -- compares its read name with the previous kept record's.  (Round 2 copied every record by itself and
         * compared every name: 66 ns per record, 5.7 s for the 86 M records of a config-4 sample on ONE thread while the 16 workers
         * and the GPU waited.) */
        size_t fill = 0, at = 0;                                      /* bytes in cur->buf; start of the next record to hop over */
        size_t prev_off = 0, prev_n = 0; int have_prev = 0;           /* read name of the previous kept record: offset in cur->buf */
        int eof = 0;
        /* paired-end: the worker skips an unaligned record singly and reads an aligned one together with the record after
         * it (emsar_functions.c:514-520), so the pairs are known from refID alone; the first mate is held until the second
         * has shown whether the pair is kept */
        int have_r1 = 0; size_t r1_off = 0; int r1_flag = 0, r1_pos = 0, r1_ok = 0;
        while (rc == EMSAR_HOST_OK && !malformed) {
            /* a whole record at `at`?  else read on (the buffer grows for a record, or a read group, longer than its slack) */
            int32_t bs = -1;
            if (fill - at >= 4) {
                bs = le32(cur->buf + at);
                if (bs < 32 || bs > (1 << 28)) { malformed = 1; break; }
            }
            if (bs < 0 || fill - at < 4 + (size_t)bs) {
                if (eof) { if (fill != at) malformed = 1; break; }
                size_t want = batch_bytes + (1 << 15);
                if (bs >= 0 && at + 4 + (size_t)bs > want) want = at + 4 + (size_t)bs;
                if (fill >= want) want = fill + (1 << 16);
                if (want > cur->cap) {
                    unsigned char *nb = (unsigned char *)realloc(cur->buf, want + (1 << 16));
                    if (!nb) { rc = EMSAR_HOST_ERR_OOM; break; }
                    cur->buf = nb; cur->cap = want + (1 << 16);
                }
                const long got = bam_get(bam, cur->buf + fill, want - fill);
                if (got < 0) { malformed = 1; break; }
                if ((size_t)got < want - fill) eof = 1;
                fill += (size_t)got;
                continue;
            }
            const unsigned char *q = cur->buf + at + 4;
            const int32_t refid = le32(q), pos = le32(q + 4);
            const int l_name = q[8], flag = q[14] | (q[15] << 8);
            const int aligned = refid >= 0 && refid < bam->n_ref;
            const int name_ok = 32 + (size_t)l_name <= (size_t)bs && l_name >= 1;
            size_t unit = at;                                         /* where the record (or its pair) begins in the batch */
            at += 4 + (size_t)bs;
            int kept;
            if (!o->pe) {
                const char strand = (flag & 0x10) ? '-' : '+';
                kept = aligned && name_ok && !(o->strand != 0 && o->strand != strand);
            } else if (!have_r1) {
                if (aligned) { have_r1 = 1; r1_off = unit; r1_flag = flag; r1_pos = pos; r1_ok = name_ok; }
                continue;
            } else {                                                  /* this record is the second mate of the pair at r1_off */
                have_r1 = 0;
                unit = r1_off;
                int p1, p2; char s1, s2; int grouped = 1;
                if ((r1_flag & 0x40) && (flag & 0x80)) { p1 = r1_pos; p2 = pos; s1 = (r1_flag & 0x10) ? '-' : '+'; s2 = (flag & 0x10) ? '-' : '+'; }
                else if ((flag & 0x40) && (r1_flag & 0x80)) { p1 = pos; p2 = r1_pos; s1 = (flag & 0x10) ? '-' : '+'; s2 = (r1_flag & 0x10) ? '-' : '+'; }
                else { grouped = 0; p1 = p2 = 0; s1 = s2 = '+'; }        /* the worker stops here: "mates are not grouped" */
                if (p2 > p1) kept = !(o->strand == '-') && (s1 == '+' && s2 == '-');
                else kept = !(o->strand == '+') && (s1 == '-' && s2 == '+');
                kept = kept && grouped && r1_ok;
            }
            if (!kept) continue;
            const size_t nm_off = unit + 4 + 32;                       /* read name of the record, or of the first mate */
            const size_t nl = strnlen((const char *)cur->buf + nm_off, (size_t)cur->buf[unit + 4 + 8]);
            if (unit >= batch_bytes) {                                /* a cut is possible here: does this record (pair) open a new read group? */
                const int new_group = !have_prev || nl != prev_n || memcmp(cur->buf + prev_off, cur->buf + nm_off, nl) != 0;
                if (new_group) {                                      /* it opens the next batch: the tail of the buffer moves there */
                    const size_t moved = fill - unit;
                    bam_batch *nb = bam_batch_get(&P);
                    if (moved + (1 << 16) > nb->cap) {
                        unsigned char *g = (unsigned char *)realloc(nb->buf, moved + (1 << 17));
                        if (!g) { bam_batch_put(&P, nb); rc = EMSAR_HOST_ERR_OOM; break; }
                        nb->buf = g; nb->cap = moved + (1 << 17);
                    }
                    memcpy(nb->buf, cur->buf + unit, moved);
                    cur->n = unit;
                    cur->index = n_batches++;
                    bam_batch_put(&P, cur);
                    cur = nb;
                    fill = moved; at -= unit;
                    prev_off = nm_off - unit; prev_n = nl; have_prev = 1;
                    pthread_mutex_lock(&P.mu);
                    const int failed = P.rc != EMSAR_HOST_OK;
                    pthread_mutex_unlock(&P.mu);
                    if (failed) { fill = at = 0; break; }             /* a batch already failed: whatever follows cannot come first */
                    continue;
                }
            }
            prev_off = nm_off; prev_n = nl; have_prev = 1;
        }
        cur->n = (rc == EMSAR_HOST_OK && !malformed) ? fill : 0;       /* (an unpaired first mate at the end of the file stays in: the worker's loop decides, as before) */
        cur->index = n_batches++;
        bam_batch_put(&P, cur);                                       /* the last batch (possibly empty) */
    }
    pthread_mutex_lock(&P.mu);
    P.closing = 1;
    pthread_cond_broadcast(&P.cv_work);
    pthread_mutex_unlock(&P.mu);
    for (int t = 0; t < nt; t++) if (w && w[t].started) pthread_join(w[t].th, NULL);

    emsar_counts *c = NULL;
    int got = 0;
    if (rc == EMSAR_HOST_OK && P.rc != EMSAR_HOST_OK) { rc = P.rc; if (err) snprintf(err, errlen, "%s", P.err); }
    if (rc == EMSAR_HOST_OK && malformed) { rc = EMSAR_HOST_ERR_FORMAT; if (err) snprintf(err, errlen, "malformed BAM record"); }
    if (rc == EMSAR_HOST_ERR_OOM && err) snprintf(err, errlen, "out of memory");
    for (int t = 0; rc == EMSAR_HOST_OK && o->collapse && t < nt; t++)
        if (w[t].c && batch_flush(r, o, w[t].c) != 0) { rc = EMSAR_HOST_ERR_IO; if (err) snprintf(err, errlen, "the collapse of read-level rows failed"); }
    if (rc == EMSAR_HOST_OK) {
        for (int t = 0; t < nt && rc == EMSAR_HOST_OK; t++) {
            got |= w[t].got;
            if (!c) { c = w[t].c; w[t].c = NULL; continue; }
            counts_add(c, w[t].c);
            if (w[t].c->readlength != r->hdr_readlength) {            /* learnt from the data (paired-end, header says -1) */
                if (c->readlength == r->hdr_readlength) c->readlength = w[t].c->readlength;
                else if (c->readlength != w[t].c->readlength) {
                    rc = EMSAR_HOST_ERR_FORMAT;
                    if (err) snprintf(err, errlen, "paired-end data with variable read length is not supported");
                }
            }
        }
        if (rc == EMSAR_HOST_OK && !got) {
            rc = EMSAR_HOST_ERR_FORMAT;
            if (err) snprintf(err, errlen, "no usable alignment in %s (the reference stops with 'NULL alignment list')", path);
        }
    }
    if (getenv("EMSAR_HOST_DEBUG")) fprintf(stderr, "emsar_count_alignments: BAM, %lld batch(es) on %d thread(s)\n", (long long)n_batches, n_started);
    for (int t = 0; w && t < nt; t++) emsar_counts_free(w[t].c);
    for (int i = 0; all && i < nq; i++) free(all[i].buf);
    free(all); free(w);
    pthread_mutex_destroy(&P.mu); pthread_cond_destroy(&P.cv_work); pthread_cond_destroy(&P.cv_free);
    bam_close(bam);
    if (rc != EMSAR_HOST_OK) { emsar_counts_free(c); return rc; }
    *out = c;
    return EMSAR_HOST_OK;
}

/* ---- text that cannot be cut into byte ranges (gzip, stdin, paired-end SAM, mates not named /1 /2) counted by the same pool ----
 * The calling thread reads the lines (zlib inflates on it), copies them into batches and decides where a batch may end: before a unit
 * (a record; paired-end: the two records the sequential loop would read together) that is KEPT and whose read id differs from the last
 * kept unit's -- the same rule as for BAM above.  "Kept" and the id are worked out here from the first four fields with the tests of
 * count_range (strand, '*' reference, mate orientation); a worker then runs count_range itself on its batches.  Whatever this thread
 * cannot judge cheaply and exactly (a malformed line, an id longer than the buffers: the worker will report the error, or not) ends the
 * cutting: everything after it goes into one last batch, which is always right. */
/* the first four tab-separated fields of [p, p + n) and min(number of fields, need), in one pass */
static int first_fields(const char *p, size_t n, int need, const char *f[4], size_t fl[4]) {
    const char *e = p + n;
    int nf = 0;
    for (;;) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
        if (nf < 4) { f[nf] = p; fl[nf] = (size_t)((t ? t : e) - p); }
        nf++;
        if (!t || nf >= need) return nf;
        p = t + 1;
    }
}
static int span_atoi(const char *p, size_t n) { char b[32]; if (n > 31) n = 31; memcpy(b, p, n); b[n] = 0; return atoi(b); }
#define TXT_ID_MAX 2048

static int count_text_parallel(const emsar_rsh *r, const char *path, const emsar_aln_opts *o, emsar_counts **out, char *err, size_t errlen) {
    const int nt = host_threads();
    if (o->format == 2 || nt <= 1) return -100;
    size_t batch_bytes = (size_t)4 << 20;
    const char *e = getenv("EMSAR_HOST_RANGE_BYTES");                 /* tests: many batches on small files */
    if (e && atoll(e) > 0) batch_bytes = (size_t)atoll(e);
    const int named = path && path[0] && strcmp(path, "-") != 0;
    if (named && !e) { struct stat st; if (stat(path, &st) == 0 && S_ISREG(st.st_mode) && st.st_size < (1 << 18)) return -100; }   /* small: not worth a pool */
    /* Not for gzip: there the reader is zlib's inflate plus this thread's copy and field scan, which together cost more than the one-thread
     * loop saves (tests/perf/text_pool_bench.py: 0.82 s against 0.61 s for 1.3 M gzipped bowtie records); EMSAR_HOST_TEXT_POOL=gz forces it
     * (tests).  A named file is looked at through a reader of its own, so that the one-thread loop can still start from the beginning. */
    const char *force = getenv("EMSAR_HOST_TEXT_POOL");
    if (named && !(force && strcmp(force, "gz") == 0)) {
        void *probe = emsar_lr_open(path);
        if (!probe) { if (err) snprintf(err, errlen, "can't open alignment file %s", path); return EMSAR_HOST_ERR_IO; }
        const int plain = emsar_lr_is_plain(probe);
        emsar_lr_close(probe);
        if (!plain) return -100;
    }
    void *lr = emsar_lr_open(path);
    if (!lr) { if (err) snprintf(err, errlen, "can't open alignment file %s", path ? path : "-"); return EMSAR_HOST_ERR_IO; }

    int rc = EMSAR_HOST_OK;
    const int nq = 2 * nt;
    bam_pool P;
    memset(&P, 0, sizeof P);
    pthread_mutex_init(&P.mu, NULL); pthread_cond_init(&P.cv_work, NULL); pthread_cond_init(&P.cv_free, NULL);
    P.r = r; P.o = o; P.master = NULL; P.rc = EMSAR_HOST_OK;
    bam_batch *all = (bam_batch *)calloc((size_t)nq, sizeof(*all));
    bam_worker *w = (bam_worker *)calloc((size_t)nt, sizeof(*w));
    char *prev = (char *)malloc(TXT_ID_MAX), *id1 = (char *)malloc(TXT_ID_MAX), *id2 = (char *)malloc(TXT_ID_MAX);
    int n_started = 0;
    if (!all || !w || !prev || !id1 || !id2) rc = EMSAR_HOST_ERR_OOM;
    for (int i = 0; i < nq && rc == EMSAR_HOST_OK; i++) {
        all[i].cap = batch_bytes + (1 << 16);
        all[i].buf = (unsigned char *)malloc(all[i].cap);
        if (!all[i].buf) rc = EMSAR_HOST_ERR_OOM;
        all[i].next = P.free_list; P.free_list = &all[i];
    }
    for (int t = 0; t < nt && rc == EMSAR_HOST_OK; t++) {
        w[t].pool = &P;
        w[t].c = counts_new(r);
        if (!w[t].c) { rc = EMSAR_HOST_ERR_OOM; break; }
        if (pthread_create(&w[t].th, NULL, bam_worker_main, &w[t]) == 0) { w[t].started = 1; n_started++; }
    }
    if (rc == EMSAR_HOST_OK && n_started == 0) rc = EMSAR_HOST_ERR_OOM;

    int64_t n_batches = 0;
    if (rc == EMSAR_HOST_OK) {
        bam_batch *cur = bam_batch_get(&P);
        size_t fill = 0;
        size_t prev_n = 0; int have_prev = 0;
        int unsure = 0;                                               /* no more cuts from here on */
        int have_1 = 0; size_t off_1 = 0, len_1 = 0;                   /* paired-end: the first record of the pair, in cur->buf */
        char *line;
        while ((line = emsar_lr_next(lr)) != NULL) {
            const size_t ll = strlen(line);
            if (fill + ll + 2 > cur->cap) {
                const size_t want = (fill + ll + 2) * 2;
                unsigned char *nb = (unsigned char *)realloc(cur->buf, want);
                if (!nb) { rc = EMSAR_HOST_ERR_OOM; break; }
                cur->buf = nb; cur->cap = want;
            }
            const size_t lo = fill;
            memcpy(cur->buf + fill, line, ll); cur->buf[fill + ll] = '\n'; fill += ll + 1;
            if (unsure) continue;
            const char *L = (const char *)cur->buf + lo;
            /* ---- the unit this line completes, if any: kept?, id, where it starts ---- */
            int kept = 0; size_t unit = lo; const char *id = NULL; size_t idn = 0;
            const char *F[4]; size_t FN[4];
            if (o->format == 1 && !have_1 && L[0] == '@') continue;   /* SAM header line: skipped by the loop */
            if (o->format == 1 && have_1 && ll > 0 && L[0] == '@') {  /* a header line where the mate should be ends the sequential loop (count_range: st == 0) */
                fill = lo;                                            /* nothing from here on is read */
                break;
            }
            const int need = o->format == 1 ? 11 : 7;
            if (first_fields(L, ll, need, F, FN) < need || FN[0] >= TXT_ID_MAX - 1 || FN[1] == 0) { unsure = 1; continue; }
            const char *f0 = F[0], *f1 = F[1], *f2 = F[2], *f3 = F[3];
            const size_t n0 = FN[0], n1 = FN[1], n2 = FN[2], n3 = FN[3];
            if (!o->pe) {
                if (o->format == 1) {
                    if (n2 == 1 && f2[0] == '*') continue;            /* unaligned: skipped */
                    const char strand = (span_atoi(f1, n1) & 0x10) ? '-' : '+';
                    kept = !(o->strand != 0 && o->strand != strand);
                    idn = n0 < 1023 ? n0 : 1023;                      /* samrec.qname holds 1023 characters */
                } else {
                    kept = !(o->strand != 0 && o->strand != f1[0]);
                    idn = n0;
                }
                id = f0;
            } else if (o->format == 1) {                              /* paired-end SAM: an unaligned record is skipped singly, an aligned one is read with its successor */
                if (!have_1) {
                    if (n2 == 1 && f2[0] == '*') continue;
                    have_1 = 1; off_1 = lo; len_1 = ll;
                    continue;
                }
                have_1 = 0; unit = off_1;
                const char *M = (const char *)cur->buf + off_1;
                const char *G4[4]; size_t GN[4];
                if (first_fields(M, len_1, 4, G4, GN) < 4) { unsure = 1; continue; }
                const char *g0 = G4[0], *g1 = G4[1], *g3 = G4[3];
                const size_t m0 = GN[0], m1 = GN[1], m3 = GN[3];
                const int fl1 = span_atoi(g1, m1), fl2 = span_atoi(f1, n1), q1 = span_atoi(g3, m3) - 1, q2 = span_atoi(f3, n3) - 1;
                int p1, p2; char s1, s2;
                if ((fl1 & 0x40) && (fl2 & 0x80)) { p1 = q1; p2 = q2; s1 = (fl1 & 0x10) ? '-' : '+'; s2 = (fl2 & 0x10) ? '-' : '+'; }
                else if ((fl2 & 0x40) && (fl1 & 0x80)) { p1 = q2; p2 = q1; s1 = (fl2 & 0x10) ? '-' : '+'; s2 = (fl1 & 0x10) ? '-' : '+'; }
                else { unsure = 1; continue; }                        /* "mates are not grouped": the worker says so */
                if (p2 > p1) kept = !(o->strand == '-') && (s1 == '+' && s2 == '-');
                else kept = !(o->strand == '+') && (s1 == '-' && s2 == '+');
                id = g0; idn = m0 < 1023 ? m0 : 1023;
            } else {                                                  /* paired-end bowtie: two records at a time */
                if (!have_1) { have_1 = 1; off_1 = lo; len_1 = ll; continue; }
                have_1 = 0; unit = off_1;
                const char *M = (const char *)cur->buf + off_1;
                const char *G4[4]; size_t GN[4];
                if (first_fields(M, len_1, 7, G4, GN) < 7 || GN[0] >= TXT_ID_MAX - 1 || GN[1] == 0) { unsure = 1; continue; }
                const char *g0 = G4[0], *g1 = G4[1], *g2 = G4[2], *g3 = G4[3];
                const size_t m0 = GN[0], m2 = GN[2], m3 = GN[3];
                memcpy(id1, g0, m0); id1[m0] = 0; memcpy(id2, f0, n0); id2[n0] = 0;   /* record 1 = f of count_range, record 2 = g */
                const int idlen = mate_id_len(id1, id2);
                if (idlen == 0) { unsure = 1; continue; }             /* "mate read IDs don't match": the worker says so */
                if (m2 == n2 && memcmp(g2, f2, n2) == 0) {            /* mates on one transcript; else no alignment */
                    /* the reference's swap (count_range): record 2 is treated as mate 1 */
                    const int p1 = span_atoi(f3, n3), p2 = span_atoi(g3, m3);
                    const char s1 = f1[0], s2 = g1[0];
                    if (p2 > p1) kept = !(o->strand == '-') && (s1 == '+' && s2 == '-');
                    else kept = !(o->strand == '+') && (s1 == '-' && s2 == '+');
                }
                id = g0; idn = (size_t)idlen < 1023 ? (size_t)idlen : 1023;
            }
            if (!kept) continue;
            const int new_group = !have_prev || idn != prev_n || memcmp(prev, id, idn) != 0;
            memcpy(prev, id, idn); prev_n = idn; have_prev = 1;      /* (id points into cur->buf: copied before the buffer changes hands) */
            if (unit >= batch_bytes && new_group) {                   /* the unit opens the next batch: the tail of the buffer moves there */
                const size_t moved = fill - unit;
                bam_batch *nb = bam_batch_get(&P);
                if (moved + 2 > nb->cap) {
                    unsigned char *g = (unsigned char *)realloc(nb->buf, moved + (1 << 16));
                    if (!g) { bam_batch_put(&P, nb); rc = EMSAR_HOST_ERR_OOM; break; }
                    nb->buf = g; nb->cap = moved + (1 << 16);
                }
                memcpy(nb->buf, cur->buf + unit, moved);
                cur->n = unit; cur->index = n_batches++;
                bam_batch_put(&P, cur);
                cur = nb; fill = moved;
                pthread_mutex_lock(&P.mu);
                const int failed = P.rc != EMSAR_HOST_OK;
                pthread_mutex_unlock(&P.mu);
                if (failed) { fill = 0; break; }                      /* a batch already failed: whatever follows cannot come first */
            }
        }
        cur->n = rc == EMSAR_HOST_OK ? fill : 0;
        cur->index = n_batches++;
        bam_batch_put(&P, cur);
    }
    pthread_mutex_lock(&P.mu);
    P.closing = 1;
    pthread_cond_broadcast(&P.cv_work);
    pthread_mutex_unlock(&P.mu);
    for (int t = 0; t < nt; t++) if (w && w[t].started) pthread_join(w[t].th, NULL);
    emsar_lr_close(lr);

    emsar_counts *c = NULL;
    int got = 0;
    if (rc == EMSAR_HOST_OK && P.rc != EMSAR_HOST_OK) { rc = P.rc; if (err) snprintf(err, errlen, "%s", P.err); }
    if (rc == EMSAR_HOST_ERR_OOM && err) snprintf(err, errlen, "out of memory");
    for (int t = 0; rc == EMSAR_HOST_OK && o->collapse && t < nt; t++)
        if (w[t].c && batch_flush(r, o, w[t].c) != 0) { rc = EMSAR_HOST_ERR_IO; if (err) snprintf(err, errlen, "the collapse of read-level rows failed"); }
    if (rc == EMSAR_HOST_OK) {
        for (int t = 0; t < nt && rc == EMSAR_HOST_OK; t++) {
            got |= w[t].got;
            if (!c) { c = w[t].c; w[t].c = NULL; continue; }
            counts_add(c, w[t].c);
            if (w[t].c->readlength != r->hdr_readlength) {            /* learnt from the data (paired-end, header says -1) */
                if (c->readlength == r->hdr_readlength) c->readlength = w[t].c->readlength;
                else if (c->readlength != w[t].c->readlength) {
                    rc = EMSAR_HOST_ERR_FORMAT;
                    if (err) snprintf(err, errlen, "paired-end data with variable read length is not supported");
                }
            }
        }
        if (rc == EMSAR_HOST_OK && !got) {
            rc = EMSAR_HOST_ERR_FORMAT;
            if (err) snprintf(err, errlen, "no usable alignment in %s (the reference stops with 'NULL alignment list')", path ? path : "-");
        }
    }
    if (getenv("EMSAR_HOST_DEBUG")) fprintf(stderr, "emsar_count_alignments: text, %lld batch(es) on %d thread(s)\n", (long long)n_batches, n_started);
    for (int t = 0; w && t < nt; t++) emsar_counts_free(w[t].c);
    for (int i = 0; all && i < nq; i++) free(all[i].buf);
    free(all); free(w); free(prev); free(id1); free(id2);
    pthread_mutex_destroy(&P.mu); pthread_cond_destroy(&P.cv_work); pthread_cond_destroy(&P.cv_free);
    if (rc != EMSAR_HOST_OK) { emsar_counts_free(c); return rc; }
    *out = c;
    return EMSAR_HOST_OK;
}

int emsar_count_alignments(const emsar_rsh *r, const char *path, const emsar_aln_opts *o, emsar_counts **out,
                           char *err, size_t errlen) {
    *out = NULL;
    {
        int prc = count_bam_parallel(r, path, o, out, err, errlen);
        if (prc != -100) return prc;
    }
    int64_t size = 0;
    const int nt = plan_ranges(path, o, &size);
    int rc = EMSAR_HOST_OK, got = 0;
    if (getenv("EMSAR_HOST_DEBUG")) fprintf(stderr, "emsar_count_alignments: %d range(s) over %lld bytes\n", nt, (long long)size);
    if (nt <= 1) {
        const int prc = count_text_parallel(r, path, o, out, err, errlen);       /* gzip, stdin, paired-end SAM: one reader, a pool of counters */
        if (prc != -100) return prc;
        rc = count_range(r, path, o, 0, -1, out, &got, err, errlen, NULL, NULL, NULL);
        if (rc == EMSAR_HOST_OK && !got) {
            emsar_counts_free(*out); *out = NULL;
            if (err) snprintf(err, errlen, "no usable alignment in %s (the reference stops with 'NULL alignment list')", path);
            return EMSAR_HOST_ERR_FORMAT;
        }
        return rc;
    }
    range_job *job = (range_job *)calloc((size_t)nt, sizeof(*job));
    pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(*th));
    if (!job || !th) { free(job); free(th); return EMSAR_HOST_ERR_OOM; }
    for (int t = 0; t < nt; t++) {
        job[t].r = r; job[t].path = path; job[t].o = o;
        job[t].begin = size / nt * t; job[t].end = t + 1 < nt ? size / nt * (t + 1) : -1;
    }
    int started = 0;
    for (int t = 1; t < nt; t++) { if (pthread_create(&th[t], NULL, range_main, &job[t]) != 0) break; started = t; }
    for (int t = started + 1; t < nt; t++) range_main(&job[t]);      /* could not spawn: run it here */
    range_main(&job[0]);
    for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
    emsar_counts *c = NULL;
    for (int t = 0; t < nt; t++) {                                   /* the first failing range (in file order) names the error */
        if (job[t].rc != EMSAR_HOST_OK && rc == EMSAR_HOST_OK) { rc = job[t].rc; if (err) snprintf(err, errlen, "%s", job[t].err); }
        got |= job[t].got;
    }
    if (rc == EMSAR_HOST_OK && !got) {
        rc = EMSAR_HOST_ERR_FORMAT;
        if (err) snprintf(err, errlen, "no usable alignment in %s (the reference stops with 'NULL alignment list')", path);
    }
    if (rc == EMSAR_HOST_OK) {
        c = job[0].c; job[0].c = NULL;
        for (int t = 1; t < nt && rc == EMSAR_HOST_OK; t++) {
            const emsar_counts *q = job[t].c;
            counts_add(c, q);
            if (q->readlength != r->hdr_readlength) {                /* learnt from the data (paired-end, header says -1) */
                if (c->readlength == r->hdr_readlength) c->readlength = q->readlength;
                else if (c->readlength != q->readlength) {
                    rc = EMSAR_HOST_ERR_FORMAT;
                    if (err) snprintf(err, errlen, "paired-end data with variable read length is not supported");
                }
            }
        }
    }
    for (int t = 0; t < nt; t++) emsar_counts_free(job[t].c);
    free(job); free(th);
    if (rc != EMSAR_HOST_OK) { emsar_counts_free(c); return rc; }
    *out = c;
    return EMSAR_HOST_OK;
}

void emsar_counts_free(emsar_counts *c) {
    if (!c) return;
    batch_free((row_batch *)c->batch);
    free(c->R); free(c->frag_counts); free(c);
}
