/* rsh.c -- reader of the EMSAR .rsh text index into flat arrays.
 *
 * Format (written by print_rsh, /root/reference/src/emsar_functions.c:2085-2127; read by
 * construct_rsh_from_rshfile / parse_rsh_*line, emsar_functions.c:1351-1510):
 *     #max_tid,max_t_size,minfrag,maxfrag,readlength          readlength -1 for single-end
 *     @<tid>\t<name>                                          one per transcript
 *     cid\tno.tids\t...                                        heading line (anything starting with 'c' is skipped)
 *     cid\tn\tfirst_tid\tother,tids,\tE0,E1,...,              one per segment; empty last field = no node
 * Row (cid) order is the one scan_rshbucket produces (emsar_functions.c:2149-2191): one single-tid row per tid
 * (present even when the transcript has no unique region), then multi-tid rows by size, first tid, list order.
 * Where the reference keeps a pointer structure (rshbucket[size-2][first_tid] -> sorted linked list), we keep
 * CSR plus a hash from the sorted tid multiset to the row.
 */
#include "emsar_host.h"

#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

/* ---------- line reader over zlib (reads plain text and .gz alike) ----------
 * gzread() in 4 MiB pieces and memchr() for the line ends: gzgets() costs more per line than the parsing of a bowtie
 * record does.  A returned line is NUL-terminated in place and stays valid until the next call. */
typedef struct { gzFile f; char *buf; size_t cap, pos, len; int eof; int64_t base /* file offset of buf[0] */, last /* of the last returned line */; } linereader;

static int lr_open(linereader *lr, const char *path) {
    lr->f = (path && strcmp(path, "-") != 0 && path[0]) ? gzopen(path, "rb") : gzdopen(0, "rb");
    lr->cap = (size_t)4 << 20;
    lr->buf = (char *)malloc(lr->cap + 1);
    lr->pos = lr->len = 0; lr->eof = 0; lr->base = 0; lr->last = 0;
    if (lr->f) gzbuffer(lr->f, 1u << 20);
    return (lr->f && lr->buf) ? 0 : -1;
}
/* returns NULL at EOF (or when out of memory / on a read error); strips the trailing newline and a CR before it */
static char *lr_next(linereader *lr) {
    for (;;) {
        char *start = lr->buf + lr->pos;
        char *nl = lr->len > lr->pos ? (char *)memchr(start, '\n', lr->len - lr->pos) : NULL;
        if (nl) {
            *nl = 0;
            lr->last = lr->base + (int64_t)lr->pos;
            lr->pos = (size_t)(nl - lr->buf) + 1;
            if (nl > start && nl[-1] == '\r') nl[-1] = 0;
            return start;
        }
        if (lr->eof) {                                   /* last line without a newline */
            if (lr->pos >= lr->len) return NULL;
            lr->buf[lr->len] = 0;
            lr->last = lr->base + (int64_t)lr->pos;
            lr->pos = lr->len;
            size_t n = strlen(start);
            if (n && start[n - 1] == '\r') start[n - 1] = 0;
            return start;
        }
        /* move the unfinished line to the front, grow if it fills the buffer, read on */
        size_t rest = lr->len - lr->pos;
        if (lr->pos > 0) { memmove(lr->buf, start, rest); lr->base += (int64_t)lr->pos; lr->pos = 0; lr->len = rest; }
        if (lr->len == lr->cap) {
            char *nb = (char *)realloc(lr->buf, lr->cap * 2 + 1);
            if (!nb) return NULL;
            lr->buf = nb; lr->cap *= 2;
        }
        int got = gzread(lr->f, lr->buf + lr->len, (unsigned)(lr->cap - lr->len));
        if (got < 0) return NULL;
        if (got == 0) lr->eof = 1;
        lr->len += (size_t)got;
    }
}
static void lr_close(linereader *lr) { if (lr->f) gzclose(lr->f); free(lr->buf); }

/* exported to align.c */
void *emsar_lr_open(const char *path) { linereader *lr = (linereader *)calloc(1, sizeof(*lr)); if (!lr) return NULL; if (lr_open(lr, path)) { lr_close(lr); free(lr); return NULL; } return lr; }
char *emsar_lr_next(void *h) { return lr_next((linereader *)h); }
void emsar_lr_close(void *h) { if (h) { lr_close((linereader *)h); free(h); } }
/* for the chunked readers of align.c: offset of the line returned last; plain (seekable, uncompressed) file?; open a
 * plain file so that the first line returned is the first one STARTING at or after `offset` */
int64_t emsar_lr_offset(void *h) { return ((linereader *)h)->last; }
int emsar_lr_is_plain(void *h) { return gzdirect(((linereader *)h)->f) ? 1 : 0; }
void *emsar_lr_open_at(const char *path, int64_t offset) {
    linereader *lr = (linereader *)emsar_lr_open(path);
    if (!lr || offset <= 0) return lr;
    if (gzseek(lr->f, (z_off_t)(offset - 1), SEEK_SET) < 0) { emsar_lr_close(lr); return NULL; }
    lr->base = offset - 1;
    if (!lr_next(lr)) { lr->eof = 1; lr->pos = lr->len; }      /* the rest of the line that holds byte offset-1 (just "\n" if a line starts at offset) */
    return lr;
}

/* ---------- name -> tid (the reference uses a character trie, stringhash.c) ---------- */
typedef struct { uint32_t cap; int32_t *slot; char **names; } name_index;

static uint64_t fnv1a(const void *p, size_t n) {
    const unsigned char *s = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ull; }
    return h;
}
static name_index *ni_build(char **names, int32_t n) {
    name_index *x = (name_index *)calloc(1, sizeof(*x));
    if (!x) return NULL;
    x->cap = 16; while (x->cap < (uint32_t)n * 2u + 2u) x->cap <<= 1;
    x->slot = (int32_t *)malloc(sizeof(int32_t) * x->cap);
    if (!x->slot) { free(x); return NULL; }
    for (uint32_t i = 0; i < x->cap; i++) x->slot[i] = -1;
    x->names = names;
    for (int32_t t = 0; t < n; t++) {
        if (!names[t]) continue;
        uint32_t h = (uint32_t)fnv1a(names[t], strlen(names[t])) & (x->cap - 1);
        while (x->slot[h] >= 0) h = (h + 1) & (x->cap - 1);
        x->slot[h] = t;
    }
    return x;
}
int32_t emsar_rsh_tid_of(const emsar_rsh *r, const char *name) {
    const name_index *x = (const name_index *)r->name_index;
    uint32_t h = (uint32_t)fnv1a(name, strlen(name)) & (x->cap - 1);
    while (x->slot[h] >= 0) {
        if (strcmp(x->names[x->slot[h]], name) == 0) return x->slot[h];
        h = (h + 1) & (x->cap - 1);
    }
    return -1;
}

/* ---------- sorted tid multiset -> row ---------- */
typedef struct { uint64_t cap; int64_t *slot; } set_index;

static set_index *si_build(const emsar_rsh *r) {
    set_index *x = (set_index *)calloc(1, sizeof(*x));
    if (!x) return NULL;
    x->cap = 16; while (x->cap < (uint64_t)r->n_rows * 2u + 2u) x->cap <<= 1;
    x->slot = (int64_t *)malloc(sizeof(int64_t) * x->cap);
    if (!x->slot) { free(x); return NULL; }
    for (uint64_t i = 0; i < x->cap; i++) x->slot[i] = -1;
    for (int64_t c = r->n_tx; c < r->n_rows; c++) {           /* multi-tid rows only; singles are found by tid */
        const int32_t *t = r->col_idx + r->row_ptr[c];
        size_t n = (size_t)(r->row_ptr[c + 1] - r->row_ptr[c]);
        uint64_t h = fnv1a(t, n * sizeof(int32_t)) & (x->cap - 1);
        int dup = 0;
        while (x->slot[h] >= 0) {
            int64_t o = x->slot[h];
            if (r->row_ptr[o + 1] - r->row_ptr[o] == n && memcmp(r->col_idx + r->row_ptr[o], t, n * sizeof(int32_t)) == 0) { dup = 1; break; }
            h = (h + 1) & (x->cap - 1);
        }
        if (!dup) x->slot[h] = c;     /* a repeated segment: the reference's list walk stops at the first one too */
    }
    return x;
}
int64_t emsar_rsh_row_of(const emsar_rsh *r, const int32_t *t, int n) {
    if (n == 1) return (t[0] >= 0 && t[0] < r->n_tx && r->has_node[t[0]]) ? t[0] : -1;   /* update_rshbucket_single, 1528-1536 */
    if (n > r->max_t_size) return -1;                                                      /* update_rshbucket 'r', 1599 */
    const set_index *x = (const set_index *)r->set_index;
    uint64_t h = fnv1a(t, (size_t)n * sizeof(int32_t)) & (x->cap - 1);
    while (x->slot[h] >= 0) {
        int64_t o = x->slot[h];
        if (r->row_ptr[o + 1] - r->row_ptr[o] == (uint64_t)n && memcmp(r->col_idx + r->row_ptr[o], t, (size_t)n * sizeof(int32_t)) == 0) return o;
        h = (h + 1) & (x->cap - 1);
    }
    return -1;
}

/* ---------- parsing ---------- */
typedef struct { int32_t size, tid0; int64_t seq; int32_t *tids; int32_t *euma; } mrow;

static int cmp_mrow(const void *a, const void *b) {
    const mrow *x = (const mrow *)a, *y = (const mrow *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    if (x->tid0 != y->tid0) return x->tid0 < y->tid0 ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

/* split "a,b,c," into ints; returns count.  atoi semantics like the reference (leading blanks, a sign, digits, the rest
 * of the field ignored), without the libc call per number: the EUMA lists are most of a paired-end rsh */
static int parse_int_list(const char *s, int32_t *out, int max) {
    int n = 0;
    while (*s && n < max) {
        const char *p = s;
        while (*p == ' ' || (*p >= '\t' && *p <= '\r')) p++;
        int neg = 0;
        if (*p == '-') { neg = 1; p++; } else if (*p == '+') p++;
        uint32_t v = 0;
        while (*p >= '0' && *p <= '9') v = v * 10u + (uint32_t)(*p++ - '0');
        out[n++] = neg ? (int32_t)(0u - v) : (int32_t)v;
        while (*p && *p != ',') p++;
        if (!*p) break;
        s = p + 1;
    }
    return n;
}

#define FAIL(code, ...) do { if (err) snprintf(err, errlen, __VA_ARGS__); rc = (code); goto done; } while (0)

/* What one worker collects from its part of the file (the whole body when the file is read on one thread): the
 * multi-tid nodes, the single-tid nodes and the index lines, each in file order.  emsar_rsh_read applies them part by
 * part, so "the last line for a tid wins" (1488) and the list order of equal (size, first tid) nodes stay those of a
 * sequential read. */
typedef struct { int32_t tid; int32_t *eu; } srow;
typedef struct { int32_t tid; char *name; } nrow;
typedef struct int_chunk { struct int_chunk *next; size_t used, cap; int32_t v[]; } int_chunk;
typedef struct {
    int_chunk *ints;                                  /* tids and EUMA vectors of the part: a few big chunks, not two mallocs per line */
    const char *path; int64_t begin, end;            /* lines STARTING in [begin, end); end < 0: to the end of the file */
    linereader *lr;                                   /* given: continue on this reader instead of opening the file */
    int32_t n_tx, nfl;
    mrow *multi; size_t n_multi, cap_multi;
    srow *single; size_t n_single, cap_single;
    nrow *names; size_t n_names, cap_names;
    int32_t max_t_size;
    int rc; char err[256];
} rsh_part;

static int32_t *part_ints(rsh_part *j, size_t n, int zero) {
    if (!j->ints || j->ints->cap - j->ints->used < n) {
        size_t cap = n > ((size_t)1 << 20) ? n : ((size_t)1 << 20);
        int_chunk *c = (int_chunk *)malloc(sizeof(int_chunk) + cap * sizeof(int32_t));
        if (!c) return NULL;
        c->next = j->ints; c->used = 0; c->cap = cap; j->ints = c;
    }
    int32_t *p = j->ints->v + j->ints->used;
    j->ints->used += n;
    if (zero) memset(p, 0, n * sizeof(int32_t));
    return p;
}

static void *rsh_part_main(void *a) {
    rsh_part *j = (rsh_part *)a;
    char *err = j->err; const size_t errlen = sizeof j->err;
    int rc = EMSAR_HOST_OK;
    linereader *lr = j->lr ? j->lr : (linereader *)emsar_lr_open_at(j->path, j->begin);
    if (!lr) FAIL(EMSAR_HOST_ERR_IO, "can't open input rsh file %s", j->path);
    char *line;
    while ((line = lr_next(lr))) {
        if (j->end >= 0 && lr->last >= j->end) break;
        if (line[0] == '#') {
            FAIL(EMSAR_HOST_ERR_FORMAT, "more than one rsh header line");
        } else if (line[0] == '@') {                                       /* parse_rsh_indexline, 1381-1403 */
            char *tab = strchr(line, '\t');
            int tid = atoi(line + 1);
            if (!tab || tid < 0 || tid >= j->n_tx) FAIL(EMSAR_HOST_ERR_FORMAT, "bad rsh index line: %.60s", line);
            char *e = strchr(tab + 1, '\t'); if (e) *e = 0;
            if (j->n_names == j->cap_names) {
                j->cap_names = j->cap_names ? j->cap_names * 2 : 4096;
                nrow *nn = (nrow *)realloc(j->names, j->cap_names * sizeof(nrow));
                if (!nn) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
                j->names = nn;
            }
            char *nm = strdup(tab + 1);
            if (!nm) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            j->names[j->n_names].tid = tid; j->names[j->n_names++].name = nm;
        } else if (line[0] == 'c' || line[0] == 0) {
            continue;                                                      /* column headings (1370) */
        } else {                                                           /* parse_rsh_mainline, 1432-1510 */
            char *f[5] = {line, NULL, NULL, NULL, NULL};
            int nf = 1;
            for (char *p = line; *p && nf < 5; p++) if (*p == '\t') { *p = 0; f[nf++] = p + 1; }
            if (nf < 3) FAIL(EMSAR_HOST_ERR_FORMAT, "short rsh segment line");
            int size = atoi(f[1]), tid0 = atoi(f[2]);
            const char *others = nf > 3 ? f[3] : "", *eumas = nf > 4 ? f[4] : "";
            if (f[4]) { char *e = strchr(f[4], '\t'); if (e) *e = 0; }
            if (size < 1 || tid0 < 0 || tid0 >= j->n_tx) FAIL(EMSAR_HOST_ERR_FORMAT, "bad segment line (size %d tid %d)", size, tid0);
            if (eumas[0] == 0) continue;                                   /* no EUMA -> no node (1486) */
            int32_t *eu = part_ints(j, (size_t)j->nfl, 1);
            if (!eu) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            parse_int_list(eumas, eu, j->nfl);
            if (size == 1) {                                               /* rshbucket_single[tid0]=q: last wins (1488) */
                if (j->n_single == j->cap_single) {
                    j->cap_single = j->cap_single ? j->cap_single * 2 : 4096;
                    srow *ns = (srow *)realloc(j->single, j->cap_single * sizeof(srow));
                    if (!ns) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
                    j->single = ns;
                }
                j->single[j->n_single].tid = tid0; j->single[j->n_single++].eu = eu;
                continue;
            }
            if (j->n_multi == j->cap_multi) {
                j->cap_multi = j->cap_multi ? j->cap_multi * 2 : 1024;
                mrow *nm = (mrow *)realloc(j->multi, j->cap_multi * sizeof(mrow));
                if (!nm) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
                j->multi = nm;
            }
            mrow *m = &j->multi[j->n_multi];
            m->size = size; m->tid0 = tid0; m->seq = (int64_t)j->n_multi; m->euma = eu;
            m->tids = part_ints(j, (size_t)size, 0);
            if (!m->tids) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            j->n_multi++;
            m->tids[0] = tid0;
            int got = parse_int_list(others, m->tids + 1, size - 1);
            if (got != size - 1) FAIL(EMSAR_HOST_ERR_FORMAT, "segment line lists %d other tids, %d expected", got, size - 1);
            for (int i = 1; i < size; i++) if (m->tids[i] < 0 || m->tids[i] >= j->n_tx) FAIL(EMSAR_HOST_ERR_FORMAT, "tid out of range in segment line");
            if (size > j->max_t_size) j->max_t_size = size;
        }
    }
done:
    if (lr && !j->lr) emsar_lr_close(lr);
    j->rc = rc;
    return NULL;
}

static void rsh_part_free(rsh_part *j) {
    for (int_chunk *c = j->ints; c;) { int_chunk *n = c->next; free(c); c = n; }
    for (size_t i = 0; i < j->n_names; i++) free(j->names[i].name);
    free(j->multi); free(j->single); free(j->names);
}

/* rows [lo, hi) of the finished table copied from the parts' chunks, several row ranges side by side */
typedef struct { emsar_rsh *r; const mrow *multi; int32_t *const *single; int64_t lo, hi; } pack_job;
static void *pack_main(void *a) {
    pack_job *p = (pack_job *)a;
    emsar_rsh *r = p->r;
    const size_t nfl = (size_t)r->nfl;
    for (int64_t c = p->lo; c < p->hi; c++) {
        int32_t *dst = r->euma + (size_t)c * nfl;
        if (c < r->n_tx) {
            r->col_idx[r->row_ptr[c]] = (int32_t)c;
            if (p->single[c]) { memcpy(dst, p->single[c], sizeof(int32_t) * nfl); r->has_node[c] = 1; }
            else { memset(dst, 0, sizeof(int32_t) * nfl); r->has_node[c] = 0; }
        } else {
            const mrow *m = &p->multi[c - r->n_tx];
            memcpy(r->col_idx + r->row_ptr[c], m->tids, sizeof(int32_t) * (size_t)m->size);
            memcpy(dst, m->euma, sizeof(int32_t) * nfl);
            r->has_node[c] = 1;
        }
    }
    return NULL;
}

/* The body of a plain (uncompressed, seekable) rsh is parsed in byte ranges by up to 16 threads (EMSAR_HOST_THREADS), one
 * range per 16 MiB at least (EMSAR_HOST_RANGE_BYTES); .gz and stdin on the calling thread. */
int emsar_rsh_read(const char *path, emsar_rsh **out, char *err, size_t errlen) {
    int rc = EMSAR_HOST_OK;
    linereader lr = {0};
    emsar_rsh *r = (emsar_rsh *)calloc(1, sizeof(*r));
    rsh_part *part = NULL; int n_part = 0;
    pthread_t *th = NULL;
    mrow *multi = NULL; size_t n_multi = 0;
    int32_t **single = NULL;   /* EUMA vector of the single-tid node of each tid, NULL if none */
    int have_hdr = 0;
    const int dbg = getenv("EMSAR_HOST_DEBUG") != NULL;
    struct timespec ts0, ts1, ts2, ts3;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    *out = NULL;
    if (!r) return EMSAR_HOST_ERR_OOM;
    if (lr_open(&lr, path)) FAIL(EMSAR_HOST_ERR_IO, "can't open input rsh file %s", path);
    char *line;
    while (!have_hdr && (line = lr_next(&lr))) {
        if (line[0] == '#') {                                              /* parse_rsh_headerline, 1406-1430 */
            int a[5] = {0, 0, 0, 0, -1};
            if (sscanf(line + 1, "%d,%d,%d,%d,%d", &a[0], &a[1], &a[2], &a[3], &a[4]) < 4) FAIL(EMSAR_HOST_ERR_FORMAT, "bad rsh header line");
            if (a[0] < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "bad max_tid in rsh header");
            r->n_tx = a[0] + 1; r->max_t_size = a[1]; r->hdr_minfrag = a[2]; r->hdr_maxfrag = a[3]; r->hdr_readlength = a[4];
            /* determine_fraglength_range, 2471-2475 */
            r->frag_min = r->hdr_minfrag > r->hdr_readlength ? r->hdr_minfrag : r->hdr_readlength;
            r->frag_max = r->hdr_maxfrag >= r->frag_min ? r->hdr_maxfrag : r->frag_min;
            r->nfl = r->frag_max - r->frag_min + 1;
            if (r->nfl < 1 || r->hdr_minfrag < 0) FAIL(EMSAR_HOST_ERR_FORMAT, "bad fragment range in rsh header");
            r->names = (char **)calloc((size_t)r->n_tx, sizeof(char *));
            single = (int32_t **)calloc((size_t)r->n_tx, sizeof(int32_t *));
            if (!r->names || !single) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            have_hdr = 1;
        } else if (line[0] == '@') FAIL(EMSAR_HOST_ERR_FORMAT, "rsh index line before the header");
        else if (line[0] == 'c' || line[0] == 0) continue;
        else FAIL(EMSAR_HOST_ERR_FORMAT, "rsh segment line before the header");
    }
    if (!have_hdr) FAIL(EMSAR_HOST_ERR_FORMAT, "rsh file has no header line");

    /* ---- the body: one part on this reader, or byte ranges side by side ---- */
    {
        const int64_t body = lr.base + (int64_t)lr.pos;                    /* offset of the line after the header */
        int nt = 1;
        struct stat st;
        if (path && path[0] && strcmp(path, "-") != 0 && gzdirect(lr.f) && stat(path, &st) == 0 && S_ISREG(st.st_mode)) {
            nt = emsar_host_threads();
            const char *e;
            int64_t min_bytes = (int64_t)16 << 20;
            if ((e = getenv("EMSAR_HOST_RANGE_BYTES")) && atoll(e) > 0) min_bytes = atoll(e);
            const int64_t by_size = ((int64_t)st.st_size - body) / min_bytes;
            if (by_size < nt) nt = by_size < 1 ? 1 : (int)by_size;
        }
        if (dbg) fprintf(stderr, "emsar_rsh_read: %d part(s)\n", nt);
        part = (rsh_part *)calloc((size_t)nt, sizeof(*part));
        th = (pthread_t *)calloc((size_t)nt, sizeof(*th));
        if (!part || !th) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
        n_part = nt;
        for (int t = 0; t < nt; t++) { part[t].path = path; part[t].n_tx = r->n_tx; part[t].nfl = r->nfl; part[t].max_t_size = r->max_t_size; }
        if (nt == 1) { part[0].lr = &lr; part[0].begin = body; part[0].end = -1; rsh_part_main(&part[0]); }
        else {
            const int64_t span = ((int64_t)st.st_size - body) / nt;
            for (int t = 0; t < nt; t++) { part[t].begin = body + span * t; part[t].end = t + 1 < nt ? body + span * (t + 1) : -1; }
            int started = 0;
            for (int t = 1; t < nt; t++) { if (pthread_create(&th[t], NULL, rsh_part_main, &part[t]) != 0) break; started = t; }
            for (int t = started + 1; t < nt; t++) rsh_part_main(&part[t]);  /* could not spawn: run it here */
            rsh_part_main(&part[0]);
            for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
        }
        for (int t = 0; t < nt; t++)                                       /* the first failing part (in file order) names the error */
            if (part[t].rc != EMSAR_HOST_OK) { if (err) snprintf(err, errlen, "%s", part[t].err); rc = part[t].rc; goto done; }
    }
    /* ---- apply the parts in file order ---- */
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    for (int t = 0; t < n_part; t++) {
        rsh_part *j = &part[t];
        for (size_t i = 0; i < j->n_names; i++) { free(r->names[j->names[i].tid]); r->names[j->names[i].tid] = j->names[i].name; j->names[i].name = NULL; }
        for (size_t i = 0; i < j->n_single; i++) single[j->single[i].tid] = j->single[i].eu;      /* the vectors stay in the part's chunks */
        if (j->max_t_size > r->max_t_size) r->max_t_size = j->max_t_size;
        n_multi += j->n_multi;
    }
    for (int32_t t = 0; t < r->n_tx; t++) if (!r->names[t]) FAIL(EMSAR_HOST_ERR_FORMAT, "no @ line for tid %d", t);
    multi = (mrow *)malloc(sizeof(mrow) * (n_multi ? n_multi : 1));
    if (!multi) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    {
        size_t k = 0;
        for (int t = 0; t < n_part; t++) {
            for (size_t i = 0; i < part[t].n_multi; i++) { multi[k] = part[t].multi[i]; multi[k].seq = (int64_t)k; k++; }
        }
    }
    /* scan order of the multi-tid nodes: size, first tid, list (= file) order */
    qsort(multi, n_multi, sizeof(mrow), cmp_mrow);
    r->n_rows = (int64_t)r->n_tx + (int64_t)n_multi;
    uint64_t nnz = (uint64_t)r->n_tx;
    for (size_t i = 0; i < n_multi; i++) nnz += (uint64_t)multi[i].size;
    r->row_ptr = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)r->n_rows + 1));
    r->col_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)nnz);
    r->euma = (int32_t *)malloc(sizeof(int32_t) * (size_t)r->n_rows * (size_t)r->nfl + 1);
    r->has_node = (uint8_t *)malloc((size_t)r->n_rows + 1);
    if (!r->row_ptr || !r->col_idx || !r->euma || !r->has_node) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    {
        uint64_t k = 0;
        for (int32_t t = 0; t < r->n_tx; t++) r->row_ptr[t] = k++;
        for (size_t i = 0; i < n_multi; i++) { r->row_ptr[(size_t)r->n_tx + i] = k; k += (uint64_t)multi[i].size; }
        r->row_ptr[r->n_rows] = k;
        int np = n_part;                                                   /* as many packers as there were parsers */
        if (np > 64) np = 64;
        pack_job pj[64];
        pthread_t pt[64];
        int started = 0;
        for (int t = 0; t < np; t++) pj[t] = (pack_job){r, multi, single, r->n_rows * t / np, r->n_rows * (t + 1) / np};
        for (int t = 1; t < np; t++) { if (pthread_create(&pt[t], NULL, pack_main, &pj[t]) != 0) break; started = t; }
        for (int t = started + 1; t < np; t++) pack_main(&pj[t]);
        pack_main(&pj[0]);
        for (int t = 1; t <= started; t++) pthread_join(pt[t], NULL);
    }
    clock_gettime(CLOCK_MONOTONIC, &ts2);
    r->name_index = ni_build(r->names, r->n_tx);
    r->set_index = si_build(r);
    if (!r->name_index || !r->set_index) FAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    clock_gettime(CLOCK_MONOTONIC, &ts3);
#define SECS(a, b) ((double)((b).tv_sec - (a).tv_sec) + 1e-9 * (double)((b).tv_nsec - (a).tv_nsec))
    if (dbg) fprintf(stderr, "emsar_rsh_read: parse %.3f s, order + pack %.3f s, indexes %.3f s\n", SECS(ts0, ts1), SECS(ts1, ts2), SECS(ts2, ts3));
#undef SECS
done:
    lr_close(&lr);
    free(single);
    free(multi);
    for (int t = 0; t < n_part; t++) rsh_part_free(&part[t]);
    free(part); free(th);
    if (rc != EMSAR_HOST_OK) { emsar_rsh_free(r); return rc; }
    *out = r;
    return EMSAR_HOST_OK;
}

/* ---- binary cache of a parsed rsh ---------------------------------------------------------------
 * A human paired-end rsh is 10^6+ text lines of up to 400 integers; the text is parsed once and the flat arrays are
 * written next to it.  File = header, names (NUL-terminated, back to back), row_ptr, col_idx, euma, has_node; all
 * native-endian, fixed-width.  The header carries the size and mtime of the text it was made from: a stale or
 * foreign file is refused and the caller falls back to the text. */
typedef struct {
    char magic[8];                    /* "EMSARSH1" */
    uint32_t version, endian;         /* 1, 0x01020304 */
    int32_t n_tx, hdr_minfrag, hdr_maxfrag, hdr_readlength, max_t_size, frag_min, frag_max, nfl;
    int64_t n_rows, nnz, names_bytes, src_size, src_mtime;
} rsh_cache_header;

static int src_stamp(const char *path, int64_t *size, int64_t *mtime) {
    struct stat st;
    if (stat(path, &st) != 0) return -1;
    *size = (int64_t)st.st_size; *mtime = (int64_t)st.st_mtime;
    return 0;
}

int emsar_rsh_write_cache(const emsar_rsh *r, const char *src_path, const char *cache_path) {
    rsh_cache_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "EMSARSH1", 8);
    h.version = 1; h.endian = 0x01020304u;
    h.n_tx = r->n_tx; h.hdr_minfrag = r->hdr_minfrag; h.hdr_maxfrag = r->hdr_maxfrag; h.hdr_readlength = r->hdr_readlength;
    h.max_t_size = r->max_t_size; h.frag_min = r->frag_min; h.frag_max = r->frag_max; h.nfl = r->nfl;
    h.n_rows = r->n_rows; h.nnz = (int64_t)r->row_ptr[r->n_rows];
    for (int32_t t = 0; t < r->n_tx; t++) h.names_bytes += (int64_t)strlen(r->names[t]) + 1;
    if (src_stamp(src_path, &h.src_size, &h.src_mtime)) return EMSAR_HOST_ERR_IO;
    size_t plen = strlen(cache_path);
    char *tmp = (char *)malloc(plen + 16);
    if (!tmp) return EMSAR_HOST_ERR_OOM;
    snprintf(tmp, plen + 16, "%s.tmp%d", cache_path, (int)getpid());
    FILE *f = fopen(tmp, "wb");
    int ok = f != NULL;
    if (ok) ok = fwrite(&h, sizeof h, 1, f) == 1;
    for (int32_t t = 0; ok && t < r->n_tx; t++) { size_t n = strlen(r->names[t]) + 1; ok = fwrite(r->names[t], 1, n, f) == n; }
    if (ok) ok = fwrite(r->row_ptr, sizeof(uint64_t), (size_t)r->n_rows + 1, f) == (size_t)r->n_rows + 1;
    if (ok && h.nnz) ok = fwrite(r->col_idx, sizeof(int32_t), (size_t)h.nnz, f) == (size_t)h.nnz;
    size_t ne = (size_t)r->n_rows * (size_t)r->nfl;
    if (ok && ne) ok = fwrite(r->euma, sizeof(int32_t), ne, f) == ne;
    if (ok && r->n_rows) ok = fwrite(r->has_node, 1, (size_t)r->n_rows, f) == (size_t)r->n_rows;
    if (f && fclose(f) != 0) ok = 0;
    if (ok && rename(tmp, cache_path) != 0) ok = 0;      /* readers never see a half-written file */
    if (!ok) remove(tmp);
    free(tmp);
    return ok ? EMSAR_HOST_OK : EMSAR_HOST_ERR_IO;
}

int emsar_rsh_read_cache(const char *src_path, const char *cache_path, emsar_rsh **out, char *err, size_t errlen) {
    *out = NULL;
    int rc = EMSAR_HOST_OK;
    char *blob = NULL;
    emsar_rsh *r = NULL;
    FILE *f = fopen(cache_path, "rb");
    if (!f) { if (err) snprintf(err, errlen, "can't open %s", cache_path); return EMSAR_HOST_ERR_IO; }
#define CFAIL(code, msg) do { rc = (code); if (err) snprintf(err, errlen, "%s: %s", cache_path, (msg)); goto done; } while (0)
    rsh_cache_header h;
    if (fread(&h, sizeof h, 1, f) != 1) CFAIL(EMSAR_HOST_ERR_FORMAT, "short header");
    if (memcmp(h.magic, "EMSARSH1", 8) != 0 || h.version != 1 || h.endian != 0x01020304u) CFAIL(EMSAR_HOST_ERR_FORMAT, "not an emsar rsh cache of this version");
    int64_t ssz = 0, smt = 0;
    if (src_path && (src_stamp(src_path, &ssz, &smt) || ssz != h.src_size || smt != h.src_mtime)) CFAIL(EMSAR_HOST_ERR_FORMAT, "stale: the rsh text has changed");
    if (h.n_tx <= 0 || h.n_rows < h.n_tx || h.nnz < 0 || h.nfl <= 0 || h.nfl != h.frag_max - h.frag_min + 1 || h.names_bytes < h.n_tx)
        CFAIL(EMSAR_HOST_ERR_FORMAT, "inconsistent header");
    r = (emsar_rsh *)calloc(1, sizeof(*r));
    if (!r) CFAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    r->n_tx = h.n_tx; r->hdr_minfrag = h.hdr_minfrag; r->hdr_maxfrag = h.hdr_maxfrag; r->hdr_readlength = h.hdr_readlength;
    r->max_t_size = h.max_t_size; r->frag_min = h.frag_min; r->frag_max = h.frag_max; r->nfl = h.nfl; r->n_rows = h.n_rows;
    blob = (char *)malloc((size_t)h.names_bytes);
    r->names = (char **)calloc((size_t)h.n_tx, sizeof(char *));
    r->row_ptr = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)h.n_rows + 1));
    r->col_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(h.nnz ? h.nnz : 1));
    size_t ne = (size_t)h.n_rows * (size_t)h.nfl;
    r->euma = (int32_t *)malloc(sizeof(int32_t) * (ne ? ne : 1));
    r->has_node = (uint8_t *)malloc((size_t)h.n_rows);
    if (!blob || !r->names || !r->row_ptr || !r->col_idx || !r->euma || !r->has_node) CFAIL(EMSAR_HOST_ERR_OOM, "out of memory");
    if (fread(blob, 1, (size_t)h.names_bytes, f) != (size_t)h.names_bytes || blob[h.names_bytes - 1] != 0) CFAIL(EMSAR_HOST_ERR_FORMAT, "truncated names");
    {
        const char *q = blob, *end = blob + h.names_bytes;
        for (int32_t t = 0; t < h.n_tx; t++) {
            if (q >= end) CFAIL(EMSAR_HOST_ERR_FORMAT, "fewer names than transcripts");
            size_t n = strlen(q);
            r->names[t] = (char *)malloc(n + 1);
            if (!r->names[t]) CFAIL(EMSAR_HOST_ERR_OOM, "out of memory");
            memcpy(r->names[t], q, n + 1);
            q += n + 1;
        }
    }
    if (fread(r->row_ptr, sizeof(uint64_t), (size_t)h.n_rows + 1, f) != (size_t)h.n_rows + 1) CFAIL(EMSAR_HOST_ERR_FORMAT, "truncated row_ptr");
    if (h.nnz && fread(r->col_idx, sizeof(int32_t), (size_t)h.nnz, f) != (size_t)h.nnz) CFAIL(EMSAR_HOST_ERR_FORMAT, "truncated col_idx");
    if (ne && fread(r->euma, sizeof(int32_t), ne, f) != ne) CFAIL(EMSAR_HOST_ERR_FORMAT, "truncated euma");
    if (fread(r->has_node, 1, (size_t)h.n_rows, f) != (size_t)h.n_rows) CFAIL(EMSAR_HOST_ERR_FORMAT, "truncated has_node");
    /* the arrays index each other: check before anything walks them */
    if (r->row_ptr[0] != 0 || r->row_ptr[h.n_rows] != (uint64_t)h.nnz) CFAIL(EMSAR_HOST_ERR_FORMAT, "row_ptr does not span col_idx");
    for (int64_t c = 0; c < h.n_rows; c++) if (r->row_ptr[c + 1] < r->row_ptr[c]) CFAIL(EMSAR_HOST_ERR_FORMAT, "row_ptr not monotone");
    for (int64_t k = 0; k < h.nnz; k++) if (r->col_idx[k] < 0 || r->col_idx[k] >= h.n_tx) CFAIL(EMSAR_HOST_ERR_FORMAT, "tid out of range");
    r->name_index = ni_build(r->names, r->n_tx);
    r->set_index = si_build(r);
    if (!r->name_index || !r->set_index) CFAIL(EMSAR_HOST_ERR_OOM, "out of memory");
#undef CFAIL
done:
    fclose(f);
    free(blob);
    if (rc != EMSAR_HOST_OK) { emsar_rsh_free(r); return rc; }
    *out = r;
    return EMSAR_HOST_OK;
}

void emsar_rsh_free(emsar_rsh *r) {
    if (!r) return;
    if (r->names) { for (int32_t t = 0; t < r->n_tx; t++) free(r->names[t]); free(r->names); }
    free(r->row_ptr); free(r->col_idx); free(r->euma); free(r->has_node);
    if (r->name_index) { free(((name_index *)r->name_index)->slot); free(r->name_index); }
    if (r->set_index) { free(((set_index *)r->set_index)->slot); free(r->set_index); }
    free(r);
}
