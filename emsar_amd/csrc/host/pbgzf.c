/* pbgzf.c -- parallel BGZF inflate for BAM input.
 *
 * BGZF (SAM/BAM specification section 4.1) is a series of gzip members of at most 64 KiB, each carrying its own
 * compressed size in a 'BC' extra subfield, so the blocks of a file inflate independently.  The reference reads BAM
 * through its vendored samtools 0.1.19 (/root/reference/src/bgzf.c: one block at a time on the calling thread);
 * here a batch of blocks is read from the file, a pool of threads inflates them side by side (raw deflate, CRC32 and
 * ISIZE of every block verified), and the caller consumes the bytes in file order.  The batch after the one being
 * consumed is read and inflated while the caller parses, so inflate overlaps with record parsing.
 *
 * Not BGZF (plain gzip, stdin): pbgzf_open returns NULL and the caller keeps zlib's gzread.
 *
 * Inflate engine: zlib by default in the build (the only one with headers in this image); when the system has libdeflate's shared
 * library (libdeflate.so.0, the decoder htslib itself prefers) it is loaded at run time through its four public entry points
 * (libdeflate.h: alloc / free decompressor, deflate_decompress, crc32) and used instead -- same bytes, same CRC and ISIZE checks, about
 * half the core-seconds per block.  EMSAR_HOST_INFLATE=zlib keeps zlib; emsar_pbgzf_engine() names the engine in use.
 */
#include "emsar_host.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>

#define PB_BLOCKS 512                 /* blocks per batch: <= 32 MiB inflated */
#define PB_MAXBLK 65536

typedef struct {
    unsigned char *cbuf;              /* compressed bytes of the batch, back to back */
    size_t ccap;
    unsigned char *ubuf;              /* PB_BLOCKS * 64 KiB */
    uint32_t coff[PB_BLOCKS], clen[PB_BLOCKS], ulen[PB_BLOCKS], crc[PB_BLOCKS];
    int n;                            /* blocks in the batch */
    int bad;                          /* set by a worker */
} pb_batch;

struct emsar_pbgzf {
    FILE *fp;
    int n_threads;
    pb_batch b[2];
    int cur;                          /* batch being consumed */
    int blk; uint32_t off;            /* cursor inside it */
    int eof;                          /* no more bytes in the file after the loaded batches */
    int have_next;                    /* b[cur ^ 1] is loaded (or being inflated by bg) */
    pthread_t bg; int bg_running;
    pb_batch *bg_batch; int bg_rc;    /* the background job: fill + inflate bg_batch */
    int err;
};

typedef struct { pb_batch *b; int first, step; } pb_job;

/* libdeflate, if the system has it (public API of libdeflate.h, unchanged since 1.0) */
typedef void *(*ld_alloc_fn)(void);
typedef void (*ld_free_fn)(void *);
typedef int (*ld_inflate_fn)(void *d, const void *in, size_t in_n, void *out, size_t out_avail, size_t *out_n);     /* 0 = LIBDEFLATE_SUCCESS */
typedef uint32_t (*ld_crc_fn)(uint32_t crc, const void *buf, size_t n);
static struct { ld_alloc_fn alloc; ld_free_fn free_; ld_inflate_fn inflate; ld_crc_fn crc; } g_ld;
static pthread_once_t g_ld_once = PTHREAD_ONCE_INIT;
static void ld_load(void) {
    const char *e = getenv("EMSAR_HOST_INFLATE");
    if (e && strcmp(e, "zlib") == 0) return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    ld_alloc_fn a = (ld_alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
    ld_free_fn f = (ld_free_fn)dlsym(h, "libdeflate_free_decompressor");
    ld_inflate_fn i = (ld_inflate_fn)dlsym(h, "libdeflate_deflate_decompress");
    ld_crc_fn c = (ld_crc_fn)dlsym(h, "libdeflate_crc32");
    if (a && f && i && c) { g_ld.free_ = f; g_ld.inflate = i; g_ld.crc = c; g_ld.alloc = a; }     /* (the handle stays open for the life of the process) */
    else dlclose(h);
}
const char *emsar_pbgzf_engine(void) { pthread_once(&g_ld_once, ld_load); return g_ld.alloc ? "libdeflate" : "zlib"; }

static void *pb_worker(void *a) {
    pb_job *j = (pb_job *)a;
    pb_batch *b = j->b;
    pthread_once(&g_ld_once, ld_load);
    void *ld = g_ld.alloc ? g_ld.alloc() : NULL;            /* one decompressor per worker and batch: 11 KiB, no state kept between blocks */
    if (ld) {
        for (int i = j->first; i < b->n; i += j->step) {
            unsigned char *out = b->ubuf + (size_t)i * PB_MAXBLK;
            size_t got = 0;
            if (g_ld.inflate(ld, b->cbuf + b->coff[i], b->clen[i], out, PB_MAXBLK, &got) != 0 || got != b->ulen[i] ||
                g_ld.crc(0u, out, got) != b->crc[i]) { b->bad = 1; break; }
        }
        g_ld.free_(ld);
        return NULL;
    }
    for (int i = j->first; i < b->n; i += j->step) {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { b->bad = 1; break; }
        z.next_in = b->cbuf + b->coff[i]; z.avail_in = b->clen[i];
        z.next_out = b->ubuf + (size_t)i * PB_MAXBLK; z.avail_out = PB_MAXBLK;
        int rc = inflate(&z, Z_FINISH);
        uLong got = z.total_out;
        inflateEnd(&z);
        if (rc != Z_STREAM_END || got != b->ulen[i] ||
            (uint32_t)crc32(crc32(0L, Z_NULL, 0), b->ubuf + (size_t)i * PB_MAXBLK, (uInt)got) != b->crc[i]) { b->bad = 1; break; }
    }
    return NULL;
}

/* read up to PB_BLOCKS blocks from the file into batch b (no inflate).  Returns 0, or -1 on a malformed block. */
static int pb_fill(struct emsar_pbgzf *p, pb_batch *b) {
    b->n = 0; b->bad = 0;
    size_t used = 0;
    while (b->n < PB_BLOCKS) {
        unsigned char h[12];
        size_t got = fread(h, 1, 12, p->fp);
        if (got == 0) { p->eof = 1; break; }
        if (got != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return -1;
        unsigned xlen = h[10] | ((unsigned)h[11] << 8);
        unsigned char x[65536];
        if (xlen < 6 || fread(x, 1, xlen, p->fp) != xlen) return -1;
        int bsize = -1;
        for (unsigned q = 0; q + 4 <= xlen;) {
            unsigned sl = x[q + 2] | ((unsigned)x[q + 3] << 8);
            if (x[q] == 'B' && x[q + 1] == 'C' && sl == 2 && q + 6 <= xlen) bsize = x[q + 4] | ((int)x[q + 5] << 8);
            q += 4 + sl;
        }
        if (bsize < 0) return -1;
        long rest = (long)bsize + 1 - 12 - (long)xlen;          /* deflate data + CRC32 + ISIZE */
        if (rest < 8) return -1;
        if (used + (size_t)rest > b->ccap) {
            size_t nc = b->ccap ? b->ccap * 2 : (size_t)PB_BLOCKS * 24576;
            while (nc < used + (size_t)rest) nc *= 2;
            unsigned char *nb = (unsigned char *)realloc(b->cbuf, nc);
            if (!nb) return -1;
            b->cbuf = nb; b->ccap = nc;
        }
        if (fread(b->cbuf + used, 1, (size_t)rest, p->fp) != (size_t)rest) return -1;
        const unsigned char *t = b->cbuf + used + rest - 8;
        uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        uint32_t isz = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (isz > PB_MAXBLK) return -1;
        b->coff[b->n] = (uint32_t)used; b->clen[b->n] = (uint32_t)(rest - 8); b->ulen[b->n] = isz; b->crc[b->n] = crc;
        b->n++;
        used += (size_t)rest;
    }
    return 0;
}

static void pb_inflate(struct emsar_pbgzf *p, pb_batch *b) {
    int nt = p->n_threads < b->n ? p->n_threads : b->n;
    if (nt <= 1) { pb_job j = {b, 0, 1}; pb_worker(&j); return; }
    pthread_t th[64]; pb_job job[64];
    int started = 0;
    for (int t = 0; t < nt; t++) {
        job[t].b = b; job[t].first = t; job[t].step = nt;
        if (pthread_create(&th[t], NULL, pb_worker, &job[t]) != 0) break;
        started++;
    }
    for (int t = started; t < nt; t++) { job[t].b = b; job[t].first = t; job[t].step = nt; pb_worker(&job[t]); }   /* could not spawn: do it here */
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

static void *pb_bg_main(void *a) {
    struct emsar_pbgzf *p = (struct emsar_pbgzf *)a;
    p->bg_rc = pb_fill(p, p->bg_batch);
    if (p->bg_rc == 0) pb_inflate(p, p->bg_batch);
    return NULL;
}

struct emsar_pbgzf *emsar_pbgzf_open(const char *path) {
    if (!path || !path[0] || strcmp(path, "-") == 0) return NULL;
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    unsigned char h[18];
    int ok = fread(h, 1, 18, fp) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
    if (!ok || fseek(fp, 0, SEEK_SET) != 0) { fclose(fp); return NULL; }
    struct emsar_pbgzf *p = (struct emsar_pbgzf *)calloc(1, sizeof(*p));
    if (!p) { fclose(fp); return NULL; }
    p->fp = fp;
    p->n_threads = emsar_host_threads();
    for (int i = 0; i < 2; i++) {
        p->b[i].ubuf = (unsigned char *)malloc((size_t)PB_BLOCKS * PB_MAXBLK);
        if (!p->b[i].ubuf) { emsar_pbgzf_close(p); return NULL; }
    }
    if (pb_fill(p, &p->b[0]) != 0) { emsar_pbgzf_close(p); return NULL; }
    pb_inflate(p, &p->b[0]);
    if (p->b[0].bad) { emsar_pbgzf_close(p); return NULL; }
    return p;
}

/* start loading the other batch in the background, unless the file has ended.  While the job runs only it touches
 * the file, p->eof and that batch; the consumer looks at them again after the join. */
static void pb_prefetch(struct emsar_pbgzf *p) {
    if (p->eof || p->have_next) return;
    p->bg_batch = &p->b[p->cur ^ 1]; p->bg_rc = 0;
    if (p->n_threads > 1 && pthread_create(&p->bg, NULL, pb_bg_main, p) == 0) p->bg_running = 1;
    else pb_bg_main(p);
    p->have_next = 1;
}

long emsar_pbgzf_read(struct emsar_pbgzf *p, void *dst, size_t n) {
    unsigned char *out = (unsigned char *)dst;
    size_t done = 0;
    if (p->err) return -1;
    while (done < n) {
        pb_batch *b = &p->b[p->cur];
        if (p->blk >= b->n) {                       /* batch consumed: switch to the prefetched one */
            if (!p->have_next) { if (p->eof) break; pb_prefetch(p); }
            if (p->bg_running) { pthread_join(p->bg, NULL); p->bg_running = 0; }
            if (p->bg_rc != 0 || p->b[p->cur ^ 1].bad) { p->err = 1; return -1; }
            p->cur ^= 1; p->blk = 0; p->off = 0; p->have_next = 0;
            if (p->b[p->cur].n == 0) break;         /* nothing was left in the file */
            continue;
        }
        if (!p->have_next) pb_prefetch(p);          /* overlap the next batch with the caller's parsing */
        uint32_t avail = b->ulen[p->blk] - p->off;
        if (avail == 0) { p->blk++; p->off = 0; continue; }
        size_t take = n - done < avail ? n - done : avail;
        memcpy(out + done, b->ubuf + (size_t)p->blk * PB_MAXBLK + p->off, take);
        done += take; p->off += (uint32_t)take;
    }
    return (long)done;
}

void emsar_pbgzf_close(struct emsar_pbgzf *p) {
    if (!p) return;
    if (p->bg_running) pthread_join(p->bg, NULL);
    if (p->fp) fclose(p->fp);
    for (int i = 0; i < 2; i++) { free(p->b[i].cbuf); free(p->b[i].ubuf); }
    free(p);
}
