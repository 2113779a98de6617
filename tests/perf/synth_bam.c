/* synth_bam.c -- test infrastructure: writes a synthetic single-end BAM for the -M (BASELINE config 4) measurements.
 *
 *   synth_bam <segments.bin> <reads.bin> <out.bam> <read_length> <threads>
 *
 * segments.bin   int64 n_tx, int64 n_seg, uint64 row_ptr[n_seg + 1], int32 col_idx[nnz]      (the index's segments: tid lists)
 * reads.bin      int64 n_reads, int32 seg_of_read[n_reads]                                   (which segment every read was drawn from)
 *
 * Read i becomes one record per transcript of its segment (refID = tid, FLAG 0, MD:Z:<L>: no mismatches), all named
 * "r<i>" and adjacent -- what an aligner run with -k 100 emits.  Reference names are ENST%07d (1000 + tid), the names the index
 * generator uses.  BGZF: every thread formats and deflates (zlib level 1) a contiguous range of reads into a part file of complete
 * BGZF blocks; the parts are concatenated behind the header block and closed with the EOF block.  Plain C + zlib + pthreads. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static int64_t n_tx, n_seg, n_reads;
static uint64_t *row_ptr;
static int32_t *col_idx, *seg_of;
static int L;
static const char *out_path;

typedef struct { unsigned char *buf; size_t n; FILE *f; } blk;

static void put32(unsigned char *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
static void put16(unsigned char *p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

/* one BGZF block from up to 65280 bytes of payload */
static int bgzf_block(FILE *f, const unsigned char *src, size_t n) {
    static const unsigned char head[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    unsigned char out[70000];
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
    z.next_in = (unsigned char *)src; z.avail_in = (uInt)n;
    z.next_out = out + 18; z.avail_out = sizeof out - 26;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) { deflateEnd(&z); return -1; }
    const size_t clen = z.total_out;
    deflateEnd(&z);
    memcpy(out, head, 12);
    out[12] = 'B'; out[13] = 'C'; put16(out + 14, 2); put16(out + 16, (unsigned)(clen + 25));
    put32(out + 18 + clen, (uint32_t)crc32(crc32(0L, NULL, 0), src, (uInt)n));
    put32(out + 22 + clen, (uint32_t)n);
    return fwrite(out, 1, clen + 26, f) == clen + 26 ? 0 : -1;
}
static int blk_put(blk *b, const void *p, size_t n) {
    const unsigned char *s = (const unsigned char *)p;
    while (n) {
        size_t k = 65280 - b->n;
        if (k > n) k = n;
        memcpy(b->buf + b->n, s, k);
        b->n += k; s += k; n -= k;
        if (b->n == 65280) { if (bgzf_block(b->f, b->buf, b->n)) return -1; b->n = 0; }
    }
    return 0;
}
static int blk_flush(blk *b) { int rc = b->n ? bgzf_block(b->f, b->buf, b->n) : 0; b->n = 0; return rc; }

typedef struct { int id, nt; int rc; } job;
static void *worker(void *arg) {
    job *j = (job *)arg;
    char path[4096];
    snprintf(path, sizeof path, "%s.part%d", out_path, j->id);
    blk b = {malloc(65536), 0, fopen(path, "wb")};
    if (!b.buf || !b.f) { j->rc = -1; return NULL; }
    const int64_t lo = n_reads * j->id / j->nt, hi = n_reads * (j->id + 1) / j->nt;
    unsigned char rec[1024];
    const int seq_bytes = (L + 1) / 2;
    for (int64_t i = lo; i < hi && !j->rc; i++) {
        const int32_t s = seg_of[i];
        char name[32];
        const int l_name = snprintf(name, sizeof name, "r%lld", (long long)i) + 1;
        const int32_t pos = (int32_t)((uint64_t)i * 2654435761u % 1000u);
        char md[16];
        const int l_md = snprintf(md, sizeof md, "MDZ%d", L) + 1;                  /* MD:Z:<L> -- no mismatches (the reference dereferences the tag) */
        const int body = 32 + l_name + 4 + seq_bytes + L + l_md;
        for (uint64_t k = row_ptr[s]; k < row_ptr[s + 1] && !j->rc; k++) {
            unsigned char *p = rec;
            put32(p, (uint32_t)body); p += 4;
            put32(p, (uint32_t)col_idx[k]); put32(p + 4, (uint32_t)pos);
            p[8] = (unsigned char)l_name; p[9] = 255; put16(p + 10, 4680);        /* l_read_name, MAPQ, bin */
            put16(p + 12, 1); put16(p + 14, 0);                                    /* one CIGAR op, FLAG 0 (forward strand) */
            put32(p + 16, (uint32_t)L); put32(p + 20, 0xFFFFFFFFu); put32(p + 24, 0xFFFFFFFFu); put32(p + 28, 0);
            p += 32;
            memcpy(p, name, (size_t)l_name); p += l_name;
            put32(p, (uint32_t)L << 4); p += 4;                                    /* <L>M */
            memset(p, 0x11, (size_t)seq_bytes); p += seq_bytes;                    /* AAAA... */
            memset(p, 0xFF, (size_t)L); p += L;                                    /* no qualities */
            memcpy(p, md, (size_t)l_md); p += l_md;
            if (blk_put(&b, rec, (size_t)(p - rec))) j->rc = -1;
        }
    }
    if (blk_flush(&b)) j->rc = -1;
    if (fclose(b.f)) j->rc = -1;
    free(b.buf);
    return NULL;
}

static void *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc((size_t)sz + 1);
    if (p && fread(p, 1, (size_t)sz, f) != (size_t)sz) { free(p); p = NULL; }
    fclose(f);
    if (n) *n = (size_t)sz;
    return p;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: synth_bam segments.bin reads.bin out.bam read_length threads\n"); return 2; }
    size_t ns = 0, nr = 0;
    unsigned char *S = slurp(argv[1], &ns), *R = slurp(argv[2], &nr);
    if (!S || !R || ns < 16 || nr < 8) { fprintf(stderr, "cannot read the inputs\n"); return 1; }
    memcpy(&n_tx, S, 8); memcpy(&n_seg, S + 8, 8);
    row_ptr = (uint64_t *)(S + 16);
    col_idx = (int32_t *)(S + 16 + 8 * (size_t)(n_seg + 1));
    memcpy(&n_reads, R, 8);
    seg_of = (int32_t *)(R + 8);
    if (ns < 16 + 8 * (size_t)(n_seg + 1) + 4 * (size_t)row_ptr[n_seg] || nr < 8 + 4 * (size_t)n_reads) { fprintf(stderr, "truncated input\n"); return 1; }
    for (int64_t i = 0; i < n_reads; i++) if (seg_of[i] < 0 || seg_of[i] >= n_seg) { fprintf(stderr, "segment id out of range\n"); return 1; }
    out_path = argv[3];
    L = atoi(argv[4]);
    int nt = atoi(argv[5]);
    if (L < 1 || L > 400 || nt < 1 || nt > 64) return 2;
    /* header: magic, text, references */
    FILE *out = fopen(out_path, "wb");
    if (!out) return 1;
    {
        blk b = {malloc(65536), 0, out};
        /* header text with one @SQ line per reference (the vendored samtools of the reference looks the names up there) */
        const char *hd = "@HD\tVN:1.0\tSO:unsorted\n";
        size_t l_text = strlen(hd);
        char line[96];
        for (int64_t t = 0; t < n_tx; t++) l_text += (size_t)snprintf(line, sizeof line, "@SQ\tSN:ENST%07lld\tLN:100000\n", (long long)(1000 + t));
        unsigned char w[8];
        blk_put(&b, "BAM\1", 4);
        put32(w, (uint32_t)l_text); blk_put(&b, w, 4); blk_put(&b, hd, strlen(hd));
        for (int64_t t = 0; t < n_tx; t++) {
            const int l = snprintf(line, sizeof line, "@SQ\tSN:ENST%07lld\tLN:100000\n", (long long)(1000 + t));
            blk_put(&b, line, (size_t)l);
        }
        put32(w, (uint32_t)n_tx); blk_put(&b, w, 4);
        for (int64_t t = 0; t < n_tx; t++) {
            char name[32];
            const int l = snprintf(name, sizeof name, "ENST%07lld", (long long)(1000 + t)) + 1;
            put32(w, (uint32_t)l); blk_put(&b, w, 4); blk_put(&b, name, (size_t)l);
            put32(w, 100000u); blk_put(&b, w, 4);
        }
        if (blk_flush(&b)) return 1;
        free(b.buf);
    }
    pthread_t th[64];
    job jobs[64];
    for (int t = 0; t < nt; t++) { jobs[t].id = t; jobs[t].nt = nt; jobs[t].rc = 0; pthread_create(&th[t], NULL, worker, &jobs[t]); }
    int rc = 0;
    for (int t = 0; t < nt; t++) { pthread_join(th[t], NULL); rc |= jobs[t].rc; }
    unsigned char *cp = malloc(1 << 22);
    for (int t = 0; t < nt && !rc; t++) {
        char path[4096];
        snprintf(path, sizeof path, "%s.part%d", out_path, t);
        FILE *f = fopen(path, "rb");
        if (!f) { rc = -1; break; }
        size_t k;
        while ((k = fread(cp, 1, 1 << 22, f)) > 0) if (fwrite(cp, 1, k, out) != k) { rc = -1; break; }
        fclose(f);
        remove(path);
    }
    static const unsigned char eof_block[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!rc && fwrite(eof_block, 1, 28, out) != 28) rc = -1;
    if (fclose(out)) rc = -1;
    return rc ? 1 : 0;
}
