/* host_count.c -- count the alignments of one file with the host library and print the totals.  Built by hand with
 * -fsanitize=thread / address over the host sources to check the parser's thread pools on the CPU:
 *   gcc -O1 -g -fsanitize=thread -Iemsar_amd/csrc/host tools/host_count.c emsar_amd/csrc/host/{rsh,align,pbgzf,model,output}.c -lz -lm -lpthread -o /tmp/host_count
 *   EMSAR_HOST_THREADS=8 EMSAR_HOST_RANGE_BYTES=2000 /tmp/host_count index.rsh reads.bam 2
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "emsar_host.h"

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: host_count index.rsh alignments format(0 bowtie,1 sam,2 bam) [pe]\n"); return 2; }
    char err[512];
    emsar_rsh *r = NULL;
    if (emsar_rsh_read(argv[1], &r, err, sizeof err) != 0) { fprintf(stderr, "rsh: %s\n", err); return 1; }
    emsar_aln_opts o = {argc > 4 ? atoi(argv[4]) : 0, 0, 100, atoi(argv[3])};
    emsar_counts *c = NULL;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    int rc = emsar_count_alignments(r, argv[2], &o, &c, err, sizeof err);
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (rc != 0) { fprintf(stderr, "count: rc %d: %s\n", rc, err); emsar_rsh_free(r); return 1; }
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < c->n_rows; i++) { h ^= (uint64_t)(uint32_t)c->R[i]; h *= 1099511628211ull; }
    printf("reads %lld seen %lld over_k %lld no_segment %lld  R hash %016llx  %.3f s\n", (long long)c->total_reads, (long long)c->reads_seen,
           (long long)c->reads_over_k, (long long)c->reads_no_segment, (unsigned long long)h,
           (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec));
    emsar_counts_free(c);
    emsar_rsh_free(r);
    return 0;
}
