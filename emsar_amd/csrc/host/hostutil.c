/* hostutil.c -- the host thread budget shared by the readers (align.c, pbgzf.c, rsh.c).
 *
 * Every parallel host step (alignment text in byte ranges, BAM record batches, BGZF inflate, the rsh body) sizes its pool
 * with emsar_host_threads(): min(online cores, 16) unless the caller set a budget -- emsar-hip -M divides the cores by its
 * worker count so that G workers do not start G x 16 threads on one host -- or EMSAR_HOST_THREADS says otherwise (tests).
 */
#include "emsar_host.h"

#include <stdlib.h>
#include <unistd.h>

static int g_budget = 0;      /* 0 = default; written once by main() before any worker thread exists */

void emsar_host_set_thread_budget(int n) { g_budget = n > 0 ? (n > 64 ? 64 : n) : 0; }

int emsar_host_threads(void) {
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = nc > 16 ? 16 : nc < 1 ? 1 : (int)nc;
    if (g_budget > 0) nt = g_budget;
    const char *e = getenv("EMSAR_HOST_THREADS");
    if (e && atoi(e) > 0) nt = atoi(e) > 64 ? 64 : atoi(e);
    return nt;
}
