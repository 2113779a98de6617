/* fixed_time.c -- TEST TOOL.  LD_PRELOAD shim that pins time() so that the compiled reference's
 * srand(time(NULL)) (emsar_main.c:441) becomes reproducible:  EMSAR_FIXED_TIME=12345 LD_PRELOAD=... emsar -p 1 ...
 * Used only by tests/golden/make_golden.py to produce the seeded fixture that pins
 * oracle_mle_pattern_search bit-for-bit. */
#include <stdlib.h>
#include <time.h>
time_t time(time_t *t) {
    const char *s = getenv("EMSAR_FIXED_TIME");
    time_t v = s ? (time_t)atoll(s) : (time_t)1;
    if (t) *t = v;
    return v;
}
