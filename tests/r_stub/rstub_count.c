/*
 * rstub_count.c — the shim's device allocations, counted (TEST INFRASTRUCTURE, see rstub.c / README.md).
 * The harness compiles r/src/chicdiff_hip_shim.c with -Dchicdiff_hip_malloc=rstub_counted_malloc and
 * -Dchicdiff_hip_free=rstub_counted_free; this file (compiled WITHOUT those) forwards to the real library, counts, and can make
 * the k-th allocation from now fail — which turns every allocation site of every .Call routine into a forced Rf_error().
 */
#include <stdint.h>

#include "chicdiff_hip.h"

static long n_malloc = 0, n_free = 0;
static long fail_in = -1; /* the allocation that fails: 0 = the next one; < 0 = none */

int rstub_counted_malloc(chicdiff_hip_ctx *ctx, uint64_t bytes, void **d_ptr) {
    if (fail_in == 0) {
        fail_in = -1;
        if (d_ptr) *d_ptr = 0;
        return CHICDIFF_E_NOMEM; /* (the library's last_error text is whatever it was: the shim only formats it) */
    }
    if (fail_in > 0) fail_in--;
    const int rc = chicdiff_hip_malloc(ctx, bytes, d_ptr);
    if (rc == CHICDIFF_OK) n_malloc++;
    return rc;
}
int rstub_counted_free(chicdiff_hip_ctx *ctx, void *d_ptr) {
    const int rc = chicdiff_hip_free(ctx, d_ptr);
    if (rc == CHICDIFF_OK && d_ptr) n_free++;
    return rc;
}
long rstub_mallocs(void) { return n_malloc; }
long rstub_frees(void) { return n_free; }
void rstub_fail_malloc_in(long k) { fail_in = k; }
long rstub_fail_pending(void) { return fail_in; }
