/*
 * rstub.c — a small FUNCTIONAL stand-in for the slice of R's C API that r/src/chicdiff_hip_shim.c uses, plus the harness
 * entry points tests/test_r_shim_exec.py drives through ctypes.
 *
 * TEST INFRASTRUCTURE.  R is absent from the authoring image and the GPU box; this file lets the real shim be linked against
 * the real libchicdiff_hip.so and every one of its .Call routines be EXECUTED on the GPU: results against the ctypes path,
 * PROTECT balance, device-memory balance after a forced Rf_error(), finalizers in any order.  It finds what the first
 * `R CMD SHLIB` + `.Call` would find (a crash, a wrong pointer, an unbalanced PROTECT, a leak on the error path); it pins
 * NOTHING about R itself — what R's allocator, garbage collector or coercions really do is restated from "Writing R
 * Extensions" (sections 5, 6) as plainly as possible:
 *   - SEXPs are malloc'ed records that live until rstub_free_all(): nothing is ever collected, so a missing PROTECT cannot
 *     be seen here (only a wrong COUNT can);
 *   - PROTECT / UNPROTECT are a counted stack; an UNPROTECT below the depth at .Call entry or a depth that differs at exit is
 *     recorded as a fault (rstub_fault());
 *   - Rf_error() formats its message and longjmp()s to rstub_call(), which — as R does — rewinds the PROTECT stack to the
 *     depth at entry and releases the R_alloc() blocks;
 *   - finalizers registered with R_RegisterCFinalizerEx() are kept in a list that the test runs in the order it chooses
 *     (registration order, reverse, contexts first, contexts last), once each, whether or not the shim already invoked the
 *     function itself: R gives no order either.
 */
#include <limits.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

struct SEXPREC {
    int type;
    R_xlen_t len;
    void *data;      /* int[] / double[] / SEXP[] (VECSXP, STRSXP) / char[] (CHARSXP, SYMSXP) */
    SEXP names, dim; /* the two attributes the shim touches */
    void *addr;      /* EXTPTRSXP */
    SEXP tag, prot;
    struct SEXPREC *next;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_NamesSymbol, R_DimSymbol;
double R_NaReal;
int R_NaInt = INT_MIN;

static SEXP arena = NULL;    /* every record but the symbols */
static SEXP symbols = NULL;  /* interned: pointer equality is name equality (the shim compares tags with ==) */
static long n_live = 0;

static char errbuf[2048], faultbuf[2048];
static jmp_buf *cur_jmp = NULL;
static int had_fault = 0;

#define PP_MAX 100000
static SEXP pp_stack[PP_MAX];
static int pp_top = 0, pp_floor = 0, pp_high = 0;

static void fault(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (!had_fault) vsnprintf(faultbuf, sizeof faultbuf, fmt, ap);
    va_end(ap);
    had_fault = 1;
}

static void init_once(void) {
    static int done = 0;
    if (done) return;
    done = 1;
    union { double d; uint32_t w[2]; } u; /* R's NA_real_: a quiet NaN whose low word is 1954 */
    u.d = NAN;
    u.w[0] = 1954; /* little endian: w[0] is the low word */
    u.w[1] = 0x7ff00000u | (u.w[1] & 0x80000000u);
    R_NaReal = u.d;
    R_NamesSymbol = Rf_install("names");
    R_DimSymbol = Rf_install("dim");
}
__attribute__((constructor)) static void rstub_ctor(void) { init_once(); }

int R_IsNA(double x) {
    union { double d; uint32_t w[2]; } u;
    u.d = x;
    return isnan(x) && u.w[0] == 1954;
}

static SEXP new_rec(int type, R_xlen_t len, size_t bytes) {
    SEXP s = (SEXP)calloc(1, sizeof *s);
    if (!s) abort();
    s->type = type;
    s->len = len;
    s->names = s->dim = s->tag = s->prot = R_NilValue;
    if (bytes) {
        s->data = calloc(1, bytes);
        if (!s->data) abort();
    }
    s->next = arena;
    arena = s;
    n_live++;
    return s;
}

SEXP Rf_install(const char *name) {
    for (SEXP s = symbols; s; s = s->next)
        if (!strcmp((const char *)s->data, name)) return s;
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = SYMSXP;
    s->data = strdup(name);
    s->names = s->dim = s->tag = s->prot = R_NilValue;
    s->next = symbols;
    symbols = s;
    return s;
}
SEXP Rf_mkChar(const char *str) {
    SEXP s = new_rec(CHARSXP, (R_xlen_t)strlen(str), strlen(str) + 1);
    memcpy(s->data, str, strlen(str) + 1);
    return s;
}
SEXP Rf_allocVector(unsigned int type, R_xlen_t n) {
    if (n < 0) Rf_error("negative length vectors are not allowed");
    size_t es;
    switch (type) {
    case INTSXP: case LGLSXP: es = 4; break;
    case REALSXP: es = 8; break;
    case STRSXP: case VECSXP: es = sizeof(SEXP); break;
    default: Rf_error("rstub: allocVector of type %u is not part of the stand-in", type);
    }
    SEXP s = new_rec((int)type, n, (size_t)(n > 0 ? n : 1) * es);
    if (type == VECSXP || type == STRSXP)
        for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue; /* (R fills STRSXP with "", VECSXP with NULL) */
    return s;
}
int TYPEOF(SEXP x) { return x->type; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
int LENGTH(SEXP x) {
    if (x->len > INT_MAX) Rf_error("long vectors not supported yet");
    return (int)x->len;
}
int Rf_length(SEXP x) { return x == R_NilValue ? 0 : LENGTH(x); }
int *INTEGER(SEXP x) {
    if (x->type != INTSXP && x->type != LGLSXP) Rf_error("INTEGER() can only be applied to a 'integer', not a type %d", x->type);
    return (int *)x->data;
}
double *REAL(SEXP x) {
    if (x->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not a type %d", x->type);
    return (double *)x->data;
}
SEXP STRING_ELT(SEXP x, R_xlen_t i) {
    if (x->type != STRSXP || i < 0 || i >= x->len) Rf_error("STRING_ELT: bad argument");
    return ((SEXP *)x->data)[i];
}
const char *CHAR(SEXP x) {
    if (x->type != CHARSXP) Rf_error("CHAR() can only be applied to a 'CHARSXP'");
    return (const char *)x->data;
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) {
    if (x->type != STRSXP || i < 0 || i >= x->len || v->type != CHARSXP) Rf_error("SET_STRING_ELT: bad argument");
    ((SEXP *)x->data)[i] = v;
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) {
    if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("SET_VECTOR_ELT: bad argument");
    ((SEXP *)x->data)[i] = v;
    return v;
}
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) {
    if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("VECTOR_ELT: bad argument");
    return ((SEXP *)x->data)[i];
}
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) {
    if (name == R_NamesSymbol) x->names = val;
    else if (name == R_DimSymbol) x->dim = val;
    else Rf_error("rstub: attribute '%s' is not part of the stand-in", (const char *)name->data);
    return val;
}
SEXP Rf_getAttrib(SEXP x, SEXP name) {
    if (name == R_NamesSymbol) return x->names;
    if (name == R_DimSymbol) return x->dim;
    return R_NilValue;
}
Rboolean Rf_isInteger(SEXP x) { return x->type == INTSXP; }
Rboolean Rf_isReal(SEXP x) { return x->type == REALSXP; }
Rboolean Rf_isNull(SEXP x) { return x->type == NILSXP; }
Rboolean Rf_isString(SEXP x) { return x->type == STRSXP; }
int Rf_asInteger(SEXP x) { /* first element, NA for anything else; doubles truncate, out of range / NaN -> NA */
    if (x->len >= 1) {
        if (x->type == INTSXP || x->type == LGLSXP) return ((int *)x->data)[0];
        if (x->type == REALSXP) {
            const double d = ((double *)x->data)[0];
            if (isnan(d) || d >= (double)INT_MAX + 1.0 || d <= (double)INT_MIN) return NA_INTEGER;
            return (int)d;
        }
    }
    return NA_INTEGER;
}
double Rf_asReal(SEXP x) {
    if (x->len >= 1) {
        if (x->type == REALSXP) return ((double *)x->data)[0];
        if (x->type == INTSXP || x->type == LGLSXP) {
            const int v = ((int *)x->data)[0];
            return v == NA_INTEGER ? NA_REAL : (double)v;
        }
    }
    return NA_REAL;
}
SEXP Rf_ScalarReal(double v) {
    SEXP s = Rf_allocVector(REALSXP, 1);
    REAL(s)[0] = v;
    return s;
}
SEXP Rf_ScalarInteger(int v) {
    SEXP s = Rf_allocVector(INTSXP, 1);
    INTEGER(s)[0] = v;
    return s;
}

/* ---- PROTECT stack --------------------------------------------------------------------------------------------------- */
SEXP Rf_protect(SEXP x) {
    if (pp_top >= PP_MAX) Rf_error("protect(): protection stack overflow");
    pp_stack[pp_top++] = x;
    if (pp_top > pp_high) pp_high = pp_top;
    return x;
}
void Rf_unprotect(int n) {
    if (n < 0 || pp_top - n < pp_floor) {
        fault("UNPROTECT(%d) with %d entries above the .Call's entry depth", n, pp_top - pp_floor);
        pp_top = pp_floor;
        return;
    }
    pp_top -= n;
}

/* ---- external pointers and finalizers -------------------------------------------------------------------------------------- */
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot) {
    SEXP s = new_rec(EXTPTRSXP, 1, 0);
    s->addr = p;
    s->tag = tag;
    s->prot = prot;
    return s;
}
static SEXP extptr(SEXP s, const char *who) {
    if (s->type != EXTPTRSXP) Rf_error("%s: argument of type %d is not an external pointer", who, s->type);
    return s;
}
void *R_ExternalPtrAddr(SEXP s) { return extptr(s, "R_ExternalPtrAddr")->addr; }
SEXP R_ExternalPtrTag(SEXP s) { return extptr(s, "R_ExternalPtrTag")->tag; }
SEXP R_ExternalPtrProtected(SEXP s) { return extptr(s, "R_ExternalPtrProtected")->prot; }
void R_ClearExternalPtr(SEXP s) { extptr(s, "R_ClearExternalPtr")->addr = NULL; }

typedef struct { SEXP obj; R_CFinalizer_t fun; int ran; } fin_t;
static fin_t *fins = NULL;
static int n_fins = 0, cap_fins = 0;
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) {
    (void)onexit;
    extptr(s, "R_RegisterCFinalizerEx");
    if (n_fins == cap_fins) {
        cap_fins = cap_fins ? 2 * cap_fins : 256;
        fins = (fin_t *)realloc(fins, (size_t)cap_fins * sizeof *fins);
        if (!fins) abort();
    }
    fins[n_fins].obj = s;
    fins[n_fins].fun = fun;
    fins[n_fins].ran = 0;
    n_fins++;
}

/* ---- errors, transient memory ------------------------------------------------------------------------------------------------ */
typedef struct ralloc { struct ralloc *next; } ralloc_t;
static ralloc_t *rallocs = NULL;
char *R_alloc(size_t n, int size) {
    ralloc_t *b = (ralloc_t *)malloc(sizeof(ralloc_t) + 16 + n * (size_t)size);
    if (!b) Rf_error("cannot allocate memory block of size %zu", n * (size_t)size);
    b->next = rallocs;
    rallocs = b;
    return (char *)b + sizeof(ralloc_t) + (16 - sizeof(ralloc_t) % 16) % 16;
}
static void free_rallocs(void) {
    while (rallocs) {
        ralloc_t *b = rallocs;
        rallocs = b->next;
        free(b);
    }
}
void Rf_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(errbuf, sizeof errbuf, fmt, ap);
    va_end(ap);
    if (!cur_jmp) {
        fprintf(stderr, "rstub: Rf_error outside a .Call: %s\n", errbuf);
        abort();
    }
    longjmp(*cur_jmp, 1);
}

/* ---- routine registration ------------------------------------------------------------------------------------------------------ */
static const R_CallMethodDef *routines = NULL;
static int dynamic_symbols = 1;
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e) {
    (void)info; (void)c; (void)f; (void)e;
    routines = call;
    return 1;
}
int R_useDynamicSymbols(DllInfo *info, int value) {
    (void)info;
    const int old = dynamic_symbols;
    dynamic_symbols = value;
    return old;
}

/* ================= harness entry points (ctypes) ================================================================================== */
void R_init_chicdiffhip(DllInfo *dll); /* the shim's */
int rstub_load(void) { /* what library.dynam() does after dlopen(): returns the number of registered .Call routines */
    init_once();
    routines = NULL;
    R_init_chicdiffhip(NULL);
    int n = 0;
    if (routines)
        while (routines[n].name) n++;
    return dynamic_symbols ? -n : n; /* the shim must switch dynamic lookup off */
}
const char *rstub_routine_name(int i) { return routines[i].name; }
int rstub_routine_nargs(int i) { return routines[i].numArgs; }
const char *rstub_last_error(void) { return errbuf; }
const char *rstub_fault(void) { return had_fault ? faultbuf : NULL; }
void rstub_clear_fault(void) { had_fault = 0; faultbuf[0] = 0; }
int rstub_protect_depth(void) { return pp_top; }
int rstub_protect_high_water(void) { const int h = pp_high; pp_high = pp_top; return h; }
long rstub_live_records(void) { return n_live; }

typedef SEXP (*fn1)(SEXP);
typedef SEXP (*fn2)(SEXP, SEXP);
typedef SEXP (*fn3)(SEXP, SEXP, SEXP);
typedef SEXP (*fn4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn10)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn11)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn12)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* .Call(name, args...): NULL when the routine raised an error (rstub_last_error()) or is not registered with that arity */
SEXP rstub_call(const char *name, int nargs, SEXP *a) {
    const R_CallMethodDef *volatile r = routines;
    errbuf[0] = 0;
    while (r && r->name && strcmp(r->name, name)) r++;
    if (!r || !r->name) {
        snprintf(errbuf, sizeof errbuf, "rstub: C symbol name \"%s\" not in the registered routines", name);
        return NULL;
    }
    if (r->numArgs != nargs) {
        snprintf(errbuf, sizeof errbuf, "rstub: Incorrect number of arguments (%d), expecting %d for '%s'", nargs, r->numArgs, name);
        return NULL;
    }
    jmp_buf jb;
    const int depth0 = pp_top, floor0 = pp_floor;
    pp_floor = depth0;
    cur_jmp = &jb;
    SEXP volatile out = NULL;
    if (setjmp(jb)) { /* Rf_error(): R rewinds the protect stack to the context's depth and drops the R_alloc blocks */
        pp_top = depth0;
        pp_floor = floor0;
        cur_jmp = NULL;
        free_rallocs();
        return NULL;
    }
    DL_FUNC f = r->fun;
    switch (nargs) {
    case 1: out = ((fn1)f)(a[0]); break;
    case 2: out = ((fn2)f)(a[0], a[1]); break;
    case 3: out = ((fn3)f)(a[0], a[1], a[2]); break;
    case 4: out = ((fn4)f)(a[0], a[1], a[2], a[3]); break;
    case 5: out = ((fn5)f)(a[0], a[1], a[2], a[3], a[4]); break;
    case 6: out = ((fn6)f)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case 7: out = ((fn7)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 8: out = ((fn8)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
    case 9: out = ((fn9)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
    case 10: out = ((fn10)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
    case 11: out = ((fn11)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
    case 12: out = ((fn12)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]); break;
    default:
        snprintf(errbuf, sizeof errbuf, "rstub: %d arguments", nargs);
        break;
    }
    cur_jmp = NULL;
    free_rallocs();
    if (pp_top != depth0) {
        fault("%s: PROTECT depth %d at exit, %d at entry", name, pp_top, depth0);
        pp_top = depth0;
    }
    pp_floor = floor0;
    return out;
}

/* finalizers that have not run yet; order: 0 registration, 1 reverse, 2 contexts (tag chicdiff_hip_ctx) first, 3 contexts last.
 * Returns how many ran; an Rf_error() inside a finalizer is recorded as a fault (R would print a warning and go on). */
int rstub_run_finalizers(int order) {
    volatile int ran = 0;
    SEXP ctxtag = Rf_install("chicdiff_hip_ctx");
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < n_fins; k++) {
            const int i = order == 1 ? n_fins - 1 - k : k;
            fin_t *volatile f = &fins[i];
            if (f->ran) continue;
            const int is_ctx = f->obj->tag == ctxtag;
            if (order == 2 && pass == 0 && !is_ctx) continue;
            if (order == 3 && pass == 0 && is_ctx) continue;
            f->ran = 1;
            jmp_buf jb;
            cur_jmp = &jb;
            if (setjmp(jb)) {
                fault("Rf_error inside a finalizer: %s", errbuf);
            } else {
                f->fun(f->obj);
                ran++;
            }
            cur_jmp = NULL;
        }
        if (order < 2) break;
    }
    return ran;
}
int rstub_pending_finalizers(void) {
    int n = 0;
    for (int i = 0; i < n_fins; i++) n += !fins[i].ran;
    return n;
}
/* drop every record (after the finalizers have run) */
void rstub_free_all(void) {
    while (arena) {
        SEXP s = arena;
        arena = s->next;
        free(s->data);
        free(s);
    }
    n_live = 0;
    n_fins = 0;
    pp_top = pp_floor = pp_high = 0;
}

/* constructors / accessors for the Python side */
SEXP rstub_nil(void) { return R_NilValue; }
SEXP rstub_mk_int(const int *v, R_xlen_t n) {
    SEXP s = Rf_allocVector(INTSXP, n);
    if (n) memcpy(s->data, v, (size_t)n * 4);
    return s;
}
SEXP rstub_mk_real(const double *v, R_xlen_t n) {
    SEXP s = Rf_allocVector(REALSXP, n);
    if (n) memcpy(s->data, v, (size_t)n * 8);
    return s;
}
SEXP rstub_mk_string(const char *str) {
    SEXP s = Rf_allocVector(STRSXP, 1);
    SET_STRING_ELT(s, 0, Rf_mkChar(str));
    return s;
}
SEXP rstub_mk_list(R_xlen_t n) { return Rf_allocVector(VECSXP, n); }
void rstub_set_elt(SEXP l, R_xlen_t i, SEXP v) { ((SEXP *)l->data)[i] = v; }
void rstub_set_dim(SEXP x, const int *d, int nd) { x->dim = rstub_mk_int(d, nd); }
double rstub_na_real(void) { return R_NaReal; }
int rstub_typeof(SEXP x) { return x->type; }
R_xlen_t rstub_length(SEXP x) { return x->len; }
void *rstub_data(SEXP x) { return x->data; }
SEXP rstub_elt(SEXP l, R_xlen_t i) { return ((SEXP *)l->data)[i]; }
const char *rstub_name(SEXP l, R_xlen_t i) {
    if (l->names == R_NilValue || i >= l->names->len) return NULL;
    return (const char *)((SEXP *)l->names->data)[i]->data;
}
void *rstub_extptr_addr(SEXP x) { return x->type == EXTPTRSXP ? x->addr : NULL; }
