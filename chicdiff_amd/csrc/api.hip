// api.hip — host side of include/chicdiff_hip.h: context, workspace, and the launch sequence of
// one fit.  Everything is enqueued on one HIP stream.  On a single rank the host blocks once, to read
// the fit's scalars back at the end; sharded, it also looks at the trend's `finished` flag once per
// batch of IRLS passes and at the verdict of each median's candidate gather.  With world_size > 1
// every global sum goes through a sum-all-reduce on the same stream: the library's own RCCL
// communicator (chicdiff_hip_rccl_init) or the caller's callback (chicdiff_hip_set_allreduce).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <unordered_set>

#include <dlfcn.h>

#include "common.h"
#include "prior_mc.h"

using namespace cd;

namespace cd {  // chinput.hip
struct ChinputCols {
    std::vector<int32_t> bait, oe, N;
    std::string error;
};
int64_t chinput_parse(const char *path, int nthreads, ChinputCols &c);
}  // namespace cd

struct KTimer {
    std::string name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double ms = 0;
    int launches = 0;
    double bytes = 0;  // collectives: payload this rank handed to the transport
};

struct chicdiff_hip_ctx {
    int device = 0;
    bool no_persistent_trend = false;  // set after a grid-barrier timeout (see fit_dev_impl)
    // tuning / test options (chicdiff_hip_set_option); the defaults are what the benchmarks run
    int opt_chunk = 0;  // line search: rows per dequeue (0 = automatic)
    int opt_prio = 0;   // line search: s_setprio by search age (0 = off)
    int opt_classes_a = 0;  // gene-wise line search: classes of the schedule dealt out statically (0 = default)
    int opt_spread = 1, opt_min_waves = 0, opt_select_rounds = 0, opt_trend_multilaunch = 0, opt_schedule = 1, opt_deal = 0;
    int opt_trend_gather = 1;  // sharded fits: gather the rows of the trend on every rank (two collectives) instead of one all-reduce per IRLS pass
    char *tg_buf = nullptr;    // ... the gathered rows (grow-only)
    size_t tg_bytes = 0;
    double shard_n[kSelMaxWorld] = {0};  // rows of every rank's shard, exchanged with the argument verdicts at the start of a sharded call
    bool shard_n_valid = false;
    // set by gathered_trend for the rest of the fit: the whole fit's (baseMean, dispGeneEst) rows are on this rank
    int64_t tg_total = 0;
    double *tg_x = nullptr, *tg_y = nullptr, *tg_resid = nullptr;
    int32_t *tg_flags = nullptr;
    int32_t *h_flag = nullptr;  // pinned: sel_overflow of the size-factor select (read with the call's last synchronisation)
    int opt_no_local_substitute = 0;  // 1: a failed parametric trend is reported (CHICDIFF_ST_TREND_FAILED), not replaced by the local fit
    // test hook (option "fault_inject", one-shot bits, consumed by the next call that reaches the step): 1 = this rank's fit reports a
    // select candidate-list overflow, 2 = this rank's persistent trend kernel reports a grid-barrier timeout, 4 = this rank's
    // size-factor select reports an overflow.  Each verdict is all-reduced, so every rank of a sharded fit must re-enter together.
    int opt_trend_blocks = 0;  // persistent trend kernel: cap on its workgroups (0 = one per CU)
    // set by an entry point for the fit it is about to make (d_nf = d_nf_tmp): the offsets are formed from FullMean inside the fit's
    // first kernel instead of by a launch of their own (common.h FusedOffsets); cleared when the fit returns
    FusedOffsets fuse;
    int opt_fuse_offsets = 1;  // (option "fuse_offsets": 0 = offsets always as a launch of their own, 2 = always inside prep: the bit-identity test)
    // bench hook (option "bench_fake_world", a 1-rank communicator only): the trend's rows are gathered as if N ranks had each sent
    // this rank's block — the single-launch trend + MAD kernel then runs on N x n rows, which is what EVERY rank of an N-GPU fit
    // does (bench.py's rehearsal of a rank's step at its share of the rows; the coefficients are those of the n rows up to rounding)
    int opt_fake_world = 0;
    int opt_mad_in_kernel = 1; // the persistent trend kernel also takes the median / MAD of the residuals (0: separate launches, as round 3)
    int opt_fault = 0;
    int refits = 0;               // refits the last call went through (select overflow / barrier timeout / local substitute), for the tests
    bool sf_overflow_seen = false;  // fit_dev_impl: some rank's size-factor select (run by the caller just before) overflowed
    int32_t *d_carry = nullptr;   // device word that survives the fit's clearing of its scalars: sel_overflow of the size-factor select
    // host-buffer entry point: device arena + pinned staging, both grow-only (no allocation per call once warm)
    char *io_dev = nullptr, *io_pin = nullptr;
    size_t io_dev_bytes = 0, io_pin_bytes = 0;
    int opt_host_threads = 12;              // host threads that move caller buffers to / from the pinned staging area
    ChinputCols *chin = nullptr;            // columns of the .chinput file read last (chicdiff_hip_chinput_read)
    // host-buffer entry point: columns that are final once the MAP dispersions exist leave for the host on a second
    // stream while the Wald stage runs (set for the duration of one chicdiff_hip_nbglm_fit call)
    struct EarlyCopy {
        int n = 0;
        struct Item { int col; char *pin; size_t bytes; } item[10];  // col: 0 baseMean 1 baseVar 2 dispGeneEst 3 dispFit 4 dispMAP 5 dispersion 6 dispGeneIter 7 dispIter 8 dispOutlier 9 allZero
        hipEvent_t ready = nullptr, done = nullptr;
        bool issued = false;
    } *early = nullptr;
    hipStream_t copy_stream = nullptr;      // also the second stream of the independent-filtering sorts
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<chicdiff_hip_ctx *> lanes;  // theta grid: child contexts (own stream + workspace), one per concurrent fit
    int opt_grid_lanes = 5;                 // theta grid: fits in flight at once (1 = one after the other)
    int cu_count = 0;  // compute units of the device (the persistent trend kernel needs one resident workgroup per CU it launches)
    hipStream_t own_stream = nullptr, stream = nullptr;
    chicdiff_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    chicdiff_allgather_fn allgather = nullptr;  // optional: without it the trend rows are gathered by a sum-all-reduce over zero-filled arrays
    void *allgather_user = nullptr;
    int world = 1, rank = 0;
    // direct RCCL path (chicdiff_hip_rccl_init): librccl is dlopen'ed, never linked
    void *rccl_lib = nullptr, *rccl_comm = nullptr;
    int (*rccl_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*rccl_allgather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*rccl_comm_destroy)(void *) = nullptr;
    const char *(*rccl_error_string)(int) = nullptr;
    char err[512] = {0};
    // workspace
    int64_t cap_n = 0;
    int cap_S = 0;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    char *aux = nullptr;  // scratch of the post-processing entry points (sort / scan temporaries)
    size_t aux_bytes = 0;
    FitWork w{};
    double *d_sf = nullptr;   // kMaxS doubles
    double *d_logfact = nullptr;  // kLogFactN doubles: log(k!) (wald_prep)
    PmcTable *d_pmc[4] = {nullptr, nullptr, nullptr, nullptr};  // simulated residual densities + loess operator per d.f. (prior_mc.h)
    double *d_nf_tmp = nullptr;
    FitScalars *h_sc = nullptr;  // pinned
    double *h_sf = nullptr;      // pinned, kMaxS
    // timing
    int timing = 0;  // 0 off, 1 every stage, 2 the three fit kernels only (disp_gene, disp_map, wald_irls)
    std::vector<KTimer> timers;
    std::vector<std::pair<int, hipEvent_t>> pending;
    std::vector<hipEvent_t> event_pool;  // recycled: creating events per launch cost ~0.9 ms per step
    int scope_depth = 0;                 // nested scopes are folded into the outermost one
    // chicdiff_hip_malloc: the caller's device vectors, so that destroying the context releases what its host forgot or could not
    // release any more (R runs the finalizers of one garbage collection in no particular order: a context can go before its vectors)
    std::unordered_set<void *> user_allocs;
    std::mutex user_mu;
};

static char g_create_err[512];
constexpr int64_t kFuseOffsetsMaxRows = 1 << 18;  // fits up to this many rows form their offsets inside prep (see wald_test_dev)

// simulated residual densities + loess operator of one d.f. (prior_mc.h): constants, built once per process
static const PmcTable &pmc_table(int df) {
    static PmcTable tables[4];
    static bool ready[4] = {false, false, false, false};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!ready[df]) {
        pmc_build(df, tables[df]);
        ready[df] = true;
    }
    return tables[df];
}

static int fail(chicdiff_hip_ctx *c, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_err, 512, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(c, CHICDIFF_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

const char *chicdiff_hip_last_error(const chicdiff_hip_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

void chicdiff_hip_default_opts(chicdiff_nbglm_opts *o) {
    o->minDisp = 1e-8;
    o->dispTol = 1e-6;
    o->kappa0 = 1.0;
    o->maxit = 100;
    o->betaMaxit = 100;
    o->betaTol = 1e-8;
    o->minmu = 0.5;
    o->outlierSD = 2.0;
    o->dispPriorVar = NAN;
    o->trendCoef[0] = o->trendCoef[1] = NAN;
    o->fitType = 0;
    o->_pad = 0;
}

int chicdiff_hip_set_option(chicdiff_hip_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return CHICDIFF_E_INVALID;
    const std::string k(name);
    if (k == "line_search_spread" && value >= 0 && value <= 3) c->opt_spread = (int)value;  // 2: samples across lanes without the lean tick of the launch's end (bit-identity tests)
    else if (k == "line_search_min_waves" && (value == 0 || (value >= 2 && value <= 4))) c->opt_min_waves = (int)value;
    else if (k == "line_search_prio" && value >= 0 && value <= 100) c->opt_prio = (int)value;
    else if (k == "line_search_chunk" && (value == 0 || (value >= 8 && value <= 64))) c->opt_chunk = (int)value;
    else if (k == "line_search_classes_a" && value >= 0 && value <= 6) c->opt_classes_a = (int)value;
    else if (k == "line_search_schedule" && (value == 0 || value == 1)) c->opt_schedule = (int)value;
    else if (k == "line_search_deal" && value >= 0 && value <= 64) c->opt_deal = (int)value;
    else if (k == "local_trend_substitute" && (value == 0 || value == 1)) c->opt_no_local_substitute = value ? 0 : 1;
    else if (k == "sharded_trend_gather" && (value == 0 || value == 1)) c->opt_trend_gather = (int)value;
    else if (k == "theta_grid_concurrency" && value >= 1 && value <= 16) c->opt_grid_lanes = (int)value;
    else if (k == "host_copy_threads" && value >= 1 && value <= 64) c->opt_host_threads = (int)value;
    else if (k == "select_all_rounds" && (value == 0 || value == 1)) c->opt_select_rounds = (int)value;
    else if (k == "trend_one_launch_per_pass" && (value == 0 || value == 1)) c->opt_trend_multilaunch = (int)value;
    else if (k == "fault_inject" && value >= 0 && value <= 7) c->opt_fault = (int)value;
    else if (k == "trend_persistent_blocks" && value >= 0 && value <= 256) c->opt_trend_blocks = (int)value;
    else if (k == "fuse_offsets" && value >= 0 && value <= 2) c->opt_fuse_offsets = (int)value;
    else if (k == "bench_fake_world" && value >= 0 && value <= kGatherMaxWorld) {
        // a rehearsal hook of bench.py, not an option of the product: refused unless the process says it is that rehearsal
        // (it makes a 1-rank fit's trend run on N copies of its rows — a fit nobody asked for)
        if (value > 1 && !getenv("CHICDIFF_BENCH_FAKE_WORLD"))
            return fail(c, CHICDIFF_E_INVALID, "set_option: bench_fake_world is a rehearsal hook of bench.py (CHICDIFF_BENCH_FAKE_WORLD unset)");
        c->opt_fake_world = (int)value;
    }
    else if (k == "trend_mad_in_kernel" && (value == 0 || value == 1)) c->opt_mad_in_kernel = (int)value;
    else return fail(c, CHICDIFF_E_INVALID, "set_option: unknown option or value (%s = %lld)", name, (long long)value);
    return CHICDIFF_OK;
}

int chicdiff_hip_create(chicdiff_hip_ctx **out, int32_t device) {
    if (!out) return fail(nullptr, CHICDIFF_E_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CHICDIFF_E_HIP, "no HIP device available (%s): the HIP path cannot run, and there is no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, CHICDIFF_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    chicdiff_hip_ctx *c = new chicdiff_hip_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_sc, sizeof(FitScalars))) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_sf, sizeof(double) * (kMaxS + 1))) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_sf, sizeof(double) * (kMaxS + 1 + kSelMaxWorld + 8))) != hipSuccess ||
        (e = hipMemset(c->d_sf, 0, sizeof(double) * (kMaxS + 1 + kSelMaxWorld + 8))) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_logfact, sizeof(double) * kLogFactN)) != hipSuccess) {
        fail(nullptr, CHICDIFF_E_HIP, "context setup: %s", hipGetErrorString(e));
        delete c;
        return CHICDIFF_E_HIP;
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    }
    {  // log(k!) for k < kLogFactN: lgamma(y + 1) of the NB log-likelihood's constant part is a table look-up for ordinary counts
        std::vector<double> lf(kLogFactN);
        for (int k = 0; k < kLogFactN; k++) lf[k] = lgamma((double)k + 1.0);
        if ((e = hipMemcpy(c->d_logfact, lf.data(), sizeof(double) * kLogFactN, hipMemcpyHostToDevice)) != hipSuccess) {
            fail(nullptr, CHICDIFF_E_HIP, "context setup: %s", hipGetErrorString(e));
            chicdiff_hip_destroy(c);
            return CHICDIFF_E_HIP;
        }
    }
    c->d_carry = reinterpret_cast<int32_t *>(c->d_sf + kMaxS + 1 + kSelMaxWorld);  // (zeroed above)
    c->stream = c->own_stream;
    *out = c;
    return CHICDIFF_OK;
}

void chicdiff_hip_destroy(chicdiff_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    delete c->chin;
    for (void *p : c->user_allocs) (void)hipFree(p);
    c->user_allocs.clear();
    if (c->io_dev) (void)hipFree(c->io_dev);
    if (c->io_pin) (void)hipHostFree(c->io_pin);
    for (auto *l : c->lanes) chicdiff_hip_destroy(l);
    c->lanes.clear();
    if (c->ws) (void)hipFree(c->ws);
    if (c->aux) (void)hipFree(c->aux);
    if (c->tg_buf) (void)hipFree(c->tg_buf);
    if (c->d_sf) (void)hipFree(c->d_sf);
    if (c->d_logfact) (void)hipFree(c->d_logfact);
    for (int k = 0; k < 4; k++)
        if (c->d_pmc[k]) (void)hipFree(c->d_pmc[k]);
    if (c->h_sc) (void)hipHostFree(c->h_sc);
    if (c->h_sf) (void)hipHostFree(c->h_sf);
    if (c->rccl_comm && c->rccl_comm_destroy) (void)c->rccl_comm_destroy(c->rccl_comm);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int chicdiff_hip_set_stream(chicdiff_hip_ctx *c, void *s) {
    if (!c) return CHICDIFF_E_INVALID;
    c->stream = (hipStream_t)s;  // NULL is HIP's null stream (what torch calls the default stream)
    return CHICDIFF_OK;
}

int chicdiff_hip_set_allreduce(chicdiff_hip_ctx *c, chicdiff_allreduce_fn fn, void *user, int32_t world, int32_t rank) {
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn))
        return fail(c, CHICDIFF_E_INVALID, "set_allreduce: bad arguments");
    c->allreduce = fn;
    c->allreduce_user = user;
    c->allgather = nullptr;  // belongs to the transport that was replaced; set it again with chicdiff_hip_set_allgather
    c->allgather_user = nullptr;
    c->world = world;
    c->rank = rank;
    return CHICDIFF_OK;
}

int chicdiff_hip_set_allgather(chicdiff_hip_ctx *c, chicdiff_allgather_fn fn, void *user) {
    if (!c) return CHICDIFF_E_INVALID;
    if (fn && !c->allreduce) return fail(c, CHICDIFF_E_INVALID, "set_allgather: register the all-reduce of the same transport first");
    c->allgather = fn;
    c->allgather_user = user;
    return CHICDIFF_OK;
}

int32_t chicdiff_hip_last_refits(const chicdiff_hip_ctx *c) { return c ? c->refits : 0; }

// ---- direct RCCL: the library calls ncclAllReduce itself on its own stream (no host callback per collective) ----
struct RcclUniqueId { char internal[128]; };  // NCCL_UNIQUE_ID_BYTES
static int rccl_open(chicdiff_hip_ctx *c, const char *path) {
    if (c->rccl_lib) return CHICDIFF_OK;
    void *h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(c, CHICDIFF_E_COMM, "dlopen(%s): %s", path && path[0] ? path : "librccl.so", dlerror());
    c->rccl_lib = h;
    return CHICDIFF_OK;
}
// what the library's own RCCL callbacks return after a failure: the message is in c->err already (a host callback returns any other
// non-zero value and gets the generic text — round 4 told the two apart by looking for "nccl" in c->err, which a stale message
// from an earlier call satisfied just as well: ADVICE r04)
constexpr int kCollMsgSet = 0x7e57;
static int rccl_allreduce_cb(void *user, void *dev_buf, int64_t count) {
    chicdiff_hip_ctx *c = (chicdiff_hip_ctx *)user;
    const int r = c->rccl_allreduce(dev_buf, dev_buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->rccl_comm, c->stream);
    if (r != 0) {
        fail(c, CHICDIFF_E_COMM, "ncclAllReduce: %s", c->rccl_error_string ? c->rccl_error_string(r) : "error");
        return kCollMsgSet;
    }
    return 0;
}
static int rccl_allgather_cb(void *user, const void *dev_send, void *dev_recv, int64_t count) {
    chicdiff_hip_ctx *c = (chicdiff_hip_ctx *)user;
    const int r = c->rccl_allgather(dev_send, dev_recv, (size_t)count, /*ncclFloat64*/ 8, c->rccl_comm, c->stream);
    if (r != 0) {
        fail(c, CHICDIFF_E_COMM, "ncclAllGather: %s", c->rccl_error_string ? c->rccl_error_string(r) : "error");
        return kCollMsgSet;
    }
    return 0;
}
int chicdiff_hip_rccl_unique_id(chicdiff_hip_ctx *c, const char *librccl_path, void *id128) {
    if (!c || !id128) return CHICDIFF_E_INVALID;
    int rc = rccl_open(c, librccl_path);
    if (rc) return rc;
    auto get = (int (*)(RcclUniqueId *))dlsym(c->rccl_lib, "ncclGetUniqueId");
    if (!get) return fail(c, CHICDIFF_E_COMM, "librccl lacks ncclGetUniqueId");
    const int r = get((RcclUniqueId *)id128);
    if (r != 0) return fail(c, CHICDIFF_E_COMM, "ncclGetUniqueId failed (%d)", r);
    return CHICDIFF_OK;
}
int chicdiff_hip_rccl_init(chicdiff_hip_ctx *c, const char *librccl_path, const void *id128, int32_t world, int32_t rank) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, CHICDIFF_E_INVALID, "rccl_init: bad arguments");
    int rc = rccl_open(c, librccl_path);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    auto init = (int (*)(void **, int, RcclUniqueId, int))dlsym(c->rccl_lib, "ncclCommInitRank");
    c->rccl_allreduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(c->rccl_lib, "ncclAllReduce");
    c->rccl_allgather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(c->rccl_lib, "ncclAllGather");
    c->rccl_comm_destroy = (int (*)(void *))dlsym(c->rccl_lib, "ncclCommDestroy");
    c->rccl_error_string = (const char *(*)(int))dlsym(c->rccl_lib, "ncclGetErrorString");
    if (!init || !c->rccl_allreduce || !c->rccl_comm_destroy) return fail(c, CHICDIFF_E_COMM, "librccl lacks ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    if (c->rccl_comm) {
        (void)c->rccl_comm_destroy(c->rccl_comm);
        c->rccl_comm = nullptr;
    }
    RcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const int r = init(&c->rccl_comm, world, id, rank);
    if (r != 0) {
        c->rccl_comm = nullptr;
        return fail(c, CHICDIFF_E_COMM, "ncclCommInitRank: %s", c->rccl_error_string ? c->rccl_error_string(r) : "error");
    }
    c->allreduce = rccl_allreduce_cb;
    c->allreduce_user = c;
    c->allgather = c->rccl_allgather ? rccl_allgather_cb : nullptr;  // (a librccl without it: the sum-all-reduce way of gathering)
    c->allgather_user = c;
    c->world = world;
    c->rank = rank;
    return CHICDIFF_OK;
}

int chicdiff_hip_enable_timing(chicdiff_hip_ctx *c, int32_t on) {
    if (!c) return CHICDIFF_E_INVALID;
    c->timing = (on == 2 || on == 3) ? on : (on != 0 ? 1 : 0);
    return CHICDIFF_OK;
}

}  // extern "C"

// ---- timing: one event pair per launch, accumulated per kernel name when the call ends ------
struct Scope {
    chicdiff_hip_ctx *c;
    int idx = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Scope(chicdiff_hip_ctx *c_, const char *name) : c(c_) {
        if (!c->timing) return;
        if (c->scope_depth++ > 0) return;  // inside an outer scope: no events of its own
        // mode 2: an event pair costs ~2 us on the stream and as much on the host when it is read back; a caller that times whole
        // calls (bench.py) brackets only the fit kernels — 0.09 ms less per call than bracketing all ~20 stages
        if (c->timing == 2 && strcmp(name, "disp_gene") != 0 && strcmp(name, "disp_map") != 0 && strcmp(name, "wald_irls") != 0) return;
        // mode 3: the gene-wise line search alone — the dominant kernel of every configuration measured (an event on the stream is a
        // packet of its own: each pair costs ~12 us of a 1.3 ms step at 250 k rows; rocprofv3 kernel trace, profiles/r05_kernel_gaps_*)
        if (c->timing == 3 && strcmp(name, "disp_gene") != 0) return;
        for (size_t i = 0; i < c->timers.size(); i++)
            if (c->timers[i].name == name) idx = (int)i;
        if (idx < 0) {
            KTimer t;
            t.name = name;
            c->timers.push_back(t);
            idx = (int)c->timers.size() - 1;
        }
        e0 = take();
        e1 = take();
        (void)hipEventRecord(e0, c->stream);
    }
    hipEvent_t take() {
        hipEvent_t e = nullptr;
        if (!c->event_pool.empty()) {
            e = c->event_pool.back();
            c->event_pool.pop_back();
        } else {
            (void)hipEventCreate(&e);
        }
        return e;
    }
    ~Scope() {
        if (!c->timing) return;
        c->scope_depth--;
        if (idx < 0) return;
        (void)hipEventRecord(e1, c->stream);
        c->pending.push_back({idx, e0});
        c->pending.push_back({idx, e1});
    }
};
static void timing_reset(chicdiff_hip_ctx *c) {
    for (auto &t : c->timers) {
        t.ms = 0;
        t.launches = 0;
        t.bytes = 0;
    }
}
static void timing_collect(chicdiff_hip_ctx *c) {
    if (!c->timing) return;
    (void)hipStreamSynchronize(c->stream);
    for (size_t i = 0; i + 1 < c->pending.size(); i += 2) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->pending[i].second, c->pending[i + 1].second);
        c->timers[c->pending[i].first].ms += ms;
        c->timers[c->pending[i].first].launches++;
        c->event_pool.push_back(c->pending[i].second);
        c->event_pool.push_back(c->pending[i + 1].second);
    }
    c->pending.clear();
}

extern "C" int32_t chicdiff_hip_kernel_times(chicdiff_hip_ctx *c, chicdiff_kernel_time *out, int32_t cap) {
    if (!c) return 0;
    int32_t k = 0;
    for (auto &t : c->timers) {
        if (t.launches == 0) continue;
        if (out && k < cap) {
            out[k].name = t.name.c_str();
            out[k].ms = t.ms;
            out[k].launches = t.launches;
            out[k].bytes = t.bytes;
        }
        k++;
    }
    return k;
}

// ---- workspace --------------------------------------------------------------------------
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int ensure_workspace(chicdiff_hip_ctx *c, int64_t n, int S) {
    if (c->ws && n <= c->cap_n && S <= c->cap_S) return CHICDIFF_OK;
    if (c->ws) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->ws));
        c->ws = nullptr;
    }
    const size_t nd = align256(sizeof(double) * (size_t)n), ni = align256(sizeof(int32_t) * (size_t)n);
    const size_t n_double_arrays = 15, n_int_arrays = 9;
    const size_t partials = align256(sizeof(double) * ((size_t)kRedBlocks * 72 + 128));
    const size_t hist = align256(sizeof(double) * (size_t)kMaxS * 2 * kSelBins);
    const size_t nfbytes = align256(sizeof(double) * (size_t)n * S);
    const size_t selcnt = align256(sizeof(double) * (size_t)kSelMaxWorld * kMaxS * 2);
    const size_t rowpack = align256((size_t)row_stride(S) * (size_t)n);
    const size_t start4 = align256(sizeof(double) * 4 * (size_t)n);
    const size_t total = nd * n_double_arrays + ni * n_int_arrays + partials + 2 * hist + selcnt + align256(sizeof(FitScalars)) + kQueueBytes + 1024 + nfbytes + rowpack + start4;
    hipError_t e = hipMalloc(&c->ws, total);
    if (e != hipSuccess) return fail(c, CHICDIFF_E_NOMEM, "workspace of %zu bytes: %s", total, hipGetErrorString(e));
    c->ws_bytes = total;
    char *p = (char *)c->ws;
    auto takeD = [&](double *&ptr) { ptr = (double *)p; p += nd; };
    auto takeI = [&](int32_t *&ptr) { ptr = (int32_t *)p; p += ni; };
    FitWork &w = c->w;
    takeD(w.baseMean); takeD(w.baseVar); takeD(w.gm0); takeD(w.gm1); takeD(w.rough); takeD(w.binit0); takeD(w.binit1);
    takeD(w.crow); takeD(w.dispGene); takeD(w.dispFit); takeD(w.dispMAP); takeD(w.disp); takeD(w.beta0); takeD(w.beta1);
    takeD(w.resid);
    takeI(w.allZero); takeI(w.geneIter); takeI(w.mapIter); takeI(w.outlier); takeI(w.betaIter); takeI(w.optimConv);
    takeI(w.order);
    { int32_t *q; takeI(q); w.cls = (uint8_t *)q; }
    takeI(w.gridlist);
    w.partials = (double *)p; p += partials;
    w.hist = (double *)p; p += hist;
    w.hist_local = (double *)p; p += hist;
    w.selcnt = (double *)p; p += selcnt;
    w.sc = (FitScalars *)p; p += align256(sizeof(FitScalars));
    w.logfact = c->d_logfact;
    w.queue = (unsigned long long *)p; p += kQueueBytes;
    w.barrier = (unsigned int *)p; p += 1024;
    c->d_nf_tmp = (double *)p; p += nfbytes;
    w.rowpack = p; p += rowpack;
    w.start = (double *)p;
    c->cap_n = n;
    c->cap_S = S;
    // scalars, queue heads and barrier counters start from zero: size_factors_impl runs before any fit has cleared them, and a
    // garbage sel_overflow there would send ONE rank of a sharded call into a second pass its peers do not make
    HIPCHK(c, hipMemsetAsync(w.sc, 0, align256(sizeof(FitScalars)) + kQueueBytes + 1024, c->stream));
    return CHICDIFF_OK;
}

static int ensure_aux(chicdiff_hip_ctx *c, size_t bytes) {
    if (c->aux_bytes >= bytes) return CHICDIFF_OK;
    if (c->aux) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->aux));
        c->aux = nullptr;
        c->aux_bytes = 0;
    }
    hipError_t e = hipMalloc((void **)&c->aux, bytes);
    if (e != hipSuccess) return fail(c, CHICDIFF_E_NOMEM, "scratch of %zu bytes: %s", bytes, hipGetErrorString(e));
    c->aux_bytes = bytes;
    return CHICDIFF_OK;
}

// with timing on (mode 1), every collective gets an event pair of its own, whatever scope it sits in: "allreduce" / "allgather" in
// chicdiff_hip_kernel_times = number of collectives of the call, their summed duration on the stream and the bytes this rank
// handed over (that time is ALSO inside the enclosing scope's figure — size_factors, trend_fit, mad_select — so it is a
// breakdown, not an extra term)
struct CollTimer {
    chicdiff_hip_ctx *c;
    int idx = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CollTimer(chicdiff_hip_ctx *c_, const char *name, double bytes) : c(c_) {
        if (c->timing != 1) return;
        for (size_t i = 0; i < c->timers.size(); i++)
            if (c->timers[i].name == name) idx = (int)i;
        if (idx < 0) {
            KTimer t;
            t.name = name;
            c->timers.push_back(t);
            idx = (int)c->timers.size() - 1;
        }
        c->timers[idx].bytes += bytes;
        auto take = [&]() {
            hipEvent_t e = nullptr;
            if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); } else (void)hipEventCreate(&e);
            return e;
        };
        e0 = take();
        e1 = take();
        (void)hipEventRecord(e0, c->stream);
    }
    ~CollTimer() {
        if (idx < 0) return;
        (void)hipEventRecord(e1, c->stream);
        c->pending.push_back({idx, e0});
        c->pending.push_back({idx, e1});
    }
};

static int do_allreduce(chicdiff_hip_ctx *c, double *dev, int64_t count) {
    if (!c->allreduce) return CHICDIFF_OK;  // a callback registered with world_size 1 is still called (tests)
    int rc;
    {
        CollTimer t(c, "allreduce", 8.0 * (double)count);
        rc = c->allreduce(c->allreduce_user, dev, count);
    }
    if (rc != 0) return rc == kCollMsgSet && c->allreduce == rccl_allreduce_cb ? CHICDIFF_E_COMM : fail(c, CHICDIFF_E_COMM, "all-reduce callback failed");
    return CHICDIFF_OK;
}
// every rank contributes `count` doubles at `send`; `recv` gets world x count, rank r's block at r * count
static int do_allgather(chicdiff_hip_ctx *c, const double *send, double *recv, int64_t count) {
    int rc;
    {
        CollTimer t(c, "allgather", 8.0 * (double)count);
        rc = c->allgather(c->allgather_user, send, recv, count);
    }
    if (rc != 0) return rc == kCollMsgSet && c->allgather == rccl_allgather_cb ? CHICDIFF_E_COMM : fail(c, CHICDIFF_E_COMM, "all-gather callback failed");
    return CHICDIFF_OK;
}

static double *sums_of(const FitWork &w) { return w.partials + (size_t)kRedBlocks * 72; }

// ---- HIP backend of fit_driver.h -----------------------------------------------------------
struct HipBackend {
    chicdiff_hip_ctx *c;
    FitDims d;
    Opts o;
    SelArgs sa;  // key source of the running select
    int err = 0;
    bool local = false;  // a sharded fit whose rows are all on this rank for this step (MAD over the gathered trend rows): single-rank path
    bool sharded() const { return c->allreduce && !local; }
    int world() const { return sharded() ? (c->world > 1 ? c->world : 2) : 1; }  // callback set => sharded protocol
    int allreduce(double *buf, int64_t n) { return sharded() ? do_allreduce(c, buf, n) : 0; }
    double *sums() { return c->w.partials; }  // sharded: the per-block partials themselves are all-reduced (one launch fewer per pass)
    int64_t sums_len() const { return (int64_t)trend_blocks() * kTrendSums; }
    double *hist() { return c->w.hist; }
    void trend_init() { launch_trend_init(d, c->w, o, c->stream); }
    void trend_pass(bool fused) {
        Scope t(c, "trend_pass");
        launch_trend_pass(d, c->w, o, c->stream, fused);
    }
    void trend_step() { launch_trend_step(d, c->w, o, c->stream); }
    const FitScalars *sync_scalars() {
        hipError_t e = hipMemcpyAsync(c->h_sc, c->w.sc, sizeof(FitScalars), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            err = fail(c, CHICDIFF_E_HIP, "reading fit scalars: %s", hipGetErrorString(e));
            c->h_sc->finished = 1;  // stop the driver; the caller sees `err`
        }
        return c->h_sc;
    }
    void sel_hist(const SelSpec &, int shift) {
        Scope t(c, "select_hist");
        sa.shift = shift;
        launch_sel_hist(sa, c->w, c->stream);
    }
    void sel_step(const SelSpec &, int shift) {
        Scope t(c, "select_step");
        sa.shift = shift;
        launch_sel_step(sa, c->w, c->stream);
    }
    bool sel_shortcut(const SelSpec &) {
        if (sharded() || c->opt_select_rounds) return false;  // sharded: candidates live on other ranks too
        Scope t(c, "select_shortcut");
        sa.shift = 40;
        launch_sel_shortcut(sa, c->w, c->stream);
        finished_by_shortcut = true;  // its last kernel also did what sel_finish does
        return true;
    }
    bool finished_by_shortcut = false;
    void sel_finish(const SelSpec &) {
        if (!finished_by_shortcut) launch_sel_finish(sa, c->w, c->stream);
    }
    // sharded shortcut
    int world_size() const { return c->world; }
    bool sel_can_gather() const { return c->world <= kSelMaxWorld && !c->opt_select_rounds; }
    double *sel_counts() { return c->w.selcnt; }
    void sel_keep_local_hist(const SelSpec &) { launch_sel_keep_local(sa, c->w, c->stream); }
    void sel_gather_counts(const SelSpec &) { launch_sel_gather_counts(sa, c->w, c->world, c->rank, c->stream); }
    void sel_gather_place(const SelSpec &) {
        Scope t(c, "select_gather");
        sa.shift = 40;
        launch_sel_gather_place(sa, c->w, c->world, c->rank, c->stream);
    }
    void sel_gather_finish(const SelSpec &) { launch_sel_gather_finish(sa, c->w, c->world, c->rank, c->stream); }
    // no look at the device: a list that did not fit sets FitScalars::sel_overflow, which the host sees at the call's last
    // synchronisation and answers with a refit over every histogram round (every rank takes the same decision)
    bool sel_gather_done() { return true; }
};

// exact medians by radix select; results land in w.sc (see sel_finish_kernel)
static int run_select(chicdiff_hip_ctx *c, SelArgs a, bool local = false) {
    HipBackend be{c, FitDims{}, Opts{}, a};
    be.local = local;
    SelSpec spec{a.mode, a.ncol};
    const int rc = drive_select(be, spec);
    if (rc) return c->err[0] ? CHICDIFF_E_COMM : fail(c, CHICDIFF_E_COMM, "select: all-reduce failed");
    return CHICDIFF_OK;
}

static int check_counts_group(chicdiff_hip_ctx *c, int64_t n, int32_t S, const int32_t *group, FitDims &d) {
    if (n < 1 || n > 2147483647ll || S < 2 || S > kMaxS)
        return fail(c, CHICDIFF_E_INVALID, "need 1 <= n < 2^31 and 2 <= S <= %d (got n=%lld, S=%d)", kMaxS, (long long)n, S);
    d.n = n;
    d.S = S;
    d.gmask = 0;
    d.nA = d.nB = 0;
    for (int j = 0; j < S; j++) {
        const int g = group ? group[j] : 0;
        if (g != 0 && g != 1) return fail(c, CHICDIFF_E_INVALID, "group[%d]=%d: only two-level designs (0/1) or ~1 (all 0) are supported", j, g);
        if (g) { d.gmask |= (1ull << j); d.nB++; } else d.nA++;
    }
    d.p = d.nB > 0 ? 2 : 1;
    if (d.p == 2 && d.nA == 0) return fail(c, CHICDIFF_E_INVALID, "design has no sample in the reference level");
    if (S <= d.p) return fail(c, CHICDIFF_E_INVALID, "no residual degrees of freedom (S=%d, p=%d)", S, d.p);
    return CHICDIFF_OK;
}

// opts the caller filled in: fitType must be one of DESeq2's three (a NA_integer_ from R arrives as INT_MIN)
static int check_opts(chicdiff_hip_ctx *c, const chicdiff_nbglm_opts *opts) {
    if (opts && (opts->fitType < 0 || opts->fitType > 2))
        return fail(c, CHICDIFF_E_INVALID, "opts.fitType = %d: 0 (parametric), 1 (mean) or 2 (local) expected", opts->fitType);
    return CHICDIFF_OK;
}

// Sharded fits: an argument error on ONE rank (e.g. an empty shard when n < world size) must not leave its peers blocked
// in the first collective.  Every rank therefore contributes its local verdict to one sum-all-reduce before the fit
// starts, and all return together.  (Single process: the local verdict.)
static int shard_consensus(chicdiff_hip_ctx *c, int local_rc, int64_t n = 0) {
    c->shard_n_valid = false;
    if (!c->allreduce) return local_rc;
    char keep[sizeof c->err];
    memcpy(keep, c->err, sizeof keep);
    // one all-reduce carries the verdict and — zero everywhere but in the rank's own slot — the shards' row counts, which the
    // gathered trend needs (round 2 exchanged them with a collective and two host synchronisations of their own)
    const int world = c->world > 0 ? c->world : 1, slots = world <= kSelMaxWorld ? 1 + world : 1;
    double buf[1 + kSelMaxWorld] = {0};
    buf[0] = local_rc ? 1.0 : 0.0;
    if (slots > 1 && c->rank >= 0 && c->rank < world) buf[1 + c->rank] = local_rc ? 0.0 : (double)n;
    double *d_flag = c->d_sf + kMaxS;  // spare doubles behind the size factors
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMemcpyAsync(d_flag, buf, sizeof(double) * slots, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, CHICDIFF_E_HIP, "shard consensus: %s", hipGetErrorString(e));
    if (do_allreduce(c, d_flag, slots)) return CHICDIFF_E_COMM;
    e = hipMemcpyAsync(buf, d_flag, sizeof(double) * slots, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, CHICDIFF_E_HIP, "shard consensus: %s", hipGetErrorString(e));
    const double flag = buf[0];
    if (slots > 1) {
        for (int r = 0; r < world; r++) c->shard_n[r] = buf[1 + r];
        c->shard_n_valid = flag == 0.0 && n > 0;
    }
    if (local_rc) {
        memcpy(c->err, keep, sizeof keep);
        return local_rc;
    }
    if (flag > 0) return fail(c, CHICDIFF_E_INVALID, "%d rank(s) of the sharded fit rejected their arguments (e.g. an empty shard): nothing was fitted", (int)flag);
    return CHICDIFF_OK;
}

static Opts make_opts(const chicdiff_hip_ctx *c, const chicdiff_nbglm_opts *in, int S) {
    chicdiff_nbglm_opts o;
    if (in) o = *in; else chicdiff_hip_default_opts(&o);
    Opts r;
    r.minDisp = o.minDisp; r.dispTol = o.dispTol; r.kappa0 = o.kappa0; r.betaTol = o.betaTol; r.minmu = o.minmu;
    r.outlierSD = o.outlierSD; r.dispPriorVarIn = o.dispPriorVar; r.maxit = o.maxit; r.betaMaxit = o.betaMaxit;
    r.maxDisp = S > 10 ? (double)S : 10.0;
    r.trendIn[0] = o.trendCoef[0];
    r.trendIn[1] = o.trendCoef[1];
    r.fit_type = o.fitType;
    r.spread = c->opt_spread;
    r.min_waves = c->opt_min_waves;
    r.prio = c->opt_prio;
    r.schedule = c->opt_schedule;
    r.deal = c->opt_deal;
    r.chunk = c->opt_chunk;
    r.classes_a = c->opt_classes_a;
    r.trend_blocks = c->opt_trend_blocks;
    return r;
}

// ---- DESeq2 localDispersionFit on the device (global_kernels.hip lf_*): the host grows locfit's tree ----------------
// One order statistic of key(x) or key(|x - xv|) over the rows of the fit: byte-wise radix select, eight histogram rounds
// (each a sum over rows, all-reduced when sharded).  rank is 0-based; *total gets the number of rows (first round).
static int lf_order_stat(chicdiff_hip_ctx *c, FitDims d, const Opts &o, int use_dist, double xv, double rank, double *value, double *total) {
    double *d_hist = c->w.hist;
    uint64_t prefix = 0;
    double h[256];
    for (int shift = 56; shift >= 0; shift -= 8) {
        launch_lf_hist(d, c->w, o, use_dist, xv, prefix, shift, d_hist, c->stream);
        if (int rc = do_allreduce(c, d_hist, 256)) return rc;
        HIPCHK(c, hipMemcpyAsync(h, d_hist, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double cum = 0;
        if (shift == 56 && total) {
            *total = 0;
            for (int b = 0; b < 256; b++) *total += h[b];
        }
        int b = 0;
        for (; b < 255; b++) {
            if (cum + h[b] > rank) break;
            cum += h[b];
        }
        rank -= cum;
        prefix |= (uint64_t)b << shift;
    }
    *value = value_of(prefix);
    return CHICDIFF_OK;
}
// bandwidth, value and slope of the local quadratic at xv (locfit: nbhd / kordstat, tricube, Taylor basis 1, dx, dx^2/2)
static int lf_vertex(chicdiff_hip_ctx *c, FitDims d, const Opts &o, double nfit, double xv, double *h, double *f, double *df) {
    const double k = floor(nfit * 0.7);  // alpha = 0.7
    if (k < 1) return CHICDIFF_E_NUMERIC;
    int rc = lf_order_stat(c, d, o, 1, xv, k - 1, h, nullptr);
    if (rc) return rc;
    if (!(*h > 0)) return CHICDIFF_E_NUMERIC;
    double *d_out = c->w.hist + 256, s[8];
    launch_lf_sums(d, c->w, o, xv, *h, c->w.partials, d_out, c->stream);
    if ((rc = do_allreduce(c, d_out, 8))) return rc;
    HIPCHK(c, hipMemcpyAsync(s, d_out, sizeof s, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    long double A[3][4] = {{s[0], s[1], (long double)s[2] / 2, s[5]},
                           {s[1], s[2], (long double)s[3] / 2, s[6]},
                           {(long double)s[2] / 2, (long double)s[3] / 2, (long double)s[4] / 4, (long double)s[7] / 2}};
    for (int col = 0; col < 3; col++) {
        int piv = col;
        for (int r = col + 1; r < 3; r++)
            if (fabsl(A[r][col]) > fabsl(A[piv][col])) piv = r;
        if (!(fabsl(A[piv][col]) > 0)) return CHICDIFF_E_NUMERIC;
        if (piv != col)
            for (int q = 0; q < 4; q++) std::swap(A[col][q], A[piv][q]);
        for (int r = col + 1; r < 3; r++) {
            const long double m = A[r][col] / A[col][col];
            for (int q = col; q < 4; q++) A[r][q] -= m * A[col][q];
        }
    }
    long double b[3];
    for (int r = 2; r >= 0; r--) {
        long double t = A[r][3];
        for (int q = r + 1; q < 3; q++) t -= A[r][q] * b[q];
        b[r] = t / A[r][r];
    }
    *f = (double)b[0];
    *df = (double)b[1];
    return CHICDIFF_OK;
}
struct LfTree {
    int nv = 0;
    double x[kLfMaxV], h[kLfMaxV], f[kLfMaxV], d[kLfMaxV];
};
// a new vertex at xv: its index, or -1 with *rc set (CHICDIFF_E_NUMERIC: no fit possible there / too many vertices)
static int lf_add(chicdiff_hip_ctx *c, FitDims d, const Opts &o, double nfit, LfTree &t, double xv, int *rc) {
    if (t.nv >= kLfMaxV) { *rc = CHICDIFF_E_NUMERIC; return -1; }
    if ((*rc = lf_vertex(c, d, o, nfit, xv, &t.h[t.nv], &t.f[t.nv], &t.d[t.nv]))) return -1;
    t.x[t.nv] = xv;
    return t.nv++;
}
// locfit atree_grow in one dimension: cut the cell at its midpoint while its length exceeds cut = 0.8 bandwidths
static int lf_grow(chicdiff_hip_ctx *c, FitDims d, const Opts &o, double nfit, LfTree &t, int il, int ir, double range) {
    const double le = t.x[ir] - t.x[il];
    double hmin = t.h[il] > 0 ? t.h[il] : 0;
    if (t.h[ir] > 0 && (hmin == 0 || t.h[ir] < hmin)) hmin = t.h[ir];
    const double score = hmin == 0 ? 2 * le / range : le / hmin;
    if (!(0.8 < score)) return CHICDIFF_OK;
    int rc = CHICDIFF_OK;
    const int im = lf_add(c, d, o, nfit, t, (t.x[il] + t.x[ir]) / 2, &rc);
    if (im < 0) return rc;
    if ((rc = lf_grow(c, d, o, nfit, t, il, im, range))) return rc;
    return lf_grow(c, d, o, nfit, t, im, ir, range);
}
// returns CHICDIFF_OK with dispFit[] filled and the trend marked local, or CHICDIFF_E_NUMERIC when no fit is possible
static int local_trend_fit(chicdiff_hip_ctx *c, FitDims d, const Opts &o) {
    double lo, hi, nfit = 0, dummy;
    int rc = lf_order_stat(c, d, o, 0, 0.0, 0.0, &lo, &nfit);  // min(x) and the number of rows in the fit
    if (rc) return rc;
    if (nfit < 4) return CHICDIFF_E_NUMERIC;
    if ((rc = lf_order_stat(c, d, o, 0, 0.0, nfit - 1, &hi, &dummy))) return rc;  // max(x)
    if (!(hi > lo)) return CHICDIFF_E_NUMERIC;
    LfTree t;
    const int il = lf_add(c, d, o, nfit, t, lo, &rc), ir = il < 0 ? -1 : lf_add(c, d, o, nfit, t, hi, &rc);
    if (il < 0 || ir < 0) return rc;
    if ((rc = lf_grow(c, d, o, nfit, t, il, ir, hi - lo))) return rc;
    LfVerts v{};
    std::vector<int> order(t.nv);
    for (int i = 0; i < t.nv; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int p, int q) { return t.x[p] < t.x[q]; });
    v.nv = t.nv;
    for (int i = 0; i < t.nv; i++) { v.x[i] = t.x[order[i]]; v.f[i] = t.f[order[i]]; v.d[i] = t.d[order[i]]; }
    launch_lf_eval(d, c->w, v, c->stream);
    return CHICDIFF_OK;
}

// Sharded fits: the parametric trend on ALL ranks' rows, on every rank.  The fit is ~20 dependent IRLS passes over two
// doubles per row; sharding it costs one latency-bound all-reduce per pass (20 of a sharded fit's ~38 collectives), while
// the rows themselves are small: 16 bytes each.  So the ranks exchange them once — their sizes (with the argument verdicts),
// then the rows: one all-gather of the ranks' padded (x | y) blocks (ncclAllGather, 32 MB received at 2 M rows), or, on a
// transport without an all-gather hook, a sum-all-reduce over zero-initialised all-ranks arrays (twice the bytes) — and each
// runs the single-launch persistent kernel on the whole set: one collective instead of twenty, and the trend of a sharded
// fit is bit-identical to the single-rank one (same rows, same order, same kernel).
static int gathered_trend(chicdiff_hip_ctx *c, FitDims d, const Opts &o) {
    hipStream_t st = c->stream;
    FitWork &w = c->w;
    const int real_world = c->world > 0 ? c->world : 1, rank = c->rank;
    const int fake = (real_world == 1 && c->opt_fake_world > 1 && c->allgather != nullptr) ? c->opt_fake_world : 0;  // bench hook, see opt_fake_world
    const int world = fake ? fake : real_world;
    std::vector<double> cnt((size_t)world, 0.0);
    int rc;
    if (fake) {
        for (int r = 0; r < world; r++) cnt[(size_t)r] = (double)d.n;
    } else if (c->shard_n_valid && world <= kSelMaxWorld && (int64_t)c->shard_n[rank] == d.n) {
        for (int r = 0; r < world; r++) cnt[(size_t)r] = c->shard_n[r];  // exchanged with the argument verdicts: no collective, no host stop
    } else {
        cnt[(size_t)rank] = (double)d.n;
        double *d_cnt = w.hist;
        HIPCHK(c, hipMemcpyAsync(d_cnt, cnt.data(), sizeof(double) * world, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));  // cnt lives on this frame
        if ((rc = do_allreduce(c, d_cnt, world))) return rc;
        HIPCHK(c, hipMemcpyAsync(cnt.data(), d_cnt, sizeof(double) * world, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    int64_t off = 0, total = 0, maxn = 0;
    for (int r = 0; r < world; r++) {
        if (r < rank) off += (int64_t)cnt[(size_t)r];
        total += (int64_t)cnt[(size_t)r];
        if ((int64_t)cnt[(size_t)r] > maxn) maxn = (int64_t)cnt[(size_t)r];
    }
    // layout of the gather buffer: x[total], y[total], flags[total] (int32, zero), resid[total], then — all-gather transport —
    // this rank's padded block (x[maxn] | y[maxn]) and the ranks' blocks as they arrive (world x 2 maxn)
    const bool by_gather = c->allgather != nullptr && world <= kGatherMaxWorld;
    const size_t nd = align256(sizeof(double) * (size_t)total), ni = align256(sizeof(int32_t) * (size_t)total);
    const size_t blk = align256(sizeof(double) * 2 * (size_t)maxn);
    const size_t need = 3 * nd + ni + (by_gather ? blk * (size_t)(world + 1) : 0);
    if (c->tg_bytes < need) {
        if (c->tg_buf) { HIPCHK(c, hipStreamSynchronize(st)); HIPCHK(c, hipFree(c->tg_buf)); }
        c->tg_buf = nullptr;
        c->tg_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&c->tg_buf, need));
        c->tg_bytes = need;
    }
    double *xg = (double *)c->tg_buf, *yg = (double *)(c->tg_buf + nd);
    int32_t *flags = (int32_t *)(c->tg_buf + 2 * nd);
    c->tg_total = total;
    c->tg_x = xg;
    c->tg_y = yg;
    c->tg_flags = flags;
    c->tg_resid = (double *)(c->tg_buf + 2 * nd + ni);
    if (by_gather) {
        // one ncclAllGather of the ranks' (x | y) blocks, padded to the largest shard, then a copy into the contiguous all-ranks
        // arrays: half the bytes of the sum-all-reduce over zero-filled arrays it replaces, and no floating-point adds
        double *send = (double *)(c->tg_buf + 3 * nd + ni), *recv = (double *)(c->tg_buf + 3 * nd + ni + blk);
        HIPCHK(c, hipMemsetAsync(flags, 0, ni, st));
        launch_trend_gather(d, w, o, send, send + maxn, st);
        if ((rc = do_allgather(c, send, recv, (int64_t)(blk / sizeof(double))))) return rc;
        for (int r = 1; r < fake; r++)  // the other "ranks'" blocks: copies of this one's
            HIPCHK(c, hipMemcpyAsync((char *)recv + (size_t)r * blk, recv, blk, hipMemcpyDeviceToDevice, st));
        GatherLayout gl{};
        gl.world = world;
        gl.block = (int64_t)(blk / sizeof(double));
        gl.maxn = maxn;
        int64_t o2 = 0;
        for (int r = 0; r < world; r++) { gl.off[r] = o2; o2 += (int64_t)cnt[(size_t)r]; }
        gl.off[world] = o2;
        launch_trend_compact(gl, recv, xg, yg, st);
    } else {
        HIPCHK(c, hipMemsetAsync(c->tg_buf, 0, 2 * nd + ni, st));
        launch_trend_gather(d, w, o, xg + off, yg + off, st);
        if ((rc = do_allreduce(c, xg, (int64_t)(2 * nd / sizeof(double))))) return rc;  // x and y are contiguous (padding included)
    }
    FitDims dg = d;
    dg.n = total;
    FitWork wg = w;
    wg.baseMean = xg;
    wg.dispGene = yg;
    wg.allZero = flags;  // all zero: a row that does not take part carries y = NaN
    wg.resid = c->tg_resid;
    launch_trend_persistent(dg, wg, o, st, c->opt_mad_in_kernel != 0);
    return CHICDIFF_OK;
}

// Per-row workspace columns the caller wants back are written where the caller wants them (round 3 copied them out when the fit
// ended: two 16 MB device copies per bench step): for the duration of one fit the workspace pointers point into the caller's
// buffers; restored on every way out.
struct WorkRedirect {
    chicdiff_hip_ctx *c;
    FitWork saved;
    WorkRedirect(chicdiff_hip_ctx *c_, const chicdiff_nbglm_out *o) : c(c_), saved(c_->w) {
        if (!o) return;
        FitWork &w = c->w;
        if (o->baseMean) w.baseMean = o->baseMean;
        if (o->baseVar) w.baseVar = o->baseVar;
        if (o->dispGeneEst) w.dispGene = o->dispGeneEst;
        if (o->dispFit) w.dispFit = o->dispFit;
        if (o->dispMAP) w.dispMAP = o->dispMAP;
        if (o->dispersion) w.disp = o->dispersion;
        if (o->dispGeneIter) w.geneIter = o->dispGeneIter;
        if (o->dispIter) w.mapIter = o->dispIter;
        if (o->dispOutlier) w.outlier = o->dispOutlier;
        if (o->allZero) w.allZero = o->allZero;
    }
    ~WorkRedirect() { c->w = saved; }
};

static int fit_dev_impl(chicdiff_hip_ctx *c, const int32_t *d_counts, const double *d_nf, FitDims d, Opts o,
                        const chicdiff_nbglm_out *d_out, chicdiff_nbglm_scalars *scalars) {
    int rc;
    hipStream_t st = c->stream;
    WorkRedirect redirect(c, d_out);
    FitWork &w = c->w;
    c->tg_total = 0;
    // the scalars, the queue heads and the barrier counters sit next to each other in the workspace: one fill
    HIPCHK(c, hipMemsetAsync(w.sc, 0, align256(sizeof(FitScalars)) + kQueueBytes + 1024, st));
    // sharded: the column sums of nf travel as the ranks' double-double pairs (zero-filled slots + sum-all-reduce = an exact gather)
    // and are added in rank order — the correctly rounded exact sum, as on one rank: same xim, same start values, same bits downstream
    const int world = c->world > 0 ? c->world : 1;
    const bool col_slots = c->allreduce && world <= kSelMaxWorld;
    const size_t slot_doubles = (size_t)(d.S + 1) * 2;
    double *slots = col_slots ? w.hist : nullptr;  // (the select histograms are idle here)
    if (col_slots) HIPCHK(c, hipMemsetAsync(slots, 0, sizeof(double) * slot_doubles * world, st));
    {
        Scope t(c, "prep");
        const bool fused = c->fuse.fm != nullptr && d.S <= 16 && d_nf == c->d_nf_tmp;
        launch_prep(d_counts, const_cast<double *>(d_nf), d, w, o, st, fused ? c->fuse : FusedOffsets());
        launch_prep_finish(d, w, col_slots ? slots + slot_doubles * c->rank : nullptr, st);
    }
    if (col_slots) {
        if ((rc = do_allreduce(c, slots, (int64_t)(slot_doubles * world)))) return rc;
    } else if ((rc = do_allreduce(c, w.sc->colsum, kMaxS + 1)))  // colsum[kMaxS] + nnz are contiguous
        return rc;
    if (c->allreduce) launch_xim(d, w, slots, world, st);  // (the ranks' column sums, exchanged above, added in rank order -> xim)
    {
        Scope t(c, "disp_gene");
        Opts og = o;
        og.xim_here = c->allreduce ? 0 : 1;  // single rank: disp_init forms xim from the column sums itself (one dependent launch less)
        launch_disp_gene(d_counts, d_nf, d, w, og, st);
    }
    // trend: fit_driver.h runs batches of IRLS passes and polls the finished flag between batches
    bool mad_in_kernel = false;  // the persistent trend kernel went on to the residuals, their median and MAD, and the closed-form prior variance
    if (o.trendIn[0] == o.trendIn[0] && o.trendIn[1] == o.trendIn[1]) {  // caller-supplied dispersion function
        launch_trend_init(d, w, o, st);
        HIPCHK(c, hipMemcpyAsync(w.sc->coefs, o.trendIn, sizeof(double) * 2, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));  // o.trendIn lives on this frame
    } else if (o.fit_type == 1) {
        // fitType = "mean" (DESeq2 estimateDispersionsFit): one value for every row, the 0.1 %-trimmed mean of the gene-wise
        // estimates above 10 minDisp — an explicit, rarely used alternative, so the order statistics are taken on the host
        if (c->allreduce) return fail(c, CHICDIFF_E_INVALID, "fitType \"mean\" is not available for sharded fits");
        std::vector<double> g((size_t)d.n);
        HIPCHK(c, hipMemcpyAsync(g.data(), w.dispGene, sizeof(double) * (size_t)d.n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        size_t m = 0;
        for (size_t i = 0; i < g.size(); i++)
            if (g[i] > 10 * o.minDisp) g[m++] = g[i];  // NaN (all-zero rows) fails the comparison
        if (m == 0) return fail(c, CHICDIFF_E_NUMERIC, "fitType \"mean\": no gene-wise estimate above 10 * minDisp");
        // R mean(x, trim): lo = floor(n trim) + 1, hi = n + 1 - lo, mean(sort(x)[lo:hi]) with long double accumulation
        const size_t lo = (size_t)floor((double)m * 0.001), hi = m - lo;  // 0-based half-open [lo, hi)
        if (lo > 0) {
            std::nth_element(g.begin(), g.begin() + lo, g.begin() + m);
            std::nth_element(g.begin() + lo, g.begin() + (hi - 1), g.begin() + m);
        }
        long double acc = 0;
        for (size_t i = lo; i < hi; i++) acc += g[i];
        long double mean = acc / (long double)(hi - lo), t = 0;
        for (size_t i = lo; i < hi; i++) t += g[i] - mean;
        mean += t / (long double)(hi - lo);
        const double coefs[2] = {(double)mean, 0.0};
        launch_trend_init(d, w, o, st);
        HIPCHK(c, hipMemcpyAsync(w.sc->coefs, coefs, sizeof coefs, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
    } else if (o.fit_type == 2) {
        // fitType = "local": on request, or — from the re-entry at the end of this function — DESeq2's own substitute for
        // a parametric fit that failed ("a local regression fit was automatically substituted")
        Scope t(c, "trend_fit");
        launch_trend_init(d, w, o, st);
        rc = local_trend_fit(c, d, o);
        if (rc == CHICDIFF_E_NUMERIC) {
            const int32_t one = 1;  // no local fit either (fewer than four usable rows): report the trend as failed
            HIPCHK(c, hipMemcpyAsync(&w.sc->failed, &one, sizeof one, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));
        } else if (rc)
            return rc;
    } else if (!c->allreduce && !c->no_persistent_trend && c->cu_count >= trend_persistent_blocks() && !c->opt_trend_multilaunch) {
        Scope t(c, "trend_fit");  // single rank: one persistent launch (LDS-resident rows, grid barrier per IRLS pass)
        launch_trend_persistent(d, w, o, st, c->opt_mad_in_kernel != 0);  // no host round trip: `failed` comes back with the final scalars
        mad_in_kernel = c->opt_mad_in_kernel != 0;
    } else if (c->allreduce && c->opt_trend_gather && !c->no_persistent_trend && c->cu_count >= trend_persistent_blocks() &&
               !c->opt_trend_multilaunch) {
        Scope t(c, "trend_fit");
        if ((rc = gathered_trend(c, d, o))) return rc;
        mad_in_kernel = c->opt_mad_in_kernel != 0;
    } else {
        Scope t(c, "trend_fit");
        HipBackend be{c, d, o, SelArgs{}};
        const int trc = drive_trend(be);
        if (be.err) return be.err;
        if (trc == -1) return CHICDIFF_E_COMM;
        if (trc == -2) return fail(c, CHICDIFF_E_NUMERIC, "trend state machine did not finish");
    }
    if (c->opt_fault & 2) {  // test hook: this rank's trend kernel "lost its grid barrier"
        c->opt_fault &= ~2;
        launch_poke(&w.sc->failed, 3, st);
    }
    int status = 0;
    const bool prior_by_simulation = !(o.dispPriorVarIn == o.dispPriorVarIn) && d.S - d.p <= 3 && d.S > d.p;
    // MAD of the log residuals.  A sharded fit that gathered the trend's rows has every rank's (baseMean, dispGeneEst) on this
    // rank already: the residuals of ALL rows are formed here and the two medians (and the residual histogram of the d.f. <= 3
    // prior) are taken locally, single-rank path — no collective at all instead of eight (nine) latency-bound ones, and the
    // same bits as the one-rank fit
    const bool mad_local = c->allreduce && c->tg_total > 0;
    FitDims dm = d;
    FitWork wm = w;
    if (mad_local) {
        dm.n = c->tg_total;
        wm.baseMean = c->tg_x;
        wm.dispGene = c->tg_y;
        wm.allZero = c->tg_flags;
        wm.resid = c->tg_resid;
    }
    if (mad_in_kernel) {
        if (prior_by_simulation) {  // the residuals are in wm.resid: their histogram, then the simulation-matched prior variance
            Scope t(c, "prior_mc");
            double *d_hist = sums_of(w) + 32;
            launch_resid_hist(dm, wm, d_hist, st);
            const int df = d.S - d.p;
            if (!c->d_pmc[df]) {  // built once per process, uploaded once per context
                HIPCHK(c, hipMalloc((void **)&c->d_pmc[df], sizeof(PmcTable)));
                HIPCHK(c, hipMemcpy(c->d_pmc[df], &pmc_table(df), sizeof(PmcTable), hipMemcpyHostToDevice));
            }
            launch_prior_mc(d, w, d_hist, c->d_pmc[df], st);
        }
    } else {
        launch_dispfit_resid(dm, wm, o, st);
        SelArgs sa{};
        sa.n = dm.n;
        sa.ncol = 1;
        sa.resid = wm.resid;
        {
            Scope t(c, "mad_select");
            sa.mode = SEL_RESID;
            if ((rc = run_select(c, sa, mad_local))) return rc;
            sa.mode = SEL_ABSDEV;
            if ((rc = run_select(c, sa, mad_local))) return rc;
            if (prior_by_simulation) {
                // residual d.f. <= 3: DESeq2 matches the prior variance by simulation (prior_mc.h).  The 200 x 40 simulated
                // densities are constants (built once per process and d.f.); the matching itself runs on the device
                double *d_hist = sums_of(w) + 32;
                launch_resid_hist(dm, wm, d_hist, st);
                if (!mad_local && (rc = do_allreduce(c, d_hist, kPmcBins))) return rc;
                const int df = d.S - d.p;
                if (!c->d_pmc[df]) {  // built once per process, uploaded once per context
                    HIPCHK(c, hipMalloc((void **)&c->d_pmc[df], sizeof(PmcTable)));
                    HIPCHK(c, hipMemcpy(c->d_pmc[df], &pmc_table(df), sizeof(PmcTable), hipMemcpyHostToDevice));
                }
                launch_prior_mc(d, w, d_hist, c->d_pmc[df], st);
            } else {
                launch_prior_var(d, w, o, st);
            }
        }
    }
    if (c->opt_fault & 1) {  // test hook: this rank's select "could not fit its candidate list"
        c->opt_fault &= ~1;
        launch_poke(&w.sc->sel_overflow, 1, st);
    }
    {
        Scope t(c, "disp_map");
        launch_disp_map(d_counts, d_nf, d, w, o, st);
    }
    if (c->early && c->early->n > 0) {  // these columns are final: off to the host while the Wald stage runs
        const void *src[10] = {w.baseMean, w.baseVar, w.dispGene, w.dispFit, w.dispMAP, w.disp, w.geneIter, w.mapIter, w.outlier, w.allZero};
        HIPCHK(c, hipEventRecord(c->early->ready, st));
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->early->ready, 0));
        for (int k = 0; k < c->early->n; k++)
            HIPCHK(c, hipMemcpyAsync(c->early->item[k].pin, src[c->early->item[k].col], c->early->item[k].bytes, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(c, hipEventRecord(c->early->done, c->copy_stream));
        c->early->issued = true;
    }
    static const chicdiff_nbglm_out none{};
    const chicdiff_nbglm_out &out = d_out ? *d_out : none;
    if (d.p == 2) {
        {
            Scope t(c, "wald_prep");
            launch_wald_prep(d_counts, d_nf, d, w, o, st);
        }
        {
            Scope t(c, "wald_irls");
            launch_wald_irls(d_counts, d_nf, d, w, o, st);
        }
        Scope t(c, "wald_final");
        launch_wald_final(d_counts, d_nf, d, w, o, out, st);
    } else {
        Scope t(c, "wald_intercept");
        launch_wald_intercept(d_counts, d_nf, d, w, o, out, st);
    }
    launch_dev_sum_finish(d, w, c->d_carry, c->d_sf, st);
    // deviance sum, non-converged rows, all-zero rows, then four verdicts as rank counts: trend kernel timed out, negative / NA count,
    // a select's candidate list overflowed in this fit, ... in the size-factor select the caller ran before it
    if ((rc = do_allreduce(c, w.sc->final_sums, 7))) return rc;
    // one read brings the scalars, the sums and the size factors back (the per-row columns the caller asked for were written in place)
    HIPCHK(c, hipMemcpyAsync(c->h_sc, w.sc, sizeof(FitScalars), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    const double *hs = c->h_sc->final_sums;
    // hs[4] is all-reduced like the other sums: a negative / NA count on ANY rank has entered everybody's size factors, trend and
    // prior, so every rank refuses the fit together (and none goes on to the retry below, whose collectives the others would miss)
    if (hs[4] > 0)
        return fail(c, CHICDIFF_E_INVALID, c->h_sc->neg_counts ? "counts contain a negative value or NA_integer_"
                                                               : "counts contain a negative value or NA_integer_ (on another rank of the sharded fit)");
    // The verdicts below are sums over the ranks (hs[3..6]): whatever ONE rank saw, every rank takes the same branch and re-enters
    // the fit together — none is left waiting in a collective its peers never issue.
    c->sf_overflow_seen = hs[6] > 0;  // the caller (wald_test_dev) repeats size factors + fit with every histogram round
    if (c->sf_overflow_seen && !c->opt_select_rounds) return CHICDIFF_OK;  // (results of this pass are discarded there)
    if (hs[5] > 0 && !c->opt_select_rounds) {  // a sharded select's candidate list did not fit: every histogram round instead
        c->opt_select_rounds = 1;
        c->refits++;
        rc = fit_dev_impl(c, d_counts, d_nf, d, o, d_out, scalars);
        c->opt_select_rounds = 0;  // (it was 0: checked above)
        return rc;
    }
    if (c->h_sc->failed == 3 || hs[3] > 0) {
        // the persistent trend kernel's workgroups were not all resident within the barrier's patience (a GPU shared
        // with other work): fit again with one launch per IRLS pass, and stay with that for this context
        if (c->no_persistent_trend) return fail(c, CHICDIFF_E_HIP, "trend fit: grid barrier timed out");
        c->no_persistent_trend = true;
        c->refits++;
        return fit_dev_impl(c, d_counts, d_nf, d, o, d_out, scalars);
    }
    if (c->h_sc->failed && o.fit_type == 0 && !(o.trendIn[0] == o.trendIn[0]) && !c->opt_no_local_substitute) {
        // estimateDispersionsFit: the parametric fit failed -> fitType <- "local", and the fit is done again from the trend on
        // (everything before it is recomputed too: a rare path, kept simple)
        o.fit_type = 2;
        c->refits++;
        return fit_dev_impl(c, d_counts, d_nf, d, o, d_out, scalars);
    }
    if (c->h_sc->failed) status |= CHICDIFF_ST_TREND_FAILED;
    if (c->h_sc->trend_local) status |= CHICDIFF_ST_TREND_LOCAL;
    if (scalars) {
        const FitScalars *s = c->h_sc;
        scalars->trendCoef[0] = s->coefs[0];
        scalars->trendCoef[1] = s->coefs[1];
        scalars->varLogDispEsts = s->varLogDispEsts;
        scalars->dispPriorVar = s->dispPriorVar;
        scalars->nAllZero = (int64_t)hs[2];
        scalars->sumDeviance = hs[2] > 0 ? NAN : hs[0];  // sum() without na.rm, chicdiff.R:1647
        scalars->trendOuterIter = s->outer_it;
        if (hs[2] > 0) status |= CHICDIFF_ST_ALLZERO_ROWS;
        if (hs[1] > 0) status |= CHICDIFF_ST_BETA_NONCONV;
        if (prior_by_simulation) status |= CHICDIFF_ST_PRIORVAR_MC;
        scalars->status = status;
    }
    return CHICDIFF_OK;
}

extern "C" {

int chicdiff_hip_nbglm_fit_dev(chicdiff_hip_ctx *c, const int32_t *d_counts, const double *d_nf, int64_t n, int32_t S,
                               const int32_t *group, const chicdiff_nbglm_opts *opts, const chicdiff_nbglm_out *d_out,
                               chicdiff_nbglm_scalars *scalars) {
    if (!c) return CHICDIFF_E_INVALID;
    FitDims d;
    int rc = (!d_counts || !d_nf) ? fail(c, CHICDIFF_E_INVALID, "counts / nf pointer is NULL") : check_counts_group(c, n, S, group, d);
    if (!rc) rc = check_opts(c, opts);
    timing_reset(c);  // (before the verdicts' collective: it is one of the call's)
    if ((rc = shard_consensus(c, rc, n))) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, n, S))) return rc;
    c->refits = 0;
    HIPCHK(c, hipMemsetAsync(c->d_carry, 0, sizeof(int32_t), c->stream));  // no size-factor select belongs to this call
    rc = fit_dev_impl(c, d_counts, d_nf, d, make_opts(c, opts, S), d_out, scalars);
    timing_collect(c);
    return rc;
}

// Host-buffer entry point (what R's .Call shim passes: INTEGER(counts), REAL(nf)).  The caller's pageable buffers go
// through a pinned staging area in slices: host threads copy a slice in (checking the counts for NA / negatives on
// the way) and enqueue its DMA at once, so the CPU copy of slice k+1 overlaps the PCIe transfer of slice k; results
// come back the same way.  Device arena and staging area live in the context and only ever grow.
static int ensure_io(chicdiff_hip_ctx *c, size_t dev_bytes, size_t pin_bytes) {
    if (c->io_dev_bytes < dev_bytes) {
        if (c->io_dev) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->io_dev); c->io_dev = nullptr; c->io_dev_bytes = 0; }
        hipError_t e = hipMalloc((void **)&c->io_dev, dev_bytes);
        if (e != hipSuccess) return fail(c, CHICDIFF_E_NOMEM, "device arena of %zu bytes: %s", dev_bytes, hipGetErrorString(e));
        c->io_dev_bytes = dev_bytes;
    }
    if (c->io_pin_bytes < pin_bytes) {
        if (c->io_pin) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->io_pin); c->io_pin = nullptr; c->io_pin_bytes = 0; }
        hipError_t e = hipHostMalloc((void **)&c->io_pin, pin_bytes);
        if (e != hipSuccess) return fail(c, CHICDIFF_E_NOMEM, "pinned staging of %zu bytes: %s", pin_bytes, hipGetErrorString(e));
        c->io_pin_bytes = pin_bytes;
    }
    return CHICDIFF_OK;
}

int chicdiff_hip_nbglm_fit(chicdiff_hip_ctx *c, const int32_t *counts, const double *nf, int64_t n, int32_t S,
                           const int32_t *group, const chicdiff_nbglm_opts *opts, const chicdiff_nbglm_out *out,
                           chicdiff_nbglm_scalars *scalars) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!counts || !nf) return fail(c, CHICDIFF_E_INVALID, "counts / nf pointer is NULL");
    if (n < 1 || S < 2 || S > kMaxS) return fail(c, CHICDIFF_E_INVALID, "need n >= 1 and 2 <= S <= %d", kMaxS);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t cb = sizeof(int32_t) * (size_t)n * S, fb = sizeof(double) * (size_t)n * S;
    const size_t nd = align256(sizeof(double) * (size_t)n);
    static const chicdiff_nbglm_out none{};
    const chicdiff_nbglm_out &ho = out ? *out : none;
    double *const *hd[] = {&ho.baseMean, &ho.baseVar, &ho.dispGeneEst, &ho.dispFit, &ho.dispMAP, &ho.dispersion, &ho.log2FoldChange,
                           &ho.lfcSE, &ho.stat, &ho.pvalue, &ho.intercept, &ho.interceptSE, &ho.deviance, &ho.maxCooks};
    int32_t *const *hi[] = {&ho.dispGeneIter, &ho.dispIter, &ho.dispOutlier, &ho.betaConv, &ho.betaIter, &ho.allZero, &ho.cooksArgmax};
    int nout = 0;
    for (int k = 0; k < 14; k++) nout += *hd[k] != nullptr;
    for (int k = 0; k < 7; k++) nout += *hi[k] != nullptr;
    const size_t in_bytes = align256(cb) + align256(fb), out_bytes = nd * (size_t)nout;
    int rc = ensure_io(c, in_bytes + out_bytes, in_bytes > out_bytes ? in_bytes : out_bytes);
    if (rc) return rc;
    int32_t *d_counts = (int32_t *)c->io_dev;
    double *d_nf = (double *)(c->io_dev + align256(cb));
    char *po = c->io_dev + in_bytes;
    chicdiff_nbglm_out dout{};
    double **dd[] = {&dout.baseMean, &dout.baseVar, &dout.dispGeneEst, &dout.dispFit, &dout.dispMAP, &dout.dispersion,
                     &dout.log2FoldChange, &dout.lfcSE, &dout.stat, &dout.pvalue, &dout.intercept, &dout.interceptSE,
                     &dout.deviance, &dout.maxCooks};
    int32_t **di[] = {&dout.dispGeneIter, &dout.dispIter, &dout.dispOutlier, &dout.betaConv, &dout.betaIter, &dout.allZero, &dout.cooksArgmax};
    for (int k = 0; k < 14; k++) if (*hd[k]) { *dd[k] = (double *)po; po += nd; }
    for (int k = 0; k < 7; k++) if (*hi[k]) { *di[k] = (int32_t *)po; po += nd; }

    // ---- H2D: slices of the two input matrices, copied and enqueued by a few host threads ----
    const size_t slice = (size_t)8 << 20;
    struct Piece { const char *src; char *pin, *dev; size_t bytes; bool is_counts; };
    std::vector<Piece> pieces;
    for (size_t off = 0; off < cb; off += slice)
        pieces.push_back({(const char *)counts + off, c->io_pin + off, (char *)d_counts + off, cb - off < slice ? cb - off : slice, true});
    for (size_t off = 0; off < fb; off += slice)
        pieces.push_back({(const char *)nf + off, c->io_pin + align256(cb) + off, (char *)d_nf + off, fb - off < slice ? fb - off : slice, false});
    int nthreads = c->opt_host_threads;
    if ((size_t)nthreads > pieces.size()) nthreads = (int)pieces.size();
    if (nthreads < 1) nthreads = 1;
    std::vector<int64_t> bad(nthreads, -1);
    std::vector<int> trc(nthreads, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
            th.emplace_back([&, t]() {
                if (hipSetDevice(c->device) != hipSuccess) { trc[t] = 1; return; }
                for (size_t k = t; k < pieces.size(); k += nthreads) {
                    const Piece &p = pieces[k];
                    if (p.is_counts) {  // copy + NA / negative check in one pass over the slice
                        const int32_t *s = (const int32_t *)p.src;
                        int32_t *d = (int32_t *)p.pin;
                        const size_t m = p.bytes / 4;
                        int32_t acc = 0;
                        for (size_t i = 0; i < m; i++) { d[i] = s[i]; acc |= s[i]; }
                        if (acc < 0 && bad[t] < 0)
                            for (size_t i = 0; i < m; i++)
                                if (s[i] < 0) { bad[t] = (int64_t)(s - counts) + (int64_t)i; break; }
                    } else
                        memcpy(p.pin, p.src, p.bytes);
                    if (hipMemcpyAsync(p.dev, p.pin, p.bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) trc[t] = 1;
                }
            });
        for (auto &w : th) w.join();
    }
    for (int t = 0; t < nthreads; t++) {
        if (bad[t] >= 0) {
            (void)hipStreamSynchronize(c->stream);
            return fail(c, CHICDIFF_E_INVALID, "counts[%lld] is negative or NA_integer_", (long long)bad[t]);
        }
        if (trc[t]) return fail(c, CHICDIFF_E_HIP, "H2D copy failed");
    }
    // ---- D2H: one DMA per column into the staging area, host threads copy out as the columns land.  The ten columns
    //      of the dispersion stage are final once the MAP estimates exist: they go over a second stream, straight from
    //      the workspace, while the Wald stage still runs (fit_dev_impl issues them); the others follow the fit ----
    struct Col { void *host; const void *dev; char *pin; size_t bytes; hipEvent_t ev; bool early; };
    std::vector<Col> cols;
    chicdiff_hip_ctx::EarlyCopy early;
    char *pp = c->io_pin;
    static const int early_of_d[14] = {0, 1, 2, 3, 4, 5, -1, -1, -1, -1, -1, -1, -1, -1}, early_of_i[7] = {6, 7, 8, -1, -1, 9, -1};
    for (int k = 0; k < 14; k++) if (*hd[k]) { cols.push_back({*hd[k], *dd[k], pp, sizeof(double) * (size_t)n, nullptr, early_of_d[k] >= 0}); if (early_of_d[k] >= 0) { early.item[early.n++] = {early_of_d[k], pp, sizeof(double) * (size_t)n}; *dd[k] = nullptr; } pp += nd; }
    for (int k = 0; k < 7; k++) if (*hi[k]) { cols.push_back({*hi[k], *di[k], pp, sizeof(int32_t) * (size_t)n, nullptr, early_of_i[k] >= 0}); if (early_of_i[k] >= 0) { early.item[early.n++] = {early_of_i[k], pp, sizeof(int32_t) * (size_t)n}; *di[k] = nullptr; } pp += nd; }
    hipError_t e = hipSuccess;
    if (early.n > 0) {
        if (!c->copy_stream) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&early.ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&early.done, hipEventDisableTiming);
        if (e != hipSuccess) return fail(c, CHICDIFF_E_HIP, "D2H setup: %s", hipGetErrorString(e));
        c->early = &early;
    }
    rc = chicdiff_hip_nbglm_fit_dev(c, d_counts, d_nf, n, S, group, opts, &dout, scalars);  // ends with a sync of the fit's stream
    c->early = nullptr;
    auto cleanup = [&]() {
        if (early.issued) (void)hipStreamSynchronize(c->copy_stream);  // the staging area must be quiet before it is reused
        if (early.ready) (void)hipEventDestroy(early.ready);
        if (early.done) (void)hipEventDestroy(early.done);
    };
    if (rc) { cleanup(); return rc; }
    for (auto &q : cols) {
        if (q.early) continue;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&q.ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMemcpyAsync(q.pin, q.dev, q.bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(q.ev, c->stream);
    }
    if (e == hipSuccess && !cols.empty()) {
        std::sort(cols.begin(), cols.end(), [](const Col &x, const Col &y) { return x.early > y.early; });  // those already on the host first
        int nt = c->opt_host_threads < (int)cols.size() ? c->opt_host_threads : (int)cols.size();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (size_t k = t; k < cols.size(); k += nt) {
                    if (hipEventSynchronize(cols[k].early ? early.done : cols[k].ev) != hipSuccess) { trc[0] = 1; continue; }
                    memcpy(cols[k].host, cols[k].pin, cols[k].bytes);
                }
            });
        for (auto &w : th) w.join();
    }
    for (auto &q : cols) if (q.ev) (void)hipEventDestroy(q.ev);
    cleanup();
    if (e != hipSuccess || trc[0]) return fail(c, CHICDIFF_E_HIP, "D2H copy: %s", e != hipSuccess ? hipGetErrorString(e) : "event wait failed");
    return CHICDIFF_OK;
}

// size factors -> c->d_sf (device) ; no host synchronisation.  The select's overflow verdict is left in c->d_carry (a device word
// the fit does not clear), from where the fit's last all-reduce — or sf_overflow_consensus — makes it every rank's verdict.
static int size_factors_impl(chicdiff_hip_ctx *c, const int32_t *d_counts, int64_t n, int32_t S) {
    Scope t(c, "size_factors");
    // (the offsets buffer is free until the size factors exist; the same kernel clears the select's overflow flag, c->d_carry — a
    // device word the fit does not clear: the fit's last kernel carries it into the all-reduced verdicts)
    launch_row_ratio(d_counts, n, S, c->d_nf_tmp, c->d_carry, c->stream);
    SelArgs sa{};
    sa.mode = SEL_SIZEFACTOR;
    sa.ncol = S;
    sa.n = n;
    sa.ratio = c->d_nf_tmp;
    sa.S = S;
    sa.sf_out = c->d_sf;  // the finishing step of the select writes the size factors there
    sa.overflow_out = c->d_carry;
    const int rc = run_select(c, sa);
    if (rc) return rc;
    if (c->opt_fault & 4) {  // test hook: this rank's size-factor select "could not fit its candidate list"
        c->opt_fault &= ~4;
        launch_poke(c->d_carry, 1, c->stream);
    }
    return CHICDIFF_OK;
}

// did the size-factor select overflow on ANY rank?  (one 1-double sum-all-reduce when sharded; blocks)
static int sf_overflow_consensus(chicdiff_hip_ctx *c, bool *overflow) {
    double *d_flag = c->d_sf + kMaxS;  // spare doubles behind the size factors
    launch_flag_to_double(c->d_carry, d_flag, c->stream);
    int rc = do_allreduce(c, d_flag, 1);
    if (rc) return rc;
    double h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, d_flag, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *overflow = h > 0;
    return CHICDIFF_OK;
}

int chicdiff_hip_size_factors_dev(chicdiff_hip_ctx *c, const int32_t *d_counts, int64_t n, int32_t S, double *sf_host) {
    if (!c) return CHICDIFF_E_INVALID;
    int rc = (!d_counts || !sf_host || n < 1 || S < 1 || S > kMaxS) ? fail(c, CHICDIFF_E_INVALID, "size_factors: bad arguments") : CHICDIFF_OK;
    timing_reset(c);  // (before the verdicts' collective: it is one of the call's)
    if ((rc = shard_consensus(c, rc, n))) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, n, S))) return rc;
    c->refits = 0;
    const int saved_rounds = c->opt_select_rounds;
    auto body = [&]() -> int {
        for (int attempt = 0; attempt < 2; attempt++) {
            int r = size_factors_impl(c, d_counts, n, S);
            if (r) return r;
            bool overflow = false;
            if ((r = sf_overflow_consensus(c, &overflow))) return r;
            if (!overflow || c->opt_select_rounds) break;
            c->opt_select_rounds = 1;  // a candidate list of the sharded select did not fit on some rank: every histogram round instead
            c->refits++;
        }
        HIPCHK(c, hipMemcpyAsync(c->h_sc, c->w.sc, sizeof(FitScalars), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return CHICDIFF_OK;
    };
    rc = body();
    c->opt_select_rounds = saved_rounds;  // whatever happened above, error returns included
    timing_collect(c);
    if (rc) return rc;
    for (int j = 0; j < S; j++) {
        if (c->h_sc->sel_count[j] <= 0)
            return fail(c, CHICDIFF_E_NUMERIC, "every gene contains at least one zero, cannot compute log geometric means");
        sf_host[j] = c->h_sc->sel_value[2 * j];
    }
    return CHICDIFF_OK;
}

// a5 + a4 + a6 + a7 in one enqueue: estimateSizeFactors -> sc(theta) -> estimateDispersions -> nbinomWaldTest
// (chicdiff.R:1561-1562, 1666-1674) with the size factors and offsets never leaving HBM.
int chicdiff_hip_wald_test_dev(chicdiff_hip_ctx *c, const int32_t *d_counts, const double *d_fullMean, int64_t n, int32_t S,
                               const int32_t *group, double theta, const chicdiff_nbglm_opts *opts,
                               const chicdiff_nbglm_out *d_out, chicdiff_nbglm_scalars *scalars, double *sf_host) {
    if (!c) return CHICDIFF_E_INVALID;
    FitDims d;
    int rc = !d_counts ? fail(c, CHICDIFF_E_INVALID, "counts pointer is NULL") : check_counts_group(c, n, S, group, d);
    if (!rc) rc = check_opts(c, opts);
    timing_reset(c);  // (before the verdicts' collective: it is one of the call's)
    if ((rc = shard_consensus(c, rc, n))) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, n, S))) return rc;
    c->refits = 0;
    const int saved_rounds = c->opt_select_rounds;  // restored below whatever happens (a user's "select_all_rounds" included)
    auto body = [&]() -> int {
        for (int attempt = 0; attempt < 2; attempt++) {
            int r = size_factors_impl(c, d_counts, n, S);
            if (r) return r;
            const int mix = theta == theta;
            // the offsets formed inside the fit's first kernel — where a launch is worth more than the arithmetic: measured (round 5), offsets +
            // prep against the fused prep: 18.5 + 28 -> 47 us at 250 k x 8 (and one launch, ~5 us + its gap, less), 30 + 40 -> 76 at 500 k,
            // 67 + 130 -> 291 at 2 M (the fused kernel carries 2 S logarithms per row at the four waves per SIMD its LDS tiles allow)
            if (d_fullMean && S <= 16 && c->opt_fuse_offsets && (c->opt_fuse_offsets == 2 || n <= kFuseOffsetsMaxRows)) {
                c->fuse.fm = d_fullMean;
                c->fuse.sf = c->d_sf;
                c->fuse.theta = mix ? theta : 0.0;
                c->fuse.mix = mix;
            } else {
                Scope t(c, "offsets");
                launch_offsets(d_fullMean, c->d_sf, n, S, mix ? theta : 0.0, mix, c->d_nf_tmp, c->stream);
            }
            r = fit_dev_impl(c, d_counts, c->d_nf_tmp, d, make_opts(c, opts, S), d_out, scalars);  // ends with a stream sync; the size factors come back with its scalars
            c->fuse = FusedOffsets();
            // the sharded size-factor select ran without a host look at its candidate lists: one that did not fit on ANY rank (massive
            // ties) shows in the fit's last all-reduce, on every rank alike, and the call is repeated with every histogram round
            if (r || !c->sf_overflow_seen || c->opt_select_rounds) return r;
            c->opt_select_rounds = 1;
            c->refits++;
        }
        return CHICDIFF_OK;
    };
    rc = body();
    c->opt_select_rounds = saved_rounds;
    timing_collect(c);
    if (rc) return rc;
    for (int j = 0; j < S; j++) {
        if (!(c->h_sc->final_sf[j] == c->h_sc->final_sf[j]))
            return fail(c, CHICDIFF_E_NUMERIC, "every gene contains at least one zero, cannot compute log geometric means");
        if (sf_host) sf_host[j] = c->h_sc->final_sf[j];
    }
    return CHICDIFF_OK;
}

int chicdiff_hip_offsets_dev(chicdiff_hip_ctx *c, const double *d_fullMean, const double *sf_host, int64_t n, int32_t S,
                             double theta, double *d_nf_out) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!sf_host || !d_nf_out || n < 1 || S < 1 || S > kMaxS) return fail(c, CHICDIFF_E_INVALID, "offsets: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    HIPCHK(c, hipMemcpyAsync(c->d_sf, sf_host, sizeof(double) * S, hipMemcpyHostToDevice, c->stream));
    const int mix = theta == theta;
    {
        Scope t(c, "offsets");
        launch_offsets(d_fullMean, c->d_sf, n, S, mix ? theta : 0.0, mix, d_nf_out, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // sf_host may be a temporary
    timing_collect(c);
    return CHICDIFF_OK;
}

int chicdiff_hip_window_sums_dev(chicdiff_hip_ctx *c, const int32_t *d_fragN, const double *d_fragFM, int64_t nfrag,
                                 int32_t S, const int64_t *d_region_ptr, int64_t n, int32_t *d_N, double *d_FM) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_region_ptr || n < 1 || S < 1 || (d_fragN && !d_N) || (d_fragFM && !d_FM))
        return fail(c, CHICDIFF_E_INVALID, "window_sums: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    {
        Scope t(c, "window_sums");
        launch_window_sums(d_fragN, d_fragFM, nfrag, S, d_region_ptr, n, d_N, d_FM, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

int chicdiff_hip_count_join_dev(chicdiff_hip_ctx *c, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                const int64_t *d_keys, const int32_t *d_vals, int64_t nkeys, int32_t *d_out) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_ru_bait || !d_ru_oe || !d_out || nru < 0 || nkeys < 0 || (nkeys > 0 && (!d_keys || !d_vals)))
        return fail(c, CHICDIFF_E_INVALID, "count_join: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (nru > 0) {
        if (int rc = ensure_aux(c, count_join_scratch_bytes(nkeys))) return rc;
        Scope t(c, "count_join");
        launch_count_join(d_ru_bait, d_ru_oe, nru, d_keys, d_vals, nkeys, d_out, c->aux, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

int chicdiff_hip_theta_grid_dev(chicdiff_hip_ctx *c, const int32_t *d_counts, const double *d_fullMean, const double *sf_host,
                                int64_t n, int32_t S, const double *thetas, int32_t ntheta, const chicdiff_nbglm_opts *opts,
                                double *deviances_host) {
    if (!c) return CHICDIFF_E_INVALID;
    FitDims d;
    int rc = (!d_counts || !d_fullMean || !sf_host || !thetas || !deviances_host || ntheta < 1)
                 ? fail(c, CHICDIFF_E_INVALID, "theta_grid: bad arguments")
                 : check_counts_group(c, n, S, nullptr, d);  // design ~ 1  ("sic!", chicdiff.R:1629-1631)
    if (!rc) rc = check_opts(c, opts);
    if ((rc = shard_consensus(c, rc, n))) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    const Opts o = make_opts(c, opts, S);
    // Single rank: the |Grid| fits are independent, so they run on child contexts (own stream + workspace), up to
    // opt_grid_lanes at once, each driven by its own host thread: one fit's straggler tail and its latency-bound
    // global steps (trend barriers, selects) overlap with the other fits' line searches.  Sharded: one after the other
    // (every rank must issue its collectives in the same order).
    const size_t ws_per_lane = sizeof(double) * (size_t)n * (26 + S) + (size_t)row_stride(S) * (size_t)n + (64u << 20);
    int lanes = c->allreduce ? 1 : (ntheta < c->opt_grid_lanes ? ntheta : c->opt_grid_lanes);
    bool squeezed = false;  // the lanes' workspaces do not fit: their memory is given back before the thetas are fitted one after the other
    if (lanes > 1) {
        // the lanes' workspaces must fit what the device has free NOW (a shared GPU, the caller's own tensors, a smaller
        // part); lanes that already hold a workspace of this size cost nothing more
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        int have = 0;
        for (chicdiff_hip_ctx *l : c->lanes) have += (l->cap_n >= n && l->cap_S >= S) ? 1 : 0;
        while (lanes > 1 && lanes > have && (double)ws_per_lane * (lanes - have) > 0.9 * (double)free_b) {
            lanes--;
            squeezed = lanes <= 1;
        }
    }
    while (lanes > 1 && (int)c->lanes.size() < lanes) {
        chicdiff_hip_ctx *l = nullptr;
        if (chicdiff_hip_create(&l, c->device)) { lanes = 1; squeezed = true; break; }
        c->lanes.push_back(l);
    }
    for (int k = 0; k < lanes && lanes > 1; k++)
        if (ensure_workspace(c->lanes[k], n, S)) {  // no room after all
            snprintf(c->err, sizeof c->err, "theta_grid: concurrent fits given up (%s)", c->lanes[k]->err);  // kept for the caller's log; the call goes on
            lanes = 1;
            squeezed = true;
        }
    if (lanes <= 1) {
        // (a caller who simply asked for one lane — ntheta == 1, theta_grid_concurrency 1 — keeps the cached lanes: re-creating
        // them costs streams, GBs of hipMalloc and event-pool warm-up on the next multi-theta call)
        if (squeezed && !c->lanes.empty()) {
            for (chicdiff_hip_ctx *l : c->lanes) chicdiff_hip_destroy(l);
            c->lanes.clear();
        }
        rc = ensure_workspace(c, n, S);
        if (rc == CHICDIFF_E_NOMEM && !c->lanes.empty()) {  // the cached lanes hold whole workspaces of their own: give them up and try once more
            for (chicdiff_hip_ctx *l : c->lanes) chicdiff_hip_destroy(l);
            c->lanes.clear();
            rc = ensure_workspace(c, n, S);
        }
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_carry, 0, sizeof(int32_t), c->stream));  // no size-factor select belongs to this call
        HIPCHK(c, hipMemcpyAsync(c->d_sf, sf_host, sizeof(double) * S, hipMemcpyHostToDevice, c->stream));
        for (int t = 0; t < ntheta; t++) {
            if (S <= 16 && c->opt_fuse_offsets && (c->opt_fuse_offsets == 2 || n <= kFuseOffsetsMaxRows)) {
                c->fuse.fm = d_fullMean;
                c->fuse.sf = c->d_sf;
                c->fuse.theta = thetas[t];
                c->fuse.mix = 1;
            } else {
                Scope s(c, "offsets");
                launch_offsets(d_fullMean, c->d_sf, n, S, thetas[t], 1, c->d_nf_tmp, c->stream);
            }
            chicdiff_nbglm_scalars sc;
            rc = fit_dev_impl(c, d_counts, c->d_nf_tmp, d, o, nullptr, &sc);
            c->fuse = FusedOffsets();
            if (rc) return rc;
            deviances_host[t] = sc.sumDeviance;
        }
        timing_collect(c);
        return CHICDIFF_OK;
    }
    hipEvent_t ready;  // the inputs are produced on the caller's stream
    HIPCHK(c, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(ready, c->stream));
    std::vector<int> lane_rc(lanes, CHICDIFF_OK);
    std::vector<std::thread> workers;
    for (int k = 0; k < lanes; k++) {
        chicdiff_hip_ctx *l = c->lanes[k];
        l->opt_spread = c->opt_spread;
        l->opt_min_waves = c->opt_min_waves;
        l->opt_prio = c->opt_prio;
        l->opt_schedule = c->opt_schedule ? 2 : 0;  // concurrent fits: class order through the queue, nothing dealt out statically
        l->opt_deal = c->opt_deal;
        l->opt_chunk = c->opt_chunk;
        l->opt_no_local_substitute = c->opt_no_local_substitute;
        l->opt_trend_gather = c->opt_trend_gather;
        l->opt_select_rounds = c->opt_select_rounds;
        l->opt_trend_multilaunch = c->opt_trend_multilaunch;
        l->opt_mad_in_kernel = c->opt_mad_in_kernel;
        l->opt_trend_blocks = c->opt_trend_blocks;
        // a grid barrier needs its workgroups co-resident: not guaranteed beside other fits.  (Measured, round 4: the lanes' trend as
        // the single-launch kernel capped at 16 / 32 / 64 workgroups — 2 M x 8, five thetas: 2 lanes 16.5 -> 18.2 ms, 3 lanes 17.2 ->
        // 17.9, 5 lanes 16.3 ms -> 5-13 SECONDS: the kernel's workgroups wait for LDS that the other lanes' line-search workgroups
        // hold until their launch ends, and the barrier spins meanwhile.  With the uncapped kernel and a host mutex so that only one
        // lane's trend kernel is in flight at a time: 5 lanes 16.2 -> 18.1 ms.  A kernel trace of the grid (tools/theta_grid_trace.py)
        // shows why nothing small overlaps: while a line-search launch is resident, the other streams' small kernels do not start at
        // all — they run in the gaps between the big launches.  One launch per IRLS pass it stays.)
        l->no_persistent_trend = true;
        workers.emplace_back([=, &lane_rc]() {
            int r = CHICDIFF_OK;
            if (hipSetDevice(l->device) != hipSuccess || hipStreamWaitEvent(l->stream, ready, 0) != hipSuccess) r = CHICDIFF_E_HIP;
            if (!r) r = ensure_workspace(l, n, S);
            if (!r && hipMemcpyAsync(l->d_sf, sf_host, sizeof(double) * S, hipMemcpyHostToDevice, l->stream) != hipSuccess) r = CHICDIFF_E_HIP;
            for (int t = k; t < ntheta && !r; t += lanes) {
                if (S <= 16 && c->opt_fuse_offsets && (c->opt_fuse_offsets == 2 || n <= kFuseOffsetsMaxRows)) {
                    l->fuse.fm = d_fullMean;
                    l->fuse.sf = l->d_sf;
                    l->fuse.theta = thetas[t];
                    l->fuse.mix = 1;
                } else {
                    launch_offsets(d_fullMean, l->d_sf, n, S, thetas[t], 1, l->d_nf_tmp, l->stream);
                }
                chicdiff_nbglm_scalars sc;
                r = fit_dev_impl(l, d_counts, l->d_nf_tmp, d, o, nullptr, &sc);
                l->fuse = FusedOffsets();
                if (!r) deviances_host[t] = sc.sumDeviance;
            }
            lane_rc[k] = r;
        });
    }
    for (auto &w : workers) w.join();
    (void)hipEventDestroy(ready);
    for (int k = 0; k < lanes; k++)
        if (lane_rc[k]) return fail(c, lane_rc[k], "theta_grid: %s", c->lanes[k]->err[0] ? c->lanes[k]->err : "a concurrent fit failed");
    timing_collect(c);
    return CHICDIFF_OK;
}

}  // extern "C"

// small extra export used by the parity tests: p = 2*pnorm(-|stat|) on device (Cody, as R)
extern "C" int chicdiff_hip_wald_pvalues_dev(chicdiff_hip_ctx *c, const double *d_stat, int64_t n, double *d_p) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_stat || !d_p || n < 0) return fail(c, CHICDIFF_E_INVALID, "wald_pvalues: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0) launch_pvalues(d_stat, n, d_p, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CHICDIFF_OK;
}

// device-math self test: op 0 flog, 1 tlog (table), 2 rcp, 3 lgamma, 4 digamma, 5 2*pnorm(-|x|), 6 / 7 raw v_rcp_f64 (+ one Newton step), 8 texp (table)
extern "C" int chicdiff_hip_selftest_math_dev(chicdiff_hip_ctx *c, int32_t op, const double *d_x, int64_t n, double *d_out) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_x || !d_out || n < 0 || op < 0 || op > 8) return fail(c, CHICDIFF_E_INVALID, "selftest_math: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0) launch_math_selftest(op, d_x, n, d_out, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CHICDIFF_OK;
}

// host-side self tests of the simulation-matched prior variance (no device involved)
extern "C" int chicdiff_hip_selftest_r_random(int32_t kind, uint32_t seed, double a, double b, int64_t n, double *out) {
    if (!out || n < 0 || kind < 0 || kind > 3) return CHICDIFF_E_INVALID;
    if (kind == 3 && !(a > 0 && b > 0)) return CHICDIFF_E_INVALID;
    RStream r(seed);
    for (int64_t i = 0; i < n; i++) out[i] = kind == 0 ? r.unif() : kind == 1 ? r.norm() : kind == 2 ? r.expo() : r.gamma(a, b);
    return CHICDIFF_OK;
}
extern "C" int chicdiff_hip_selftest_prior_mc(int32_t df, const double *hist40, double *dens_out, double *prior_var_out) {
    if (df < 1 || df > 3) return CHICDIFF_E_INVALID;
    const PmcTable &t = pmc_table(df);
    if (dens_out) memcpy(dens_out, t.dens, sizeof t.dens);
    if (prior_var_out) {
        if (!hist40) return CHICDIFF_E_INVALID;
        *prior_var_out = pmc_prior_var(hist40, t);
    }
    return CHICDIFF_OK;
}

// a3 — per-fragment background (Bmean, Tmean, FullMean) for every RU row and replicate
extern "C" int chicdiff_hip_fragment_background_dev(chicdiff_hip_ctx *c, const int32_t *d_bait, const int32_t *d_oe, int64_t nru,
                                                    int32_t id_min, int32_t nid, const int64_t *d_midsum, int32_t S,
                                                    const double *d_sj, const double *d_si, const int32_t *d_tblb,
                                                    const int32_t *d_tlb, const double *d_T, int32_t ntblb, int32_t ntlb,
                                                    const double *distfun_host, double *d_bmean, double *d_tmean,
                                                    double *d_fullmean) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_bait || !d_oe || !d_midsum || !d_sj || !d_si || !d_tblb || !d_tlb || !d_T || !distfun_host || nru < 0 || nid < 1 ||
        S < 1 || S > kMaxS || ntblb < 1 || ntlb < 1)
        return fail(c, CHICDIFF_E_INVALID, "fragment_background: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_aux(c, sizeof(double) * 10 * kMaxS);  // (the context's scratch: no allocation per call once warm)
    if (rc) return rc;
    double *d_df = reinterpret_cast<double *>(c->aux);
    hipError_t e = hipMemcpyAsync(d_df, distfun_host, sizeof(double) * 10 * S, hipMemcpyHostToDevice, c->stream);
    timing_reset(c);
    if (e == hipSuccess && nru > 0) {
        Scope t(c, "fragment_background");
        launch_fragment_background(d_bait, d_oe, nru, id_min, nid, d_midsum, S, d_sj, d_si, d_tblb, d_tlb, d_T, ntblb, ntlb, d_df,
                                   d_bmean, d_tmean, d_fullmean, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    timing_collect(c);
    if (e != hipSuccess) return fail(c, CHICDIFF_E_HIP, "fragment_background: %s", hipGetErrorString(e));
    return CHICDIFF_OK;
}

// ---- f1 / f3 / f4 (post_kernels.hip) ------------------------------------------------------------------------
extern "C" int chicdiff_hip_bh_adjust_dev(chicdiff_hip_ctx *c, const double *d_p, int64_t n, double *d_padj) {
    if (!c) return CHICDIFF_E_INVALID;
    if (n < 0 || n >= (1ll << 32) || (n > 0 && (!d_p || !d_padj))) return fail(c, CHICDIFF_E_INVALID, "bh_adjust: bad arguments");
    if (n == 0) return CHICDIFF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_aux(c, bh_workspace_bytes(n));
    if (rc) return rc;
    timing_reset(c);
    {
        Scope t(c, "bh_adjust");
        if (launch_bh_adjust(d_p, n, d_padj, c->aux, c->stream)) return fail(c, CHICDIFF_E_HIP, "bh_adjust: sort/scan failed");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

extern "C" int chicdiff_hip_ihw_apply_dev(chicdiff_hip_ctx *c, const double *d_avDist, const double *d_pvalue, int64_t n,
                                          const double *breaks_host, const double *avWeights_host, int32_t ngroups,
                                          int32_t *d_group, double *d_weight, double *d_wp, double *d_wpadj) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_avDist || !d_pvalue || !breaks_host || !avWeights_host || !d_wpadj || n < 1 || n >= (1ll << 32) || ngroups < 1 || ngroups > 256)
        return fail(c, CHICDIFF_E_INVALID, "ihw_apply: bad arguments");
    for (int k = 0; k < ngroups; k++)
        if (!(breaks_host[k] < breaks_host[k + 1])) return fail(c, CHICDIFF_E_INVALID, "ihw_apply: breaks must be strictly ascending ('breaks' are not unique)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bhb = bh_workspace_bytes(n), ihwb = (ihw_workspace_bytes() + 255) / 256 * 256, col = (sizeof(double) * (size_t)n + 255) / 256 * 256;
    int rc = ensure_aux(c, bhb + ihwb + 2 * col);
    if (rc) return rc;
    double *partials = (double *)(c->aux + bhb);
    double *weight = d_weight ? d_weight : (double *)(c->aux + bhb + ihwb);
    double *wp = d_wp ? d_wp : (double *)(c->aux + bhb + ihwb + col);
    timing_reset(c);
    {
        Scope t(c, "ihw_apply");
        launch_ihw_apply(d_avDist, d_pvalue, n, breaks_host, avWeights_host, ngroups, d_group, weight, wp, partials, c->stream);
        if (launch_bh_adjust(wp, n, d_wpadj, c->aux, c->stream)) return fail(c, CHICDIFF_E_HIP, "ihw_apply: sort/scan failed");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

extern "C" int chicdiff_hip_region_universe_count_dev(chicdiff_hip_ctx *c, const int32_t *d_bait, const int32_t *d_oe, int64_t n,
                                                      int32_t RUexpand, const int32_t *d_chr_of, int32_t maxfrag,
                                                      int64_t *d_region_ptr, int32_t *d_minOE, int32_t *d_maxOE,
                                                      int64_t *total_host) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_bait || !d_oe || !d_chr_of || !d_region_ptr || !total_host || n < 1 || RUexpand < 0 || RUexpand > (1 << 20) || maxfrag < 1)
        return fail(c, CHICDIFF_E_INVALID, "region_universe: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t scan = ru_scan_bytes(n);
    int rc = ensure_aux(c, 256 + scan);
    if (rc) return rc;
    int *bad = (int *)c->aux;
    timing_reset(c);
    {
        Scope t(c, "region_universe");
        if (launch_ru_count(d_bait, d_oe, n, RUexpand, d_chr_of, maxfrag, d_region_ptr, d_minOE, d_maxOE, bad, c->aux + 256, scan, c->stream))
            return fail(c, CHICDIFF_E_HIP, "region_universe: scan failed");
    }
    int h_bad = 0;
    HIPCHK(c, hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(total_host, d_region_ptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    if (h_bad) return fail(c, CHICDIFF_E_INVALID, "region_universe: Invalid parameters (a peak with baitID == oeID)");
    return CHICDIFF_OK;
}

extern "C" int chicdiff_hip_region_universe_fill_dev(chicdiff_hip_ctx *c, const int32_t *d_bait, const int32_t *d_oe, int64_t n,
                                                     int32_t RUexpand, const int32_t *d_chr_of, int32_t maxfrag,
                                                     const int64_t *d_region_ptr, int32_t *d_ru_bait, int32_t *d_ru_region,
                                                     int32_t *d_ru_oe) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_bait || !d_oe || !d_chr_of || !d_region_ptr || !d_ru_bait || !d_ru_region || !d_ru_oe || n < 1 || RUexpand < 0 || maxfrag < 1)
        return fail(c, CHICDIFF_E_INVALID, "region_universe: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    {
        Scope t(c, "region_universe");
        launch_ru_fill(d_bait, d_oe, n, RUexpand, d_chr_of, maxfrag, d_region_ptr, d_ru_bait, d_ru_region, d_ru_oe, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

// both steps in one call (round 5): the caller gives room for the upper bound n max(2 RUexpand + 1, 2) rows, so that nothing on the host
// stands between the scan and the fill — one synchronisation and one read-back instead of two of each
extern "C" int chicdiff_hip_region_universe_dev(chicdiff_hip_ctx *c, const int32_t *d_bait, const int32_t *d_oe, int64_t n, int32_t RUexpand,
                                                const int32_t *d_chr_of, int32_t maxfrag, int64_t *d_region_ptr, int32_t *d_minOE,
                                                int32_t *d_maxOE, int32_t *d_ru_bait, int32_t *d_ru_region, int32_t *d_ru_oe,
                                                int64_t capacity, int64_t *total_host) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_bait || !d_oe || !d_chr_of || !d_region_ptr || !d_ru_bait || !d_ru_region || !d_ru_oe || !total_host || n < 1 || RUexpand < 0 ||
        RUexpand > (1 << 20) || maxfrag < 1)
        return fail(c, CHICDIFF_E_INVALID, "region_universe: bad arguments");
    // rows per peak: (oe - s):(oe + s) = 2 s + 1 — or, for RUexpand = 0 and a peak right beside its bait, R's DESCENDING (bait + 2):(oe + 0): two
    const int64_t per_peak = RUexpand > 0 ? 2 * (int64_t)RUexpand + 1 : 2;
    if (capacity < n * per_peak)
        return fail(c, CHICDIFF_E_INVALID, "region_universe: room for n max(2 RUexpand + 1, 2) = %lld rows needed, %lld given",
                    (long long)(n * per_peak), (long long)capacity);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t scan = (ru_scan_bytes(n) + 255) / 256 * 256;
    int rc = ensure_aux(c, 256 + scan + sizeof(unsigned int) * (size_t)n);
    if (rc) return rc;
    int *bad = (int *)c->aux;
    unsigned int *masks = reinterpret_cast<unsigned int *>(c->aux + 256 + scan);  // which candidates each region keeps: counted once, read by the fill
    timing_reset(c);
    {
        Scope t(c, "region_universe");
        if (launch_ru_count(d_bait, d_oe, n, RUexpand, d_chr_of, maxfrag, d_region_ptr, d_minOE, d_maxOE, bad, c->aux + 256, scan, c->stream, masks))
            return fail(c, CHICDIFF_E_HIP, "region_universe: scan failed");
        launch_ru_fill(d_bait, d_oe, n, RUexpand, d_chr_of, maxfrag, d_region_ptr, d_ru_bait, d_ru_region, d_ru_oe, c->stream, masks);
    }
    int h_bad = 0;
    HIPCHK(c, hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(total_host, d_region_ptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    if (h_bad) return fail(c, CHICDIFF_E_INVALID, "region_universe: Invalid parameters (a peak with baitID == oeID)");
    return CHICDIFF_OK;
}

extern "C" int chicdiff_hip_count_table_dev(chicdiff_hip_ctx *c, const int32_t *d_bait, const int32_t *d_oe, const int32_t *d_N,
                                            int64_t nrows, const uint8_t *d_bait_in_RU, int32_t max_id, int64_t *d_keys,
                                            int32_t *d_vals, int64_t *nkeys_host) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_bait || !d_oe || !d_N || !d_keys || !d_vals || !nkeys_host || nrows < 1 || nrows >= (1ll << 32) || (d_bait_in_RU && max_id < 0))
        return fail(c, CHICDIFF_E_INVALID, "count_table: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_aux(c, ct_workspace_bytes(nrows));
    if (rc) return rc;
    timing_reset(c);
    {
        Scope t(c, "count_table");
        if (launch_count_table(d_bait, d_oe, d_N, nrows, d_bait_in_RU, max_id, d_keys, d_vals, c->aux, c->stream))
            return fail(c, CHICDIFF_E_HIP, "count_table: sort failed");
    }
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, c->aux, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    *nkeys_host = (int64_t)h;
    return CHICDIFF_OK;
}

// IHWcorrection's covariate (chicdiff.R:1965-1967): per-region mean of distSign over the CSR of RU rows
extern "C" int chicdiff_hip_region_avdist_dev(chicdiff_hip_ctx *c, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                              const int64_t *d_region_ptr, int64_t n, int32_t id_min, int32_t nid,
                                              const int64_t *d_midsum, const int32_t *d_chr, double *d_avDist) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_region_ptr || !d_midsum || !d_avDist || n < 1 || nru < 0 || nid < 1 || (nru > 0 && (!d_ru_bait || !d_ru_oe)))
        return fail(c, CHICDIFF_E_INVALID, "region_avdist: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    {
        Scope t(c, "region_avdist");
        launch_region_avdist(d_ru_bait, d_ru_oe, d_region_ptr, n, id_min, nid, d_midsum, d_chr, d_avDist, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

// a1 for all replicates in one pass (chicdiff.R:843-858, the loop over the replicates): out = S x nru, column s = merge(RU, table s, all.x = TRUE)
extern "C" int chicdiff_hip_count_join_multi_dev(chicdiff_hip_ctx *c, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                                 int32_t S, const int64_t *const *d_keys, const int32_t *const *d_vals,
                                                 const int64_t *nkeys, int32_t *d_out) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_ru_bait || !d_ru_oe || !d_out || !d_keys || !d_vals || !nkeys || nru < 0 || S < 1 || S > kMaxS)
        return fail(c, CHICDIFF_E_INVALID, "count_join_multi: bad arguments");
    for (int s = 0; s < S; s++)
        if (nkeys[s] < 0 || (nkeys[s] > 0 && (!d_keys[s] || !d_vals[s])))
            return fail(c, CHICDIFF_E_INVALID, "count_join_multi: bad table %d", s);
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (nru > 0) {
        if (int rc = ensure_aux(c, count_join_multi_scratch_bytes(S, nkeys))) return rc;
        Scope t(c, "count_join_multi");
        launch_count_join_multi(d_ru_bait, d_ru_oe, nru, S, d_keys, d_vals, nkeys, d_out, c->aux, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

// a1 without chinput files (chicdiff.R:774-807): N of every RU row from the replicates' Chicago tables, inner-merged
extern "C" int chicdiff_hip_count_join_inner_dev(chicdiff_hip_ctx *c, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                                 int32_t S, const int64_t *const *d_keys, const int32_t *const *d_vals,
                                                 const int64_t *nkeys, int32_t *d_out) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_ru_bait || !d_ru_oe || !d_out || !d_keys || !d_vals || !nkeys || nru < 0 || S < 1 || S > kMaxS)
        return fail(c, CHICDIFF_E_INVALID, "count_join_inner: bad arguments");
    for (int s = 0; s < S; s++)
        if (nkeys[s] < 0 || (nkeys[s] > 0 && (!d_keys[s] || !d_vals[s])))
            return fail(c, CHICDIFF_E_INVALID, "count_join_inner: bad table %d", s);
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (nru > 0) {
        Scope t(c, "count_join_inner");
        launch_count_join_inner(d_ru_bait, d_ru_oe, nru, S, d_keys, d_vals, nkeys, d_out, c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}

// host-side self test of the .chinput parser (no device, no context): the three columns of up to `cap` rows
extern "C" int chicdiff_hip_selftest_chinput(const char *path, int32_t nthreads, int64_t cap, int32_t *bait, int32_t *oe, int32_t *N,
                                             int64_t *nrows, char *err, int32_t errcap) {
    if (!path || !nrows) return CHICDIFF_E_INVALID;
    ChinputCols cols;
    const int64_t n = chinput_parse(path, nthreads, cols);
    if (n < 0) {
        if (err && errcap > 0) snprintf(err, (size_t)errcap, "%s", cols.error.c_str());
        return CHICDIFF_E_INVALID;
    }
    *nrows = n;
    const size_t m = (size_t)(n < cap ? n : cap);
    if (bait) memcpy(bait, cols.bait.data(), 4 * m);
    if (oe) memcpy(oe, cols.oe.data(), 4 * m);
    if (N) memcpy(N, cols.N.data(), 4 * m);
    return CHICDIFF_OK;
}

// f2 — the text side: fread(chinput) + column pick (chicdiff.R:828, :849) with host threads (chinput.hip), then the key
// table of the count join on the device
extern "C" int chicdiff_hip_chinput_read(chicdiff_hip_ctx *c, const char *path, int32_t nthreads, int64_t *nrows_host) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!path || !nrows_host) return fail(c, CHICDIFF_E_INVALID, "chinput_read: bad arguments");
    if (!c->chin) c->chin = new ChinputCols();
    const int64_t n = chinput_parse(path, nthreads > 0 ? nthreads : c->opt_host_threads, *c->chin);
    if (n < 0) return fail(c, CHICDIFF_E_INVALID, "chinput_read: %s", c->chin->error.c_str());
    *nrows_host = n;
    return CHICDIFF_OK;
}
extern "C" int chicdiff_hip_chinput_table_dev(chicdiff_hip_ctx *c, const uint8_t *d_bait_in_RU, int32_t max_id, int64_t *d_keys,
                                              int32_t *d_vals, int64_t *nkeys_host) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!c->chin) return fail(c, CHICDIFF_E_INVALID, "chinput_table: nothing read (call chicdiff_hip_chinput_read first)");
    if (!nkeys_host) return fail(c, CHICDIFF_E_INVALID, "chinput_table: bad arguments");
    if (c->chin->bait.empty()) {  // a header without data rows: fread gives an empty table, merge(all.x = TRUE) then N = 0 for every RU row
        *nkeys_host = 0;
        return CHICDIFF_OK;
    }
    const size_t n = c->chin->bait.size(), col = align256(sizeof(int32_t) * n);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_io(c, 3 * col, 0);
    if (rc) return rc;
    int32_t *d_b = (int32_t *)c->io_dev, *d_o = (int32_t *)(c->io_dev + col), *d_n = (int32_t *)(c->io_dev + 2 * col);
    HIPCHK(c, hipMemcpyAsync(d_b, c->chin->bait.data(), 4 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_o, c->chin->oe.data(), 4 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_n, c->chin->N.data(), 4 * n, hipMemcpyHostToDevice, c->stream));
    return chicdiff_hip_count_table_dev(c, d_b, d_o, d_n, (int64_t)n, d_bait_in_RU, max_id, d_keys, d_vals, nkeys_host);
}

// ---- device memory for hosts without their own GPU arrays (the R shim) ---------------------------------------
extern "C" int chicdiff_hip_malloc(chicdiff_hip_ctx *c, uint64_t bytes, void **d_ptr) {
    if (!c || !d_ptr) return CHICDIFF_E_INVALID;
    *d_ptr = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? (size_t)bytes : 1);
    if (e != hipSuccess) return fail(c, CHICDIFF_E_NOMEM, "device allocation of %llu bytes: %s", (unsigned long long)bytes, hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(c->user_mu);
    c->user_allocs.insert(*d_ptr);
    return CHICDIFF_OK;
}
extern "C" int chicdiff_hip_free(chicdiff_hip_ctx *c, void *d_ptr) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_ptr) return CHICDIFF_OK;
    {
        std::lock_guard<std::mutex> lock(c->user_mu);
        if (!c->user_allocs.erase(d_ptr)) return fail(c, CHICDIFF_E_INVALID, "chicdiff_hip_free: not an allocation of this context (or freed already)");
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d_ptr));
    return CHICDIFF_OK;
}
extern "C" int64_t chicdiff_hip_outstanding_allocations(chicdiff_hip_ctx *c) {
    if (!c) return -1;
    std::lock_guard<std::mutex> lock(c->user_mu);
    return (int64_t)c->user_allocs.size();
}
extern "C" int chicdiff_hip_memcpy_h2d(chicdiff_hip_ctx *c, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!c || (bytes && (!d_dst || !h_src))) return c ? fail(c, CHICDIFF_E_INVALID, "memcpy_h2d: NULL pointer") : CHICDIFF_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CHICDIFF_OK;
}
extern "C" int chicdiff_hip_memcpy_d2h(chicdiff_hip_ctx *c, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!c || (bytes && (!h_dst || !d_src))) return c ? fail(c, CHICDIFF_E_INVALID, "memcpy_d2h: NULL pointer") : CHICDIFF_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CHICDIFF_OK;
}

// ---- a9: results() on device -----------------------------------------------------------------------------------
extern "C" int chicdiff_hip_cooks_filter_dev(chicdiff_hip_ctx *c, const int32_t *d_counts, int64_t n, int32_t S, const int32_t *group,
                                             const double *d_maxCooks, const int32_t *d_cooksArgmax, double cutoff, double *d_pvalue,
                                             int64_t *n_outliers_host) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_counts || !d_maxCooks || !d_cooksArgmax || !d_pvalue || !n_outliers_host) return fail(c, CHICDIFF_E_INVALID, "cooks_filter: NULL pointer");
    FitDims d;
    int rc = check_counts_group(c, n, S, group, d);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure_aux(c, 256))) return rc;
    timing_reset(c);
    {
        Scope t(c, "cooks_filter");
        launch_cooks_filter(d_counts, n, S, d.p, d_maxCooks, d_cooksArgmax, cutoff, d_pvalue, (unsigned long long *)c->aux, c->stream);
    }
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, c->aux, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    *n_outliers_host = (int64_t)h;
    return CHICDIFF_OK;
}

extern "C" int chicdiff_hip_independent_filtering_dev(chicdiff_hip_ctx *c, const double *d_baseMean, const double *d_pvalue, int64_t n,
                                                      double alpha, double *d_padj, chicdiff_results_info *info) {
    if (!c) return CHICDIFF_E_INVALID;
    if (!d_baseMean || !d_pvalue || !d_padj || !info || n < 1 || n >= (1ll << 32) || !(alpha > 0 && alpha < 1))
        return fail(c, CHICDIFF_E_INVALID, "independent_filtering: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_aux(c, if_workspace_bytes(n));
    if (rc) return rc;
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (!c->ev_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    timing_reset(c);
    {
        Scope t(c, "independent_filtering");
        if (run_independent_filtering(d_baseMean, d_pvalue, n, alpha, d_padj, c->aux, c->stream, c->copy_stream, c->ev_fork, c->ev_join, info))
            return fail(c, CHICDIFF_E_HIP, "independent_filtering: %s", hipGetErrorString(hipGetLastError()));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return CHICDIFF_OK;
}
