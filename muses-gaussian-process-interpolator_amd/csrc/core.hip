// Context, device matrices and error plumbing of libmgpi_hip.so.
#include "mgpi_internal.h"
#include <stdarg.h>
#include <math.h>
#include <new>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void mgpi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mgpi_last_error(void) { return g_err; }

extern "C" int mgpi_device_count(int* count) {
    MGPI_REQUIRE(count != nullptr, "mgpi_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return 0;
}

extern "C" int mgpi_ctx_create(int device_id, int nb, mgpi_ctx** out) {
    MGPI_REQUIRE(out != nullptr, "mgpi_ctx_create: null out pointer");
    *out = nullptr;
    if (nb == 0) nb = 512;
    MGPI_REQUIRE(nb >= MGPI_TILE && nb % MGPI_TILE == 0 && nb <= 4096,
                 "mgpi_ctx_create: nb=%d must be a multiple of %d in [%d, 4096]", nb, MGPI_TILE, MGPI_TILE);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        mgpi_set_error("mgpi_ctx_create: no HIP device available (%s); this library has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return MGPI_ERR_HIP;
    }
    MGPI_REQUIRE(device_id >= 0 && device_id < ndev, "mgpi_ctx_create: device %d out of range [0,%d)", device_id, ndev);
    MGPI_HIP_TRY(hipSetDevice(device_id));
    mgpi_ctx* c = new (std::nothrow) mgpi_ctx();
    if (!c) { mgpi_set_error("mgpi_ctx_create: out of host memory"); return MGPI_ERR_ALLOC; }
    memset((void*)c, 0, sizeof(*c));
    c->prof = new mgpi_prof_slot[MGPI_PROF_SLOTS]();
    c->device = device_id;
    c->nb = nb;
    MGPI_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        int lo = 0, hi = 0;   // numerically lowest value = highest priority
        MGPI_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        MGPI_HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi));
    }
    MGPI_HIP_TRY(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    MGPI_HIP_TRY(hipEventCreateWithFlags(&c->ev_fwd, hipEventDisableTiming));
    MGPI_HIP_TRY(hipEventCreateWithFlags(&c->ev_upd, hipEventDisableTiming));
    MGPI_HIP_TRY(hipEventCreateWithFlags(&c->ev_panel, hipEventDisableTiming));
    MGPI_HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    c->streams_overlap = -1;
    c->skinny_tiles = 448;
    c->no_lookahead = getenv("MGPI_NO_LOOKAHEAD") && atoi(getenv("MGPI_NO_LOOKAHEAD")) != 0;
    c->pivot_tol = getenv("MGPI_PIVOT_TOL") ? atof(getenv("MGPI_PIVOT_TOL")) : 100.0 * 2.220446049250313e-16;
    MGPI_HIP_TRY(hipEventCreate(&c->ev0));
    MGPI_HIP_TRY(hipEventCreate(&c->ev1));
    MGPI_HIP_TRY(hipMalloc((void**)&c->d_info, 64));
    MGPI_HIP_TRY(hipMalloc((void**)&c->d_scal, 8192 * sizeof(double)));
    MGPI_HIP_TRY(hipMemset(c->d_info, 0, 64));
    if (hipHostMalloc(&c->h_pinned, MGPI_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) {   // optional: pageable read-backs still work
        (void)hipGetLastError();
        c->h_pinned = nullptr;
    }
    c->pool = new mgpi_pool();
    {
        size_t free_b = 0, total_b = 0;
        MGPI_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        c->pool->cap = total_b / 4;
        if (const char* env = getenv("MGPI_POOL_MB")) c->pool->cap = (size_t)atoll(env) << 20;
    }
    *out = c;
    return 0;
}

// ---- device buffer pool -------------------------------------------------------------------
// All users of a pooled buffer are ordered on ctx->stream (the look-ahead stream is joined back into it before any
// launcher returns), and the first touch of a reused buffer is the hipMemsetAsync of mgpi_mat_alloc on that same
// stream, so handing a block over needs no host synchronisation.
static void pool_release(mgpi_ctx* ctx) {
    if (!ctx->pool || ctx->pool->blocks.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->pool->blocks) (void)hipFree(b.ptr);
    ctx->pool->blocks.clear();
    ctx->pool->held = 0;
}

static int pool_alloc(mgpi_ctx* ctx, size_t bytes, double** out, size_t* got) {
    mgpi_pool* pool = ctx->pool;
    const int best = mgpi_pool_pick(pool->blocks, bytes);
    if (best >= 0) {
        *out = (double*)pool->blocks[best].ptr;
        *got = pool->blocks[best].bytes;
        pool->held -= pool->blocks[best].bytes;
        pool->blocks.erase(pool->blocks.begin() + best);
        return 0;
    }
    static const bool pool_debug = getenv("MGPI_POOL_DEBUG") && atoi(getenv("MGPI_POOL_DEBUG")) != 0;
    if (pool_debug) fprintf(stderr, "mgpi pool: miss, hipMalloc of %.1f MB (%zu blocks pooled, %.1f MB)\n", bytes / 1048576.0, pool->blocks.size(), pool->held / 1048576.0);
    hipError_t e = hipMalloc((void**)out, bytes);
    if (e != hipSuccess && !pool->blocks.empty()) {          // give the pooled memory back and try once more
        (void)hipGetLastError();
        pool_release(ctx);
        e = hipMalloc((void**)out, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        mgpi_set_error("mgpi_mat_alloc: hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return MGPI_ERR_ALLOC;
    }
    *got = bytes;
    return 0;
}

static void pool_free(mgpi_ctx* ctx, void* ptr, size_t bytes) {
    mgpi_pool* pool = ctx->pool;
    if (bytes > pool->cap) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ptr);
        return;
    }
    while (pool->held + bytes > pool->cap && !pool->blocks.empty()) {      // evict the oldest blocks
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(pool->blocks.front().ptr);
        pool->held -= pool->blocks.front().bytes;
        pool->blocks.erase(pool->blocks.begin());
    }
    pool->blocks.push_back({ptr, bytes});
    pool->held += bytes;
}

// pooled scratch blocks for the other translation units (the inverse diagonal blocks of a factor, chol.hip)
int mgpi_pool_alloc(mgpi_ctx* ctx, size_t bytes, double** out, size_t* got) { return pool_alloc(ctx, bytes, out, got); }
void mgpi_pool_free(mgpi_ctx* ctx, void* ptr, size_t bytes) { pool_free(ctx, ptr, bytes); }

extern "C" int mgpi_ctx_trim(mgpi_ctx* ctx) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_trim: null pointer");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    pool_release(ctx);
    return 0;
}

extern "C" int mgpi_ctx_destroy(mgpi_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    pool_release(ctx);
    delete ctx->pool;
    if (ctx->work) (void)hipFree(ctx->work);
    if (ctx->d_diag) (void)hipFree(ctx->d_diag);
    (void)hipFree(ctx->d_info);
    (void)hipFree(ctx->d_scal);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    for (int s = 0; s < MGPI_PROF_SLOTS; ++s) {
        for (hipEvent_t e : ctx->prof[s].begin) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->prof[s].end) (void)hipEventDestroy(e);
    }
    delete[] ctx->prof;
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) { (void)hipStreamSynchronize(ctx->stream3); (void)hipStreamDestroy(ctx->stream3); }
    if (ctx->ev_fwd) (void)hipEventDestroy(ctx->ev_fwd);
    (void)hipEventDestroy(ctx->ev_upd);
    (void)hipEventDestroy(ctx->ev_panel);
    (void)hipEventDestroy(ctx->ev_join);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

extern "C" int mgpi_ctx_sync(mgpi_ctx* ctx) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_sync: null ctx");
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// everything queued on `signaler`'s stream so far happens before whatever `waiter` enqueues from now on (two contexts
// of one device: the main and the look-ahead stream of the multi-GPU driver); nothing waits on the host
extern "C" int mgpi_ctx_wait(mgpi_ctx* waiter, mgpi_ctx* signaler) {
    MGPI_REQUIRE(waiter && signaler, "mgpi_ctx_wait: null ctx");
    MGPI_REQUIRE(waiter->device == signaler->device, "mgpi_ctx_wait: contexts on different devices (%d, %d)", waiter->device,
                 signaler->device);
    if (waiter == signaler) return 0;
    MGPI_HIP_TRY(hipSetDevice(signaler->device));
    MGPI_HIP_TRY(hipEventRecord(signaler->ev_join, signaler->stream));
    MGPI_HIP_TRY(hipStreamWaitEvent(waiter->stream, signaler->ev_join, 0));
    return 0;
}

// The context's launches go to its HIGH-PRIORITY stream from now on (the two streams trade places; call it on an idle context).
// For the look-ahead context of the multi-GPU driver: its panels and broadcasts share the GPU with the main context's trailing
// update, whose tens of thousands of queued workgroups a normal-priority queue waits behind (measured: a panel step took 200 us
// there against 100 us on the single-GPU path's high-priority look-ahead stream).
extern "C" int mgpi_ctx_prefer_high_priority(mgpi_ctx* ctx) {
    MGPI_REQUIRE(ctx != nullptr && ctx->stream2 != nullptr, "mgpi_ctx_prefer_high_priority: null ctx");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream2));
    hipStream_t t = ctx->stream;
    ctx->stream = ctx->stream2;
    ctx->stream2 = t;
    return 0;
}

extern "C" void* mgpi_ctx_stream(mgpi_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int mgpi_ctx_tic(mgpi_ctx* ctx) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_tic: null ctx");
    MGPI_HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

extern "C" int mgpi_ctx_toc(mgpi_ctx* ctx, double* ms) {
    MGPI_REQUIRE(ctx != nullptr && ms != nullptr, "mgpi_ctx_toc: null pointer");
    MGPI_HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    MGPI_HIP_TRY(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    MGPI_HIP_TRY(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = (double)f;
    return 0;
}

extern "C" int mgpi_ctx_set_lookahead(mgpi_ctx* ctx, int on) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_set_lookahead: null ctx");
    ctx->no_lookahead = on ? 0 : 1;
    return 0;
}

extern "C" int mgpi_ctx_set_pivot_tol(mgpi_ctx* ctx, double tol) {
    MGPI_REQUIRE(ctx != nullptr && tol >= 0.0 && tol < 1.0, "mgpi_ctx_set_pivot_tol: tolerance %g outside [0, 1)", tol);
    ctx->pivot_tol = tol;
    return 0;
}

extern "C" int mgpi_ctx_set_profile(mgpi_ctx* ctx, int on);
// which slots record HIP-event pairs (bit s = slot s); every recorded event is a marker between two kernels of its stream,
// ~5 - 9 us each on a latency chain - bench.py keeps only the dominant kernel's slot on inside its timed region
extern "C" int mgpi_ctx_set_profile_slots(mgpi_ctx* ctx, unsigned mask) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_set_profile_slots: null ctx");
    MGPI_TRY(mgpi_ctx_set_profile(ctx, 0));                     // drains the stream, clears the slots
    ctx->profile = (int)(mask & ((1u << MGPI_PROF_SLOTS) - 1u));
    return 0;
}

extern "C" int mgpi_ctx_set_profile(mgpi_ctx* ctx, int on) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_set_profile: null ctx");
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->profile = on ? (1 << MGPI_PROF_SLOTS) - 1 : 0;
    for (int s = 0; s < MGPI_PROF_SLOTS; ++s) {
        ctx->prof[s].used = 0;
        ctx->prof[s].ms = 0.0;
        ctx->prof[s].launches = 0;
        ctx->prof[s].flops = 0.0;
    }
    return 0;
}

// sums the recorded event pairs (after draining the stream)
extern "C" int mgpi_ctx_get_profile(mgpi_ctx* ctx, int which, double* ms, int64_t* launches) {
    MGPI_REQUIRE(ctx != nullptr && which >= 0 && which < MGPI_PROF_SLOTS, "mgpi_ctx_get_profile: bad argument");
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    mgpi_prof_slot& p = ctx->prof[which];
    double total = 0.0;
    for (size_t i = 0; i < p.used; ++i) {
        float f = 0.f;
        MGPI_HIP_TRY(hipEventElapsedTime(&f, p.begin[i], p.end[i]));
        total += f;
    }
    p.ms = total;
    p.launches = (int64_t)p.used;
    if (ms) *ms = total;
    if (launches) *launches = p.launches;
    return 0;
}

extern "C" int mgpi_ctx_get_profile_flops(mgpi_ctx* ctx, int which, double* flops) {
    MGPI_REQUIRE(ctx != nullptr && flops != nullptr && which >= 0 && which < MGPI_PROF_SLOTS, "mgpi_ctx_get_profile_flops: bad argument");
    *flops = ctx->prof[which].flops;
    return 0;
}

void mgpi_prof_flops(mgpi_ctx* ctx, int slot, double flops) {
    if ((ctx->profile >> slot) & 1) ctx->prof[slot].flops += flops;
}

// HIP-event pair around a launch (or a group of launches) on the ctx stream; nothing waits on the host
void mgpi_prof_begin(mgpi_ctx* ctx, int slot) {
    if (!((ctx->profile >> slot) & 1)) return;
    mgpi_prof_slot& p = ctx->prof[slot];
    if (p.used == p.begin.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        p.begin.push_back(a);
        p.end.push_back(b);
    }
    (void)hipEventRecord(p.begin[p.used], ctx->stream);
}
void mgpi_prof_end(mgpi_ctx* ctx, int slot) {
    if (!((ctx->profile >> slot) & 1)) return;
    mgpi_prof_slot& p = ctx->prof[slot];
    if (p.used >= p.end.size()) return;
    (void)hipEventRecord(p.end[p.used], ctx->stream);
    p.used += 1;
}

int mgpi_ensure_work(mgpi_ctx* ctx, size_t bytes) {
    if (ctx->work_bytes >= bytes) return 0;
    if (ctx->work) {
        MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
        MGPI_HIP_TRY(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
    }
    hipError_t e = hipMalloc((void**)&ctx->work, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        mgpi_set_error("workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return MGPI_ERR_ALLOC;
    }
    ctx->work_bytes = bytes;
    return 0;
}

// ---- matrices ---------------------------------------------------------------------------

static int mat_alloc_impl(mgpi_ctx* ctx, int64_t rows, int64_t cols, int wide, mgpi_mat** out) {
    MGPI_REQUIRE(ctx != nullptr && out != nullptr, "mgpi_mat_alloc: null pointer");
    MGPI_REQUIRE(rows >= 1 && cols >= 1, "mgpi_mat_alloc: empty shape %lld x %lld", (long long)rows, (long long)cols);
    *out = nullptr;
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    mgpi_mat* m = new (std::nothrow) mgpi_mat();
    if (!m) { mgpi_set_error("mgpi_mat_alloc: out of host memory"); return MGPI_ERR_ALLOC; }
    m->ctx = ctx;
    m->rows = rows;
    m->cols = cols;
    m->prows = mgpi_roundup(rows, MGPI_TILE);
    // narrow matrices (coordinates, vectors) keep their natural row stride; wide ones are padded to the tile
    m->pcols = (!wide && cols <= MGPI_MAX_D) ? cols : mgpi_roundup(cols, MGPI_TILE);
    m->ld = m->pcols;
    m->owns = 1;
    m->factored = 0;
    m->diag_min = m->diag_max = 0.0;
    size_t bytes = (size_t)m->prows * (size_t)m->pcols * sizeof(double);
    if (int rc = pool_alloc(ctx, bytes, &m->d, &m->alloc_bytes)) {
        delete m;
        return rc;
    }
    MGPI_HIP_TRY(hipMemsetAsync(m->d, 0, bytes, ctx->stream));
    *out = m;
    return 0;
}

extern "C" int mgpi_mat_alloc(mgpi_ctx* ctx, int64_t rows, int64_t cols, mgpi_mat** out) {
    return mat_alloc_impl(ctx, rows, cols, 0, out);
}

extern "C" int mgpi_mat_alloc_wide(mgpi_ctx* ctx, int64_t rows, int64_t cols, mgpi_mat** out) {
    return mat_alloc_impl(ctx, rows, cols, 1, out);
}

// the inverse-diagonal-block cache belongs to the handle that built it (views and wraps never share it)
static void mat_drop_dinv(mgpi_mat* m) {
    if (m->dinv) {
        (void)hipSetDevice(m->ctx->device);
        pool_free(m->ctx, m->dinv, m->dinv_bytes);
    }
    m->dinv = nullptr;
    m->dinv_nb = 0;
    m->dinv_bytes = 0;
}

// the contents of m are about to change: it is no longer a factor and its cached inverse blocks are stale
void mgpi_mat_invalidate(mgpi_mat* m) {
    m->factored = 0;
    m->diag_min = m->diag_max = 0.0;
    mat_drop_dinv(m);
}

extern "C" int mgpi_mat_free(mgpi_mat* m) {
    if (!m) return 0;
    mat_drop_dinv(m);
    if (m->owns && m->d) {
        (void)hipSetDevice(m->ctx->device);
        pool_free(m->ctx, m->d, m->alloc_bytes);
    }
    delete m;
    return 0;
}

extern "C" int mgpi_mat_view(const mgpi_mat* m, int64_t r0, int64_t c0, int64_t rows, int64_t cols, mgpi_mat** out) {
    MGPI_REQUIRE(m != nullptr && out != nullptr, "mgpi_mat_view: null pointer");
    MGPI_REQUIRE(r0 >= 0 && c0 >= 0 && rows >= 1 && cols >= 1 && r0 + rows <= m->prows && c0 + cols <= m->pcols,
                 "mgpi_mat_view: window (%lld,%lld)+(%lld x %lld) outside padded %lld x %lld", (long long)r0,
                 (long long)c0, (long long)rows, (long long)cols, (long long)m->prows, (long long)m->pcols);
    mgpi_mat* v = new (std::nothrow) mgpi_mat();
    if (!v) { mgpi_set_error("mgpi_mat_view: out of host memory"); return MGPI_ERR_ALLOC; }
    v->ctx = m->ctx;
    v->rows = rows;
    v->cols = cols;
    v->prows = rows;   // a view exposes exactly its window
    v->pcols = cols;
    v->ld = m->ld;
    v->d = m->d + r0 * m->ld + c0;
    v->owns = 0;
    v->alloc_bytes = 0;
    // a window of a factor is a factor only when it is the whole (padded) matrix; the parent's cache of inverse
    // diagonal blocks is never shared (a view that needs one builds and owns its own)
    v->factored = (m->factored && r0 == 0 && c0 == 0 && rows == m->prows && cols == m->pcols) ? 1 : 0;
    v->diag_min = v->factored ? m->diag_min : 0.0;
    v->diag_max = v->factored ? m->diag_max : 0.0;
    v->dinv = nullptr;
    v->dinv_nb = 0;
    v->dinv_bytes = 0;
    *out = v;
    return 0;
}

// the factor inside a riders matrix (mgpi_posv_riders_dev) as a square handle of its own: rows = cols = n, padded pn x pn,
// non-owning - what the entry points that take "the Cholesky factor L" expect (mgpi_potrs_dev, mgpi_predict_var_dev, ..)
extern "C" int mgpi_mat_square_view(const mgpi_mat* m, mgpi_mat** out) {
    MGPI_REQUIRE(m != nullptr && out != nullptr, "mgpi_mat_square_view: null pointer");
    MGPI_REQUIRE(m->prows >= m->pcols && m->cols <= m->pcols, "mgpi_mat_square_view: %lld x %lld (padded %lld x %lld) has no leading square",
                 (long long)m->rows, (long long)m->cols, (long long)m->prows, (long long)m->pcols);
    mgpi_mat* v = new (std::nothrow) mgpi_mat();
    if (!v) { mgpi_set_error("mgpi_mat_square_view: out of host memory"); return MGPI_ERR_ALLOC; }
    *v = *m;
    v->rows = v->cols = m->cols;
    v->prows = v->pcols = m->pcols;
    v->owns = 0;
    v->alloc_bytes = 0;
    v->dinv = nullptr;
    v->dinv_nb = 0;
    v->dinv_bytes = 0;
    *out = v;
    return 0;
}

// non-owning handle over caller-provided device memory (e.g. a buffer another library allocated): exactly
// rows x cols, row stride ld; the caller guarantees the dimensions it later passes to tile kernels are padded
extern "C" int mgpi_mat_wrap(mgpi_ctx* ctx, void* devptr, int64_t rows, int64_t cols, int64_t ld, int factored, mgpi_mat** out) {
    MGPI_REQUIRE(ctx && devptr && out, "mgpi_mat_wrap: null pointer");
    MGPI_REQUIRE(rows >= 1 && cols >= 1 && ld >= cols, "mgpi_mat_wrap: bad shape %lld x %lld (ld %lld)", (long long)rows,
                 (long long)cols, (long long)ld);
    MGPI_REQUIRE(((uintptr_t)devptr % 16) == 0, "mgpi_mat_wrap: pointer must be 16-byte aligned");
    mgpi_mat* v = new (std::nothrow) mgpi_mat();
    if (!v) { mgpi_set_error("mgpi_mat_wrap: out of host memory"); return MGPI_ERR_ALLOC; }
    v->ctx = ctx;
    v->rows = v->prows = rows;
    v->cols = v->pcols = cols;
    v->ld = ld;
    v->d = (double*)devptr;
    v->owns = 0;
    v->factored = factored;
    v->diag_min = v->diag_max = 0.0;
    *out = v;
    return 0;
}

extern "C" int mgpi_mat_shape(const mgpi_mat* m, int64_t* rows, int64_t* cols, int64_t* ld) {
    MGPI_REQUIRE(m != nullptr, "mgpi_mat_shape: null matrix");
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (ld) *ld = m->ld;
    return 0;
}

// a buffer that RECEIVED a factored block (broadcast, copy from another process) is a factor too: the solves check
extern "C" int mgpi_mat_mark_factored(mgpi_mat* m, int factored) {
    MGPI_REQUIRE(m != nullptr, "mgpi_mat_mark_factored: null matrix");
    if (!factored) mgpi_mat_invalidate(m);
    else m->factored = 1;
    return 0;
}

extern "C" void* mgpi_mat_devptr(const mgpi_mat* m) { return m ? (void*)m->d : nullptr; }

extern "C" int mgpi_mat_upload(mgpi_mat* m, const double* host, int64_t host_ld) {
    MGPI_REQUIRE(m != nullptr && host != nullptr, "mgpi_mat_upload: null pointer");
    MGPI_REQUIRE(host_ld >= m->cols, "mgpi_mat_upload: host_ld %lld < cols %lld", (long long)host_ld, (long long)m->cols);
    MGPI_HIP_TRY(hipSetDevice(m->ctx->device));
    MGPI_HIP_TRY(hipMemcpy2DAsync(m->d, (size_t)m->ld * sizeof(double), host, (size_t)host_ld * sizeof(double),
                                  (size_t)m->cols * sizeof(double), (size_t)m->rows, hipMemcpyHostToDevice,
                                  m->ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    m->factored = 0;
    m->diag_min = m->diag_max = 0.0;
    mat_drop_dinv(m);
    return 0;
}

extern "C" int mgpi_mat_download(const mgpi_mat* m, double* host, int64_t host_ld) {
    MGPI_REQUIRE(m != nullptr && host != nullptr, "mgpi_mat_download: null pointer");
    MGPI_REQUIRE(host_ld >= m->cols, "mgpi_mat_download: host_ld %lld < cols %lld", (long long)host_ld, (long long)m->cols);
    MGPI_HIP_TRY(hipSetDevice(m->ctx->device));
    MGPI_HIP_TRY(hipMemcpy2DAsync(host, (size_t)host_ld * sizeof(double), m->d, (size_t)m->ld * sizeof(double),
                                  (size_t)m->cols * sizeof(double), (size_t)m->rows, hipMemcpyDeviceToHost,
                                  m->ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

extern "C" int mgpi_mat_copy(mgpi_ctx* ctx, const mgpi_mat* src, mgpi_mat* dst) {
    MGPI_REQUIRE(ctx && src && dst, "mgpi_mat_copy: null pointer");
    MGPI_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, "mgpi_mat_copy: shape mismatch");
    int64_t r = src->prows < dst->prows ? src->prows : dst->prows;
    int64_t c = src->pcols < dst->pcols ? src->pcols : dst->pcols;
    MGPI_HIP_TRY(hipMemcpy2DAsync(dst->d, (size_t)dst->ld * sizeof(double), src->d, (size_t)src->ld * sizeof(double),
                                  (size_t)c * sizeof(double), (size_t)r, hipMemcpyDeviceToDevice, ctx->stream));
    dst->factored = src->factored;
    dst->diag_min = src->diag_min;
    dst->diag_max = src->diag_max;
    mat_drop_dinv(dst);          // inverse blocks of whatever dst held before
    return 0;
}

// upper(i,j) = lower(j,i); 32x32 tiles through LDS so both sides stay coalesced
__global__ __launch_bounds__(256) void symmetrize_kernel(double* a, int64_t ld, int64_t n) {
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;   // source tile (bi, bj) with bj <= bi
    if (bj > bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int64_t i = (int64_t)bi * 32 + r, j = (int64_t)bj * 32 + tx;
        tile[r][tx] = (i < n && j < n) ? a[i * ld + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        // destination element (row = bj*32 + r, col = bi*32 + tx) = source (bi*32+tx, bj*32+r)
        int64_t i = (int64_t)bj * 32 + r, j = (int64_t)bi * 32 + tx;
        if (i < n && j < n && j > i) a[i * ld + j] = tile[tx][r];
    }
}

extern "C" int mgpi_mat_symmetrize(mgpi_ctx* ctx, mgpi_mat* m) {
    MGPI_REQUIRE(ctx && m, "mgpi_mat_symmetrize: null pointer");
    MGPI_REQUIRE(m->rows == m->cols, "mgpi_mat_symmetrize: matrix is not square");
    unsigned nt = (unsigned)((m->rows + 31) / 32);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nt, nt), dim3(256), 0, ctx->stream, m->d, m->ld, m->rows);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void tril_kernel(double* a, int64_t ld, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n && j > i) a[i * ld + j] = 0.0;
}

extern "C" int mgpi_mat_tril(mgpi_ctx* ctx, mgpi_mat* m) {
    MGPI_REQUIRE(ctx && m, "mgpi_mat_tril: null pointer");
    MGPI_REQUIRE(m->rows == m->cols, "mgpi_mat_tril: matrix is not square");
    const int64_t n = m->prows < m->pcols ? m->prows : m->pcols;
    MGPI_REQUIRE(n <= 65535 * 64, "mgpi_mat_tril: matrix too large");
    hipLaunchKernelGGL(tril_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, ctx->stream, m->d, m->ld, n);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void diag_kernel(const double* a, int64_t ld, int64_t n, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i * ld + i];
}

extern "C" int mgpi_mat_diag(mgpi_ctx* ctx, const mgpi_mat* m, mgpi_mat* diag) {
    MGPI_REQUIRE(ctx && m && diag, "mgpi_mat_diag: null pointer");
    MGPI_REQUIRE(m->rows == m->cols && diag->rows == m->rows && diag->ld == 1, "mgpi_mat_diag: shapes do not conform");
    hipLaunchKernelGGL(diag_kernel, dim3((unsigned)((m->rows + 255) / 256)), dim3(256), 0, ctx->stream, m->d, m->ld, m->rows, diag->d);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void add_diag_kernel(double* a, int64_t ld, int64_t n, double shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i * ld + i] += shift;
}

extern "C" int mgpi_mat_add_diag(mgpi_ctx* ctx, mgpi_mat* m, double shift) {
    MGPI_REQUIRE(ctx && m, "mgpi_mat_add_diag: null pointer");
    MGPI_REQUIRE(m->rows == m->cols, "mgpi_mat_add_diag: matrix is not square");
    hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)((m->rows + 255) / 256)), dim3(256), 0, ctx->stream, m->d, m->ld, m->rows, shift);
    MGPI_HIP_TRY(hipGetLastError());
    mgpi_mat_invalidate(m);
    return 0;
}

__global__ void fill_diag_kernel(double* a, int64_t ld, int64_t n, double value) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i * ld + i] = value;
}

// m[i][i] = value for i < min(rows, cols) of the handle (a view addresses a sub-diagonal): the unit diagonal of the
// padded global indices of a block-cyclic K
extern "C" int mgpi_mat_fill_diag(mgpi_ctx* ctx, mgpi_mat* m, double value) {
    MGPI_REQUIRE(ctx && m, "mgpi_mat_fill_diag: null pointer");
    const int64_t n = m->rows < m->cols ? m->rows : m->cols;
    hipLaunchKernelGGL(fill_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, m->d, m->ld, n, value);
    MGPI_HIP_TRY(hipGetLastError());
    mgpi_mat_invalidate(m);
    return 0;
}

__global__ void axpy_kernel(double a, const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fma(a, x[i], y[i]);
}

// y += a x for (n x 1) vectors
extern "C" int mgpi_axpy_dev(mgpi_ctx* ctx, double a, const mgpi_mat* x, mgpi_mat* y) {
    MGPI_REQUIRE(ctx && x && y, "mgpi_axpy_dev: null pointer");
    MGPI_REQUIRE(x->ld == 1 && y->ld == 1 && x->rows == y->rows, "mgpi_axpy_dev: x and y must be vectors of one length");
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((x->rows + 255) / 256)), dim3(256), 0, ctx->stream, a, x->d, y->d, x->rows);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void copy_blocks_kernel(const double* __restrict__ src, int64_t lds_, int64_t src_step,
                                                          double* __restrict__ dst, int64_t ldd, int64_t dst_step, int64_t rows,
                                                          int64_t cols) {
    const double* s = src + (int64_t)blockIdx.z * src_step * lds_;
    double* d = dst + (int64_t)blockIdx.z * dst_step * ldd;
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2, r = blockIdx.y;
    if (c < cols && r < rows) *reinterpret_cast<double2*>(d + r * ldd + c) = *reinterpret_cast<const double2*>(s + r * lds_ + c);
}

// block j (j < count) = `rows` full-width rows: dst rows [dst_r0 + j dst_step, ..) <- src rows [src_r0 + j src_step, ..).
// One launch gathers the blocks a peer wants out of a row panel (they sit lcm(P,Q)/P blocks apart) or scatters them.
extern "C" int mgpi_copy_blocks_dev(mgpi_ctx* ctx, const mgpi_mat* src, int64_t src_r0, int64_t src_step, mgpi_mat* dst,
                                    int64_t dst_r0, int64_t dst_step, int64_t rows, int64_t count) {
    MGPI_REQUIRE(ctx && src && dst, "mgpi_copy_blocks_dev: null pointer");
    if (count == 0 || rows == 0) return 0;
    MGPI_REQUIRE(rows > 0 && count > 0 && count < 65536 && rows < 65536 && src_step >= 0 && dst_step >= 0 && src_r0 >= 0 && dst_r0 >= 0,
                 "mgpi_copy_blocks_dev: bad block description");
    const int64_t cols = src->pcols < dst->pcols ? src->pcols : dst->pcols;
    MGPI_REQUIRE(src->pcols == dst->pcols && cols % 2 == 0 && src->ld % 2 == 0 && dst->ld % 2 == 0,
                 "mgpi_copy_blocks_dev: the matrices must have one (even) width");
    MGPI_REQUIRE(src_r0 + (count - 1) * src_step + rows <= src->prows && dst_r0 + (count - 1) * dst_step + rows <= dst->prows,
                 "mgpi_copy_blocks_dev: blocks reach outside the matrices");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(copy_blocks_kernel, dim3((unsigned)((cols / 2 + 255) / 256), (unsigned)rows, (unsigned)count), dim3(256), 0,
                       ctx->stream, src->d + src_r0 * src->ld, src->ld, src_step, dst->d + dst_r0 * dst->ld, dst->ld, dst_step, rows,
                       cols);
    MGPI_HIP_TRY(hipGetLastError());
    mgpi_mat_invalidate(dst);
    return 0;
}

__global__ void unscale_kernel(double* a, int64_t ld, int64_t n, const double* __restrict__ s) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n) a[i * ld + j] = a[i * ld + j] / (s[i] * s[j]);
}

extern "C" int mgpi_mat_unscale(mgpi_ctx* ctx, mgpi_mat* m, const mgpi_mat* s) {
    MGPI_REQUIRE(ctx && m && s, "mgpi_mat_unscale: null pointer");
    MGPI_REQUIRE(m->rows == m->cols && s->rows == m->rows && s->ld == 1, "mgpi_mat_unscale: shapes do not conform");
    MGPI_REQUIRE(m->rows <= 65535, "mgpi_mat_unscale: matrix too large");
    hipLaunchKernelGGL(unscale_kernel, dim3((unsigned)((m->rows + 255) / 256), (unsigned)m->rows), dim3(256), 0, ctx->stream,
                       m->d, m->ld, m->rows, s->d);
    MGPI_HIP_TRY(hipGetLastError());
    m->factored = 0;
    m->diag_min = m->diag_max = 0.0;
    mat_drop_dinv(m);
    return 0;
}

int mgpi_make_terms(const mgpi_kernel_desc* terms, int nterms, int d, mgpi_terms_dev* out) {
    MGPI_REQUIRE(terms != nullptr && out != nullptr, "kernel terms: null pointer");
    MGPI_REQUIRE(nterms >= 1 && nterms <= MGPI_MAX_TERMS, "kernel terms: nterms=%d outside [1,%d]", nterms, MGPI_MAX_TERMS);
    MGPI_REQUIRE(d >= 1 && d <= MGPI_MAX_D, "kernel terms: d=%d outside [1,%d]", d, MGPI_MAX_D);
    memset(out, 0, sizeof(*out));
    out->nterms = nterms;
    out->d = d;
    int exact = 1;
    for (int t = 0; t < nterms; ++t) {
        MGPI_REQUIRE(terms[t].type == MGPI_TERM_SE || terms[t].type == MGPI_TERM_WHITE || terms[t].type == MGPI_TERM_MATERN,
                     "kernel terms: term %d has unsupported type %d", t, terms[t].type);
        out->type[t] = terms[t].type;
        out->sig2[t] = terms[t].sigma * terms[t].sigma;          // sigma**2
        if (terms[t].type == MGPI_TERM_SE || terms[t].type == MGPI_TERM_MATERN)
            for (int k = 0; k < d; ++k) {
                const double l2 = terms[t].length[k] * terms[t].length[k];     // lengths**2 (kernels.py:185)
                out->l2[t][k] = l2;
                out->inv_l2[t][k] = 1.0 / l2;
                if (!(l2 >= 2.3e-308 && l2 <= 1e300 && out->inv_l2[t][k] >= 2.3e-308)) exact = 0;   // zero / inf / NaN / subnormal: divide for real
            }
        if (terms[t].type == MGPI_TERM_MATERN) {
            const double o = terms[t].order;
            MGPI_REQUIRE(o > 0.0 && o <= 120.0, "kernel terms: Matern order %g outside (0, 120]", o);
            const int nl = (int)(o + 0.5);
            const double mu = o - nl;                             // |mu| <= 1/2
            out->mat_nl[t] = nl;
            out->mat_mu[t] = mu;
            out->mat_order[t] = o;
            out->mat_two_o[t] = 2.0 * o;
            out->mat_norm[t] = pow(2.0, 1.0 - o) / tgamma(o);     // 2**(1-o)/gamma(o)  (kernels.py:162)
            const double gampl = 1.0 / tgamma(1.0 + mu), gammi = 1.0 / tgamma(1.0 - mu);
            out->mat_gampl[t] = gampl;
            out->mat_gammi[t] = gammi;
            out->mat_gam2[t] = 0.5 * (gammi + gampl);
            if (fabs(mu) > 1e-3) {
                out->mat_gam1[t] = (gammi - gampl) / (2.0 * mu);
            } else {   // 1/Gamma(1+z) = 1 + c1 z + c2 z^2 + ...  =>  (rg(1-mu) - rg(1+mu)) / (2 mu) = -(c1 + c3 mu^2 + c5 mu^4)
                const double c1 = 0.5772156649015329, c3 = -0.0420026350340952, c5 = -0.0421977345555443, m2 = mu * mu;
                out->mat_gam1[t] = -(c1 + m2 * (c3 + m2 * c5));
            }
            // o = p + 1/2: K_o is elementary and the shape function collapses to
            //   exp(-x) * p!/(2p)! * sum_{i=0..p} (2p - i)! / ((p - i)! i!) * (2 x)^i      (1, 1 + x, 1 + x + x^2/3, ...)
            out->mat_half_p[t] = -1;
            const double twice = 2.0 * o;
            if (twice == floor(twice) && ((long long)twice & 1) && o <= 0.5 + (MGPI_MATERN_POLY - 1)) {
                const int p = (int)((twice - 1.0) * 0.5);
                out->mat_half_p[t] = p;
                for (int i = 0; i < MGPI_MATERN_POLY; ++i) out->mat_poly[t][i] = 0.0;
                for (int i = 0; i <= p; ++i)
                    out->mat_poly[t][i] = exp(lgamma(p + 1.0) - lgamma(2.0 * p + 1.0) + lgamma(2.0 * p - i + 1.0) - lgamma(p - i + 1.0) -
                                              lgamma(i + 1.0) + i * log(2.0));
            }
        }
    }
    out->exact_div = exact;
    out->fast_mix = 0;
    if (exact && nterms == 1 && terms[0].type == MGPI_TERM_SE) out->fast_mix = 1;
    if (exact && nterms == 2 && terms[0].type == MGPI_TERM_SE && terms[1].type == MGPI_TERM_WHITE) out->fast_mix = 2;
    if (getenv("MGPI_COV_FAST") && atoi(getenv("MGPI_COV_FAST")) == 0) out->fast_mix = 0;      // developer switch: the general loop
    return 0;
}
