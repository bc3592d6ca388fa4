// yh_api.hip — the extern "C" boundary of libyacht_hip.so (declared in include/yacht_hip.h).
#include "yh_common.h"
#include "yh_sort.h"
#include "yh_pack.h"

#include <mutex>
#include <unordered_map>

#include <stdarg.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <utility>
#include <vector>

// ---- error plumbing ------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void yh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* yh_tune_env(const char* name) {
    static const bool open_gate = [] { const char* e = getenv("YH_DEBUG_TUNING"); return e && e[0] == '1'; }();
    return open_gate ? getenv(name) : nullptr;
}

// ---- a cache of device buffers -----------------------------------------------------------------------------------
// The temporaries of a build or a pairwise pass, and the arrays of a YH_DB_PAIRWISE_ONLY handle (which lives for one
// `yacht train` call), are taken from and returned to a per-process cache instead of the driver: a hipFree of a gigabyte
// buffer costs 0.3-1 ms and synchronizes the device, `yacht train` made thirty of them per call (3.5 of 14 ms), and
// hipFreeAsync into the device's memory pool was no cheaper (0.35 ms a call).  A block is returned with an EVENT recorded
// on the stream that used it last (events are the library's own, pooled: the stream itself may be a caller's and gone by
// the time the block is reused): the same stream may have the block back at once (stream order), any other user waits
// for the event.  What the cache holds beyond YH_POOL_KEEP (default 48 GiB) goes back to the driver at the end of a
// create / destroy.
namespace {
struct CacheBlock { void* p; size_t bytes; int device; hipStream_t owner; hipEvent_t freed; uint64_t tick; };  // tick: when it came back (trim: oldest first)
uint64_t g_cache_tick = 0;
std::mutex g_cache_mu;
std::vector<CacheBlock> g_cache;                      // free blocks
std::unordered_map<void*, size_t> g_cache_live;       // blocks handed out: their sizes
std::vector<std::pair<int, hipEvent_t>> g_cache_events;  // idle events, by device
uint64_t g_driver_allocs = 0;                         // hipMalloc calls made for handles (yh_alloc_stats)
double g_driver_alloc_ms = 0.0;                       // host time inside them
bool cache_on() {
    static const bool on = [] { const char* off = yh_tune_env("YH_NO_POOL"); return !(off && off[0] == '1'); }();
    return on;
}
hipEvent_t cache_event_take(int device) {  // (g_cache_mu held)
    for (size_t i = 0; i < g_cache_events.size(); ++i)
        if (g_cache_events[i].first == device) {
            hipEvent_t e = g_cache_events[i].second;
            g_cache_events[i] = g_cache_events.back();
            g_cache_events.pop_back();
            return e;
        }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
void cache_event_give(int device, hipEvent_t e) {  // (g_cache_mu held)
    if (e) g_cache_events.emplace_back(device, e);
}
}  // namespace
// (YH_TRACE_BUILD=1 behind the tuning gate: every device allocation that took the host more than 0.5 ms, to stderr)
static bool alloc_trace_on() {
    static const bool on = [] { const char* e = yh_tune_env("YH_TRACE_BUILD"); return e && e[0] == '1'; }();
    return on;
}
static double alloc_now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
hipError_t yh_tmalloc(yh_db* db, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (!cache_on()) return hipMalloc(p, bytes);
    bytes = (bytes + 255) & ~(size_t)255;
    hipEvent_t wait_for = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        int best = -1;
        for (size_t i = 0; i < g_cache.size(); ++i) {
            const CacheBlock& b = g_cache[i];
            if (b.device != db->device || b.bytes < bytes || b.bytes > bytes + bytes / 4 + (1u << 20)) continue;
            if (best < 0 || b.bytes < g_cache[best].bytes || (b.bytes == g_cache[best].bytes && b.owner == db->stream)) best = (int)i;
        }
        if (best >= 0) {
            const CacheBlock b = g_cache[best];
            g_cache[best] = g_cache.back();
            g_cache.pop_back();
            g_cache_live[b.p] = b.bytes;
            *p = b.p;
            if (b.freed && b.owner != db->stream) wait_for = b.freed;  // (another stream used it last: its work up to the free)
            else { cache_event_give(b.device, b.freed); return hipSuccess; }
        }
    }
    if (wait_for) {
        const hipError_t we = hipEventSynchronize(wait_for);
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            cache_event_give(db->device, wait_for);
        }
        if (we != hipSuccess) { (void)hipGetLastError(); return hipDeviceSynchronize(); }
        return hipSuccess;
    }
    const double t0 = alloc_now_ms();
    hipError_t e = hipMalloc(p, bytes);
    const double took = alloc_now_ms() - t0;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        ++g_driver_allocs;
        g_driver_alloc_ms += took;
    }
    if (alloc_trace_on() && took > 0.5) fprintf(stderr, "[yh alloc] cache miss: hipMalloc(%.1f MB) took %.3f ms\n", bytes / 1e6, took);
    if (e == hipErrorOutOfMemory) {  // give the driver back what the cache holds, and once more
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (const CacheBlock& b : g_cache) { drop.push_back(b.p); cache_event_give(b.device, b.freed); }
            g_cache.clear();
        }
        for (void* q : drop) (void)hipFree(q);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_cache_live[*p] = bytes;
    }
    return e;
}
void yh_tfree(yh_db* db, void* p) {
    if (!p) return;
    if (cache_on()) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_cache_live.find(p);
        if (it != g_cache_live.end()) {
            hipEvent_t ev = cache_event_take(db->device);
            if (ev && hipEventRecord(ev, db->stream) != hipSuccess) {  // (cannot mark the point: wait for the stream now instead)
                (void)hipGetLastError();
                (void)hipStreamSynchronize(db->stream);
                cache_event_give(db->device, ev);
                ev = nullptr;
            } else if (!ev) {
                (void)hipStreamSynchronize(db->stream);
            }
            g_cache.push_back(CacheBlock{p, it->second, db->device, db->stream, ev, ++g_cache_tick});
            g_cache_live.erase(it);
            return;
        }
    }
    (void)hipFree(p);
}
// (the handle's stream has drained) blocks it used last are anyone's now; the excess over the bar goes back to the driver
void yh_pool_trim(yh_db* db) {
    if (!cache_on()) return;
    // The bar: a SIXTH OF THE DEVICE'S MEMORY (48 GB of an MI355X's 288: what a handle at GTDB scale and its build's
    // temporaries leave behind -- 28 GB -- comes back without the driver), never more than half of what is free right now
    // plus what the cache already holds, so that torch / RCCL in the same process, or another process on the GPU, are not
    // starved by blocks nobody uses; YH_POOL_KEEP=<bytes> overrides it (no tuning gate: a deployment knob),
    // yh_pool_release() gives everything back on demand.
    static const long long keep_env = [] { const char* e = getenv("YH_POOL_KEEP"); return e && e[0] ? atoll(e) : -1ll; }();
    size_t keep = 0;
    if (keep_env >= 0) {
        keep = (size_t)keep_env;
    } else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)288 << 30; free_b = total_b; }
        size_t held_now = 0;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (const CacheBlock& b : g_cache) held_now += b.bytes;
        }
        keep = std::min(total_b / 6, (free_b + held_now) / 2);
    }
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        size_t held = 0;
        for (CacheBlock& b : g_cache) {
            if (b.owner == db->stream) {  // (drained: nothing to wait for any more)
                b.owner = nullptr;
                cache_event_give(b.device, b.freed);
                b.freed = nullptr;
            }
            held += b.bytes;
        }
        while (held > keep) {  // the idle block that came back longest ago first (round 4: the largest first -- a process that went from one
            // database to another of a slightly different size then lost the NEW handle's big blocks at every destroy and asked the
            // driver for them again at every create: 493 ms per build on a box whose hipMalloc stalls, profiles/r05/hot_kmers_first.json)
            int big = -1;
            for (size_t i = 0; i < g_cache.size(); ++i)
                if (!g_cache[i].owner && (big < 0 || g_cache[i].tick < g_cache[big].tick)) big = (int)i;
            if (big < 0) break;
            held -= g_cache[big].bytes;
            drop.push_back(g_cache[big].p);
            g_cache[big] = g_cache.back();
            g_cache.pop_back();
        }
    }
    for (void* q : drop) (void)hipFree(q);
}

// everything the cache holds idle goes back to the driver (blocks whose last user has not finished are waited for first)
static uint64_t pool_release_all() {
    std::vector<CacheBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        drop.swap(g_cache);
    }
    uint64_t bytes = 0;
    for (CacheBlock& b : drop) {
        if (b.freed) {
            (void)hipSetDevice(b.device);
            if (hipEventSynchronize(b.freed) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
            std::lock_guard<std::mutex> lk(g_cache_mu);
            cache_event_give(b.device, b.freed);
        }
        (void)hipFree(b.p);
        bytes += b.bytes;
    }
    return bytes;
}

int yh_dmalloc(yh_db* db, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    // Every array of a handle comes from the cache too (round 4; before: only those of YH_DB_PAIRWISE_ONLY handles): a
    // process that creates handle after handle -- tests, the shares of a scaling model, `yacht train` -- then stops asking
    // the driver for gigabytes it has just given back.  The driver's hipMalloc of a multi-GB block now and then takes
    // SECONDS on this pool right after such frees (profiles/r04/malloc_probe.txt: 0.2 ms, 0.2 ms, 3 951 ms for the same
    // call in a plain HIP program; build_trace.txt: one create of six at 3.5 s, all of it inside one allocation).
    const bool cached = true;
    const double t0 = alloc_trace_on() ? alloc_now_ms() : 0.0;
    hipError_t e = cached ? yh_tmalloc(db, p, bytes) : hipMalloc(p, bytes);
    if (alloc_trace_on() && alloc_now_ms() - t0 > 0.5)
        fprintf(stderr, "[yh alloc] %s(%.1f MB) took %.3f ms\n", cached ? "yh_tmalloc" : "hipMalloc", bytes / 1e6, alloc_now_ms() - t0);
    if (e != hipSuccess) {
        *p = nullptr;
        yh_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return YH_ERR_OOM;
    }
    db->device_bytes += bytes;
    return YH_OK;
}
void yh_dfree(yh_db* db, void* p) { yh_tfree(db, p); }  // (a block the cache did not hand out goes to hipFree)

static void ring_create(EventRing& r) {
    if (r.created) return;
    for (int i = 0; i < TIMING_RING; ++i) {
        (void)hipEventCreate(&r.beg[i]);
        (void)hipEventCreate(&r.end[i]);
    }
    r.created = true;
}
static void ring_destroy(EventRing& r) {
    if (!r.created) return;
    for (int i = 0; i < TIMING_RING; ++i) {
        (void)hipEventDestroy(r.beg[i]);
        (void)hipEventDestroy(r.end[i]);
    }
    r.created = false;
}
// Every event record is a barrier packet between two kernels of the step (~1-2 us each), so the
// rings sample: YH_TIMING_EVERY=n records every n-th launch of a ring (default 8, 1 = all, 0 = none;
// measured step time at rs214 scale: 0.237 ms recording every launch, 0.227 every 4th, 0.224 never).
static int timing_every() {
    static int every = -1;
    if (every < 0) {
        const char* e = yh_tune_env("YH_TIMING_EVERY");
        every = e ? atoi(e) : 32;  // (an event pair costs the stream ~5 us: every 8th launch was 1.2 us of a 45 us step)
        if (every < 0) every = 0;
    }
    return every;
}
// ---- page-locked staging buffers (yh_common.h: YhPin) -----------------------------------------------------------------
namespace {
struct PinBuf { void* p = nullptr; u64 cap = 0; bool busy = false; };
std::mutex g_pin_mu;
PinBuf g_pin[4];
}
void* yh_pin_acquire(u64 bytes) {
    static const bool off = [] { const char* e = yh_tune_env("YH_NO_PIN"); return e && e[0] == '1'; }();  // (tests: the pageable paths)
    if (off || bytes == 0 || bytes > ((u64)64 << 20)) return nullptr;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (PinBuf& b : g_pin)
        if (!b.busy && b.p && b.cap >= bytes) { b.busy = true; return b.p; }
    PinBuf* pick = nullptr;
    for (PinBuf& b : g_pin)
        if (!b.busy && !b.p) { pick = &b; break; }
    if (!pick)
        for (PinBuf& b : g_pin)
            if (!b.busy) { pick = &b; break; }  // (too small for this caller: replaced)
    if (!pick) return nullptr;
    if (pick->p) { (void)hipHostFree(pick->p); pick->p = nullptr; pick->cap = 0; }
    u64 cap = (u64)64 << 10;
    while (cap < bytes) cap <<= 1;
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    pick->p = p;
    pick->cap = cap;
    pick->busy = true;
    return p;
}
void yh_pin_release(void* p) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (PinBuf& b : g_pin)
        if (b.p == p) { b.busy = false; return; }
}
static void pin_release_idle() {  // (yh_pool_release: the idle ones back to the system)
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (PinBuf& b : g_pin)
        if (b.p && !b.busy) { (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
}

void yh_ring_record_begin(yh_db* db, EventRing& r, hipStream_t st) {
    if (!r.created) {
        if (!r.wanted || timing_every() <= 0) return;
        ring_create(r);  // lazily: 2 x TIMING_RING hipEventCreate calls were ~1.5 ms of every yh_db_create with three eager rings of 256
    }
    const int every = timing_every();
    // every n-th launch, and always the first one after the ring was read (a short measurement still gets a sample)
    r.armed = every > 0 && ((r.calls++ % (unsigned)every) == 0 || r.pending == 0);
    if (!r.armed) return;
    (void)hipEventRecord(r.beg[r.head], st ? st : db->stream);
}
void yh_ring_record_end(yh_db* db, EventRing& r, hipStream_t st) {
    if (!r.created || !r.armed) return;
    r.armed = false;
    (void)hipEventRecord(r.end[r.head], st ? st : db->stream);
    r.head = (r.head + 1) % TIMING_RING;
    if (r.pending < TIMING_RING) ++r.pending;
}
// mean elapsed ms over the launches recorded since the last read (stream must be idle)
static float ring_read(EventRing& r) {
    if (!r.created || r.pending == 0) return 0.f;
    double acc = 0.0;
    int n = 0;
    for (int k = 0; k < r.pending; ++k) {
        const int slot = (r.head - 1 - k + 2 * TIMING_RING) % TIMING_RING;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.beg[slot], r.end[slot]) == hipSuccess) { acc += ms; ++n; }
    }
    r.pending = 0;
    return n ? (float)(acc / n) : 0.f;
}

// Stages of pipelined steps (yh_run_device_pipelined) that have not run yet: one or two draining launches on the
// handle's stream, behind which every output of every step queued so far is complete.
static int pipe_join(yh_db* db) {
    while (db->pend_red >= 0 || db->pend_excl >= 0) YH_TRY(yh_q_step_fused(db, nullptr, 0, nullptr, nullptr, nullptr));
    return YH_OK;
}

// A query that is not one of the two halves of a step context re-uses the current context's subset bits and work
// list: a context whose first half is queued loses them (yh_run_finish_device reports it).
// the pending stages run, and the step queued last is no longer a pipelined one
static int pipe_leave(yh_db* db) {
    YH_TRY(pipe_join(db));
    db->pipe_last_ctx = -1;
    return YH_OK;
}
// the handle's stream behind the last second half queued on the finish stream (yh_db_set_batch_finish_stream): the second
// halves use the work list and the subset bits every other query uses
static void fin_join(yh_db* db) {
    if (!db->fin_pending) return;
    db->fin_pending = false;
    if (db->ev_fin) (void)hipStreamWaitEvent(db->stream, db->ev_fin, 0);
}
static void note_other_query(yh_db* db, bool join_fin = true) {
    (void)pipe_leave(db);
    if (db->ctx_open[db->ctx_now]) db->ctx_clobbered[db->ctx_now] = true;
    if (join_fin) fin_join(db);
}

// HIP events around the copies of the synchronous host-pointer queries (yh_timing.ms_h2d / ms_d2h); dir 0 = up, 1 = down
static void xfer_mark(yh_db* db, int dir, int end) {
    if (!db->ev_xfer[dir][0]) {
        if (hipEventCreate(&db->ev_xfer[dir][0]) != hipSuccess || hipEventCreate(&db->ev_xfer[dir][1]) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
    }
    (void)hipEventRecord(db->ev_xfer[dir][end], db->stream);
    if (end) db->xfer_recorded[dir] = true;
}

static bool db_ok(yh_db* db) {
    if (!db) { yh_set_error("null yh_db handle"); return false; }
    return true;
}
static int db_select(yh_db* db) {
    YH_HIP(hipSetDevice(db->device));
    return YH_OK;
}

static int ensure_sample_tmp(yh_db* db, u64 n) {
    if (n <= db->sample_tmp_cap) return YH_OK;
    if (db->d_sample_tmp) { yh_dfree(db, db->d_sample_tmp); db->d_sample_tmp = nullptr; db->sample_tmp_cap = 0; }
    const u64 cap = std::max<u64>(n, 1024);
    YH_HIP(hipMalloc((void**)&db->d_sample_tmp, cap * sizeof(u64)));
    db->sample_tmp_cap = cap;
    return YH_OK;
}

extern "C" {

const char* yh_last_error(void) { return g_err; }
int yh_abi_version(void) { return YH_ABI_VERSION; }

int yh_alloc_stats(uint64_t* n_driver_allocs, double* ms_in_driver, uint64_t* bytes_idle) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (n_driver_allocs) *n_driver_allocs = g_driver_allocs;
    if (ms_in_driver) *ms_in_driver = g_driver_alloc_ms;
    if (bytes_idle) {
        uint64_t b = 0;
        for (const CacheBlock& c : g_cache) b += c.bytes;
        *bytes_idle = b;
    }
    return YH_OK;
}
int yh_pool_release(uint64_t* bytes_released) {
    const uint64_t b = cache_on() ? pool_release_all() : 0;
    pin_release_idle();  // (the page-locked staging buffers too; host memory, not counted in *bytes_released)
    if (bytes_released) *bytes_released = b;
    return YH_OK;
}
int yh_device_count(int* n_devices) {
    if (!n_devices) { yh_set_error("n_devices is null"); return YH_ERR_INVALID_ARG; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_devices = 0;
        yh_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return YH_ERR_NO_DEVICE;
    }
    *n_devices = n;
    return YH_OK;
}

static int db_create_common(const u64* values, const u64* offsets, bool on_device, u64 n_refs, int device_id,
                            uint32_t flags, yh_db** out, const YhPackedCsr* pk = nullptr) {
    if (!out) { yh_set_error("out is null"); return YH_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!offsets) { yh_set_error("offsets is null"); return YH_ERR_INVALID_ARG; }
    if (n_refs > 0x7ffffff0ull) { yh_set_error("too many references (at most 2^31 - 16: the top bit of a reference id is a flag)"); return YH_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        yh_set_error("no HIP device available (libyacht_hip has no CPU fallback)");
        return YH_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) {
        yh_set_error("device_id %d out of range [0, %d)", device_id, ndev);
        return YH_ERR_NO_DEVICE;
    }
    YH_HIP(hipSetDevice(device_id));

    yh_db* db = new yh_db();
    db->device = device_id;
    db->flags = flags;
    db->n_refs = n_refs;
    int rc = YH_OK;
    u64* d_values_in = nullptr;   // arrays the build reads (borrowed or temporary)
    u64* d_offsets_in = nullptr;
    bool own_inputs = false, chunked = false, pooled_inputs = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    do {
        // a BLOCKING stream: ordered with the legacy default stream, where a caller that never heard of streams
        // (and torch, by default) fills the device buffers it hands over and reads the results back
        if (hipStreamCreateWithFlags(&db->own_stream, hipStreamDefault) != hipSuccess) {
            yh_set_error("hipStreamCreate failed"); rc = YH_ERR_HIP; break;
        }
        db->stream = db->own_stream;
        (void)hipEventCreate(&ev0);
        (void)hipEventCreate(&ev1);

        u64 H = 0;
        // (a device CSR with references: offsets[0] and offsets[n_refs] come back with the extents pass's verdict below --
        // two blocking 8-byte copies here were ~50 us of a 1.1 ms create)
        const bool ends_from_extents = on_device && n_refs > 0;
        if (ends_from_extents) {
            if (!values) {  // (null is only right for a database without a hash, which nothing here knows yet)
                u64 last = 0;
                if (hipMemcpy(&last, offsets + n_refs, sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) { yh_set_error("cannot read d_offsets[n_refs]"); rc = YH_ERR_HIP; break; }
                if (last) { yh_set_error("values is null"); rc = YH_ERR_INVALID_ARG; break; }
            }
        } else if (on_device) {
            if (hipMemcpy(&H, offsets + n_refs, sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) {
                yh_set_error("cannot read d_offsets[n_refs]"); rc = YH_ERR_HIP; break;
            }
            u64 first = 0;
            (void)hipMemcpy(&first, offsets, sizeof(u64), hipMemcpyDeviceToHost);
            if (first != 0) { yh_set_error("offsets[0] must be 0"); rc = YH_ERR_INVALID_ARG; break; }
        } else {
            if (offsets[0] != 0) { yh_set_error("offsets[0] must be 0"); rc = YH_ERR_INVALID_ARG; break; }
            H = offsets[n_refs];
        }
        if (H && !values && !pk) { yh_set_error("values is null"); rc = YH_ERR_INVALID_ARG; break; }
        db->n_hashes = H;

        if (on_device) {
            d_values_in = const_cast<u64*>(values);
            d_offsets_in = const_cast<u64*>(offsets);
        } else {
            own_inputs = true;
            pooled_inputs = !(flags & YH_DB_KEEP_CSR);  // (buffers the handle keeps are plain allocations)
            if ((pooled_inputs ? yh_tmalloc(db, (void**)&d_values_in, std::max<u64>(H, 2) * sizeof(u64))
                               : hipMalloc((void**)&d_values_in, std::max<u64>(H, 2) * sizeof(u64))) != hipSuccess ||
                (pooled_inputs ? yh_tmalloc(db, (void**)&d_offsets_in, (n_refs + 1) * sizeof(u64))
                               : hipMalloc((void**)&d_offsets_in, (n_refs + 1) * sizeof(u64))) != hipSuccess) {
                yh_set_error("device allocation for the CSR upload failed"); rc = YH_ERR_OOM; break;
            }
            (void)hipStreamSynchronize(db->stream);  // (stream-ordered allocations: there before another stream copies into them)
            // a large host database goes up in chunks that are sorted and merged while the next one crosses the bus
            // (yh_build_upload_sorted); a small one in one piece
            static const u64 chunk_min = [] { const char* e = yh_tune_env("YH_UPLOAD_CHUNK_MIN"); return e ? (u64)atoll(e) : (u64)12 << 20; }();
            chunked = pk ? true : (H >= chunk_min && n_refs >= 4);  // (a packed database always: yh_db_create_packed unpacks small ones on the host)
            const double t_up = alloc_now_ms();
            if (hipMemcpy(d_offsets_in, offsets, (n_refs + 1) * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess ||
                (!chunked && H && hipMemcpy(d_values_in, values, H * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess)) {
                yh_set_error("CSR upload failed"); rc = YH_ERR_HIP; break;
            }
            db->ms_h2d_create = (float)(alloc_now_ms() - t_up);  // (a chunked upload adds its own time: yh_build_upload_sorted)
        }

        u64* d_sk_pre = nullptr;
        u32* d_sv_pre = nullptr;
        if (chunked) {
            rc = pk ? yh_build_upload_sorted_packed(db, pk, d_values_in, d_offsets_in, &d_sk_pre, &d_sv_pre)
                    : yh_build_upload_sorted(db, values, offsets, d_values_in, d_offsets_in, &d_sk_pre, &d_sv_pre);
            if (rc != YH_OK) break;
            (void)hipEventRecord(ev0, db->stream);
        } else {
            (void)hipEventRecord(ev0, db->stream);
            // sizes + largest hash now; every sketch's ordering is checked by the sort's first level on its way through
            // (yh_build_index), or by a pass of its own where that sort does not apply
            rc = yh_build_validate_extents(db, d_values_in, d_offsets_in);  // (sets db->n_hashes of a device CSR: offsets[n_refs])
            if (rc != YH_OK) break;
            H = db->n_hashes;
        }
        rc = yh_build_index(db, d_values_in, d_offsets_in, d_sk_pre, d_sv_pre);  // (overlap-only handles too: the delta stream comes out of the same sort)
        if (rc != YH_OK) break;
        (void)hipEventRecord(ev1, db->stream);
        (void)hipEventSynchronize(ev1);
        (void)hipEventElapsedTime(&db->ms_db_build, ev0, ev1);
        db->ms_db_build += db->ms_upload_kernels;  // (device time of the chunk sorts and merges that ran under the upload)

        const u64 N = n_refs;
        if ((rc = yh_dmalloc(db, (void**)&db->d_mask, std::max<u64>(N, 1))) != YH_OK) break;
        if ((rc = yh_dmalloc(db, (void**)&db->d_excl_e, 3 * std::max<u64>(N, 1) * sizeof(u32) + 16)) != YH_OK) break;
        db->d_excl_m = db->d_excl_e + N;
        db->d_ovsh = db->d_excl_e + 2 * N;
        if ((rc = yh_dmalloc(db, (void**)&db->d_maskbits, ((N + 255) / 256) * 32 + 16)) != YH_OK) break;  // whole 256-thread blocks write it
        if ((rc = yh_dmalloc(db, (void**)&db->d_overlap_tmp, std::max<u64>(N, 1) * sizeof(u32))) != YH_OK) break;

        if (flags & YH_DB_KEEP_CSR) {
            if (own_inputs) {  // adopt the upload buffers
                db->d_values = d_values_in;
                db->d_offsets = d_offsets_in;
                db->device_bytes += std::max<u64>(H, 2) * sizeof(u64) + (n_refs + 1) * sizeof(u64);
                own_inputs = false;
            } else {
                if ((rc = yh_dmalloc(db, (void**)&db->d_values, std::max<u64>(H, 2) * sizeof(u64))) != YH_OK) break;
                if ((rc = yh_dmalloc(db, (void**)&db->d_offsets, (n_refs + 1) * sizeof(u64))) != YH_OK) break;
                // (on the handle's stream and waited for: a device-to-device hipMemcpy may return before it is done)
                if ((H && hipMemcpyAsync(db->d_values, values, H * sizeof(u64), hipMemcpyDeviceToDevice, db->stream) != hipSuccess) ||
                    hipMemcpyAsync(db->d_offsets, offsets, (n_refs + 1) * sizeof(u64), hipMemcpyDeviceToDevice, db->stream) != hipSuccess ||
                    hipStreamSynchronize(db->stream) != hipSuccess) {
                    yh_set_error("CSR copy failed"); rc = YH_ERR_HIP; break;
                }
            }
        }
        db->ev_overlap.wanted = db->ev_excl.wanted = db->ev_pair.wanted = true;
    } while (0);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (own_inputs) {
        if (pooled_inputs) { yh_tfree(db, d_values_in); yh_tfree(db, d_offsets_in); }
        else { (void)hipFree(d_values_in); (void)hipFree(d_offsets_in); }
    }
    if (db->stream) { (void)hipStreamSynchronize(db->stream); yh_pool_trim(db); }
    if (rc != YH_OK) {
        yh_db_destroy(db);
        return rc;
    }
    *out = db;
    return YH_OK;
}

int yh_db_create(const uint64_t* values, const uint64_t* offsets, uint64_t n_refs, int device_id, uint32_t flags,
                 yh_db** out) {
    return db_create_common((const u64*)values, (const u64*)offsets, false, n_refs, device_id, flags, out);
}

int yh_db_create_device(const uint64_t* d_values, const uint64_t* d_offsets, uint64_t n_refs, int device_id,
                        uint32_t flags, yh_db** out) {
    return db_create_common((const u64*)d_values, (const u64*)d_offsets, true, n_refs, device_id, flags, out);
}

// A database that arrives packed (yh_csr_pack): ~5.7 instead of 8 bytes per hash cross the bus, the chunks are expanded in HBM in
// front of their ordering checks (yh_build.hip).  Small ones (what yh_db_create uploads in one piece) are unpacked on the host.
int yh_db_create_packed(const void* packed, uint64_t packed_bytes, int device_id, uint32_t flags, yh_db** out) {
    if (!out) { yh_set_error("out is null"); return YH_ERR_INVALID_ARG; }
    *out = nullptr;
    try {  // (the view's and the host unpack's vectors are sized by the blob's header: no exception crosses the C boundary)
        YhPackedCsr v;
        YH_TRY(yh_csr_view(packed, packed_bytes, &v));
        static const u64 chunk_min = [] { const char* e = yh_tune_env("YH_UPLOAD_CHUNK_MIN"); return e ? (u64)atoll(e) : (u64)12 << 20; }();
        if (!(v.n_hashes >= chunk_min && v.n_refs >= 4)) {
            std::vector<uint64_t> values(std::max<u64>(v.n_hashes, 1)), offsets(v.n_refs + 1);
            uint64_t h = 0, n = 0;
            YH_TRY(yh_csr_unpack(packed, packed_bytes, values.data(), v.n_hashes, offsets.data(), v.n_refs, &h, &n));
            return db_create_common((const u64*)values.data(), (const u64*)offsets.data(), false, v.n_refs, device_id, flags, out);
        }
        return db_create_common(nullptr, v.offsets, false, v.n_refs, device_id, flags, out, &v);
    } catch (const std::bad_alloc&) {
        yh_set_error("yh_db_create_packed: out of host memory");
        return YH_ERR_OOM;
    } catch (const std::exception& ex) {
        yh_set_error("yh_db_create_packed: %s", ex.what());
        return YH_ERR_INVALID_ARG;
    }
}

int yh_db_destroy(yh_db* db) {
    if (!db) return YH_OK;
    if (db->device >= 0) (void)hipSetDevice(db->device);
    if (db->fin_stream) (void)hipStreamSynchronize(db->fin_stream);  // (the caller's: second halves may still be running there)
    if (db->stream) (void)hipStreamSynchronize(db->stream);
    if (db->ev_fin) (void)hipEventDestroy(db->ev_fin);
    for (yh_db::BatchSlot& bs : db->batch) if (bs.ev_first) (void)hipEventDestroy(bs.ev_first);
    if (db->tmp_psort) { yh_psort_destroy(db, db->tmp_psort); db->tmp_psort = nullptr; }  // (a create that failed half way)
    if (db->ctx_bits[0]) {  // the step contexts: back to the handle's own arrays, the second set freed here
        db->d_maskbits = db->ctx_bits[0];
        db->d_work = db->ctx_work[0];
        db->d_work_count = db->ctx_count[0];
        for (int c = 1; c < YH_RUN_CONTEXTS; ++c) {
            yh_dfree(db, db->ctx_bits[c]);
            yh_dfree(db, db->ctx_work[c]);
            yh_dfree(db, db->ctx_count[c]);
        }
    }
    void* ptrs[] = {db->d_values, db->d_offsets, db->d_sizes, db->d_g, db->d_po, db->d_pr, db->d_pg, db->d_nshared,
                    db->d_dh, db->d_dref, db->d_dir, db->d_bkt, db->d_cbkt, db->d_ovf_keys, db->d_ovf_vals,
                    db->d_rpo, db->d_rg, db->d_rrec, db->d_rrecx, db->d_filter, db->d_hrec, db->d_hrecx, db->d_hmult, db->d_hpo,
                    db->d_work, db->d_work_count, db->d_mask, db->d_maskbits, db->d_hit, db->d_excl_e, db->d_overlap_tmp,
                    db->d_sample_tmp, db->d_out_tmp, db->d_flag, db->d_reps, db->batch[0].d_scratch, db->batch[1].d_scratch, db->batch[2].d_scratch, db->d_sdelta, db->d_shdr, db->d_srec,
                    db->d_wg_key, db->d_ghost_src, db->d_bad_word, db->d_prank, db->d_fz_rec, db->d_fz_list, db->d_fz_list2, db->d_fz_off, db->d_fz_tab};
    for (void* p : ptrs)
        if (p) yh_dfree(db, p);
    for (auto& pair : db->ev_xfer)
        for (hipEvent_t e : pair) if (e) (void)hipEventDestroy(e);
    ring_destroy(db->ev_overlap);
    ring_destroy(db->ev_excl);
    ring_destroy(db->ev_pair);
    for (hipStream_t q : db->st_in) if (q) (void)hipStreamSynchronize(q);
    for (RunSlot& s : db->slots) {
        yh_dfree(db, s.d_sample);  // (yh_dfree: cache blocks back to the cache, plain allocations to hipFree)
        yh_dfree(db, s.d_packed);
        yh_dfree(db, s.d_rows);
        yh_dfree(db, s.d_out);
        yh_dfree(db, s.d_bad);
        if (s.h_bad) (void)hipHostFree(s.h_bad);
        if (s.ev_up) (void)hipEventDestroy(s.ev_up);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    }
    for (hipStream_t q : db->st_in) if (q) (void)hipStreamDestroy(q);
    if (db->stream) { (void)hipStreamSynchronize(db->stream); yh_pool_trim(db); }
    if (db->own_stream) (void)hipStreamDestroy(db->own_stream);
    free(db->h_pw_i);
    free(db->h_pw_j);
    free(db->h_pw_c);
    delete db;
    return YH_OK;
}

int yh_db_get_info(yh_db* db, yh_db_info* info) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!info) { yh_set_error("info is null"); return YH_ERR_INVALID_ARG; }
    memset(info, 0, sizeof(*info));
    info->n_refs = db->n_refs;
    info->n_hashes = db->n_hashes;
    info->max_hash = db->max_hash;
    info->n_distinct = db->n_distinct;
    info->n_shared_distinct = db->n_shared;
    info->n_shared_postings = db->n_postings;
    info->device_bytes = db->device_bytes;
    info->device_id = db->device;
    info->flags = db->flags;
    info->n_holder_sets = db->d_hrec ? db->n_sets : 0;
    info->filter_bytes = db->d_filter ? db->filter_bits / 8 : 0;
    info->sort_path = db->sort_path;
    info->n_spilled_buckets = db->n_spilled_buckets;
    info->n_spilled_pairs = db->n_spilled_pairs;
    if (db->d_sdelta) {
        info->stream_layout = YH_STREAM_DELTA;
        info->stream_shift = db->sshift;
        info->stream_bytes = db->slen + (db->slen / STREAM_BLOCK + 1) * sizeof(u64);
    }
    return YH_OK;
}

int yh_db_set_stream(yh_db* db, void* hip_stream) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    YH_TRY(db_select(db));
    YH_TRY(pipe_join(db));
    YH_HIP(hipStreamSynchronize(db->stream));
    yh_pool_trim(db);  // (the old stream has drained: cached blocks it used last are anyone's now -- it may not outlive this call)
    db->stream = hip_stream ? (hipStream_t)hip_stream : db->own_stream;
    return YH_OK;
}

int yh_db_set_batch_finish_stream(yh_db* db, void* hip_stream) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    YH_TRY(db_select(db));
    if (db->fin_stream) YH_HIP(hipStreamSynchronize(db->fin_stream));
    db->fin_pending = false;
    db->fin_stream = (hipStream_t)hip_stream;
    if (db->fin_stream) {
        if (!db->ev_fin) YH_HIP(hipEventCreateWithFlags(&db->ev_fin, hipEventDisableTiming));
        for (yh_db::BatchSlot& bs : db->batch)
            if (!bs.ev_first) YH_HIP(hipEventCreateWithFlags(&bs.ev_first, hipEventDisableTiming));
    }
    return YH_OK;
}

int yh_db_synchronize(yh_db* db) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    YH_TRY(db_select(db));
    YH_TRY(pipe_join(db));
    if (db->fin_stream) YH_HIP(hipStreamSynchronize(db->fin_stream));
    db->fin_pending = false;
    YH_HIP(hipStreamSynchronize(db->stream));
    return YH_OK;
}

int yh_db_get_timing(yh_db* db, yh_timing* t) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!t) { yh_set_error("t is null"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(pipe_join(db));
    YH_HIP(hipStreamSynchronize(db->stream));
    t->ms_overlap_kernel = ring_read(db->ev_overlap);
    t->ms_exclusive_kernels = ring_read(db->ev_excl);
    t->ms_pairwise_kernels = ring_read(db->ev_pair);
    t->ms_db_build = db->ms_db_build;
    t->ms_h2d = db->ms_h2d_create;
    t->ms_d2h = 0.f;
    for (int dir = 0; dir < 2; ++dir) {
        float ms = 0.f;
        if (db->xfer_recorded[dir] && hipEventElapsedTime(&ms, db->ev_xfer[dir][0], db->ev_xfer[dir][1]) == hipSuccess)
            (dir ? t->ms_d2h : t->ms_h2d) = ms;
        else
            (void)hipGetLastError();
    }
    return YH_OK;
}

// ---- which lookup answers a sample query ---------------------------------------------------------------
// Both are exact.  The streaming kernel reads every reference hash (one delta byte each) and stages the sample:
// ~12 us + 0.24 ns per reference hash + 14 ns per 1 000 sample hashes.  The sample-driven kernel reads a presence
// bit and (mostly for hashes that are there) one 64-byte bucket per SAMPLE hash: ~8 us + 30 ns per 1 000 sample
// hashes up to 2e6 of them, 9 ns per 1 000 beyond (a sorted sample that dense finds neighbouring buckets cached).
// Measured on MI355X, whole step, 3.3e8 reference hashes (scripts/probes/crossover.py): samples of 1e6 / 4e6 / 3.2e7
// hashes: 47 / 77 / 308 us sample-driven against 99 / 152 / 533 us streaming; 8.3e4 hashes: 30 against 104;
// 2.2e9 reference hashes, 1e7-hash sample: 313 against 568 us.  The streaming kernel wins where the
// database is small against the sample (a shard of a strongly scaled run).  YH_LOOKUP=stream|indexed in the
// environment, or yh_db_set_lookup, force one.
static bool prefer_indexed(const yh_db* db, u64 n_sample) {
    if (!db->has_dir || !db->has_index || n_sample == 0) return false;
    if (db->lookup_mode == YH_LOOKUP_STREAM) return false;
    if (db->lookup_mode == YH_LOOKUP_INDEXED) return true;
    static const int env = [] {
        const char* e = yh_tune_env("YH_LOOKUP");
        return !e ? 0 : (strcmp(e, "stream") == 0 ? 1 : (strcmp(e, "indexed") == 0 ? 2 : 0));
    }();
    if (env) return env == 2;
    if (!db->d_sdelta) return true;
    const double ns = (double)n_sample;
    const double t_stream = 12.0 + 0.24e-3 * (double)db->n_hashes + 0.014e-3 * ns;
    const double t_index = 8.0 + 0.030e-3 * std::min(ns, 2e6) + 0.009e-3 * std::max(ns - 2e6, 0.0);
    return t_index < 0.9 * t_stream;  // (a margin for hit-dense samples, where the sample-driven kernel loses most)
}

int yh_db_lookup_choice(yh_db* db, uint64_t n_sample) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    return prefer_indexed(db, n_sample) ? YH_LOOKUP_INDEXED : YH_LOOKUP_STREAM;
}

int yh_db_set_lookup(yh_db* db, int mode) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (mode < YH_LOOKUP_AUTO || mode > YH_LOOKUP_INDEXED) { yh_set_error("mode must be YH_LOOKUP_AUTO, _STREAM or _INDEXED"); return YH_ERR_INVALID_ARG; }
    if (mode == YH_LOOKUP_INDEXED && !db->has_dir) { yh_set_error("this handle has no directory of its distinct hashes"); return YH_ERR_UNSUPPORTED; }
    db->lookup_mode = mode;
    return YH_OK;
}

// ---- overlap ---------------------------------------------------------------------------------------
int yh_overlap_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    note_other_query(db);
    if (prefer_indexed(db, n_sample)) return yh_q_overlap_indexed(db, (const u64*)d_sample, n_sample, d_overlap, false);
    return yh_q_overlap(db, (const u64*)d_sample, n_sample, d_overlap, false, false);
}

int yh_overlap_indexed_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    note_other_query(db);
    return yh_q_overlap_indexed(db, (const u64*)d_sample, n_sample, d_overlap, false);
}

int yh_run_indexed_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap,
                          uint32_t* d_n_excl, uint32_t* d_n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || !d_n_excl || !d_n_match || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    note_other_query(db);
    const int rc = yh_q_overlap_indexed(db, (const u64*)d_sample, n_sample, d_overlap, true, d_n_excl, d_n_match);
    if (rc != YH_OK) return rc == 2 ? YH_OK : rc;
    return yh_q_exclusive(db, db->d_mask, (const u64*)d_sample, n_sample, d_overlap, d_n_excl, d_n_match, db->d_maskbits);
}

int yh_run_batch_device(yh_db* db, const uint64_t* d_samples, const uint64_t* d_sample_offsets, uint32_t n_samples,
                        uint64_t total_hashes, uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_sample_offsets || !d_overlap || !d_n_excl || !d_n_match || (total_hashes && !d_samples)) {
        yh_set_error("null device pointer");
        return YH_ERR_INVALID_ARG;
    }
    YH_TRY(db_select(db));
    note_other_query(db);
    if (db->batch[0].open) db->batch[0].clobbered = true;  // (whole-batch calls run in slot 0)
    return yh_q_run_batch(db, (const u64*)d_samples, (const u64*)d_sample_offsets, n_samples, total_hashes, d_overlap,
                          d_n_excl, d_n_match);
}

int yh_run_batch(yh_db* db, const uint64_t* samples, const uint64_t* sample_offsets, uint32_t n_samples,
                 uint32_t* overlap, uint32_t* n_excl, uint32_t* n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!sample_offsets || n_samples < 1 || n_samples > YH_BATCH_MAX_SAMPLES) { yh_set_error("1..%d samples and their offsets", YH_BATCH_MAX_SAMPLES); return YH_ERR_INVALID_ARG; }
    const u64 total = sample_offsets[n_samples];
    const u64 N = db->n_refs;
    if (sample_offsets[0] != 0 || (total && !samples) || (N && (!overlap || !n_excl || !n_match))) {
        yh_set_error("bad sample arrays or null output");
        return YH_ERR_INVALID_ARG;
    }
    for (uint32_t s = 0; s < n_samples; ++s) {
        if (sample_offsets[s + 1] < sample_offsets[s] ||
            yh_q_check_sorted_host((const u64*)samples + sample_offsets[s], sample_offsets[s + 1] - sample_offsets[s]) != YH_OK) {
            yh_set_error("sample %u is not strictly ascending", s);
            return YH_ERR_UNSORTED;
        }
    }
    YH_TRY(db_select(db));
    if (N == 0) return YH_OK;
    const u64 BN = (u64)n_samples * N;
    u64 *d_s = nullptr, *d_o = nullptr;
    u32* d_out = nullptr;
    int rc = YH_OK;
    do {
        // (from the buffer cache: a caller that batches sample after sample does not go to the driver every time)
        if (yh_tmalloc(db, (void**)&d_s, std::max<u64>(total, 2) * sizeof(u64)) != hipSuccess ||
            yh_tmalloc(db, (void**)&d_o, (u64)(n_samples + 1) * sizeof(u64)) != hipSuccess ||
            yh_tmalloc(db, (void**)&d_out, 3 * BN * sizeof(u32)) != hipSuccess) { yh_set_error("device allocation failed"); rc = YH_ERR_OOM; break; }
        xfer_mark(db, 0, 0);
        const bool up_ok = (!total || hipMemcpyAsync(d_s, samples, total * sizeof(u64), hipMemcpyHostToDevice, db->stream) == hipSuccess) &&
                           hipMemcpyAsync(d_o, sample_offsets, (u64)(n_samples + 1) * sizeof(u64), hipMemcpyHostToDevice, db->stream) == hipSuccess;
        xfer_mark(db, 0, 1);
        if (!up_ok) { yh_set_error("sample upload failed"); rc = YH_ERR_HIP; break; }
        note_other_query(db);
        if (db->batch[0].open) db->batch[0].clobbered = true;
        if ((rc = yh_q_run_batch(db, d_s, d_o, n_samples, total, d_out, d_out + BN, d_out + 2 * BN)) != YH_OK) break;
        xfer_mark(db, 1, 0);
        const bool down_ok = hipMemcpyAsync(overlap, d_out, BN * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess &&
                             hipMemcpyAsync(n_excl, d_out + BN, BN * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess &&
                             hipMemcpyAsync(n_match, d_out + 2 * BN, BN * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess;
        xfer_mark(db, 1, 1);
        if (!down_ok || hipStreamSynchronize(db->stream) != hipSuccess) {
            yh_set_error("batch download failed: %s", hipGetErrorString(hipGetLastError())); rc = YH_ERR_HIP;
        }
    } while (0);
    yh_tfree(db, d_s); yh_tfree(db, d_o); yh_tfree(db, d_out);
    return rc;
}

int yh_overlap_bsearch_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    return yh_q_overlap_bsearch(db, (const u64*)d_sample, n_sample, d_overlap);
}

// flag = 1 when a[i - 1] >= a[i] somewhere
__global__ void k_check_ascending(const u64* __restrict__ a, u64 n, u32* __restrict__ flag) {
    bool bad = false;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x + 1; i < n; i += (u64)gridDim.x * blockDim.x)
        bad |= !(a[i - 1] < a[i]);
    if (bad) *flag = 1;
}

// Sets *flag (device) and *host_flag (page-locked host memory, written through PCIe: visible to the host once the kernel
// has completed) when a[i - 1] >= a[i] somewhere.
__global__ void k_check_ascending2(const u64* __restrict__ a, u64 n, u32* __restrict__ flag, u32 gen, u32* __restrict__ host_flag) {
    bool bad = false;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x + 1; i < n; i += (u64)gridDim.x * blockDim.x)
        bad |= !(a[i - 1] < a[i]);
    if (bad) {
        *flag = gen;
        if (host_flag) *host_flag = 1;
    }
}
// device -> page-locked host memory by stores through PCIe (16 bytes per lane), in the step's own stream:
// HIP's own path for this direction is a blit kernel with ~25 us of gaps around it (traced)
__global__ void __launch_bounds__(256) k_copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst, u64 n16) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) dst[i] = src[i];
}
static u32 next_gen(yh_db* db) {
    if (++db->bad_gen == 0) ++db->bad_gen;  // (0 is what a fresh flag word holds)
    return db->bad_gen;
}
// the device address of a host buffer the GPU can store to directly, or null (pageable memory)
static void* device_view_of_host(void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
}

// Host sample -> d_sample_tmp, and the ordering check ON THE DEVICE (a 10^6-hash sample is 8 MB: the
// host loop over it was ~0.4 ms of the ~0.55 ms a host-pointer query took).  The query kernels assume
// an ascending sample, so the verdict is awaited before they are queued.
static int upload_sample(yh_db* db, const uint64_t* sample, uint64_t n_sample, bool defer_verdict = false) {
    if (n_sample && !sample) { yh_set_error("sample is null"); return YH_ERR_INVALID_ARG; }
    YH_TRY(ensure_sample_tmp(db, n_sample));
    if (n_sample < 2) {
        if (n_sample)
            YH_HIP(hipMemcpyAsync(db->d_sample_tmp, sample, n_sample * sizeof(u64), hipMemcpyHostToDevice, db->stream));
        return YH_OK;
    }
    u32 verdict = 0;
    xfer_mark(db, 0, 0);
    YH_HIP(hipMemcpyAsync(db->d_sample_tmp, sample, n_sample * sizeof(u64), hipMemcpyHostToDevice, db->stream));
    xfer_mark(db, 0, 1);
    if (defer_verdict) {
        // the kernels queued next read the verdict themselves (StreamHit::bad) and look nothing up when the
        // sample failed; the caller downloads it with the counts -- one host sync per call, not two
        k_check_ascending2<<<(unsigned)std::min<u64>((n_sample + 255) / 256, 2048), 256, 0, db->stream>>>(
            db->d_sample_tmp, n_sample, db->d_bad_word, db->bad_gen, nullptr);
        return YH_OK;
    }
    YH_HIP(hipMemsetAsync(db->d_flag, 0, sizeof(u32), db->stream));
    k_check_ascending<<<(unsigned)std::min<u64>((n_sample + 255) / 256, 2048), 256, 0, db->stream>>>(db->d_sample_tmp, n_sample,
                                                                                                  db->d_flag);
    YH_HIP(hipMemcpyAsync(&verdict, db->d_flag, sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    if (verdict) {
        yh_set_error("the sample sketch is not strictly ascending");
        return YH_ERR_UNSORTED;
    }
    return YH_OK;
}

// two [N] uint32 result arrays of the host-pointer entry points, kept with the handle
static int ensure_out_tmp(yh_db* db, u32** d_e, u32** d_m) {
    const u64 N = std::max<u64>(db->n_refs, 1);
    if (!db->d_out_tmp) YH_TRY(yh_dmalloc(db, (void**)&db->d_out_tmp, 2 * N * sizeof(u32) + 16));
    *d_e = db->d_out_tmp;
    *d_m = db->d_out_tmp + N;
    return YH_OK;
}

static int overlap_host(yh_db* db, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap, bool bsearch) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!overlap && db->n_refs) { yh_set_error("overlap is null"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(upload_sample(db, sample, n_sample));
    if (bsearch) YH_TRY(yh_q_overlap_bsearch(db, db->d_sample_tmp, n_sample, db->d_overlap_tmp));
    else YH_TRY(yh_overlap_device(db, (const uint64_t*)db->d_sample_tmp, n_sample, db->d_overlap_tmp));
    xfer_mark(db, 1, 0);
    if (db->n_refs)
        YH_HIP(hipMemcpyAsync(overlap, db->d_overlap_tmp, db->n_refs * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    xfer_mark(db, 1, 1);
    YH_HIP(hipStreamSynchronize(db->stream));
    return YH_OK;
}

int yh_overlap(yh_db* db, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap) {
    return overlap_host(db, sample, n_sample, overlap, false);
}
int yh_overlap_bsearch(yh_db* db, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap) {
    return overlap_host(db, sample, n_sample, overlap, true);
}

// ---- exclusive -------------------------------------------------------------------------------------
int yh_exclusive(yh_db* db, const uint8_t* subset_mask, const uint64_t* sample, uint64_t n_sample,
                 uint32_t* n_excl, uint32_t* n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    const u64 N = db->n_refs;
    if (N && (!subset_mask || !n_excl || !n_match)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    YH_TRY(db_select(db));
    note_other_query(db);
    YH_TRY(upload_sample(db, sample, n_sample));
    if (N == 0) return YH_OK;
    u32 *d_e = nullptr, *d_m = nullptr;
    YH_TRY(ensure_out_tmp(db, &d_e, &d_m));
    int rc = YH_OK;
    do {
        if (hipMemcpyAsync(db->d_mask, subset_mask, N, hipMemcpyHostToDevice, db->stream) != hipSuccess) { yh_set_error("mask upload failed"); rc = YH_ERR_HIP; break; }
        if ((rc = yh_q_overlap(db, db->d_sample_tmp, n_sample, db->d_overlap_tmp, true, false)) != YH_OK) break;
        rc = yh_q_exclusive(db, db->d_mask, db->d_sample_tmp, n_sample, db->d_overlap_tmp, d_e, d_m, nullptr);
        if (rc != YH_OK) break;
        xfer_mark(db, 1, 0);
        const bool down_ok = hipMemcpyAsync(n_excl, d_e, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess &&
                             hipMemcpyAsync(n_match, d_m, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess;
        xfer_mark(db, 1, 1);
        if (!down_ok || hipStreamSynchronize(db->stream) != hipSuccess) {
            yh_set_error("exclusive download failed: %s", hipGetErrorString(hipGetLastError()));
            rc = YH_ERR_HIP;
        }
    } while (0);
    return rc;
}

int yh_run_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap,
                  uint32_t* d_n_excl, uint32_t* d_n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    if ((d_n_excl == nullptr) != (d_n_match == nullptr)) { yh_set_error("pass both d_n_excl and d_n_match or neither"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    note_other_query(db);
    if (prefer_indexed(db, n_sample)) {
        if (!d_n_excl) return yh_q_overlap_indexed(db, (const u64*)d_sample, n_sample, d_overlap, false);
        return yh_run_indexed_device(db, d_sample, n_sample, d_overlap, d_n_excl, d_n_match);
    }
    if (d_n_excl) {  // three launches when the handle holds the hash-sorted stream
        const int rc = yh_q_run_fused(db, (const u64*)d_sample, n_sample, d_overlap, d_n_excl, d_n_match);
        if (rc != 1) return rc;
    }
    YH_TRY(yh_q_overlap(db, (const u64*)d_sample, n_sample, d_overlap, d_n_excl != nullptr, d_n_excl != nullptr));
    if (!d_n_excl) return YH_OK;
    return yh_q_exclusive(db, db->d_mask, (const u64*)d_sample, n_sample, d_overlap, d_n_excl, d_n_match, db->d_maskbits);
}

static int use_ctx(yh_db* db, int c);

// Throughput form of yh_run_device for a caller with many samples in HBM: every call is ONE launch (k_step_fused) that
// looks up this sample, reduces the previous one and runs the exclusive pass of the one before.  A call's three output
// rows are complete, in the order of the handle's stream, after two further pipelined calls or after
// yh_run_device_join (which every other query, yh_db_synchronize and yh_db_set_stream perform first by themselves).
int yh_run_device_pipelined(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap,
                            uint32_t* d_n_excl, uint32_t* d_n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || !d_n_excl || !d_n_match || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    // only the large-sample form of the sample-driven step is fused; everything else runs as yh_run_device does
    if (!prefer_indexed(db, n_sample) || !yh_q_step_fused_ok(db, n_sample))
        return yh_run_device(db, d_sample, n_sample, d_overlap, d_n_excl, d_n_match);
    if (!db->ctx_bits[2]) {  // the three step contexts the stages rotate through
        const int now = db->ctx_now;
        for (int c = 0; c < 3; ++c) YH_TRY(use_ctx(db, c));
        YH_TRY(use_ctx(db, now));
    }
    for (int c = 0; c < 3; ++c)
        if (db->ctx_open[c]) db->ctx_clobbered[c] = true;
    return yh_q_step_fused(db, (const u64*)d_sample, n_sample, d_overlap, d_n_excl, d_n_match);
}

int yh_run_device_join(yh_db* db) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    YH_TRY(db_select(db));
    return pipe_join(db);
}

int yh_run(yh_db* db, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap, uint32_t* n_excl,
           uint32_t* n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    const u64 N = db->n_refs;
    if (N && (!overlap || !n_excl || !n_match)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    YH_TRY(db_select(db));
    const bool defer = db->d_sdelta != nullptr;  // the stream kernel honours the device-side verdict
    if (defer) {
        if (!db->d_bad_word) {
            YH_TRY(yh_dmalloc(db, (void**)&db->d_bad_word, 16));
            YH_HIP(hipMemsetAsync(db->d_bad_word, 0, 16, db->stream));
        }
        next_gen(db);
    }
    YH_TRY(upload_sample(db, sample, n_sample, defer));
    if (N == 0) {
        if (defer) YH_HIP(hipStreamSynchronize(db->stream));
        return YH_OK;
    }
    u32 *d_e = nullptr, *d_m = nullptr;
    YH_TRY(ensure_out_tmp(db, &d_e, &d_m));
    if (defer) db->d_bad = db->d_bad_word;
    int rc = yh_run_device(db, (const uint64_t*)db->d_sample_tmp, n_sample, db->d_overlap_tmp, d_e, d_m);
    db->d_bad = nullptr;
    u32 verdict = 0;
    if (rc == YH_OK) {
        xfer_mark(db, 1, 0);
        const bool down_ok = hipMemcpyAsync(overlap, db->d_overlap_tmp, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess &&
                             hipMemcpyAsync(n_excl, d_e, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess &&
                             hipMemcpyAsync(n_match, d_m, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess &&
                             (!defer || hipMemcpyAsync(&verdict, db->d_bad_word, sizeof(u32), hipMemcpyDeviceToHost, db->stream) == hipSuccess);
        xfer_mark(db, 1, 1);
        if (!down_ok || hipStreamSynchronize(db->stream) != hipSuccess) {
            yh_set_error("run download failed: %s", hipGetErrorString(hipGetLastError()));
            rc = YH_ERR_HIP;
        }
    }
    if (rc == YH_OK && defer && verdict == db->bad_gen) {
        yh_set_error("the sample sketch is not strictly ascending");
        rc = YH_ERR_UNSORTED;
    }
    return rc;
}

// ---- sharded run: the step in two halves around the exchange of the subset bits -----------------------
int yh_db_set_ghosts(yh_db* db, uint64_t ghost_begin, uint64_t n_ghost, const uint32_t* d_ghost_src) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if ((ghost_begin & 63u) || ghost_begin + n_ghost > db->n_refs || (n_ghost && !d_ghost_src)) {
        yh_set_error("ghost range must start at a multiple of 64 and lie inside the handle's references");
        return YH_ERR_INVALID_ARG;
    }
    YH_TRY(db_select(db));
    YH_HIP(hipStreamSynchronize(db->stream));
    if (db->d_ghost_src) { yh_dfree(db, db->d_ghost_src); db->d_ghost_src = nullptr; }
    db->ghost_begin = ghost_begin;
    db->n_ghost = n_ghost;
    if (n_ghost) {
        YH_TRY(yh_dmalloc(db, (void**)&db->d_ghost_src, n_ghost * sizeof(u32)));
        YH_HIP(hipMemcpyAsync(db->d_ghost_src, d_ghost_src, n_ghost * sizeof(u32), hipMemcpyDeviceToDevice, db->stream));
        YH_HIP(hipStreamSynchronize(db->stream));
    }
    return YH_OK;
}

// point the handle's step state (subset bits, work list) at context c
static int use_ctx(yh_db* db, int c) {
    if (c < 0 || c >= YH_RUN_CONTEXTS) { yh_set_error("step context must be in [0, %d)", YH_RUN_CONTEXTS); return YH_ERR_INVALID_ARG; }
    if (!db->ctx_bits[0]) {  // first use: context 0 = the handle's own arrays
        db->ctx_bits[0] = db->d_maskbits;
        db->ctx_work[0] = db->d_work;
        db->ctx_count[0] = db->d_work_count;
    }
    if (c > 0 && !db->ctx_bits[c]) {
        const u64 N = db->n_refs;
        YH_TRY(yh_dmalloc(db, (void**)&db->ctx_bits[c], ((N + 255) / 256) * 32 + 16));
        YH_HIP(hipMemsetAsync(db->ctx_bits[c], 0, ((N + 255) / 256) * 32 + 16, db->stream));
        if (db->ctx_work[0]) {
            YH_TRY(yh_dmalloc(db, (void**)&db->ctx_work[c], ((u64)db->n_chunks + 64) * sizeof(uint4)));
            YH_TRY(yh_dmalloc(db, (void**)&db->ctx_count[c], 16));
            YH_HIP(hipMemsetAsync(db->ctx_count[c], 0, 16, db->stream));
        }
    }
    db->d_maskbits = db->ctx_bits[c];
    db->d_work = db->ctx_work[c];
    db->d_work_count = db->ctx_count[c];
    db->ctx_now = c;
    return YH_OK;
}

int yh_run_local_device(yh_db* db, int ctx, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap, uint32_t* d_n_excl,
                        uint32_t* d_n_match, uint32_t* d_bits_out) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || !d_n_excl || !d_n_match || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(pipe_leave(db));  // (pending stages of pipelined steps read the step contexts 0..2 this call may write)
    YH_TRY(use_ctx(db, ctx));
    db->ctx_open[ctx] = true;
    db->ctx_clobbered[ctx] = false;
    const int rc = yh_q_run_fused(db, (const u64*)d_sample, n_sample, d_overlap, d_n_excl, d_n_match, 1, d_bits_out, nullptr,
                                  prefer_indexed(db, n_sample));
    if (rc == 1) { yh_set_error("yh_run_local_device needs a non-empty handle in the default layout with its index"); return YH_ERR_UNSUPPORTED; }
    return rc;
}

int yh_run_finish_device(yh_db* db, int ctx, const uint32_t* d_global_bits, uint32_t* d_n_excl) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_n_excl || (db->n_ghost && !d_global_bits)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(pipe_leave(db));
    YH_TRY(use_ctx(db, ctx));
    const bool clobbered = db->ctx_open[ctx] && db->ctx_clobbered[ctx];
    db->ctx_open[ctx] = false;
    db->ctx_clobbered[ctx] = false;
    if (clobbered) {
        yh_set_error("another query ran on the handle between yh_run_local_device and yh_run_finish_device of context %d: its work list is gone", ctx);
        return YH_ERR_INVALID_ARG;
    }
    const int rc = yh_q_run_fused(db, nullptr, 1, nullptr, d_n_excl, nullptr, 2, nullptr, d_global_bits);
    if (rc == 1) { yh_set_error("yh_run_finish_device needs a non-empty handle in the default layout with its index"); return YH_ERR_UNSUPPORTED; }
    return rc;
}

// ---- hash-range shards: the step in two halves around ONE exchange, no ghosts ------------------------------------
int yh_run_local_range_device(yh_db* db, int ctx, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap,
                              uint32_t* d_n_match, uint32_t* d_bits_out) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_overlap || !d_n_match || !d_bits_out || (n_sample && !d_sample)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    if (db->n_ghost) { yh_set_error("a handle with ghosts is a reference shard, not a hash-range shard"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(pipe_leave(db));
    YH_TRY(use_ctx(db, ctx));
    db->ctx_open[ctx] = true;
    db->ctx_clobbered[ctx] = false;
    db->range_local = true;
    // (d_n_match doubles as the non-null "fused" marker; with range_local nothing is written through the n_excl slot)
    const int rc = yh_q_run_fused(db, (const u64*)d_sample, n_sample, d_overlap, d_n_match, d_n_match, 1, d_bits_out, nullptr,
                                  prefer_indexed(db, n_sample));
    db->range_local = false;
    if (rc == 1) { yh_set_error("yh_run_local_range_device needs a non-empty handle in the default layout with its index"); return YH_ERR_UNSUPPORTED; }
    return rc;
}

int yh_run_finish_range_device(yh_db* db, int ctx, const uint32_t* d_gathered_bits, uint32_t n_ranks, uint64_t stride_words,
                               uint32_t* d_n_excl) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_n_excl || !d_gathered_bits || n_ranks < 1) { yh_set_error("null device pointer / no ranks"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(pipe_leave(db));
    YH_TRY(use_ctx(db, ctx));
    const bool clobbered = db->ctx_open[ctx] && db->ctx_clobbered[ctx];
    db->ctx_open[ctx] = false;
    db->ctx_clobbered[ctx] = false;
    if (clobbered) {
        yh_set_error("another query ran on the handle between the two halves of context %d: its work list is gone", ctx);
        return YH_ERR_INVALID_ARG;
    }
    return yh_q_range_finish(db, d_gathered_bits, n_ranks, stride_words, d_n_excl);
}

// the batched run on a hash-range shard: up to YH_BATCH_MAX_SAMPLES samples per call around ONE exchange of their subset words
static_assert(YH_BATCH_SLOTS == 3, "yh_db_destroy frees the scratch of three batch slots");
static bool batch_slot_ok(int slot) {
    if (slot >= 0 && slot < YH_BATCH_SLOTS) return true;
    yh_set_error("batch slot must be in [0, %d)", YH_BATCH_SLOTS);
    return false;
}
int yh_run_batch_local_range_device(yh_db* db, int slot, const uint64_t* d_samples, const uint64_t* d_sample_offsets, uint32_t n_samples,
                                    uint64_t total_hashes, uint32_t* d_overlap, uint64_t* d_maskwords_out) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!batch_slot_ok(slot)) return YH_ERR_INVALID_ARG;
    if (!d_sample_offsets || !d_overlap || !d_maskwords_out || (total_hashes && !d_samples)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    note_other_query(db, /*join_fin=*/false);  // (a first half touches nothing a second half on the finish stream uses)
    const int rc = yh_q_run_batch(db, (const u64*)d_samples, (const u64*)d_sample_offsets, n_samples, total_hashes, d_overlap, nullptr,
                                  nullptr, 1, (u64*)d_maskwords_out, nullptr, 0, slot);
    yh_db::BatchSlot& bs = db->batch[slot];
    bs.open = rc == YH_OK;
    bs.clobbered = false;
    return rc;
}

int yh_run_batch_finish_range_device(yh_db* db, int slot, uint32_t n_samples, const uint64_t* d_gathered_maskwords, uint32_t n_ranks,
                                     const uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!batch_slot_ok(slot)) return YH_ERR_INVALID_ARG;
    if (!d_gathered_maskwords || n_ranks < 1 || !d_overlap || !d_n_excl || !d_n_match) { yh_set_error("null device pointer / no ranks"); return YH_ERR_INVALID_ARG; }
    yh_db::BatchSlot& bs = db->batch[slot];
    const bool was_open = bs.open, clobbered = bs.clobbered;
    bs.open = false;
    bs.clobbered = false;
    if (db->n_refs && (!was_open || clobbered || bs.n_samples != n_samples)) {
        yh_set_error(!was_open ? "batch slot %d holds no first half (yh_run_batch_local_range_device)"
                     : clobbered ? "another batched call ran in batch slot %d between the two halves: the first half's state is gone"
                                 : "batch slot %d: the second half names a different number of samples than the first", slot);
        return YH_ERR_INVALID_ARG;
    }
    YH_TRY(db_select(db));
    note_other_query(db, /*join_fin=*/false);  // (the second half rewrites the current step context's subset bits and work list)
    return yh_q_run_batch(db, nullptr, nullptr, n_samples, 0, const_cast<uint32_t*>(d_overlap), d_n_excl, d_n_match, 2, nullptr,
                          (const u64*)d_gathered_maskwords, n_ranks, slot);
}

// the compact form of a batch's result: one entry per set bit of the slot's subset words (yh_batch.hip)
static int batch_rows_check(yh_db* db, int slot) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!batch_slot_ok(slot)) return YH_ERR_INVALID_ARG;
    if (db->n_refs && (!db->batch[slot].words_valid || db->batch[slot].open)) {
        yh_set_error("batch slot %d holds no completed batch (yh_run_batch_device / yh_run_batch_finish_range_device first)", slot);
        return YH_ERR_INVALID_ARG;
    }
    return db_select(db);
}
int yh_run_batch_rows_pack_device(yh_db* db, int slot, const uint32_t* d_overlap, const uint32_t* d_n_excl, const uint32_t* d_n_match,
                                  uint32_t* d_vals, uint64_t cap_rows, uint32_t* d_n_rows) {
    YH_TRY(batch_rows_check(db, slot));
    if (!d_n_rows || (db->n_refs && (!d_overlap || !d_n_excl || !d_n_match)) || (cap_rows && !d_vals)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    return yh_q_batch_rows_pack(db, slot, d_overlap, d_n_excl, d_n_match, d_vals, cap_rows, d_n_rows);
}
int yh_run_batch_rows_unpack_device(yh_db* db, int slot, const uint32_t* d_vals, uint64_t cap_rows, yh_batch_row* d_rows,
                                    uint32_t* d_n_rows) {
    YH_TRY(batch_rows_check(db, slot));
    if (!d_n_rows || (cap_rows && (!d_vals || !d_rows))) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    // (runs on the handle's stream and reads the slot's subset words + rewrites its block counts, both written on the finish
    // stream by the second half and the rows pack: behind them -- ADVICE r05)
    fin_join(db);
    return yh_q_batch_rows_unpack(db, slot, d_vals, cap_rows, d_rows, d_n_rows);
}

uint64_t yh_run_batch_words_packed_len(uint64_t cap_words) { return yh_batch_words_packed_len(cap_words); }
static bool batch_planes_ok(const yh_db* db, uint32_t n_planes) {
    if (n_planes >= 1 && n_planes <= YH_BATCH_PLANES(YH_BATCH_MAX_SAMPLES) && (u64)n_planes * db->n_refs < 0xffffffffull) return true;
    yh_set_error("n_planes must be 1..%d (and n_planes * N below 2^32: an entry's id is 32 bits)", (int)YH_BATCH_PLANES(YH_BATCH_MAX_SAMPLES));
    return false;
}
int yh_run_batch_words_pack_device(yh_db* db, const uint64_t* d_words, uint32_t n_planes, uint64_t* d_packed, uint64_t cap_words) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_packed || (db->n_refs && !d_words)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    if (!batch_planes_ok(db, n_planes)) return YH_ERR_INVALID_ARG;
    YH_TRY(db_select(db));
    return yh_q_batch_words_pack(db, (const u64*)d_words, n_planes, (u64*)d_packed, cap_words);
}
int yh_run_batch_words_unpack_device(yh_db* db, const uint64_t* d_gathered, uint32_t n_ranks, uint32_t n_planes, uint64_t cap_words,
                                     uint64_t* d_words_out, uint32_t* d_overflow) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_gathered || !d_overflow || (db->n_refs && !d_words_out)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    if (n_ranks < 1 || n_ranks > 65535) { yh_set_error("n_ranks out of range"); return YH_ERR_INVALID_ARG; }
    if (!batch_planes_ok(db, n_planes)) return YH_ERR_INVALID_ARG;
    YH_TRY(db_select(db));
    return yh_q_batch_words_unpack(db, (const u64*)d_gathered, n_ranks, n_planes, cap_words, (u64*)d_words_out, d_overflow);
}

// ---- pipelined host-buffer run calls ---------------------------------------------------------------
static int slot_prepare(yh_db* db, RunSlot& s, u64 n_sample, u64 packed_bytes, bool rows_staging) {
    const u64 N = std::max<u64>(db->n_refs, 1);
    for (hipStream_t& q : db->st_in) if (!q) YH_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    if (!s.ev_up) {
        YH_HIP(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
        YH_HIP(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    }
    if (!s.d_out) YH_TRY(yh_dmalloc(db, (void**)&s.d_out, 3 * N * sizeof(u32) + 16));
    if (!s.d_bad) {
        YH_TRY(yh_dmalloc(db, (void**)&s.d_bad, 16));
        YH_HIP(hipMemsetAsync(s.d_bad, 0, 16, db->stream));
        // two page-locked words the kernels write through PCIe: [0] the ordering verdict, [1] the number of rows
        YH_HIP(hipHostMalloc((void**)&s.h_bad, 16, hipHostMallocDefault));
        YH_HIP(hipHostGetDevicePointer((void**)&s.h_bad_dev, s.h_bad, 0));
    }
    if (s.cap < n_sample) {
        if (s.d_sample) { (void)hipFree(s.d_sample); s.d_sample = nullptr; s.cap = 0; }
        const u64 cap = std::max<u64>(n_sample + n_sample / 8, 1024);
        YH_HIP(hipMalloc((void**)&s.d_sample, cap * sizeof(u64)));
        s.cap = cap;
    }
    if (s.packed_cap < packed_bytes) {
        if (s.d_packed) { (void)hipFree(s.d_packed); s.d_packed = nullptr; s.packed_cap = 0; }
        const u64 cap = packed_bytes + packed_bytes / 8 + 4096;
        YH_HIP(hipMalloc(&s.d_packed, cap));
        s.packed_cap = cap;
    }
    if (rows_staging && !s.d_rows) YH_TRY(yh_dmalloc(db, (void**)&s.d_rows, N * sizeof(uint4)));
    return YH_OK;
}

// One call in flight: upload (raw u64 hashes, or a packed sample that is expanded in HBM) on the copy-in stream; ordering
// check + kernels + the way back (three dense count rows, or the compact rows of the references with overlap > 0) on the
// handle's stream.
static int submit_common(yh_db* db, int slot, const void* sample, u64 n_or_bytes, bool packed, bool want_rows,
                         uint32_t* overlap, uint32_t* n_excl, uint32_t* n_match, yh_run_row* rows, u64 cap_rows) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (slot < 0 || slot >= YH_RUN_SLOTS) { yh_set_error("slot %d out of range [0, %d)", slot, YH_RUN_SLOTS); return YH_ERR_INVALID_ARG; }
    const u64 N = db->n_refs;
    if (!want_rows && N && (!overlap || !n_excl || !n_match)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (want_rows && cap_rows && !rows) { yh_set_error("null row buffer"); return YH_ERR_INVALID_ARG; }
    if (n_or_bytes && !sample) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    // (a database without a single hash has no stream either: every count is zero, as yh_run gives it)
    if (!db->has_index || (!db->d_sdelta && db->n_hashes)) { yh_set_error("the pipelined run calls need a handle in the default layout with its index"); return YH_ERR_UNSUPPORTED; }
    u64 n_sample = n_or_bytes, packed_bytes = 0;
    if (packed) {
        packed_bytes = n_or_bytes;
        YH_TRY(yh_pack_validate(sample, packed_bytes, &n_sample));
    }
    if (n_sample > 0xfffffff0ull) { yh_set_error("sample larger than 2^32-16 hashes"); return YH_ERR_INVALID_ARG; }
    RunSlot& s = db->slots[slot];
    if (s.busy) { yh_set_error("slot %d is in flight: yh_run_wait it first", slot); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    note_other_query(db);
    void* rows_dev = nullptr;  // where k_compact_rows stores: the caller's page-locked buffer, or the slot's staging rows
    if (want_rows && cap_rows) rows_dev = device_view_of_host(rows);
    const bool rows_staged = want_rows && cap_rows && !rows_dev;
    YH_TRY(slot_prepare(db, s, n_sample, packed_bytes, rows_staged));
    s.h_bad[0] = 0;
    s.h_bad[1] = 0;
    static const bool one_up = [] { const char* e = yh_tune_env("YH_ONE_UPLOAD_STREAM"); return e && e[0] == '1'; }();
    // (two engines only for the packed form: two raw 8 MB copies at once share the link and complete in bursts -- p90 0.35 ms)
    hipStream_t up = db->st_in[(packed && !one_up) ? (slot & 1) : 0];
    if (packed) {
        if (packed_bytes) YH_HIP(hipMemcpyAsync(s.d_packed, sample, packed_bytes, hipMemcpyHostToDevice, up));
    } else if (n_sample) {
        YH_HIP(hipMemcpyAsync(s.d_sample, sample, n_sample * sizeof(u64), hipMemcpyHostToDevice, up));
    }
    YH_HIP(hipEventRecord(s.ev_up, up));
    // (The way back stays on the handle's stream: HIP moves device -> pinned host with blit kernels, and a third stream
    // only added cross-stream waits in front of every step -- traced in round 2.)
    YH_HIP(hipStreamWaitEvent(db->stream, s.ev_up, 0));
    const u32 gen = next_gen(db);
    if (packed)
        YH_TRY(yh_pack_expand_device(db, s.d_packed, n_sample, s.d_sample, s.d_bad, gen, s.h_bad_dev));
    else if (n_sample > 1)
        k_check_ascending2<<<(unsigned)std::min<u64>((n_sample + 255) / 256, 2048), 256, 0, db->stream>>>(s.d_sample, n_sample, s.d_bad,
                                                                                                         gen, s.h_bad_dev);
    int rc = YH_OK;
    if (N) {
        db->d_bad = s.d_bad;
        rc = yh_run_device(db, (const uint64_t*)s.d_sample, n_sample, s.d_out, s.d_out + N, s.d_out + 2 * N);
        db->d_bad = nullptr;
    }
    if (rc != YH_OK) return rc;
    s.rows_cap = cap_rows;
    s.rows_mode = want_rows;
    if (want_rows) {
        YH_TRY(yh_rows_compact_device(db, s.d_out, s.d_out + N, s.d_out + 2 * N, rows_staged ? (void*)s.d_rows : rows_dev,
                                      cap_rows, nullptr, s.h_bad_dev + 1));
        if (rows_staged && N)  // pageable row buffer: as many rows as it holds (their number is not known on the host yet)
            YH_HIP(hipMemcpyAsync(rows, s.d_rows, std::min<u64>(cap_rows, N) * sizeof(uint4), hipMemcpyDeviceToHost, db->stream));
    } else if (N) {
        const bool one_block = n_excl == overlap + N && n_match == n_excl + N;  // one contiguous [3][N] host buffer
        void* dv = nullptr;
        if (one_block && (3 * N) % 4 == 0 && ((uintptr_t)overlap & 15u) == 0)
            dv = device_view_of_host(overlap);  // asked on every call: the same address may be pageable memory by now
        if (dv) {
            k_copy_out<<<(unsigned)std::min<u64>((3 * N / 4 + 255) / 256, 512), 256, 0, db->stream>>>(
                reinterpret_cast<const uint4*>(s.d_out), reinterpret_cast<uint4*>(dv), 3 * N / 4);
        } else if (one_block) {
            YH_HIP(hipMemcpyAsync(overlap, s.d_out, 3 * N * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
        } else {
            YH_HIP(hipMemcpyAsync(overlap, s.d_out, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
            YH_HIP(hipMemcpyAsync(n_excl, s.d_out + N, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
            YH_HIP(hipMemcpyAsync(n_match, s.d_out + 2 * N, N * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
        }
    }
    YH_HIP(hipEventRecord(s.ev_out, db->stream));
    s.busy = true;
    return YH_OK;
}

int yh_run_submit(yh_db* db, int slot, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap,
                  uint32_t* n_excl, uint32_t* n_match) {
    return submit_common(db, slot, sample, n_sample, false, false, overlap, n_excl, n_match, nullptr, 0);
}
int yh_run_submit_rows(yh_db* db, int slot, const uint64_t* sample, uint64_t n_sample, yh_run_row* rows, uint64_t cap_rows) {
    return submit_common(db, slot, sample, n_sample, false, true, nullptr, nullptr, nullptr, rows, cap_rows);
}
int yh_run_submit_packed(yh_db* db, int slot, const void* packed, uint64_t packed_bytes, yh_run_row* rows, uint64_t cap_rows) {
    return submit_common(db, slot, packed, packed_bytes, true, true, nullptr, nullptr, nullptr, rows, cap_rows);
}

static int wait_common(yh_db* db, int slot, uint64_t* n_rows) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (slot < 0 || slot >= YH_RUN_SLOTS) { yh_set_error("slot %d out of range [0, %d)", slot, YH_RUN_SLOTS); return YH_ERR_INVALID_ARG; }
    RunSlot& s = db->slots[slot];
    if (!s.busy) { yh_set_error("slot %d has no call in flight", slot); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    s.busy = false;
    YH_HIP(hipEventSynchronize(s.ev_out));
    if (n_rows) *n_rows = 0;
    if (((volatile u32*)s.h_bad)[0]) { yh_set_error("the sample sketch is not strictly ascending"); return YH_ERR_UNSORTED; }
    if (s.rows_mode) {
        const u64 n = ((volatile u32*)s.h_bad)[1];
        if (n_rows) *n_rows = n;
        if (n > s.rows_cap) {
            yh_set_error("row buffer holds %llu rows, %llu references overlap the sample", (u64)s.rows_cap, n);
            return YH_ERR_CAPACITY;
        }
    }
    return YH_OK;
}
int yh_run_wait(yh_db* db, int slot) { return wait_common(db, slot, nullptr); }
int yh_run_wait_rows(yh_db* db, int slot, uint64_t* n_rows) {
    if (!n_rows) { yh_set_error("n_rows is null"); return YH_ERR_INVALID_ARG; }
    return wait_common(db, slot, n_rows);
}

// the compact rows of the step that just ran on the handle (yh_run_device and its relatives): device-resident form
int yh_run_rows_device(yh_db* db, const uint32_t* d_overlap, const uint32_t* d_n_excl, const uint32_t* d_n_match,
                       yh_run_row* d_rows, uint64_t cap_rows, uint32_t* d_n_rows) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!d_n_rows || (db->n_refs && (!d_overlap || !d_n_excl || !d_n_match)) || (cap_rows && !d_rows)) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    // (the rows of a pipelined step are complete only behind its reducer and exclusive stages: run them first, so that
    // the count rows AND the subset bits the compaction reads are those of the step queued last)
    YH_TRY(pipe_join(db));
    return yh_rows_compact_device(db, d_overlap, d_n_excl, d_n_match, d_rows, cap_rows, d_n_rows, nullptr);
}

int yh_host_alloc(void** out, uint64_t bytes) {
    if (!out) { yh_set_error("out is null"); return YH_ERR_INVALID_ARG; }
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, std::max<uint64_t>(bytes, 16), hipHostMallocDefault);
    if (e != hipSuccess) { yh_set_error("hipHostMalloc(%llu) failed: %s", (u64)bytes, hipGetErrorString(e)); return YH_ERR_OOM; }
    return YH_OK;
}
int yh_host_free(void* p) {
    if (p) (void)hipHostFree(p);
    return YH_OK;
}

// ---- pairwise --------------------------------------------------------------------------------------
int yh_pairwise(yh_db* db, double c_thresh, uint64_t row_begin, uint64_t row_end, uint64_t cap, uint32_t* pair_i,
                uint32_t* pair_j, uint32_t* pair_count, uint64_t* n_out) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!n_out) { yh_set_error("n_out is null"); return YH_ERR_INVALID_ARG; }
    if (!(c_thresh >= 0.0 && c_thresh <= 1.0)) { yh_set_error("containment threshold must be between 0.0 and 1.0"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    if (row_end > db->n_refs) row_end = db->n_refs;
    if (!(db->pw_valid && db->pw_c == c_thresh && db->pw_r0 == row_begin && db->pw_r1 == row_end)) {
        try {  // (the pass sizes host vectors by N and by the survivors: no exception crosses the C boundary)
            YH_TRY(yh_q_pairwise(db, c_thresh, row_begin, row_end));
        } catch (const std::bad_alloc&) {
            yh_set_error("yh_pairwise: out of host memory");
            return YH_ERR_OOM;
        }
    }
    *n_out = db->pw_n;
    if (cap == 0) return YH_OK;
    if (cap < db->pw_n) { yh_set_error("pair buffers hold %llu entries, %llu needed", (u64)cap, db->pw_n); return YH_ERR_CAPACITY; }
    if (db->pw_n && (!pair_i || !pair_j || !pair_count)) { yh_set_error("null pair buffer"); return YH_ERR_INVALID_ARG; }
    memcpy(pair_i, db->h_pw_i, db->pw_n * sizeof(u32));
    memcpy(pair_j, db->h_pw_j, db->pw_n * sizeof(u32));
    memcpy(pair_count, db->h_pw_c, db->pw_n * sizeof(u32));
    return YH_OK;
}

int yh_db_nshared_device(yh_db* db, uint32_t* d_out) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    if (db->n_refs && !d_out) { yh_set_error("null device pointer"); return YH_ERR_INVALID_ARG; }
    YH_TRY(db_select(db));
    YH_TRY(yh_q_fz_nshared(db));  // (a fused YH_DB_PAIRWISE_ONLY handle counts them now, once)
    if (db->n_refs)
        YH_HIP(hipMemcpyAsync(d_out, db->d_nshared, db->n_refs * sizeof(u32), hipMemcpyDeviceToDevice, db->stream));
    return YH_OK;
}

int yh_index_stats(yh_db* db, uint64_t* n_distinct, uint64_t* n_singletons, uint64_t* n_index) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    if (n_distinct) *n_distinct = db->n_distinct;
    if (n_singletons) *n_singletons = db->n_distinct - db->n_shared;
    if (n_index) *n_index = db->n_shared;
    return YH_OK;
}

int yh_pairwise_row_stats(yh_db* db, uint64_t* sparse_rows, uint64_t* dense_rows) {
    if (!db_ok(db)) return YH_ERR_INVALID_ARG;
    if (sparse_rows) *sparse_rows = db->pw_sparse_rows;
    if (dense_rows) *dense_rows = db->pw_dense_rows;
    return YH_OK;
}

// ---- greedy selection (host; src/cpp/main.cpp:371-407) ----------------------------------------------
// Walk the references by ascending sketch size; drop one when a not-yet-dropped neighbour of at
// least its size exists.  The size ordering comes from libstdc++'s std::sort with a size-only
// comparator over {id, size} pairs in input order, exactly the reference's call: ties are
// resolved by that (deterministic, unstable) algorithm, and using the same call is the only way
// to resolve them identically.
int yh_train_select(const uint32_t* sizes, uint64_t n_refs, const uint32_t* pair_i, const uint32_t* pair_j,
                    uint64_t n_pairs, uint32_t* selected, uint64_t* n_selected) {
    if (!n_selected || (n_refs && (!sizes || !selected)) || (n_pairs && (!pair_i || !pair_j))) {
        yh_set_error("null argument");
        return YH_ERR_INVALID_ARG;
    }
    if (n_refs > 0x7fffffffull) { yh_set_error("too many references"); return YH_ERR_INVALID_ARG; }
    std::vector<u64> first(n_refs + 1, 0);  // neighbour lists in CSR form
    for (u64 k = 0; k < n_pairs; ++k) {
        if (pair_i[k] >= n_refs || pair_j[k] >= n_refs) { yh_set_error("pair index out of range"); return YH_ERR_INVALID_ARG; }
        if (k && (pair_i[k] < pair_i[k - 1] || (pair_i[k] == pair_i[k - 1] && pair_j[k] <= pair_j[k - 1]))) {
            yh_set_error("pairs must be sorted by (i, j) without duplicates");
            return YH_ERR_INVALID_ARG;
        }
        ++first[pair_i[k] + 1];
    }
    for (u64 i = 0; i < n_refs; ++i) first[i + 1] += first[i];

    std::vector<std::pair<int, int>> order(n_refs);
    for (u64 i = 0; i < n_refs; ++i) order[i] = {(int)i, (int)sizes[i]};
    std::sort(order.begin(), order.end(),
              [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });

    std::vector<char> dropped(n_refs, 0);
    u64 ns = 0;
    for (u64 t = 0; t < n_refs; ++t) {
        const int id = order[t].first;
        const int sz = order[t].second;
        bool keep = true;
        for (u64 k = first[id]; k < first[id + 1]; ++k) {
            const u32 o = pair_j[k];
            if (dropped[o]) continue;
            if ((int)sizes[o] >= sz) { keep = false; break; }
        }
        if (keep) selected[ns++] = (u32)id;
        else dropped[id] = 1;
    }
    *n_selected = ns;
    return YH_OK;
}

}  // extern "C"
