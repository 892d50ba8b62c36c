// Error reporting, ABI version, hipEvent profiling of kernel classes.
#include "keds_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void keds_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int keds_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        keds_set_error("%s: %s", what, hipGetErrorString(e));
        return KEDS_E_LAUNCH;
    }
    return KEDS_OK;
}

extern "C" int keds_abi_version(void) { return KEDS_ABI_VERSION; }
// the compiler flags this library was built with BEYOND the Makefile's defaults (the -D switches of a timing-only or A/B
// build: tools/ab_build.sh, tools/rounds/r05_noepi_bound.sh); "" for the product build.  keds_amd._lib.source_digest folds it into
// the digest that ties committed measurements to a build, so evidence taken on a variant build is never accepted as "this build".
#ifndef KEDS_BUILD_EXTRA
#define KEDS_BUILD_EXTRA ""
#endif
extern "C" const char* keds_build_flags(void) { return KEDS_BUILD_EXTRA; }
extern "C" const char* keds_last_error(void) { return g_err; }

// ---- side lane ---------------------------------------------------------------------------
namespace {
std::mutex g_lane_mu;
KedsSideLane g_lanes[64];
int g_lane_state[64];                          // 0 untried, 1 ready, -1 unavailable
bool is_side_stream(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_lane_mu);
    for (int d = 0; d < 64; ++d)
        if (g_lane_state[d] == 1 && g_lanes[d].s == s) return true;
    return false;
}
}  // namespace

static int g_lane_on = -1;                     // -1: take KEDS_SIDE_STREAM (default on)
bool keds_side_lane_enabled() {
    if (g_lane_on < 0) {
        const char* e = getenv("KEDS_SIDE_STREAM");
        g_lane_on = !(e && e[0] == '0');
    }
    return g_lane_on != 0;
}
extern "C" int keds_side_lane_enable(int on) {
    g_lane_on = on ? 1 : 0;
    return KEDS_OK;
}

KedsSideLane* keds_side_lane() {
    if (!keds_side_lane_enabled()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_lane_mu);
    if (g_lane_state[dev] == 0) {
        int least = 0, greatest = 0;           // the "greatest" priority is the numerically lowest value
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        bool ok = hipStreamCreateWithPriority(&g_lanes[dev].s, hipStreamNonBlocking, greatest) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&g_lanes[dev].fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&g_lanes[dev].join, hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        g_lane_state[dev] = ok ? 1 : -1;
    }
    return g_lane_state[dev] == 1 ? &g_lanes[dev] : nullptr;
}

// A stream wait binds to the event's most recent record AT THE TIME OF THE CALL, and the lane's two events are shared by
// every caller on the device (keds_session.h allows different handles on different threads): record + wait are one
// critical section, so no other thread's record of the same event can slip in between and re-aim this caller's wait.
static std::mutex g_order_mu;
// (round 5) the same critical section for an event that a LAUNCH records: keds_order_lock(); launch with the event as its stop
// event (hipExtLaunchKernelGGL: the kernel's own completion signal -- no marker packet behind it on the launching stream);
// keds_stream_wait_locked(ev, to); keds_order_unlock()
void keds_order_lock() { g_order_mu.lock(); }
void keds_order_unlock() { g_order_mu.unlock(); }
int keds_stream_wait_locked(hipEvent_t ev, hipStream_t to) {
    if (hipStreamWaitEvent(to, ev, 0) != hipSuccess) {
        keds_set_error("stream ordering failed: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    return KEDS_OK;
}
int keds_stream_order(hipStream_t from, hipEvent_t ev, hipStream_t to) {
    std::lock_guard<std::mutex> g(g_order_mu);
    if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) {
        keds_set_error("stream ordering failed: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    return KEDS_OK;
}

// ---- per-device launch state ---------------------------------------------------------------
namespace {
std::mutex g_dev_mu;
struct FuncDev {
    const void* f;
    int dev;
    int bytes;
};
std::vector<FuncDev> g_func_lds;
int g_cus[64];
struct DevScratch {
    float* p;
    size_t bytes;
};
DevScratch g_splitk_dev[64];
thread_local float* tl_splitk_p = nullptr;
thread_local size_t tl_splitk_bytes = 0;
thread_local bool tl_splitk_off = false;        // inside a scope opened with a null buffer: no GEMM of this thread splits K
}  // namespace

int keds_func_lds_once(const void* func, int bytes, const char* what) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(g_dev_mu);
    for (const FuncDev& e : g_func_lds)
        if (e.f == func && e.dev == dev && e.bytes >= bytes) return KEDS_OK;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        keds_set_error("%s: cannot raise dynamic LDS to %d bytes on device %d", what, bytes, dev);
        return KEDS_E_LAUNCH;
    }
    g_func_lds.push_back(FuncDev{func, dev, bytes});
    return KEDS_OK;
}

int keds_device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (g_cus[dev] == 0) {
        hipDeviceProp_t prop;
        g_cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0
                         ? prop.multiProcessorCount : 256;
    }
    return g_cus[dev];
}

KedsSplitKScope::KedsSplitKScope(void* p, size_t bytes) : prev_p(tl_splitk_p), prev_bytes(tl_splitk_bytes), prev_off(tl_splitk_off) {
    tl_splitk_p = (float*)p;
    tl_splitk_bytes = p ? bytes : 0;
    tl_splitk_off = p == nullptr;
}
KedsSplitKScope::~KedsSplitKScope() {
    tl_splitk_p = prev_p;
    tl_splitk_bytes = prev_bytes;
    tl_splitk_off = prev_off;
}
void keds_splitk_scratch(float** p, size_t* bytes) {
    if (tl_splitk_off) {                        // a composite call that runs GEMMs on two streams at once: never split K
        *p = nullptr;
        *bytes = 0;
        return;
    }
    if (tl_splitk_p) {
        *p = tl_splitk_p;
        *bytes = tl_splitk_bytes;
        return;
    }
    int dev = 0;
    *p = nullptr;
    *bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> g(g_dev_mu);
    *p = g_splitk_dev[dev].p;
    *bytes = g_splitk_dev[dev].bytes;
}

// Per-device fallback scratch for keds_gemm_bt* calls made outside a composite call (keds_common.h): the buffer must
// live on the CURRENT device; nullptr unregisters.  One stream at a time may run split-K shapes against it.
extern "C" int keds_gemm_set_workspace(void* ptr, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        keds_set_error("keds_gemm_set_workspace: no current device");
        return KEDS_E_ARG;
    }
    std::lock_guard<std::mutex> g(g_dev_mu);
    g_splitk_dev[dev].p = (float*)ptr;
    g_splitk_dev[dev].bytes = ptr ? bytes : 0;
    return KEDS_OK;
}

// ---- numerics guard ------------------------------------------------------------------------
static thread_local int* tl_guard = nullptr;
int* keds_numerics_guard() { return tl_guard; }
extern "C" int keds_numerics_guard_set(int32_t* device_flag) {
    tl_guard = device_flag;
    return KEDS_OK;
}

// ---- profiling ---------------------------------------------------------------------------
namespace {
struct EvPair {
    hipEvent_t a, b;
};
struct ProfState {
    unsigned mask = 0;   // bit k: record events for class k
    std::mutex mu;
    std::vector<EvPair> used[KEDS_PROF_NCLASS];
    double work[KEDS_PROF_NCLASS] = {};
    std::vector<EvPair> pool;
};
ProfState& prof() {
    static ProfState s;
    return s;
}
}  // namespace

static thread_local KedsProfScope* tl_lazy_scope = nullptr;

KedsProfScope::KedsProfScope(int k, hipStream_t s, bool lz)
    : klass(k), stream(s), slot(nullptr), lazy(lz), taken(false), work_units(0.0), ev_a(nullptr), ev_b(nullptr), outer(nullptr) {
    ProfState& p = prof();
    if (!(p.mask >> klass & 1u)) return;
    // side-lane launches overlap the caller's stream (and their event pairs would also time the wait for a free CU):
    // summing them with the main-lane durations would double-count the time, so they carry no events
    if (s && is_side_stream(s)) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (lazy && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) lazy = false;
    std::lock_guard<std::mutex> g(p.mu);
    EvPair ev;
    if (!p.pool.empty()) {
        ev = p.pool.back();
        p.pool.pop_back();
    } else {
        if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
    }
    ev_a = ev.a;
    ev_b = ev.b;
    slot = (void*)1;
    if (lazy) {                              // the launches bind the pair (KEDS_LAUNCH); it joins `used` when one did
        outer = tl_lazy_scope;
        tl_lazy_scope = this;
        return;
    }
    (void)hipEventRecord(ev.a, stream);
    p.used[klass].push_back(ev);
}

KedsProfScope::~KedsProfScope() {
    if (!slot) return;
    ProfState& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    if (lazy) {
        tl_lazy_scope = outer;
        if (taken) {
            p.used[klass].push_back(EvPair{ev_a, ev_b});
            p.work[klass] += work_units;
        } else {
            p.pool.push_back(EvPair{ev_a, ev_b});          // no launch of this scope went through KEDS_LAUNCH: nothing was timed
        }
        return;
    }
    (void)hipEventRecord(ev_b, stream);
}

void KedsProfScope::work(double units) {
    if (!slot) return;
    if (lazy) {
        work_units += units;
        return;
    }
    ProfState& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    p.work[klass] += units;
}

KedsLaunchEvents keds_prof_launch_events(hipStream_t st) {
    KedsProfScope* sc = tl_lazy_scope;
    if (!sc || sc->stream != st) return KedsLaunchEvents{nullptr, nullptr};
    const bool first = !sc->taken;
    sc->taken = true;
    return KedsLaunchEvents{first ? sc->ev_a : nullptr, sc->ev_b};
}

extern "C" int keds_prof_read_work(int klass, double* units) {
    if (klass < 0 || klass >= KEDS_PROF_NCLASS || !units) {
        keds_set_error("keds_prof_read_work: bad argument");
        return KEDS_E_ARG;
    }
    ProfState& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    *units = p.work[klass];
    return KEDS_OK;
}

extern "C" int keds_prof_enable(int on) {
    // 0: off, 1: every class, otherwise a bit mask with bit (k+1) selecting class k (so 0b110 = GEMM + ATTN)
    prof().mask = on == 0 ? 0u : (on == 1 ? 0xFFFFFFFFu : ((unsigned)on >> 1));
    return KEDS_OK;
}

extern "C" int keds_prof_reset(void) {
    ProfState& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    for (int k = 0; k < KEDS_PROF_NCLASS; ++k) {
        for (auto& e : p.used[k]) p.pool.push_back(e);
        p.used[k].clear();
        p.work[k] = 0;
    }
    return KEDS_OK;
}

extern "C" int keds_prof_read(int klass, double* total_ms, int64_t* launches) {
    if (klass < 0 || klass >= KEDS_PROF_NCLASS || !total_ms || !launches) {
        keds_set_error("keds_prof_read: bad argument");
        return KEDS_E_ARG;
    }
    ProfState& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    double t = 0;
    for (auto& e : p.used[klass]) {
        if (hipEventSynchronize(e.b) != hipSuccess) {
            keds_set_error("keds_prof_read: event sync failed");
            return KEDS_E_LAUNCH;
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e.a, e.b);
        t += ms;
    }
    *total_ms = t;
    *launches = (int64_t)p.used[klass].size();
    return KEDS_OK;
}
