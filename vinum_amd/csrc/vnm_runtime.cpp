// Runtime plumbing of libvinum_hip.so: init, errors, caching device allocator, staging, memcpy helpers.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstring>
#include <map>
#include <new>
#include <pthread.h>
#include <thread>
#include <unordered_map>

#include "vnm_common.hpp"

namespace vnm {

std::string& last_error() {
    static thread_local std::string e;
    return e;
}

int set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return 1;
}

DeviceInfo& device_info() {
    static DeviceInfo d;
    return d;
}

int ensure_init() {
    if (device_info().ready) return 0;
    return vnm_init(-1);
}

// ---- caching allocator ---------------------------------------------------------------------------
// Freed blocks are cached for reuse (a 1e9-row query recycles tens of GB of scratch between its passes and its batches).  A host
// that embeds the library never calls vnm_pool_trim, so the cache gives itself back: a reaper thread releases the cached blocks
// beyond VNM_POOL_KEEP_BYTES (default 256 MiB) once the allocator has been IDLE for VNM_POOL_IDLE_MS (default 2000 ms: no
// pool_alloc / pool_free in that time -- a query in flight, or a bench loop, touches the pool every few milliseconds and is
// never trimmed under its feet).  vnm_pool_set_idle_trim changes both at run time; idle_ms < 0 switches the reaper off.
namespace {
// (leaked on purpose: the reaper thread may still look at them while static destructors run at exit)
std::mutex& g_pool_mu = *new std::mutex;
std::multimap<size_t, void*>& g_free = *new std::multimap<size_t, void*>;               // size -> block
std::unordered_map<void*, size_t>& g_sizes = *new std::unordered_map<void*, size_t>;    // live + cached blocks
size_t g_cached_bytes = 0;
std::atomic<int64_t> g_last_activity_ms{0};
std::atomic<int64_t> g_idle_ms{-2};           // -2: not read from the environment yet
std::atomic<int64_t> g_keep_bytes{256ll << 20};
std::atomic<bool> g_reaper_started{false}, g_exiting{false};

int64_t now_ms() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

size_t round_size(size_t b) {
    if (b < 4096) return 4096;
    // round up to 1/8 of the leading power of two: bounded waste, good reuse
    size_t p = 1;
    while (p * 2 <= b) p *= 2;
    size_t step = p / 8;
    return (b + step - 1) / step * step;
}

// blocks waiting for an event before they may be reused (pool_free_after)
struct DeferredFree { hipEvent_t ev; void* p; };
std::vector<DeferredFree>& g_deferred = *new std::vector<DeferredFree>;
std::vector<hipEvent_t>& g_spare_events = *new std::vector<hipEvent_t>;
void reap_deferred_locked() {   // g_pool_mu held
    size_t keep = 0;
    for (size_t i = 0; i < g_deferred.size(); i++) {
        if (hipEventQuery(g_deferred[i].ev) == hipErrorNotReady) { g_deferred[keep++] = g_deferred[i]; continue; }
        g_spare_events.push_back(g_deferred[i].ev);
        auto it = g_sizes.find(g_deferred[i].p);
        if (it != g_sizes.end()) { g_free.emplace(it->second, g_deferred[i].p); g_cached_bytes += it->second; }
    }
    g_deferred.resize(keep);
}

// releases cached blocks, largest first, until at most `keep` bytes stay cached.  The victims are COLLECTED under the lock and
// freed after it is released: hipFree synchronises the device, and a query thread must not wait behind it for pool_alloc / pool_free.
size_t trim_to(size_t keep) {
    std::vector<void*> victims;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (!g_deferred.empty()) reap_deferred_locked();
        while (g_cached_bytes > keep && !g_free.empty()) {
            auto it = std::prev(g_free.end());
            bytes += it->first;
            g_cached_bytes -= it->first;
            g_sizes.erase(it->second);
            victims.push_back(it->second);
            g_free.erase(it);
        }
    }
    for (void* p : victims) (void)hipFree(p);
    return bytes;
}

std::thread* g_reaper = nullptr;
std::mutex& g_reaper_mu = *new std::mutex;
std::condition_variable& g_reaper_cv = *new std::condition_variable;

void reaper_main(int device) {
    if (device >= 0) (void)hipSetDevice(device);   // (the pool's blocks live on the device the process selected: vnm_init / torch's current device)
    std::unique_lock<std::mutex> lk(g_reaper_mu);
    for (;;) {
        g_reaper_cv.wait_for(lk, std::chrono::milliseconds(250));
        if (g_exiting.load()) return;
        const int64_t idle = g_idle_ms.load();
        if (idle < 0) continue;
        bool over;
        { std::lock_guard<std::mutex> g(g_pool_mu); over = g_cached_bytes > (size_t)g_keep_bytes.load(); }
        if (over && now_ms() - g_last_activity_ms.load() >= idle && !g_exiting.load()) {
            lk.unlock();
            trim_to((size_t)g_keep_bytes.load());
            lk.lock();
        }
    }
}

void stop_reaper() {   // atexit: signal and JOIN, so that the thread is never inside hipFree while the HIP runtime tears down
    g_exiting.store(true);
    { std::lock_guard<std::mutex> g(g_reaper_mu); }
    g_reaper_cv.notify_all();
    if (g_reaper && g_reaper->joinable()) g_reaper->join();
}

void after_fork_in_child() {   // the child has no reaper thread and must not believe it has (the mutexes are leaked objects: re-made)
    new (&g_pool_mu) std::mutex;
    new (&g_reaper_mu) std::mutex;
    g_reaper = nullptr;
    g_reaper_started.store(false);
    g_exiting.store(false);
}

void touch_pool() {
    g_last_activity_ms.store(now_ms(), std::memory_order_relaxed);
    if (!g_reaper_started.load(std::memory_order_relaxed) && !g_reaper_started.exchange(true)) {
        if (g_idle_ms.load() == -2) {
            const char* e = getenv("VNM_POOL_IDLE_MS");
            g_idle_ms.store(e ? atoll(e) : 2000);
            const char* k = getenv("VNM_POOL_KEEP_BYTES");
            if (k) g_keep_bytes.store(atoll(k));
        }
        static bool hooks = false;
        if (!hooks) {
            hooks = true;
            std::atexit(stop_reaper);
            (void)pthread_atfork(nullptr, nullptr, after_fork_in_child);
        }
        int device = -1;
        if (hipGetDevice(&device) != hipSuccess) device = -1;
        g_reaper = new std::thread(reaper_main, device);
    }
}
}  // namespace

void* pool_alloc(size_t bytes) {
    size_t sz = round_size(bytes ? bytes : 1);
    touch_pool();
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (!g_deferred.empty()) reap_deferred_locked();
        auto it = g_free.lower_bound(sz);
        if (it != g_free.end() && it->first <= sz * 2) {
            void* p = it->second;
            g_cached_bytes -= it->first;
            g_free.erase(it);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        pool_trim();
        e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", sz, hipGetErrorString(e));
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_sizes[p] = sz;
    return p;
}

void pool_free(void* p) {
    if (!p) return;
    touch_pool();
    std::lock_guard<std::mutex> g(g_pool_mu);
    auto it = g_sizes.find(p);
    if (it == g_sizes.end()) {
        (void)hipFree(p);
        return;
    }
    g_free.emplace(it->second, p);
    g_cached_bytes += it->second;
}

// A block that kernels already enqueued on `stream` still read (the widened key / input buffers of an aggregate, released
// right after the launches): it joins the free list only once an event recorded on that stream NOW has completed, so that an
// allocation from ANOTHER stream or thread -- the CSV reader, a second operator, torch's side streams in distributed.py -- cannot be
// handed the block while those kernels run.  The waiting blocks are looked at by every pool_alloc (hipEventQuery: no blocking).
void pool_free_after(void* p, hipStream_t stream) {
    if (!p) return;
    touch_pool();
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (!g_spare_events.empty()) { ev = g_spare_events.back(); g_spare_events.pop_back(); }
    }
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (!ev || hipEventRecord(ev, stream) != hipSuccess) {   // no event to wait for: wait for the stream itself, then a plain free
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream);
        if (ev) { std::lock_guard<std::mutex> g(g_pool_mu); g_spare_events.push_back(ev); }
        pool_free(p);
        return;
    }
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_deferred.push_back({ev, p});
}

size_t pool_trim() { return trim_to(0); }

Predicate make_predicate(int col_type, bool col_has_nulls, int op, int scalar_is_float, double dval, int64_t ival) {
    Predicate p{};
    p.enabled = 1;
    p.op = op;
    p.dval = scalar_is_float ? dval : (double)ival;
    p.ival = ival;
    if (col_type == VNM_F32) {
        p.mode = CMP_F32;  // float32 stays float32 in NumPy (NULL -> NaN handled by the F64/F32 paths)
        if (col_has_nulls) p.mode = CMP_F64, p.dval = (double)(float)p.dval;  // same result: f32 -> f64 is exact
    } else if (col_type == VNM_F64 || col_has_nulls || scalar_is_float) {
        p.mode = CMP_F64;
    } else if (col_type == VNM_U64) {
        if (ival < 0) {
            p.mode = CMP_CONST;
            p.const_result = (op == VNM_GT || op == VNM_GE || op == VNM_NE);
        } else {
            p.mode = CMP_U64;
        }
    } else {
        p.mode = CMP_I64;
    }
    return p;
}

// ---- route notes -----------------------------------------------------------------------------------------
namespace {
std::mutex& g_route_mu = *new std::mutex;
std::map<std::string, int64_t>& g_route_counts = *new std::map<std::string, int64_t>;
thread_local std::string t_route_last;
}  // namespace

void route_note(const char* route, const char* reason_fmt, ...) {
    char buf[512];
    buf[0] = 0;
    if (reason_fmt) {
        va_list ap;
        va_start(ap, reason_fmt);
        vsnprintf(buf, sizeof(buf), reason_fmt, ap);
        va_end(ap);
    }
    t_route_last = std::string(route) + (buf[0] ? std::string(": ") + buf : std::string());
    static const bool trace = getenv("VNM_AGG_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[route] %s\n", t_route_last.c_str());
    std::lock_guard<std::mutex> g(g_route_mu);
    g_route_counts[route]++;
}

// ---- kernel timing ------------------------------------------------------------------------------------
namespace {
struct TimedSpan { std::string name; hipEvent_t a, b; };
bool g_profiling = false;
std::vector<TimedSpan> g_spans;
}  // namespace

KernelTimer::KernelTimer(const char* name, hipStream_t s) : on(g_profiling), slot(-1), stream(s) {
    static const bool trace = getenv("VNM_TRACE") != nullptr;  // debugging aid: name every timed launch on stderr
    if (trace) { fprintf(stderr, "[vnm] launch %s\n", name); fflush(stderr); (void)hipStreamSynchronize(s); }
    if (!on) return;
    TimedSpan sp;
    sp.name = name;
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(sp.a, s);
    g_spans.push_back(sp);
    slot = (int)g_spans.size() - 1;
}
KernelTimer::~KernelTimer() {
    if (on && slot >= 0) (void)hipEventRecord(g_spans[slot].b, stream);
}


// ---- pinned double-buffered H2D staging ------------------------------------------------------------------
// Arrow buffers are pageable host memory.  Copying them with hipMemcpyAsync directly makes the runtime bounce
// every chunk through its own staging synchronously; here two pinned 32 MiB buffers alternate: while the DMA
// engine drains one, the CPU fills the other (copy stream separate from the compute stream of the caller).
namespace {
constexpr size_t STAGE_BYTES = 32u << 20;
struct Stager {
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t done[2];
    hipStream_t copy = nullptr;
    bool ok = false;
    bool init() {
        if (ok) return true;
        if (hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) != hipSuccess) return false;
        for (int i = 0; i < 2; i++) {
            if (hipHostMalloc(&pin[i], STAGE_BYTES, hipHostMallocDefault) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        ok = true;
        return true;
    }
};
Stager& stager() { static Stager s; return s; }
std::mutex g_stage_mu;

// One thread fills a pinned buffer at ~13 GB/s, the DMA engine drains it at ~50: a few helper threads copy
// slices of every chunk so the PCIe link, not memcpy, bounds the staging (VNM_STAGE_THREADS, default 4).
class CopyPool {
public:
    explicit CopyPool(int helpers) {
        for (int i = 0; i < helpers; i++) workers_.emplace_back([this, i] { run(i); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // fn(part, parts) on every helper thread and on the caller (part 0); returns when all are done
    void parallel(const std::function<void(int, int)>& fn) {
        const int parts = (int)workers_.size() + 1;
        if (parts == 1) { fn(0, 1); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn;
            pending_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        fn(0, parts);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
    void copy(void* dst, const void* src, size_t n) {
        const int parts = (int)workers_.size() + 1;
        if (parts == 1 || n < (4u << 20)) { memcpy(dst, src, n); return; }
        const size_t slice = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = (uint8_t*)dst; src_ = (const uint8_t*)src; n_ = n; slice_ = slice;
            pending_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        memcpy(dst, src, slice < n ? slice : n);  // slice 0 on the calling thread
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return pending_ == 0; });
    }
private:
    void run(int idx) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            const size_t lo = slice_ * (size_t)(idx + 1);
            uint8_t* d = dst_; const uint8_t* s = src_;
            const size_t n = n_, sl = slice_;
            const std::function<void(int, int)>* fn = fn_;
            g.unlock();
            if (fn) (*fn)(idx + 1, (int)workers_.size() + 1);
            else if (lo < n) memcpy(d + lo, s + lo, lo + sl < n ? sl : n - lo);
            g.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    uint8_t* dst_ = nullptr; const uint8_t* src_ = nullptr;
    size_t n_ = 0, slice_ = 0;
    const std::function<void(int, int)>* fn_ = nullptr;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};
CopyPool& copy_pool() {
    static CopyPool* p = [] {
        const char* e = getenv("VNM_STAGE_THREADS");
        int t = e ? atoi(e) : 4;
        unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)t > hw) t = (int)hw;
        if (t < 1) t = 1;
        return new CopyPool(t - 1);  // leaked on purpose: joining threads from a static destructor is fragile
    }();
    return *p;
}
}  // namespace

// host (pageable) -> device, through the pinned ring; returns when the last DMA has been enqueued AND completed
// for the caller's stream (the caller's kernels may start right after: we make `stream` wait on the copy stream)
static int staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return 0;
    std::lock_guard<std::mutex> g(g_stage_mu);
    Stager& st = stager();
    if (bytes < (1u << 20) || !st.init()) {  // small copies: not worth the ring
        VNM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return 0;
    }
    size_t off = 0;
    int slot = 0;
    bool used[2] = {false, false};
    while (off < bytes) {
        size_t n = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
        if (used[slot]) VNM_HIP(hipEventSynchronize(st.done[slot]));  // the DMA that read this buffer is finished
        copy_pool().copy(st.pin[slot], (const uint8_t*)src + off, n);
        VNM_HIP(hipMemcpyAsync((uint8_t*)dst + off, st.pin[slot], n, hipMemcpyHostToDevice, st.copy));
        VNM_HIP(hipEventRecord(st.done[slot], st.copy));
        used[slot] = true;
        off += n;
        slot ^= 1;
    }
    // order the caller's stream after the copies, and keep the pinned buffers safe for the next call
    for (int i = 0; i < 2; i++)
        if (used[i]) { VNM_HIP(hipStreamWaitEvent(stream, st.done[i], 0)); VNM_HIP(hipEventSynchronize(st.done[i])); }
    return 0;
}

// Several host ranges laid end to end into ONE device range: the pinned buffers are filled from as many chunks as fit (the copy
// threads share the chunks), one DMA per filled buffer.  For the record batches an operator kept while they were small: 10 000-row
// batches are 80 KB per column -- a DMA (or a host-side concatenation first) per chunk cost more than the bytes.
int stage_chunks(void* dst, const void* const* srcs, const size_t* sizes, size_t n_chunks, hipStream_t stream) {
    std::lock_guard<std::mutex> g(g_stage_mu);
    Stager& st = stager();
    if (!st.init()) return set_error("staging: no pinned buffers");
    struct Piece { size_t at; const uint8_t* src; size_t n; };
    std::vector<Piece> pieces;
    size_t dst_off = 0, chunk = 0, chunk_off = 0;
    int slot = 0;
    bool used[2] = {false, false};
    while (chunk < n_chunks) {
        pieces.clear();
        size_t fill = 0;
        while (chunk < n_chunks && fill < STAGE_BYTES) {
            const size_t left = sizes[chunk] - chunk_off;
            const size_t n = left < STAGE_BYTES - fill ? left : STAGE_BYTES - fill;
            if (n) pieces.push_back(Piece{fill, (const uint8_t*)srcs[chunk] + chunk_off, n});
            fill += n; chunk_off += n;
            if (chunk_off == sizes[chunk]) { chunk++; chunk_off = 0; }
        }
        if (!fill) break;
        if (used[slot]) VNM_HIP(hipEventSynchronize(st.done[slot]));   // the DMA that read this buffer is finished
        uint8_t* pin = (uint8_t*)st.pin[slot];
        const size_t np = pieces.size();
        const std::function<void(int, int)> work = [&](int part, int parts) {
            for (size_t i = np * (size_t)part / (size_t)parts; i < np * (size_t)(part + 1) / (size_t)parts; i++) memcpy(pin + pieces[i].at, pieces[i].src, pieces[i].n);
        };
        if (fill < (4u << 20)) work(0, 1); else copy_pool().parallel(work);
        VNM_HIP(hipMemcpyAsync((uint8_t*)dst + dst_off, pin, fill, hipMemcpyHostToDevice, st.copy));
        VNM_HIP(hipEventRecord(st.done[slot], st.copy));
        used[slot] = true;
        dst_off += fill;
        slot ^= 1;
    }
    for (int i = 0; i < 2; i++)
        if (used[i]) { VNM_HIP(hipStreamWaitEvent(stream, st.done[i], 0)); VNM_HIP(hipEventSynchronize(st.done[i])); }
    return 0;
}

}  // namespace vnm

using namespace vnm;

extern "C" {
// "route=count" lines of every route taken since the library was loaded (or vnm_route_reset); returns the bytes needed incl. the 0
int64_t vnm_route_counts(char* buf, int64_t cap) {
    std::string out;
    {
        std::lock_guard<std::mutex> g(g_route_mu);
        for (auto& kv : g_route_counts) out += kv.first + "=" + std::to_string(kv.second) + "\n";
    }
    if (buf && cap > 0) { const size_t n = std::min<size_t>(out.size(), (size_t)cap - 1); memcpy(buf, out.data(), n); buf[n] = 0; }
    return (int64_t)out.size() + 1;
}
// the last note of the calling thread: "route: reason"
int64_t vnm_route_last(char* buf, int64_t cap) {
    if (buf && cap > 0) { const size_t n = std::min<size_t>(t_route_last.size(), (size_t)cap - 1); memcpy(buf, t_route_last.data(), n); buf[n] = 0; }
    return (int64_t)t_route_last.size() + 1;
}
void vnm_route_reset(void) {
    std::lock_guard<std::mutex> g(g_route_mu);
    g_route_counts.clear();
}

int vnm_set_profiling(int on) {
    g_profiling = on != 0;
    for (auto& sp : g_spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    g_spans.clear();
    return 0;
}
// total milliseconds and launch count of the timed spans whose name matches `name` since
// vnm_set_profiling(1); synchronises the recorded events
int vnm_profile_query(const char* name, double* total_ms, int64_t* count) {
    double tot = 0;
    int64_t n = 0;
    for (auto& sp : g_spans) {
        if (name && sp.name != name) continue;
        if (hipEventSynchronize(sp.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { tot += ms; n++; }
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    return 0;
}
}

extern "C" {

int vnm_init(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error("vnm_init: no HIP device available (%s); libvinum_hip has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    int dev = device_id;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= n) return set_error("vnm_init: device %d out of range (%d devices)", dev, n);
    VNM_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    VNM_HIP(hipGetDeviceProperties(&prop, dev));
    DeviceInfo& d = device_info();
    d.device = dev;
    d.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    d.ready = true;
    return 0;
}

const char* vnm_last_error(void) { return last_error().c_str(); }

int vnm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vnm_device_synchronize(void) {
    VNM_HIP(hipDeviceSynchronize());
    return 0;
}

void* vnm_malloc(int64_t bytes) {
    if (ensure_init()) return nullptr;
    return pool_alloc((size_t)bytes);
}

int vnm_free(void* p) {
    pool_free(p);
    return 0;
}

int64_t vnm_pool_cached_bytes(void) {
    std::lock_guard<std::mutex> g(g_pool_mu);
    return (int64_t)g_cached_bytes;
}

int vnm_pool_set_idle_trim(int64_t idle_ms, int64_t keep_bytes) {
    g_idle_ms.store(idle_ms < 0 ? -1 : idle_ms);
    if (keep_bytes >= 0) g_keep_bytes.store(keep_bytes);
    return 0;
}

int64_t vnm_pool_trim(void) {
    return (int64_t)pool_trim();
}

int vnm_memcpy_h2d(void* dst, const void* src, int64_t bytes) {
    VNM_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return 0;
}

int vnm_memcpy_d2h(void* dst, const void* src, int64_t bytes) {
    VNM_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

int vnm_memset(void* dst, int value, int64_t bytes) {
    VNM_HIP(hipMemset(dst, value, (size_t)bytes));
    return 0;
}

// Stage one host Arrow column into HBM.  Only the bytes the slice covers are copied; the device view
// keeps the sub-byte validity offset so no bitmap realignment pass is needed.
int vnm_stage_column(const void* host_values, const uint8_t* host_validity, int64_t offset, int64_t length,
                     int32_t type, vnm_dcol* out, void* stream) {
    VNM_TRY(ensure_init());
    int w = type_width(type);
    hipStream_t s = as_stream(stream);
    memset(out, 0, sizeof(*out));
    out->type = type;
    out->length = length;
    size_t vbytes = (size_t)length * w;
    // With a validity bitmap the values are staged with (offset & 7) elements of left padding: the bitmap is
    // copied from its first BYTE, so bit (offset & 7) of the device bitmap belongs to logical element 0, and one
    // offset then serves both buffers.
    const int shift = host_validity ? (int)(offset & 7) : 0;
    size_t pbytes = ((size_t)length + shift) * w;
    void* dv = pool_alloc(pbytes ? pbytes : 1);
    if (!dv) return 1;
    if (vbytes)
        VNM_TRY(staged_h2d((uint8_t*)dv + (size_t)shift * w, (const uint8_t*)host_values + (size_t)offset * w, vbytes, s));
    out->values = dv;
    out->offset = shift;
    if (host_validity) {
        int64_t first_byte = offset >> 3;
        int64_t last_byte = (offset + length + 7) >> 3;
        size_t nb = (size_t)(last_byte - first_byte);
        void* db = pool_alloc(nb ? nb : 1);
        if (!db) return 1;
        if (nb) VNM_TRY(staged_h2d(db, host_validity + first_byte, nb, s));
        out->validity = (const uint8_t*)db;
    }
    return 0;
}

int vnm_free_column(vnm_dcol* col) {
    if (!col) return 0;
    pool_free(const_cast<void*>(col->values));
    pool_free(const_cast<uint8_t*>(col->validity));
    col->values = nullptr;
    col->validity = nullptr;
    return 0;
}

}  // extern "C"
