// A stand-in for the HIP runtime on the CPU, for ThreadSanitizer runs of the library's HOST-side context code (context.hip):
// device and pinned memory are malloc'ed, every stream is a worker thread that executes its copies in order and ASYNCHRONOUSLY
// (so a staging buffer shared by two callers really is overwritten while the first still reads it), queries and events follow
// the stream's progress.  Only what context.hip references is provided.  Never linked into the product.
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <chrono>

namespace {
struct StubStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t submitted = 0, done = 0;
    bool stop = false;
    std::thread worker;
    StubStream() : worker([this] { run(); }) {}
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front()); q.pop_front();
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));      // a copy takes a while: widens every race window
            f();
            { std::lock_guard<std::mutex> lk(mu); done++; }
        }
    }
    uint64_t push(std::function<void()> f) { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); cv.notify_one(); return ++submitted; }
    bool idle() { std::lock_guard<std::mutex> lk(mu); return done == submitted; }
    bool reached(uint64_t ticket) { std::lock_guard<std::mutex> lk(mu); return done >= ticket; }
    ~StubStream() { { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_one(); } worker.join(); }
};
struct StubEvent { StubStream* s = nullptr; uint64_t ticket = 0; std::mutex mu; };
thread_local int t_device = 0;
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { t_device = d; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 8); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 8); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t) new StubStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete (StubStream*)s; return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { return ((StubStream*)s)->idle() ? hipSuccess : hipErrorNotReady; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    ((StubStream*)s)->push([dst, src, n] { memcpy(dst, src, n); });
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub HIP error"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new StubEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (StubEvent*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    StubEvent* ev = (StubEvent*)e; StubStream* st = (StubStream*)s;
    const uint64_t t = st->push([] {});
    std::lock_guard<std::mutex> lk(ev->mu); ev->s = st; ev->ticket = t;
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    StubEvent* ev = (StubEvent*)e;
    std::lock_guard<std::mutex> lk(ev->mu);
    return (!ev->s || ev->s->reached(ev->ticket)) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { memset(a, 0, sizeof *a); a->device = 0; return hipSuccess; }
}
