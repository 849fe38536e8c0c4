// The host side of a library context: error reporting, the stream-ordered caching pool, the pinned staging buffers, scope timing,
// lifetime (reference counting) and the copy entry points of the C ABI.  No kernels here: this file's host pass is also built with
// -fsanitize=thread against a stub HIP runtime (tools/sanitizer/ctx_race.cpp), because the one data corruption of round 2 was a
// host-side race in exactly this code (one pinned buffer shared by the threads that read a circuit's Merkle cap through its context).
#include "context.hpp"
#include <atomic>
#include <memory>
#include <cstdio>
#include <cstring>
#include <cstdlib>

thread_local std::string g_gl_last_error;

int gl_fail(int code, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s (%s:%d)", what, file, line);
    g_gl_last_error = buf;
    return code;
}

// ------------------------------------------------------------------------------------------ context
int gl_ctx::activate() {
    int cur = -1;
    GL_CHECK_HIP(hipGetDevice(&cur));
    if (cur != device) GL_CHECK_HIP(hipSetDevice(device));
    return GL_OK;
}
int gl_ctx::ensure_scratch(size_t elems) {
    if (elems <= scratch_elems) return GL_OK;
    if (scratch) { GL_CHECK_HIP(gl_stream_wait(stream)); GL_CHECK_HIP(hipFree(scratch)); scratch = nullptr; scratch_elems = 0; }
    GL_CHECK_HIP(hipMalloc((void**)&scratch, elems * sizeof(gl_t)));
    scratch_elems = elems;
    return GL_OK;
}
int gl_ctx::pin_acquire(size_t bytes, void** out, size_t* cap) {
    {
        std::lock_guard<std::mutex> lk(pin_mu);
        for (size_t i = 0; i < pin_free.size(); i++)
            if (pin_free[i].second >= bytes) { *out = pin_free[i].first; *cap = pin_free[i].second; pin_free.erase(pin_free.begin() + i); return GL_OK; }
    }
    const size_t sz = bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes;
    GL_CHECK_HIP(hipHostMalloc(out, sz, hipHostMallocDefault));
    *cap = sz;
    return GL_OK;
}
void gl_ctx::pin_release(void* p, size_t cap) {
    std::lock_guard<std::mutex> lk(pin_mu);
    pin_free.emplace_back(p, cap);
}
int gl_ctx::ensure_dev_small(size_t bytes) {
    if (bytes <= dev_small_bytes) return GL_OK;
    if (dev_small) { GL_CHECK_HIP(gl_stream_wait(stream)); GL_CHECK_HIP(hipFree(dev_small)); dev_small = nullptr; dev_small_bytes = 0; }
    size_t sz = bytes < (1u << 20) ? (1u << 20) : bytes;
    GL_CHECK_HIP(hipMalloc((void**)&dev_small, sz));
    dev_small_bytes = sz;
    return GL_OK;
}
int gl_ctx::pool_alloc(size_t bytes, void** out) {
    const size_t gran = bytes >= (size_t(1) << 20) ? (size_t(1) << 20) : (size_t(1) << 12);
    const size_t want = ((bytes ? bytes : 8) + gran - 1) / gran * gran;
    {
        std::lock_guard<std::mutex> lk(pool_mu);
        auto it = pool_free_blocks.lower_bound(want);
        if (it != pool_free_blocks.end() && it->first <= want + want / 4) {
            *out = it->second;
            pool_free_blocks.erase(it);
            return GL_OK;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {          // out of memory: drop the cache and retry once
        pool_trim();
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) return gl_fail(GL_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__);
    std::lock_guard<std::mutex> lk(pool_mu);
    pool_block_size[p] = want;
    pool_bytes += want;
    *out = p;
    return GL_OK;
}
void gl_ctx::pool_release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(pool_mu);
    auto it = pool_block_size.find(p);
    if (it == pool_block_size.end()) { (void)hipFree(p); return; }
    pool_free_blocks.emplace(it->second, p);
}
void gl_ctx::pool_trim() {
    std::lock_guard<std::mutex> lk(pool_mu);
    (void)gl_stream_wait(stream);
    for (auto& kv : pool_free_blocks) { pool_bytes -= kv.first; pool_block_size.erase(kv.second); (void)hipFree(kv.second); }
    pool_free_blocks.clear();
}

void gl_ctx::timing_begin(const char* name) {
    if (!timing_enabled) return;
    TimingRec r; r.name = name;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    (void)hipEventRecord(r.start, stream);
    timing_stack.push_back(timing_recs.size());
    timing_recs.push_back(r);
}
void gl_ctx::timing_end() {
    if (!timing_enabled || timing_stack.empty()) return;
    (void)hipEventRecord(timing_recs[timing_stack.back()].stop, stream);
    timing_stack.pop_back();
}

extern "C" int gl_ctx_timing_enable(gl_ctx* c, int on) {
    GL_REQUIRE(c, GL_ERR_ARG, "null ctx");
    c->timing_enabled = on != 0;
    return GL_OK;
}
extern "C" int gl_ctx_timing_reset(gl_ctx* c) {
    GL_REQUIRE(c, GL_ERR_ARG, "null ctx");
    GL_TRY(c->activate());
    GL_CHECK_HIP(gl_stream_wait(c->stream));
    for (auto& r : c->timing_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    c->timing_recs.clear();
    c->timing_stack.clear();
    return GL_OK;
}
// writes a JSON object {"scope": {"count": n, "ms": total}, ...} into buf (NUL-terminated)
extern "C" int gl_ctx_timing_report(gl_ctx* c, char* buf, size_t cap) {
    GL_REQUIRE(c && buf && cap > 2, GL_ERR_ARG, "bad argument");
    GL_TRY(c->activate());
    GL_CHECK_HIP(gl_stream_wait(c->stream));
    std::map<std::string, std::pair<uint64_t, double>> agg;
    std::vector<std::string> order;
    for (auto& r : c->timing_recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
        if (!agg.count(r.name)) order.push_back(r.name);
        auto& a = agg[r.name];
        a.first++; a.second += ms;
    }
    std::string out = "{";
    for (size_t i = 0; i < order.size(); i++) {
        char line[256];
        snprintf(line, sizeof line, "%s\"%s\": {\"count\": %llu, \"ms\": %.6f}", i ? ", " : "", order[i].c_str(),
                 (unsigned long long)agg[order[i]].first, agg[order[i]].second);
        out += line;
    }
    out += "}";
    GL_REQUIRE(out.size() + 1 <= cap, GL_ERR_ARG, "timing report buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return GL_OK;
}

int gl_ctx::get_offsets_table(const uint64_t* host, size_t len, const uint64_t** d_out) {
    std::vector<uint64_t> key(host, host + len);
    auto it = offset_tables.find(key);
    if (it != offset_tables.end()) { *d_out = it->second; return GL_OK; }
    uint64_t* d = nullptr;
    GL_CHECK_HIP(hipMalloc((void**)&d, len * sizeof(uint64_t)));
    GL_CHECK_HIP(hipMemcpyAsync(d, host, len * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    GL_CHECK_HIP(gl_stream_wait(stream));   // first use only
    offset_tables[key] = d;
    *d_out = d;
    return GL_OK;
}

// the last reference is gone: nothing points at the context any more
static void gl_ctx_teardown(gl_ctx* c) {
    (void)hipSetDevice(c->device);
    (void)gl_stream_wait(c->stream);
    for (auto& r : c->timing_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (int d = 0; d < 2; d++) if (c->tw_local[d]) (void)hipFree(c->tw_local[d]);
    c->pool_trim();
    for (auto& kv : c->pool_block_size) (void)hipFree(kv.first);      // only blocks leaked by a caller that dropped a handle without freeing it
    for (auto& kv : c->pow_tables) (void)hipFree(kv.second.lo);
    for (gl_t* t : c->retired_tables) (void)hipFree(t);
    for (auto& kv : c->pass_tables) (void)hipFree(kv.second);
    for (auto& kv : c->offset_tables) (void)hipFree(kv.second);
    if (c->scratch) (void)hipFree(c->scratch);
    for (auto& pb : c->pin_free) (void)hipHostFree(pb.first);
    if (c->dev_small) (void)hipFree(c->dev_small);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
void gl_ctx_release(gl_ctx* c) {
    if (c && c->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) gl_ctx_teardown(c);
}
// Drops the creator's reference.  Handles created on the context keep it alive (and usable through them) until the last of
// them is freed; the caller must not pass `c` to any entry point after this call.
extern "C" void gl_ctx_destroy(gl_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)gl_stream_wait(c->stream);
    gl_ctx_release(c);
}
extern "C" int gl_ctx_synchronize(gl_ctx* c) {
    GL_REQUIRE(c, GL_ERR_ARG, "null ctx");
    GL_TRY(c->activate());
    GL_CHECK_HIP(gl_stream_wait(c->stream));
    return GL_OK;
}
extern "C" int gl_ctx_set_scratch_elems(gl_ctx* c, size_t elems) {
    GL_REQUIRE(c && elems >= (size_t(1) << 13), GL_ERR_ARG, "bad scratch size");      // at least one NTT tile (ntt.cuh NTT_TILE_LOG)
    c->scratch_target = elems;
    return GL_OK;
}
extern "C" const char* gl_last_error(void) { return g_gl_last_error.c_str(); }
extern "C" int gl_dev_alloc(gl_ctx* c, size_t bytes, void** d_out) {
    GL_REQUIRE(c && d_out, GL_ERR_ARG, "null argument");
    GL_TRY(c->activate());
    GL_CHECK_HIP(hipMalloc(d_out, bytes ? bytes : 8));
    return GL_OK;
}
extern "C" int gl_dev_free(gl_ctx* c, void* d_ptr) {
    // plain device memory: a null context (already destroyed by the caller) is accepted, the whole device is drained instead
    if (!d_ptr) return GL_OK;
    if (c) { GL_TRY(c->activate()); GL_CHECK_HIP(gl_stream_wait(c->stream)); }
    else {
        // the device that OWNS the block, not whichever is current in the calling thread: kernels on its streams may still use it
        hipPointerAttribute_t at;
        GL_CHECK_HIP(hipPointerGetAttributes(&at, d_ptr));
        int cur = -1;
        GL_CHECK_HIP(hipGetDevice(&cur));
        if (cur != at.device) GL_CHECK_HIP(hipSetDevice(at.device));
        GL_CHECK_HIP(hipDeviceSynchronize());
        hipError_t e = hipFree(d_ptr);
        if (cur != at.device && cur >= 0) (void)hipSetDevice(cur);
        GL_CHECK_HIP(e);
        return GL_OK;
    }
    GL_CHECK_HIP(hipFree(d_ptr));
    return GL_OK;
}
extern "C" int gl_copy_h2d(gl_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    GL_REQUIRE(c && d_dst && h_src, GL_ERR_ARG, "null argument");
    GL_TRY(c->activate());
    GL_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    GL_CHECK_HIP(gl_stream_wait(c->stream));   // pageable source must not be reused before the copy lands
    return GL_OK;
}
extern "C" int gl_copy_d2h(gl_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    GL_REQUIRE(c && h_dst && d_src, GL_ERR_ARG, "null argument");
    GL_TRY(c->activate());
    // A copy to pageable memory makes the runtime wait for the stream INSIDE hipMemcpyAsync, spinning (measured: each of 16
    // proofs in flight kept a core at 100 % there).  Small results go through one of the context's pinned buffers -- the copy is then
    // really asynchronous and the wait sleeps between polls; large ones wait for the stream first.
    if (bytes <= (size_t(4) << 20)) {
        void* stage = nullptr; size_t cap = 0;
        GL_TRY(c->pin_acquire(bytes, &stage, &cap));
        hipError_t e = hipMemcpyAsync(stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = gl_stream_wait(c->stream);
        if (e == hipSuccess) memcpy(h_dst, stage, bytes);
        c->pin_release(stage, cap);
        GL_CHECK_HIP(e);
        return GL_OK;
    }
    GL_CHECK_HIP(gl_stream_wait(c->stream));
    GL_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    GL_CHECK_HIP(gl_stream_wait(c->stream));
    return GL_OK;
}

