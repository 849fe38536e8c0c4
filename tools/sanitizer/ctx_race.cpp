// ThreadSanitizer harness for the host side of a library context (plonky2_demo_amd/csrc/context.hip) over the stub HIP runtime of
// hip_stub.cpp -- CPU only, never on the GPU box.  The pattern is bench.py's and tools/soak.py's: 16 lanes, each with its own
// context, prove in a loop (pool blocks taken and returned, small device-to-host results copied through the context's pinned
// staging buffers), and every lane now and then reads the SHARED circuit's Merkle cap through the circuit's context while that
// context is busy on its own thread (gl_batch_cap -> gl_copy_d2h on another context's stream).  Handles retain and release the
// contexts from several threads; the contexts are destroyed while handles still live.
//   round 2's bug (one pinned staging buffer per context) is reproduced with -DCTX_RACE_SINGLE_BUFFER: TSan reports the race and
//   the copied bytes come back wrong.
#include "../../plonky2_demo_amd/csrc/context.hpp"
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static gl_ctx* make_ctx() {                 // gl_ctx_create without the twiddle-table kernel launches
    gl_ctx* c = new gl_ctx();
    c->device = 0;
    if (hipStreamCreateWithFlags(&c->stream, 0) != hipSuccess) abort();
    c->own_stream = true;
    return c;
}

int main(int argc, char** argv) {
    const int lanes = argc > 1 ? atoi(argv[1]) : 16, iters = argc > 2 ? atoi(argv[2]) : 60;
    gl_ctx* circuit_ctx = make_ctx();
    std::vector<gl_ctx*> ctxs(lanes);
    ctxs[0] = circuit_ctx;                                   // the first lane proves on the circuit's own context, as in bench.py
    for (int i = 1; i < lanes; i++) ctxs[i] = make_ctx();
    // the circuit's "Merkle cap" (device memory of the circuit's context) and one distinct result per lane
    const size_t cap_words = 64;
    uint64_t* d_cap = nullptr;
    if (gl_dev_alloc(circuit_ctx, cap_words * 8, (void**)&d_cap) != GL_OK) return 2;
    for (size_t k = 0; k < cap_words; k++) d_cap[k] = 0xCA9000 + k;
    std::vector<uint64_t*> d_res(lanes);
    for (int i = 0; i < lanes; i++) {
        if (gl_dev_alloc(ctxs[i], 4096 * 8, (void**)&d_res[i]) != GL_OK) return 2;
        for (size_t k = 0; k < 4096; k++) d_res[i][k] = ((uint64_t)i << 32) | k;
    }
    std::atomic<int> bad{0};
    auto lane = [&](int me) {
        gl_ctx* c = ctxs[me];
        c->retain();                                         // a handle (gl_batch) created on this context
        std::vector<uint64_t> host(4096), cap(cap_words);
        for (int it = 0; it < iters; it++) {
            void* blocks[4];
            for (int b = 0; b < 4; b++) if (c->pool_alloc((size_t)(1 + (it + b) % 3) << 16, &blocks[b]) != GL_OK) bad++;
            const size_t n = 256 + 64 * (size_t)((it * 7 + me) % 60);
#ifdef CTX_RACE_SINGLE_BUFFER
            {   // round 2's first version of the staging: ONE pinned buffer per context
                static std::vector<void*> one(64, nullptr);
                gl_ctx* cc = (it % 7 == 3) ? circuit_ctx : c;
                int idx = 0; for (int k = 0; k < (int)ctxs.size(); k++) if (ctxs[k] == cc) idx = k;
                static std::mutex mk; { std::lock_guard<std::mutex> lk(mk); if (!one[idx]) (void)hipHostMalloc(&one[idx], 1 << 20, 0); }
                const void* src = (cc == circuit_ctx && it % 7 == 3) ? (const void*)d_cap : (const void*)d_res[me];
                const size_t bytes = (cc == circuit_ctx && it % 7 == 3) ? cap_words * 8 : n * 8;
                (void)hipMemcpyAsync(one[idx], src, bytes, hipMemcpyDeviceToHost, cc->stream);
                (void)gl_stream_wait(cc->stream);
                memcpy((it % 7 == 3) ? (void*)cap.data() : (void*)host.data(), one[idx], bytes);
            }
#else
            if (it % 7 == 3) {                               // every 7th proof is verified: the circuit's cap through the CIRCUIT's context
                if (gl_copy_d2h(circuit_ctx, cap.data(), d_cap, cap_words * 8) != GL_OK) bad++;
            } else if (gl_copy_d2h(c, host.data(), d_res[me], n * 8) != GL_OK) bad++;
#endif
            if (it % 7 == 3) { for (size_t k = 0; k < cap_words; k++) if (cap[k] != 0xCA9000 + k) { bad++; break; } }
            else for (size_t k = 0; k < n; k++) if (host[k] != (((uint64_t)me << 32) | k)) { bad++; break; }
            for (int b = 0; b < 4; b++) c->pool_release(blocks[b]);
            if (it % 16 == 5) c->pool_trim();
        }
        (void)gl_ctx_synchronize(c);
        gl_ctx_release(c);                                   // the handle dies
    };
    std::vector<std::thread> ths;
    for (int i = 0; i < lanes; i++) ths.emplace_back(lane, i);
    // meanwhile the creator drops its references to half of the contexts: handles keep them alive until the lanes finish
    for (int i = 1; i < lanes; i += 2) gl_ctx_destroy(ctxs[i]);
    for (auto& t : ths) t.join();
    for (int i = 0; i < lanes; i++) (void)gl_dev_free(i % 2 ? nullptr : ctxs[i], d_res[i]);
    (void)gl_dev_free(circuit_ctx, d_cap);
    for (int i = 0; i < lanes; i += 2) gl_ctx_destroy(ctxs[i]);
    printf("lanes %d, iterations %d: %d wrong copies or failed calls\n", lanes, iters, bad.load());
    return bad.load() ? 1 : 0;
}
