// Does a SALU instruction that reads an SGPR pair a VALU instruction has just written (a compare result or a carry-out) see the
// new value on gfx950, and after how many wait states?  glx_shl_c<36/48/60> returned want + EPS on x = h 2^(96-E):
// its s_andn2_b64 read the borrow mask ONE instruction after v_subbrev_co_u32_e64 wrote it.  Every probe first sets the SGPR pair
// to the complement of what the VALU instruction will write, so a stale read is visible in every lane.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/sgpr_hazard.hip -o /tmp/sgpr_hazard && /tmp/sgpr_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

// PROD: 0 = v_cmp_gt_u32_e64, 1 = v_sub_co_u32_e64 (borrow), 2 = v_subbrev_co_u32_e64 with vcc in, 3 = v_mad_u64_u32 carry
// GAP: text placed between the VALU producer and the SALU consumer
#define PROBE(NAME, GAP)                                                                                                 \
    template <int PROD>                                                                                                  \
    __device__ __forceinline__ uint32_t NAME(uint32_t x, uint32_t y) {                                                   \
        uint64_t m, o; uint32_t r, d0; uint64_t d1;                                                                      \
        if constexpr (PROD == 0)                                                                                         \
            asm volatile("v_cmp_le_u32_e64 %[m], %[x], %[y]\n\ts_nop 7\n\t"                                             \
                         "v_cmp_gt_u32_e64 %[m], %[x], %[y]\n\t" GAP "s_and_b64 %[o], %[m], exec\n\ts_nop 7\n\t"        \
                         "v_cndmask_b32_e64 %[r], 0, 1, %[o]"                                                            \
                         : [m] "=&s"(m), [o] "=&s"(o), [r] "=&v"(r) : [x] "v"(x), [y] "v"(y) : "scc", "vcc");                   \
        else if constexpr (PROD == 1)                                                                                    \
            asm volatile("v_cmp_ge_u32_e64 %[m], %[y], %[x]\n\ts_nop 7\n\t"                                             \
                         "v_sub_co_u32_e64 %[d], %[m], %[y], %[x]\n\t" GAP "s_and_b64 %[o], %[m], exec\n\ts_nop 7\n\t"  \
                         "v_cndmask_b32_e64 %[r], 0, 1, %[o]"                                                            \
                         : [m] "=&s"(m), [o] "=&s"(o), [r] "=&v"(r), [d] "=&v"(d0) : [x] "v"(x), [y] "v"(y) : "scc", "vcc");    \
        else if constexpr (PROD == 2)                                                                                    \
            asm volatile("v_cmp_le_u32_e64 %[m], %[x], %[y]\n\t"                                                        \
                         "v_cmp_gt_u32_e64 vcc, %[x], %[y]\n\ts_nop 7\n\t"                                              \
                         "v_subbrev_co_u32_e64 %[d], %[m], 0, 0, vcc\n\t" GAP "s_and_b64 %[o], %[m], exec\n\ts_nop 7\n\t" \
                         "v_cndmask_b32_e64 %[r], 0, 1, %[o]"                                                            \
                         : [m] "=&s"(m), [o] "=&s"(o), [r] "=&v"(r), [d] "=&v"(d0) : [x] "v"(x), [y] "v"(y) : "scc", "vcc"); \
        else                                                                                                             \
            asm volatile("v_cmp_le_u32_e64 %[m], %[x], %[y]\n\ts_nop 7\n\t"                                             \
                         "v_mad_u64_u32 %[d], %[m], %[x], 1, %[a]\n\t" GAP "s_and_b64 %[o], %[m], exec\n\ts_nop 7\n\t"   \
                         "v_cndmask_b32_e64 %[r], 0, 1, %[o]"                                                            \
                         : [m] "=&s"(m), [o] "=&s"(o), [r] "=&v"(r), [d] "=&v"(d1)                                       \
                         : [x] "v"(x), [y] "v"(y), [a] "v"(~(uint64_t)y) : "scc", "vcc");                                       \
        return r;                                                                                                        \
    }
PROBE(gap0, "")
PROBE(gap_n0, "s_nop 0\n\t")
PROBE(gap_n1, "s_nop 1\n\t")
PROBE(gap_n2, "s_nop 2\n\t")
PROBE(gap_n3, "s_nop 3\n\t")
PROBE(gap_n5, "s_nop 5\n\t")
PROBE(gap_v1, "v_mov_b32 %[r], 0\n\t")
PROBE(gap_v2, "v_mov_b32 %[r], 0\n\tv_mov_b32 %[r], 1\n\t")
PROBE(gap_v3, "v_mov_b32 %[r], 0\n\tv_mov_b32 %[r], 1\n\tv_mov_b32 %[r], 2\n\t")
PROBE(gap_cmp1, "v_cmp_gt_u32_e64 vcc, %[x], %[y]\n\t")                      // the neighbour in glx_shl_c is a compare
PROBE(gap_n0cmp1, "s_nop 0\n\tv_cmp_eq_u32_e64 vcc, %[x], %[y]\n\t")
PROBE(gap_s1, "s_mov_b64 %[o], 0\n\t")                                        // an unrelated SALU instruction first
PROBE(gap_s2, "s_mov_b64 %[o], 0\n\ts_mov_b64 %[o], 1\n\t")

template <int WHICH, int PROD>
__global__ void k_probe(const uint32_t* x, const uint32_t* y, uint32_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r;
    if constexpr (WHICH == 0) r = gap0<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 1) r = gap_n0<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 2) r = gap_n1<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 3) r = gap_n2<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 4) r = gap_n3<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 5) r = gap_n5<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 6) r = gap_v1<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 7) r = gap_v2<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 8) r = gap_v3<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 9) r = gap_cmp1<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 10) r = gap_n0cmp1<PROD>(x[i], y[i]);
    else if constexpr (WHICH == 11) r = gap_s1<PROD>(x[i], y[i]);
    else r = gap_s2<PROD>(x[i], y[i]);
    out[i] = r;
}

static const char* kGap[] = {"none", "s_nop 0", "s_nop 1", "s_nop 2", "s_nop 3", "s_nop 5", "1 v_mov", "2 v_mov", "3 v_mov", "1 v_cmp",
                             "s_nop 0 + v_cmp", "1 s_mov", "2 s_mov"};
static const char* kProd[] = {"v_cmp_gt_u32_e64", "v_sub_co_u32_e64", "v_subbrev_co_u32_e64", "v_mad_u64_u32 carry"};

template <int WHICH, int PROD>
static int run(const uint32_t* dx, const uint32_t* dy, uint32_t* dout, const std::vector<uint32_t>& hx, const std::vector<uint32_t>& hy) {
    const size_t n = hx.size();
    std::vector<uint32_t> ho(n);
    int bad = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL((k_probe<WHICH, PROD>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, dout, n);
        (void)hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; i++) {
            uint32_t want;
            if (PROD == 0) want = hx[i] > hy[i];
            else if (PROD == 1) want = hy[i] < hx[i];                                   // borrow of y - x
            else if (PROD == 2) want = hx[i] > hy[i];                                   // borrow of 0 - 0 - vcc
            else want = ((uint64_t)hx[i] + ~(uint64_t)hy[i]) < (uint64_t)hx[i];         // carry of x + ~y (64-bit): x > y
            if (ho[i] != want) bad++;
        }
    }
    printf("%-22s gap %-16s %s (%d wrong lanes of %zu)\n", kProd[PROD], kGap[WHICH], bad ? "STALE" : "ok", bad, 4 * n);
    return bad;
}
template <int PROD>
static int run_all(const uint32_t* dx, const uint32_t* dy, uint32_t* dout, const std::vector<uint32_t>& hx, const std::vector<uint32_t>& hy) {
    int bad = 0;
    bad += run<0, PROD>(dx, dy, dout, hx, hy); bad += run<1, PROD>(dx, dy, dout, hx, hy); bad += run<2, PROD>(dx, dy, dout, hx, hy);
    bad += run<3, PROD>(dx, dy, dout, hx, hy); bad += run<4, PROD>(dx, dy, dout, hx, hy); bad += run<5, PROD>(dx, dy, dout, hx, hy);
    bad += run<6, PROD>(dx, dy, dout, hx, hy); bad += run<7, PROD>(dx, dy, dout, hx, hy); bad += run<8, PROD>(dx, dy, dout, hx, hy);
    bad += run<9, PROD>(dx, dy, dout, hx, hy); bad += run<10, PROD>(dx, dy, dout, hx, hy); bad += run<11, PROD>(dx, dy, dout, hx, hy);
    bad += run<12, PROD>(dx, dy, dout, hx, hy);
    return bad;
}

// ---- the failing function itself (glx_shl_c<36> as of round 2), with GAP spliced in front of the first scalar instruction ----------
#define SHL36(NAME, GAP)                                                                                                 \
    __device__ __forceinline__ uint64_t NAME(uint64_t x) {                                                               \
        uint64_t z, c;                                                                                                   \
        const uint64_t a = x << 36, u = x >> 28;                                                                         \
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((uint32_t)u), "v"(a));                          \
        uint32_t y0 = (uint32_t)z, y1 = (uint32_t)(z >> 32), f0, f1;                                                     \
        uint64_t bw, t, g;                                                                                               \
        asm("s_nop 0\n\t"                                                                                                \
            "v_sub_co_u32 %[y0], vcc, %[y0], %[h]\n\t"                                                                   \
            "s_nop 1\n\t"                                                                                                \
            "v_subbrev_co_u32_e64 %[y1], %[bw], 0, %[y1], vcc"                                                           \
            : [y0] "+v"(y0), [y1] "+v"(y1), [bw] "=&s"(bw) : [h] "v"((uint32_t)(u >> 32)) : "vcc");                      \
        const uint64_t y = ((uint64_t)y1 << 32) | y0;                                                                    \
        asm("v_cmp_gt_u64_e64 %[g], %[y], %[pm1]\n\t" GAP                                                                \
            "s_andn2_b64 %[t], %[bw], %[c]\n\t"                                                                          \
            "s_andn2_b64 %[bw], %[c], %[bw]\n\t"                                                                         \
            "s_nop 0\n\t"                                                                                                \
            "s_andn2_b64 %[g], %[g], %[t]\n\t"                                                                           \
            "s_or_b64 %[bw], %[g], %[bw]\n\t"                                                                            \
            "v_cndmask_b32_e64 %[f1], 0, -1, %[t]\n\t"                                                                   \
            "v_cndmask_b32_e64 %[f0], 0, -1, %[bw]\n\t"                                                                  \
            "v_sub_u32 %[f0], %[f0], %[f1]"                                                                              \
            : [f0] "=&v"(f0), [f1] "=&v"(f1), [t] "=&s"(t), [g] "=&s"(g), [bw] "+s"(bw)                                  \
            : [y] "v"(y), [pm1] "s"(0xFFFFFFFF00000000ULL), [c] "s"(c)                                                   \
            : "scc");                                                                                                    \
        return y + (((uint64_t)f1 << 32) | f0);                                                                          \
    }
SHL36(shl36_asis, "")
SHL36(shl36_n0, "s_nop 0\n\t")
SHL36(shl36_n1, "s_nop 1\n\t")
SHL36(shl36_n2, "s_nop 2\n\t")
SHL36(shl36_n3, "s_nop 3\n\t")
SHL36(shl36_n4, "s_nop 4\n\t")
template <int W>
__global__ void k_shl36(const uint64_t* x, uint64_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t r;
    if constexpr (W == 0) r = shl36_asis(x[i]);
    else if constexpr (W == 1) r = shl36_n0(x[i]);
    else if constexpr (W == 2) r = shl36_n1(x[i]);
    else if constexpr (W == 3) r = shl36_n2(x[i]);
    else if constexpr (W == 4) r = shl36_n3(x[i]);
    else r = shl36_n4(x[i]);
    out[i] = r;
}
template <int W>
static void run_shl36(const char* name) {
    const uint64_t P = 0xFFFFFFFF00000001ULL;
    const size_t n = 1 << 16;
    std::vector<uint64_t> hx(n), ho(n);
    uint64_t s = 7;
    for (size_t i = 0; i < n; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t r = s ^ (s >> 29);
        if (i % 3 == 0) hx[i] = (uint64_t)(1 + (r % 15)) << 60;            // h 2^60: a = 0, m = 0, h != 0 -> borrow without carry
        else if (i % 3 == 1) hx[i] = r % P;
        else hx[i] = ((r % P) >> 40) << 40;
        if (hx[i] >= P) hx[i] -= P;
    }
    uint64_t *dx, *dout;
    (void)hipMalloc((void**)&dx, n * 8); (void)hipMalloc((void**)&dout, n * 8);
    (void)hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_shl36<W>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, dx, dout, n);
    (void)hipMemcpy(ho.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0, bad_h = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned __int128 v = hx[i];
        for (int k = 0; k < 36; k++) v = (v * 2) % P;
        if (ho[i] != (uint64_t)v) { bad++; if (i % 3 == 0) bad_h++; }
    }
    printf("glx_shl_c<36> body, gap %-10s: %d wrong of %zu (%d of them on h 2^60 inputs)\n", name, bad, n, bad_h);
    (void)hipFree(dx); (void)hipFree(dout);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = 1 << 20;
    std::vector<uint32_t> hx(n), hy(n);
    uint64_t s = 99;
    for (size_t i = 0; i < n; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL; hx[i] = (uint32_t)(s >> 33);
        s = s * 6364136223846793005ULL + 1442695040888963407ULL; hy[i] = (uint32_t)(s >> 33);
    }
    uint32_t *dx, *dy, *dout;
    (void)hipMalloc((void**)&dx, n * 4); (void)hipMalloc((void**)&dy, n * 4); (void)hipMalloc((void**)&dout, n * 4);
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, hy.data(), n * 4, hipMemcpyHostToDevice);
    int stale = 0;
    stale += run_all<0>(dx, dy, dout, hx, hy);
    stale += run_all<1>(dx, dy, dout, hx, hy);
    stale += run_all<2>(dx, dy, dout, hx, hy);
    stale += run_all<3>(dx, dy, dout, hx, hy);
    run_shl36<0>("as is"); run_shl36<1>("s_nop 0"); run_shl36<2>("s_nop 1"); run_shl36<3>("s_nop 2"); run_shl36<4>("s_nop 3"); run_shl36<5>("s_nop 4");
    printf("probes that read a stale SGPR: %s\n", stale ? "some (see above)" : "none");
    return 0;       // a probe, not a test: the exit status does not depend on the findings
}
