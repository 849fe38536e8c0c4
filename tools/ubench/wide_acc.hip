// Check of GlxWideAcc2 (gl64_gfx950.cuh: unreduced alpha-weighted sums of the quotient kernels) and of the 7 y = 8 y - y step
// against the host's unsigned __int128 arithmetic, on random and edge terms.
//   hipcc -O3 --offload-arch=gfx950 -I plonky2_demo_amd/csrc tools/ubench/wide_acc.hip -o tools/ubench/bin/wide_acc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "gl64.cuh"
#include "gl64_gfx950.cuh"

static const uint64_t P = 0xFFFFFFFF00000001ULL;
static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static uint64_t h_mod(unsigned __int128 x) { return (uint64_t)(x % P); }

__global__ void k_wide(const gl_t* terms, const gl_t* weights, uint32_t T, gl_t* out) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    GlxWideAcc2 acc; acc.clear();
#pragma unroll 1
    for (uint32_t t = 0; t < T; t++) acc.mac(terms[(size_t)lane * T + t], weights[t], weights[T + t]);
    out[2 * lane] = acc.sum(0); out[2 * lane + 1] = acc.sum(1);
}
__global__ void k_times7(const gl_t* a, gl_t* o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = glx_sub_cc(glx_shl_c<3>(a[i]), a[i]);
}

// the factor chains of the quotient's permutation argument, canonical formulation against the general one
__global__ void k_factors(const gl_t* wv_, const gl_t* sg_, const gl_t* bxk_, gl_t beta, gl_t gamma, gl_t* o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gl_t wv = wv_[i], sg = sg_[i], bxk = bxk_[i];
    // numerator factor w + (beta k x) + gamma, denominator factor w + beta sigma + gamma (vanishing_poly.rs:279-300)
    const gl_t f = glx_add_cc(glx_add_cc(wv, bxk), gamma);
    gl_t sb, u1, u2;
    glx_mul3<true>(sg, beta, 0, 0, 0, 0, sb, u1, u2);
    const gl_t g = glx_add_cc(glx_add_cc(wv, sb), gamma);
    const gl_t f_ref = gl_canon(gl_add(gl_add(wv, bxk), gamma));
    const gl_t g_ref = gl_canon(gl_add(gl_mul_add(wv, beta, sg), gamma));
    o[4 * i] = f; o[4 * i + 1] = g; o[4 * i + 2] = f_ref; o[4 * i + 3] = g_ref;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint32_t lanes = 4096, T = 300;
    const uint64_t edge[] = {0, 1, P - 1, P, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL, 0x00000000FFFFFFFFULL, 0x8000000000000000ULL};
    std::vector<uint64_t> terms((size_t)lanes * T), w(2 * T);
    uint64_t seed = 99;
    for (auto& v : terms) { v = splitmix(seed); if ((v & 15) == 0) v = edge[(v >> 4) & 7]; }
    for (uint32_t lane = 0; lane < 64; lane++) for (uint32_t t = 0; t < T; t++) terms[(size_t)lane * T + t] = edge[(lane + t) & 7] | (lane < 8 ? 0 : 0);
    for (uint32_t t = 0; t < 2 * T; t++) w[t] = (t % 7 == 0) ? edge[(t / 7) & 7] : splitmix(seed);
    for (uint32_t t = 0; t < T; t++) if (t < 40) { w[t] = 0xFFFFFFFFFFFFFFFFULL; w[T + t] = P - 1; }       // worst-case wraps
    gl_t *dt, *dw, *dout;
    (void)hipMalloc((void**)&dt, terms.size() * 8); (void)hipMalloc((void**)&dw, w.size() * 8); (void)hipMalloc((void**)&dout, lanes * 16);
    (void)hipMemcpy(dt, terms.data(), terms.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_wide, dim3(lanes / 256), dim3(256), 0, 0, dt, dw, T, dout);
    std::vector<uint64_t> out(2 * lanes);
    if (hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    int bad = 0;
    for (uint32_t lane = 0; lane < lanes; lane++)
        for (int b = 0; b < 2; b++) {
            uint64_t ref = 0;
            for (uint32_t t = 0; t < T; t++) ref = h_mod((unsigned __int128)ref + (unsigned __int128)h_mod((unsigned __int128)(terms[(size_t)lane * T + t] % P) * (w[b * T + t] % P)));
            const uint64_t got = out[2 * lane + b];
            if (got != ref) { if (bad < 5) printf("  wide sum MISMATCH lane %u sum %d got=%016llx want=%016llx\n", lane, b, (unsigned long long)got, (unsigned long long)ref); bad++; }
        }
    printf("GlxWideAcc2: %u lanes x %u terms x 2 sums: %s (%d mismatches)\n", lanes, T, bad ? "FAIL" : "ok", bad);
    std::vector<uint64_t> a(1 << 16);
    for (size_t i = 0; i < a.size(); i++) a[i] = (i < 8 ? edge[i] : splitmix(seed)) % P;
    gl_t *da, *dob;
    (void)hipMalloc((void**)&da, a.size() * 8); (void)hipMalloc((void**)&dob, a.size() * 8);
    (void)hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_times7, dim3((unsigned)(a.size() / 256)), dim3(256), 0, 0, da, dob, a.size());
    std::vector<uint64_t> o7(a.size());
    (void)hipMemcpy(o7.data(), dob, a.size() * 8, hipMemcpyDeviceToHost);
    int bad7 = 0;
    for (size_t i = 0; i < a.size(); i++) if (o7[i] != h_mod((unsigned __int128)a[i] * 7)) { if (bad7 < 5) printf("  x7 MISMATCH a=%016llx got=%016llx\n", (unsigned long long)a[i], (unsigned long long)o7[i]); bad7++; }
    printf("7 y = 8 y - y: %s (%d mismatches)\n", bad7 ? "FAIL" : "ok", bad7);
    {
        const size_t n = 1 << 16;
        std::vector<uint64_t> wv(n), sg(n), bx(n);
        for (size_t i = 0; i < n; i++) { wv[i] = splitmix(seed) % P; sg[i] = splitmix(seed) % P; bx[i] = splitmix(seed) % P; if (i < 64) { wv[i] = edge[i & 7] % P; sg[i] = edge[(i >> 3) & 7] % P; bx[i] = P - 1; } }
        gl_t *dw, *ds, *db, *dof;
        (void)hipMalloc((void**)&dw, n * 8); (void)hipMalloc((void**)&ds, n * 8); (void)hipMalloc((void**)&db, n * 8); (void)hipMalloc((void**)&dof, n * 32);
        (void)hipMemcpy(dw, wv.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(ds, sg.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, bx.data(), n * 8, hipMemcpyHostToDevice);
        const uint64_t beta = splitmix(seed) % P, gamma = P - 5;
        hipLaunchKernelGGL(k_factors, dim3((unsigned)(n / 256)), dim3(256), 0, 0, dw, ds, db, beta, gamma, dof, n);
        std::vector<uint64_t> of(4 * n);
        (void)hipMemcpy(of.data(), dof, n * 32, hipMemcpyDeviceToHost);
        int badf = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t fr = h_mod((unsigned __int128)wv[i] + bx[i] + gamma), gr = h_mod((unsigned __int128)h_mod((unsigned __int128)sg[i] * beta) + wv[i] + gamma);
            if (of[4 * i] != fr || of[4 * i + 1] != gr || of[4 * i + 2] != fr || of[4 * i + 3] != gr) { if (badf < 5) printf("  factors MISMATCH i=%zu f=%016llx f_ref=%016llx want=%016llx | g=%016llx g_ref=%016llx want=%016llx\n", i, (unsigned long long)of[4 * i], (unsigned long long)of[4 * i + 2], (unsigned long long)fr, (unsigned long long)of[4 * i + 1], (unsigned long long)of[4 * i + 3], (unsigned long long)gr); badf++; }
        }
        printf("factor chains: %s (%d mismatches)\n", badf ? "FAIL" : "ok", badf);
        bad += badf;
    }
    return (bad || bad7) ? 1 : 0;
}
