// The sigma polynomials of build() on the device (plonky2/src/plonk/circuit_builder.rs:1007-1014 `sigma_vecs`,
// plonk/permutation_argument.rs:85-170 `WirePartition::get_sigma_polys`).
//
// Input: for every routed wire the id of its copy-constraint class (the representative the reference's union-find `Forest`
// assigns; any u64, equal ids = wires constrained equal).  The reference walks each partition subset in (row, column) order and
// maps every wire to its successor (the last to the first): sigma(wire) = k_is[column(next)] * w^row(next).  Here the subsets are
// found by ONE stable radix sort of the wires by class id -- wires enumerated in (row, column) order beforehand, so that equal
// ids keep that order -- followed by a run-start scan and a gather: no forest, no pointer chasing, 2.6 M wires in a few
// hundred microseconds at m = 64 (the host restatement sorts for 0.3 s there, 5 s at m = 128).
#include "context.hpp"
#include <hipcub/hipcub.hpp>

// key[q] = class of the q-th wire in (row, column) order, val[q] = its position col * n + row
__global__ void k_sigma_keys(const uint64_t* __restrict__ cls, uint32_t n, uint32_t ncols, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * ncols) return;
    const uint32_t row = q / ncols, col = q - row * ncols, pos = col * n + row;
    key[q] = cls[pos];
    val[q] = pos;
}
// head[q] = q if q starts a run of equal keys, else 0 (an inclusive max-scan then gives every q the start of its run)
__global__ void k_sigma_heads(const uint64_t* __restrict__ key, uint32_t total, uint32_t* __restrict__ head) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    head[q] = (q == 0 || key[q] != key[q - 1]) ? q : 0u;
}
// sigma[pos[q]] = k_is[col(next)] * w^row(next), next = the successor of q in its run (cyclically)
__global__ void k_sigma_values(const uint64_t* __restrict__ key, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ start, uint32_t total,
                               uint32_t n, const gl_t* __restrict__ k_is, const gl_t* __restrict__ xpow_lo, const gl_t* __restrict__ xpow_hi,
                               gl_t* __restrict__ sigma) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const uint32_t nq = (q + 1 < total && key[q + 1] == key[q]) ? q + 1 : start[q];
    const uint32_t nb = pos[nq], col = nb / n, row = nb - col * n;
    const gl_t w = gl_mul(xpow_lo[row & 2047u], xpow_hi[row >> 11]);        // w^row from the two-level power table
    sigma[pos[q]] = gl_canon(gl_mul(k_is[col], w));
}

// d_classes[ncols][n] (device) -> d_sigma[ncols][n] VALUES (device)
int gl_sigmas_from_classes(gl_ctx* c, const uint64_t* d_classes, uint32_t lgn, uint32_t ncols, const uint64_t* h_k_is, gl_t* d_sigma) {
    GL_REQUIRE(c && d_classes && h_k_is && d_sigma && ncols >= 1 && ncols <= 80 && lgn <= 24, GL_ERR_ARG, "gl_sigmas_from_classes: bad argument");
    GL_TRY(c->activate());
    const uint32_t n = 1u << lgn, total = n * ncols;
    hipStream_t st = c->stream;
    struct Buf { gl_ctx* c; void* p = nullptr; ~Buf() { if (p) c->pool_release(p); } };
    Buf key_in{c}, key_out{c}, val_in{c}, val_out{c}, head{c}, tmp{c}, kis{c};
    GL_TRY(c->pool_alloc((size_t)total * 8, &key_in.p)); GL_TRY(c->pool_alloc((size_t)total * 8, &key_out.p));
    GL_TRY(c->pool_alloc((size_t)total * 4, &val_in.p)); GL_TRY(c->pool_alloc((size_t)total * 4, &val_out.p));
    GL_TRY(c->pool_alloc((size_t)total * 4, &head.p));
    GL_TRY(c->pool_alloc(80 * sizeof(gl_t), &kis.p));
    std::vector<gl_t> k(80, 0);
    for (uint32_t j = 0; j < ncols; j++) k[j] = gl_canon(h_k_is[j]);
    GL_CHECK_HIP(hipMemcpyAsync(kis.p, k.data(), 80 * sizeof(gl_t), hipMemcpyHostToDevice, st));
    const unsigned blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_sigma_keys, dim3(blocks), dim3(256), 0, st, d_classes, n, ncols, (uint64_t*)key_in.p, (uint32_t*)val_in.p);
    size_t tb_sort = 0, tb_scan = 0;
    GL_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, (const uint64_t*)key_in.p, (uint64_t*)key_out.p, (const uint32_t*)val_in.p,
                                                    (uint32_t*)val_out.p, (int)total, 0, 64, st));
    GL_CHECK_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, tb_scan, (const uint32_t*)head.p, (uint32_t*)val_in.p, hipcub::Max(), (int)total, st));
    GL_TRY(c->pool_alloc(tb_sort > tb_scan ? tb_sort : tb_scan, &tmp.p));
    GL_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb_sort, (const uint64_t*)key_in.p, (uint64_t*)key_out.p, (const uint32_t*)val_in.p,
                                                    (uint32_t*)val_out.p, (int)total, 0, 64, st));           // stable: equal ids stay in (row, column) order
    hipLaunchKernelGGL(k_sigma_heads, dim3(blocks), dim3(256), 0, st, (const uint64_t*)key_out.p, total, (uint32_t*)head.p);
    GL_CHECK_HIP(hipcub::DeviceScan::InclusiveScan(tmp.p, tb_scan, (const uint32_t*)head.p, (uint32_t*)val_in.p, hipcub::Max(), (int)total, st));   // val_in := run starts
    GlPowTable xt;
    GL_TRY(c->get_pow_table(gl_host_root_of_unity(lgn), 1, (n + 2047) >> 11, &xt));
    hipLaunchKernelGGL(k_sigma_values, dim3(blocks), dim3(256), 0, st, (const uint64_t*)key_out.p, (const uint32_t*)val_out.p, (const uint32_t*)val_in.p, total, n,
                       (const gl_t*)kis.p, xt.lo, xt.hi, d_sigma);
    GL_CHECK_HIP(hipGetLastError());
    GL_CHECK_HIP(gl_stream_wait(st));          // `k` (host source of the async upload) dies here
    return GL_OK;
}
