// dag_probe.hip -- prices the two ways a chain of dependent SGD updates on ONE parameter row can run on gfx950:
//  A. hand-over: every step of the chain runs on another wave (another CU / XCD): wait for the row's version counter, load the 256-byte
//     row (sc1), touch it, store it (sc1), drain, publish version + 1.  This is what a generic DAG executor pays per dependency edge.
//  B. residency: ONE lane group keeps the row in registers and streams through the chain's partner rows (random 256-byte rows, loaded
//     D steps ahead): per step a 16-lane DPP-style dot chain, an axpy on both rows, the partner row stored.  This is what a row-owner
//     ("walker") pays per step.
// Every spin is bounded (a timeout raises *err and the kernel leaves).  hipcc --offload-arch=gfx950 -O3 -o dag_probe dag_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_row_sc1(const float4 *p) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_row_sc1(float4 *p, float4 w) {
    f4 v = {w.x, w.y, w.z, w.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// A: `steps` hand-overs of one row around `nw` participating workgroups (workgroup w runs steps s with s % nw == w); 16 lanes carry the row
__global__ void k_handoff(float4 *row, unsigned *ver, int steps, int nw, int stride, unsigned *err, float *sink, const float4 *noise, long noise_n, int load) {
    const int wg = blockIdx.x;
    const int lane = threadIdx.x;
    int me = -1;
    for (int j = 0; j < nw; j++) if (wg == j * stride) me = j;
    if (me < 0) {   // bystanders: stream HBM while the chain runs (load = 1), or leave
        if (!load) return;
        float acc = 0.f;
        long i = ((long)wg * 64 + lane) % noise_n;
        while (ld_sc1(ver) < (unsigned)steps && ld_sc1(err) == 0u) {
            for (int r = 0; r < 64; r++) { float4 v = noise[i]; acc += v.x + v.w; i = (i + 104729) % noise_n; }
        }
        if (acc == 12345.f) sink[0] = acc;
        return;
    }
    for (int s = me; s < steps; s += nw) {
        long spins = 0;
        while (ld_sc1(ver) != (unsigned)s) {
            if (++spins > 20000000L) { st_sc1(err, 1u + (unsigned)s); return; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane < 16) {
            float4 v = ld_row_sc1(row + lane);
            v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
            st_row_sc1(row + lane, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_sc1(ver, (unsigned)s + 1u);
    }
}

// B: one wave = 4 lane groups of 16 lanes (k = 64: one float4 per lane); group g walks its own chain of `steps` partner rows (ids in
// idx[g * steps + t]) with its pivot row in registers, partner rows requested D steps ahead
template <int D>
__global__ void k_walker(float4 *W, const unsigned *idx, int steps, float *out) {
    const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
    const unsigned *my = idx + (long)(blockIdx.x * 4 + g) * steps;
    float4 q = make_float4(0.01f * lane, 0.02f, 0.03f, 0.04f);
    float4 ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = W[(long)my[d] * 16 + lane];
    for (int t = 0; t < steps; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const float4 p = ring[d];
            const unsigned row = my[t + d];
            if (t + d + D < steps) ring[d] = W[(long)my[t + d + D] * 16 + lane];
            float part = p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
            // the reference's dot is a serial chain over the 16 chunks: 15 dependent adds through DPP row_shr:1
            float acc = part;
#pragma unroll
            for (int j = 1; j < 16; j++) {
                float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x111, 0xf, 0xf, false));   // row_shr:1
                acc = (lane >= j) ? up + part : acc;
            }
            const float dot = __shfl(acc, (threadIdx.x & ~15) + 15);
            const float err = 3.0f - dot, s = 0.005f * err;
            float4 pn = make_float4((p.x + s * q.x) * 0.99998f, (p.y + s * q.y) * 0.99998f, (p.z + s * q.z) * 0.99998f, (p.w + s * q.w) * 0.99998f);
            q = make_float4((q.x + s * p.x) * 0.99998f, (q.y + s * p.y) * 0.99998f, (q.z + s * p.z) * 0.99998f, (q.w + s * p.w) * 0.99998f);
            W[(long)row * 16 + lane] = pn;
        }
    }
    if (lane == 0) out[blockIdx.x * 4 + g] = q.x;
}

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float4 *row; unsigned *ver, *err; float *sink; float4 *noise;
    const long noise_n = 1L << 26;   // 1 GiB of float4
    CK(hipMalloc(&row, 256)); CK(hipMalloc(&ver, 256)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&noise, noise_n * 16));
    CK(hipHostMalloc((void **)&err, 4, hipHostMallocMapped));
    CK(hipMemset(noise, 0, noise_n * 16));
    printf("== A. hand-over of a 256-byte row + version counter between workgroups (us per dependent step)\n");
    const int steps = 20000;
    for (int load = 0; load <= 1; load++)
        for (int nw : {1, 2, 8, 64, 256})
            for (int stride : {1, 8}) {
                if (nw * stride > 256 * 8) continue;
                const int grid = load ? 1024 : nw * stride;
                if (nw * stride > grid) continue;
                CK(hipMemset(row, 0, 256)); CK(hipMemset(ver, 0, 256)); *err = 0;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_handoff, dim3(grid), dim3(64), 0, 0, row, ver, steps, nw, stride, err, sink, noise, noise_n, load);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                float4 h; CK(hipMemcpy(&h, row, 16, hipMemcpyDeviceToHost));
                printf("  %-9s workgroups %3d, ids %s: %.3f us per step (row[0] = %.0f of %d, err %u)\n", load ? "streaming" : "idle", nw,
                       stride == 1 ? "0,1,2..   (neighbouring XCDs)" : "0,8,16..  (one XCD)         ", ms * 1e3 / steps, h.x, steps, *err);
            }
    printf("== B. one lane group keeps the row in registers and walks its chain (random 256-byte partner rows of a 256 MB matrix, D rows ahead)\n");
    const long rows = 1L << 20;
    float4 *W; CK(hipMalloc(&W, rows * 256)); CK(hipMemset(W, 0, rows * 256));
    const int wsteps = 4096 * 8;
    for (int waves : {1, 256, 2048, 8192}) {
        std::vector<unsigned> h((size_t)waves * 4 * wsteps);
        unsigned x = 12345u;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) % (unsigned)rows; }
        unsigned *idx; float *out;
        CK(hipMalloc(&idx, h.size() * 4)); CK(hipMalloc(&out, (size_t)waves * 16));
        CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int D : {1, 4, 8}) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (D == 1) hipLaunchKernelGGL(k_walker<1>, dim3(waves), dim3(64), 0, 0, W, idx, wsteps, out);
                if (D == 4) hipLaunchKernelGGL(k_walker<4>, dim3(waves), dim3(64), 0, 0, W, idx, wsteps, out);
                if (D == 8) hipLaunchKernelGGL(k_walker<8>, dim3(waves), dim3(64), 0, 0, W, idx, wsteps, out);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double inst = (double)waves * 4 * wsteps;
            printf("  waves %5d (x4 chains), D = %d: %.1f ns per step of a chain, %.2f G steps/s in all, %.2f TB/s of partner rows (read + write)\n", waves, D,
                   ms * 1e6 / wsteps, inst / ms / 1e6, inst * 512 / ms / 1e9);
        }
        CK(hipFree(idx)); CK(hipFree(out));
    }
    return 0;
}
