// pmc_calib.hip -- known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_EA0_* on gfx950
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel touches each row of a table exactly once through a random permutation (no reuse; the table is much larger than
// L2 + Infinity Cache), in the lane layouts the training kernels use:
//   gather_rows<LPI>   one ROW of LPI x 16 B per lane group (LPI = 16: 256-B rows = basicMF k=64; 32: 512-B rows = k=128), read only
//   update_rows<LPI>   the same rows read, scaled and written back (the SGD read-modify-write)
//   gather_words       one 4-B word per lane at a random index (bias gathers), read only
//   update_words       4-B read-modify-write at a random index (bias updates)
//   stream_read        16 B per lane, consecutive (the pattern the guide's x2 rule was measured on)
// usage: pmc_calib [rows_log2=23]   (run under rocprofv3 --kernel-trace --pmc ...)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4 *p) { const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LPI>
__global__ __launch_bounds__(256) void gather_rows(const float4 *tab, const unsigned *perm, long n, float *sink) {
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPI;
    const int l = threadIdx.x % LPI;
    if (g >= n) return;
    const float4 v = nt_load(&tab[(size_t)perm[g] * LPI + l]);
    if (v.x == 12345.678f) sink[0] = v.y;   // never true: keeps the load
}
template <int LPI>
__global__ __launch_bounds__(256) void update_rows(float4 *tab, const unsigned *perm, long n) {
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPI;
    const int l = threadIdx.x % LPI;
    if (g >= n) return;
    float4 v = nt_load(&tab[(size_t)perm[g] * LPI + l]);
    v.x *= 0.999f; v.y *= 0.999f; v.z *= 0.999f; v.w *= 0.999f;
    tab[(size_t)perm[g] * LPI + l] = v;
}
__global__ __launch_bounds__(256) void gather_words(const float *tab, const unsigned *perm, long n, float *sink) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float v = tab[(size_t)perm[g] * 64];   // one word per 256-B stride: every word on its own line
    if (v == 12345.678f) sink[0] = v;
}
__global__ __launch_bounds__(256) void update_words(float *tab, const unsigned *perm, long n) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    tab[(size_t)perm[g] * 64] = tab[(size_t)perm[g] * 64] * 0.999f;
}
__global__ __launch_bounds__(256) void gather_words_dense(const float *tab, const unsigned *perm, long n, float *sink) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float v = tab[perm[g]];   // a dense array of words (the bias arrays): random order, every word once
    if (v == 12345.678f) sink[0] = v;
}
__global__ __launch_bounds__(256) void update_words_dense(float *tab, const unsigned *perm, long n) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    tab[perm[g]] = tab[perm[g]] * 0.999f;
}
__global__ __launch_bounds__(256) void stream_read(const float4 *tab, long n4, float *sink) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n4) return;
    const float4 v = tab[g];
    if (v.x == 12345.678f) sink[0] = v.y;
}
__global__ __launch_bounds__(256) void stream_write(float4 *tab, long n4) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n4) return;
    tab[g] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 23;
    const long n = 1L << lg;                 // rows
    const size_t bytes = (size_t)n * 512;    // table of n 512-B rows (4 GiB at 2^23); the 256-B kernels use 2n rows of it
    float4 *tab; unsigned *perm, *perm2; float *sink;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&perm, n * sizeof(unsigned))); CK(hipMalloc(&perm2, 2 * n * sizeof(unsigned))); CK(hipMalloc(&sink, 64));
    CK(hipMemset(tab, 0, bytes));
    std::vector<unsigned> p((size_t)2 * n);
    std::mt19937 rng(7);
    std::iota(p.begin(), p.begin() + n, 0u); std::shuffle(p.begin(), p.begin() + n, rng);
    CK(hipMemcpy(perm, p.data(), n * sizeof(unsigned), hipMemcpyHostToDevice));
    std::iota(p.begin(), p.end(), 0u); std::shuffle(p.begin(), p.end(), rng);
    CK(hipMemcpy(perm2, p.data(), 2 * n * sizeof(unsigned), hipMemcpyHostToDevice));
    auto grid = [](long threads) { return dim3((unsigned)((threads + 255) / 256)); };
    printf("rows n = %ld (512-B) / %ld (256-B); expected bytes per kernel:\n", n, 2 * n);
    printf("  gather_rows<32>  read %zu  (+ perm %zu)\n", (size_t)n * 512, (size_t)n * 4);
    printf("  gather_rows<16>  read %zu  (+ perm %zu)\n", (size_t)2 * n * 256, (size_t)2 * n * 4);
    printf("  update_rows<32>  read %zu write %zu\n", (size_t)n * 512, (size_t)n * 512);
    printf("  update_rows<16>  read %zu write %zu\n", (size_t)2 * n * 256, (size_t)2 * n * 256);
    printf("  gather_words     %ld words, each on its own 256-B block\n", n);
    printf("  update_words     %ld words, each on its own 256-B block\n", n);
    printf("  gather_words_dense / update_words_dense   %ld words of a dense %zu-B array, random order\n", 2 * n, (size_t)2 * n * 4);
    printf("  stream_read / stream_write   %zu\n", bytes);
    hipLaunchKernelGGL(gather_rows<32>, grid(n * 32), dim3(256), 0, 0, tab, perm, n, sink);
    hipLaunchKernelGGL(gather_rows<16>, grid(2 * n * 16), dim3(256), 0, 0, tab, perm2, 2 * n, sink);
    hipLaunchKernelGGL(update_rows<32>, grid(n * 32), dim3(256), 0, 0, tab, perm, n);
    hipLaunchKernelGGL(update_rows<16>, grid(2 * n * 16), dim3(256), 0, 0, tab, perm2, 2 * n);
    hipLaunchKernelGGL(gather_words, grid(n), dim3(256), 0, 0, (const float *)tab, perm, n, sink);
    hipLaunchKernelGGL(update_words, grid(n), dim3(256), 0, 0, (float *)tab, perm, n);
    hipLaunchKernelGGL(gather_words_dense, grid(2 * n), dim3(256), 0, 0, (const float *)tab, perm2, 2 * n, sink);
    hipLaunchKernelGGL(update_words_dense, grid(2 * n), dim3(256), 0, 0, (float *)tab, perm2, 2 * n);
    hipLaunchKernelGGL(stream_read, grid((long)(bytes / 16)), dim3(256), 0, 0, tab, (long)(bytes / 16), sink);
    hipLaunchKernelGGL(stream_write, grid((long)(bytes / 16)), dim3(256), 0, 0, tab, (long)(bytes / 16));
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    return 0;
}
