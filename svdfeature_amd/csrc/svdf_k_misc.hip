// svdf_k_misc.hip -- multi-GPU delta kernels, layout queries
// (part of the gfx950 kernel set described at the top of svdf_device.h)
#include "svdf_device.h"

namespace svdf {

// ---- multi-GPU item-side delta exchange -------------------------------------
__global__ void k_delta_sub(const float *cur, const float *snap, float *delta, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = i; j < n; j += stride) delta[j] = cur[j] - snap[j];
}
__global__ void k_delta_add(float *cur, const float *snap, const float *delta, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = i; j < n; j += stride) cur[j] = snap[j] + delta[j];
}

int lanes_per_instance(int k) {
    int chunks = (k + 3) / 4;
    int lpi = 1;
    while (lpi < chunks && lpi < 64) lpi <<= 1;
    return lpi;   // 64 also for wide rows (k > 256: several float4 slots per lane)
}
int max_supported_factor() { return 1024; }
int max_fast_path_factor() { return 256; }

// One launch over ALL replicated ranges (W_item, biases, globals ...): pack = (current - snapshot) in the wire type,
// unpack = current <- snapshot + delta, optionally snapshot <- current so that the next window needs no copy.
// fp16 conversion is round-to-nearest-even (what a separate .half() pass would do).
__device__ __forceinline__ float *delta_slot(const DeltaRanges &R, long j, long &snap_pos) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < SVDF_MAX_DELTA_RANGES; q++) r += (q < R.n && j >= R.off[q]) ? 1 : 0;
    snap_pos = R.snap_off[r] + (j - R.off[r]);
    return R.base[r] + (j - R.off[r]);
}
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_pack(const DeltaRanges R, const float *snap, void *dst, long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        long sp;
        const float *cur = delta_slot(R, j, sp);
        const float d = *cur - snap[sp];
        if (HALF) reinterpret_cast<__half *>(dst)[j] = __float2half_rn(d);
        else reinterpret_cast<float *>(dst)[j] = d;
    }
}
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_unpack(const DeltaRanges R, float *snap, const void *src, long total, int refresh) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const float d = HALF ? __half2float(reinterpret_cast<const __half *>(src)[j]) : reinterpret_cast<const float *>(src)[j];
        long sp;
        float *cur = delta_slot(R, j, sp);
        const float v = snap[sp] + d;
        *cur = v;
        if (refresh) snap[sp] = v;
    }
}
void launch_delta_pack(const DeltaRanges &R, const float *snap, void *dst, int half, hipStream_t st) {
    const long total = R.off[R.n];
    if (total <= 0) return;
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (half) hipLaunchKernelGGL(k_delta_pack<true>, dim3((int)grid), dim3(256), 0, st, R, snap, dst, total);
    else hipLaunchKernelGGL(k_delta_pack<false>, dim3((int)grid), dim3(256), 0, st, R, snap, dst, total);
}
void launch_delta_unpack(const DeltaRanges &R, float *snap, const void *src, int half, int refresh, hipStream_t st) {
    const long total = R.off[R.n];
    if (total <= 0) return;
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (half) hipLaunchKernelGGL(k_delta_unpack<true>, dim3((int)grid), dim3(256), 0, st, R, snap, src, total, refresh);
    else hipLaunchKernelGGL(k_delta_unpack<false>, dim3((int)grid), dim3(256), 0, st, R, snap, src, total, refresh);
}
void launch_delta_sub(const float *cur, const float *snap, float *delta, long n, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_delta_sub, dim3((int)grid), dim3(256), 0, st, cur, snap, delta, n);
}
void launch_delta_add(float *cur, const float *snap, const float *delta, long n, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_delta_add, dim3((int)grid), dim3(256), 0, st, cur, snap, delta, n);
}

// sum of N packed delta buffers (virtual ranks / no RCCL): dst = sum_d src[d], fp32 accumulation, fp16 or fp32 storage
struct SumSrcs { const void *p[16]; int n; };
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_sum(const SumSrcs S, void *dst, long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        float acc = 0.0f;
        for (int d = 0; d < S.n; d++) acc += HALF ? __half2float(reinterpret_cast<const __half *>(S.p[d])[j]) : reinterpret_cast<const float *>(S.p[d])[j];
        if (HALF) reinterpret_cast<__half *>(dst)[j] = __float2half_rn(acc);
        else reinterpret_cast<float *>(dst)[j] = acc;
    }
}
void launch_delta_sum(const void *const *srcs, int n, void *dst, long total, int half, hipStream_t st) {
    if (total <= 0 || n <= 0) return;
    SumSrcs S;
    S.n = n > 16 ? 16 : n;
    for (int d = 0; d < S.n; d++) S.p[d] = srcs[d];
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (half) hipLaunchKernelGGL(k_delta_sum<true>, dim3((int)grid), dim3(256), 0, st, S, dst, total);
    else hipLaunchKernelGGL(k_delta_sum<false>, dim3((int)grid), dim3(256), 0, st, S, dst, total);
}

// Direct exchange between the ranks of ONE process (svdf_multi.cpp, amd:exchange = p2p): rank d owns elements [begin, end) of the
// packed delta; its kernel reads that slice from every rank's wire buffer through peer pointers (reduce-scatter over all xGMI
// links at once), sums it in rank order in fp32 -- the order of k_delta_sum, so virtual ranks and real devices agree bit for bit --
// and stores the sum back into the same slice of EVERY rank's buffer (all-gather), where the rank's own unpack / add kernel
// picks it up.  16-byte accesses over the aligned body of the slice, scalar head / tail.
struct PeerBufs { void *p[16]; int n; };
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_reduce_gather(const PeerBufs B, long begin, long end, const unsigned *err) {
    // a wait of this exchange timed out (svdf_ipc.cpp): the peers' buffers are incomplete -- nothing is summed and nothing is stored anywhere
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    const long stride = (long)gridDim.x * blockDim.x;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int PER = HALF ? 8 : 4;                       // elements per 16-byte access
    const long vb = (begin + PER - 1) / PER * PER, ve = end / PER * PER;
    if (vb < ve) {
        for (long j = vb / PER + tid; j < ve / PER; j += stride) {
            float acc[PER];
#pragma unroll
            for (int q = 0; q < PER; q++) acc[q] = 0.0f;
            for (int d = 0; d < B.n; d++) {
                const uint4 raw = reinterpret_cast<const uint4 *>(B.p[d])[j];
                if (HALF) {
                    const __half2 *h = reinterpret_cast<const __half2 *>(&raw);
#pragma unroll
                    for (int q = 0; q < 4; q++) { acc[2 * q] += __low2float(h[q]); acc[2 * q + 1] += __high2float(h[q]); }
                } else {
                    const float *f = reinterpret_cast<const float *>(&raw);
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[q] += f[q];
                }
            }
            uint4 out;
            if (HALF) {
                __half2 *h = reinterpret_cast<__half2 *>(&out);
#pragma unroll
                for (int q = 0; q < 4; q++) h[q] = __halves2half2(__float2half_rn(acc[2 * q]), __float2half_rn(acc[2 * q + 1]));
            } else {
                float *f = reinterpret_cast<float *>(&out);
#pragma unroll
                for (int q = 0; q < 4; q++) f[q] = acc[q];
            }
            for (int d = 0; d < B.n; d++) reinterpret_cast<uint4 *>(B.p[d])[j] = out;
        }
    }
    // head and tail of the slice, element by element
    const long nhead = (vb < ve ? vb : end) - begin, ntail = vb < ve ? end - ve : 0;
    for (long t = tid; t < nhead + ntail; t += stride) {
        const long j = t < nhead ? begin + t : ve + (t - nhead);
        float acc = 0.0f;
        for (int d = 0; d < B.n; d++) acc += HALF ? __half2float(reinterpret_cast<const __half *>(B.p[d])[j]) : reinterpret_cast<const float *>(B.p[d])[j];
        for (int d = 0; d < B.n; d++) {
            if (HALF) reinterpret_cast<__half *>(B.p[d])[j] = __float2half_rn(acc);
            else reinterpret_cast<float *>(B.p[d])[j] = acc;
        }
    }
}
void launch_delta_reduce_gather(void *const *bufs, int n, long begin, long end, int half, hipStream_t st, const unsigned *err) {
    if (end <= begin || n <= 0) return;
    PeerBufs B;
    B.n = n > 16 ? 16 : n;
    for (int d = 0; d < B.n; d++) B.p[d] = bufs[d];
    const long per = half ? 8 : 4;
    long grid = ((end - begin) / per + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 4096) grid = 4096;
    if (half) hipLaunchKernelGGL(k_delta_reduce_gather<true>, dim3((int)grid), dim3(256), 0, st, B, begin, end, err);
    else hipLaunchKernelGGL(k_delta_reduce_gather<false>, dim3((int)grid), dim3(256), 0, st, B, begin, end, err);
}

// ---- cross-PROCESS direct exchange (svdf_ipc.cpp): the ranks' wire buffers and flag pages are IPC-mapped into
// every process, so the peer-pointer reduce-scatter + all-gather of k_delta_reduce_gather serves one-process-per-GPU runs too.  Events do
// not order work across processes without a host rendezvous per window (waiting on an event that has not been recorded yet is a no-op), so
// the ranks meet through SEQUENCE FLAGS in device memory: rank `me` stores `seq` into word (phase, me) of every rank's flag page once its
// own phase is done (kernel boundary + system-scope fence before the store); a one-wave kernel on the waiting rank polls its own page
// (system-scope loads, s_sleep between polls, a spin limit that raises *err instead of hanging the queue).
struct IpcFlags { unsigned *page[16]; int n; };
// Once a wait has raised *err the exchange is dead: signals stop (the peers' waits then time out too instead of consuming a sum over
// incomplete buffers), the reduce and the block copies become no-ops, and the host fails at its next svdf_ipc_* call, at
// svdf_synchronize or at svdf_ipc_close, whichever comes first.
__global__ void k_ipc_signal(const IpcFlags F, int phase, int me, unsigned seq, const unsigned *err) {
    const int r = threadIdx.x;
    if (r >= F.n) return;
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    __threadfence_system();
    __hip_atomic_store(F.page[r] + (phase * 16 + me) * 32, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_ipc_wait(unsigned *page, int phase, int n, unsigned seq, unsigned *err, unsigned long long spin_limit) {
    const int r = threadIdx.x;
    if (r < n) {
        unsigned *f = page + (phase * 16 + r) * 32;
        unsigned long long spins = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (++spins > spin_limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                __hip_atomic_store(err, 1u + (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __threadfence_system();
}
void launch_ipc_signal(unsigned *const *pages, int n, int phase, int me, unsigned seq, const unsigned *err, hipStream_t st) {
    IpcFlags F;
    F.n = n;
    for (int r = 0; r < 16; r++) F.page[r] = r < n ? pages[r] : nullptr;
    hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(64), 0, st, F, phase, me, seq, err);
}
void launch_ipc_wait(unsigned *page, int phase, int n, unsigned seq, unsigned *err, unsigned long long spin_limit, hipStream_t st) {
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, st, page, phase, n, seq, err, spin_limit);
}
// a packed block (svdf_item_block_get layout) stored straight into a peer's mapped buffer: the stratified hand-over without a collective
__global__ __launch_bounds__(256) void k_ipc_copy(float *dst, const float *src, long n, const unsigned *err) {
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = src[j];
}
void launch_ipc_copy(float *dst, const float *src, long n, const unsigned *err, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_ipc_copy, dim3((int)grid), dim3(256), 0, st, dst, src, n, err);
}

// row r of dst (rows dst_first + r * dst_stride) <- row r of src (rows src_first + r * src_stride), `width` floats per row: packs the
// user rows a rank owns (ids = rank mod N) for the hand-over to rank 0 and unpacks them there (svdf_multi.cpp, save_model / get_view)
__global__ __launch_bounds__(256) void k_rows_strided_copy(float *dst, long dst_first, long dst_stride, const float *src, long src_first, long src_stride,
                                                           long nrows, int width) {
    const long total = nrows * width;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const long r = j / width, c = j - r * width;
        dst[(dst_first + r * dst_stride) * width + c] = src[(src_first + r * src_stride) * width + c];
    }
}
void launch_rows_strided_copy(float *dst, long dst_first, long dst_stride, const float *src, long src_first, long src_stride, long nrows, int width,
                              hipStream_t st) {
    if (nrows <= 0 || width <= 0) return;
    long grid = (nrows * width + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_rows_strided_copy, dim3((int)grid), dim3(256), 0, st, dst, dst_first, dst_stride, src, src_first, src_stride, nrows, width);
}

// rank pairs (user, pos, neg) -> columns of the few-row schedule: lower / higher item id with the negative's sign flipped
// (apex_svd_data.cpp:828-860), label and user value 1; *flag is raised when a pair has pos == neg
__global__ __launch_bounds__(256) void k_pairs_prepare(long n, const unsigned *pos, const unsigned *neg, unsigned *lo, unsigned *hi, float *vlo,
                                                       float *vhi, float *ones, unsigned *flag) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const unsigned p = pos[r], q = neg[r];
        if (p == q) atomicOr(flag, 1u);
        const bool pf = p < q;
        lo[r] = pf ? p : q; hi[r] = pf ? q : p;
        vlo[r] = pf ? 1.0f : -1.0f; vhi[r] = pf ? -1.0f : 1.0f;
        ones[r] = 1.0f;
    }
}
void launch_pairs_prepare(long n, const unsigned *pos, const unsigned *neg, unsigned *lo, unsigned *hi, float *vlo, float *vhi, float *ones,
                          unsigned *flag, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_pairs_prepare, dim3((int)grid), dim3(256), 0, st, n, pos, neg, lo, hi, vlo, vhi, ones, flag);
}

// ---- probe of the device expf (tests: compared with the host libm's expf bit for bit) ------------------
__global__ __launch_bounds__(256) void k_expf_probe(const float *in, float *out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) out[j] = glibc_expf(in[j]);
}
// out[j] = glibc_expf(first + j * step) over float BIT PATTERNS (sweeps of the whole input space without host arrays)
__global__ __launch_bounds__(256) void k_expf_sweep(unsigned first, unsigned step, float *out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
        out[j] = glibc_expf(__uint_as_float(first + (unsigned)j * step));
}
int device_expf(const float *in, unsigned first, unsigned step, float *out, long n) {
    if (n <= 0) return 0;
    float *din = nullptr, *dout = nullptr;
    if (hipMalloc((void **)&dout, (size_t)n * sizeof(float)) != hipSuccess) return -1;
    int rc = 0;
    long grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (in) {
        if (hipMalloc((void **)&din, (size_t)n * sizeof(float)) != hipSuccess) { (void)hipFree(dout); return -1; }
        if (hipMemcpy(din, in, (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = -1;
        hipLaunchKernelGGL(k_expf_probe, dim3((int)grid), dim3(256), 0, nullptr, din, dout, n);
    } else {
        hipLaunchKernelGGL(k_expf_sweep, dim3((int)grid), dim3(256), 0, nullptr, first, step, dout, n);
    }
    if (hipMemcpy(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    if (din) (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}

}  // namespace svdf
