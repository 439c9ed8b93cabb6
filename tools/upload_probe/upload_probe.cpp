// Host -> device hand-over of pageable columns: what the runtime's own pageable copy reaches against (b) the same copy split over threads / streams and
// (c) threads copying into pinned staging buffers + DMA from there.  build: hipcc -O2 -o upload_probe upload_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)1200 << 20;
    std::vector<char> src(bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;
    char *dst; CK(hipMalloc((void **)&dst, bytes));
    CK(hipMemcpy(dst, src.data(), 64 << 20, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpy(dst, src.data(), bytes, hipMemcpyHostToDevice));
        printf("pageable hipMemcpy, one call: %.1f GB/s\n", bytes / (now() - t0) / 1e9);
    }
    for (int T : {2, 4, 8}) {
        std::vector<hipStream_t> st(T);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                const size_t a = bytes * t / T, b = bytes * (t + 1) / T;
                (void)hipMemcpyAsync(dst + a, src.data() + a, b - a, hipMemcpyHostToDevice, st[t]);
                (void)hipStreamSynchronize(st[t]);
            });
            for (auto &x : th) x.join();
            printf("pageable, %d threads x streams: %.1f GB/s\n", T, bytes / (now() - t0) / 1e9);
        }
        // pinned staging: each thread 2 x CH buffers
        const size_t CH = 4 << 20;
        std::vector<char *> pin(2 * T);
        std::vector<hipEvent_t> ev(2 * T);
        for (int i = 0; i < 2 * T; i++) { CK(hipHostMalloc((void **)&pin[i], CH, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                const size_t a = bytes * t / T, b = bytes * (t + 1) / T;
                int c = 0;
                for (size_t off = a; off < b; off += CH, c++) {
                    const size_t len = std::min(CH, b - off);
                    const int q = 2 * t + (c & 1);
                    if (c >= 2) (void)hipEventSynchronize(ev[q]);
                    memcpy(pin[q], src.data() + off, len);
                    (void)hipMemcpyAsync(dst + off, pin[q], len, hipMemcpyHostToDevice, st[t]);
                    (void)hipEventRecord(ev[q], st[t]);
                }
                (void)hipStreamSynchronize(st[t]);
            });
            for (auto &x : th) x.join();
            printf("pinned staging, %d threads: %.1f GB/s\n", T, bytes / (now() - t0) / 1e9);
        }
        for (int i = 0; i < 2 * T; i++) { (void)hipHostFree(pin[i]); (void)hipEventDestroy(ev[i]); }
        for (auto &s : st) (void)hipStreamDestroy(s);
    }
    // verify the last copy
    std::vector<char> back(1 << 20);
    CK(hipMemcpy(back.data(), dst + bytes - back.size(), back.size(), hipMemcpyDeviceToHost));
    printf("tail equal: %d\n", memcmp(back.data(), src.data() + bytes - back.size(), back.size()) == 0);
    return 0;
}
