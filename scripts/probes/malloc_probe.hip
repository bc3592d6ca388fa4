// malloc_probe.hip -- how long hipMalloc / hipFree / a first hipMemset of large buffers take on this box, fresh and
// right after a free of the same size (the bench's yh_db_create showed 0.7-1.6 s inside its allocation phase).
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/malloc_probe.hip -o scripts/probes/malloc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <vector>
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
int main() {
    hipSetDevice(0);
    hipFree(0);
    const size_t sizes[] = {(size_t)1 << 30, (size_t)4 << 30, (size_t)8 << 30, (size_t)12 << 30};
    for (int round = 0; round < 4; ++round) {
        for (size_t sz : sizes) {
            void* p = nullptr;
            double t0 = now_ms();
            hipError_t e = hipMalloc(&p, sz);
            double t1 = now_ms();
            hipMemset(p, 0, sz);
            hipDeviceSynchronize();
            double t2 = now_ms();
            hipFree(p);
            double t3 = now_ms();
            printf("round %d  %5.1f GB: hipMalloc %9.3f ms  memset %8.3f ms  hipFree %8.3f ms  (%s)\n", round, sz / 1073741824.0, t1 - t0, t2 - t1, t3 - t2,
                   hipGetErrorString(e));
        }
    }
    // many live allocations, then a big one (a fragmented heap)
    std::vector<void*> keep;
    for (int i = 0; i < 64; ++i) { void* p = nullptr; hipMalloc(&p, (size_t)512 << 20); keep.push_back(p); }
    for (size_t i = 0; i < keep.size(); i += 2) { hipFree(keep[i]); keep[i] = nullptr; }
    for (int k = 0; k < 3; ++k) {
        void* p = nullptr;
        double t0 = now_ms();
        hipMalloc(&p, (size_t)8 << 30);
        double t1 = now_ms();
        hipFree(p);
        printf("after 32 x 512 MB holes: hipMalloc(8 GB) %9.3f ms, hipFree %8.3f ms\n", t1 - t0, now_ms() - t1);
    }
    for (void* p : keep) if (p) hipFree(p);
    return 0;
}
