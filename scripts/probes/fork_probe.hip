// Probe: what does handing a dependency from the main stream to a side stream cost the MAIN stream?
// Chain per iteration: main: K1 (tiny, like bn_bwd_finalize) -> K2 (~20 us); side: S (~20 us) must start after K1.
//   A  hipEventRecord(main) after K1 + hipStreamWaitEvent(side)
//   B  the event bound to K1's dispatch (hipExtLaunchKernelGGL stop event) + hipStreamWaitEvent(side)
//   C  K1's last workgroup bumps a counter in device memory; side: hipStreamWaitValue32(>= iteration)
//   0  no side work at all (floor of the main chain)
// Build: hipcc -O3 --offload-arch=gfx950 fork_probe.hip -o fork_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void busy(float* p, int iters) {           // ~iters * 8 ns per workgroup round
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.f) p[threadIdx.x] = v;
}
__global__ void tiny(float* p, unsigned* flag, unsigned value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { p[0] += 1.f; if (flag) { __threadfence(); atomicExch(flag, value); } }
}

int main() {
    float *a, *b, *c;
    unsigned* flag;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&c, 1 << 20));
    CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t m, s;
    CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s));
    const int N = 200, IT = 2600;
    hipEvent_t ev[256];
    for (int i = 0; i < 256; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventDisableSystemFence));
    hipEvent_t t0, t1, jn;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&jn, hipEventDisableTiming));
    unsigned epoch = 0;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 3 && !can) { printf("C: not supported\n"); continue; }
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, m));
            for (int i = 0; i < N; ++i) {
                ++epoch;
                if (mode == 2) hipExtLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, m, nullptr, ev[i & 255], 0, a, (unsigned*)nullptr, 0u);
                else hipLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, m, a, mode == 3 ? flag : nullptr, epoch);
                if (mode == 1) CK(hipEventRecord(ev[i & 255], m));
                if (mode == 1 || mode == 2) CK(hipStreamWaitEvent(s, ev[i & 255], 0));
                if (mode == 3) CK(hipStreamWaitValue32(s, flag, epoch, hipStreamWaitValueGte, 0xffffffffu));
                if (mode != 0) hipLaunchKernelGGL(busy, dim3(128), dim3(256), 0, s, c, IT);
                hipLaunchKernelGGL(busy, dim3(128), dim3(256), 0, m, b, IT);
            }
            if (mode != 0) { CK(hipEventRecord(jn, s)); CK(hipStreamWaitEvent(m, jn, 0)); }
            CK(hipEventRecord(t1, m));
            CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (ms < best) best = ms;
        }
        const char* name[] = {"0 main chain alone", "A event record after K1", "B event bound to K1", "C counter + hipStreamWaitValue32"};
        printf("%-36s %.2f us per iteration\n", name[mode], best * 1000.f / N);
    }
    return 0;
}
