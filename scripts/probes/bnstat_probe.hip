// Probe for the BatchNorm statistics hand-over: what does it cost when
//   P) G producer workgroups ADD their per-channel partial sums into `slots` replicated accumulators (fp64 atomics, or two
//      int64 fixed-point limbs = exact, order independent), all at the same moment (the end of a single-round grid), against
//      writing G partial rows with plain stores;
//   Q) every one of W consumer workgroups derives scale / shift for K channels from those accumulators in its prologue (loads
//      from L2, rsqrt, LDS), against reading two finished fp32 vectors;
//   F) the dependent-launch chain: producer -> tiny finalize -> consumer against producer -> consumer.
// Build: hipcc -O3 --offload-arch=gfx950 bnstat_probe.hip -o bnstat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---- producers: one value per (channel, statistic) and workgroup
template <int MODE>   // 0 plain rows, 1 fp64 atomics, 2 two int64 limbs
__global__ __launch_bounds__(256) void producer(float* rows, double* acc, long long* limbs, int C, int slots, float seed) {
    const int b = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < 2 * C; c += 256) {
        const float v = seed * (float)((b * 131 + c * 7) % 1000 - 500) * 1e-3f;
        if (MODE == 0) rows[(size_t)b * 2 * C + c] = v;
        if (MODE == 1) unsafeAtomicAdd(&acc[(size_t)(b % slots) * 2 * C + c], (double)v);
        if (MODE == 2) {
            // fixed point, unit 2^-50, two limbs of 50 bits
            const double s = (double)v * 1125899906842624.0;          // 2^50
            const double hi = __builtin_trunc(s * (1.0 / 1125899906842624.0));
            const long long h = (long long)hi, l = (long long)(s - hi * 1125899906842624.0);
            long long* d = limbs + ((size_t)(b % slots) * 2 * C + c) * 2;
            atomicAdd((unsigned long long*)d, (unsigned long long)l);
            if (h) atomicAdd((unsigned long long*)(d + 1), (unsigned long long)h);
        }
    }
}

// ---- consumers: derive K coefficient pairs into LDS, then touch them
template <int MODE>   // 0 finished vectors, 1 fp64 accumulators, 2 limbs
__global__ __launch_bounds__(256) void consumer(const float* scale, const float* shift, const double* acc, const long long* limbs,
                                                const float* gamma, const float* beta, int K, int slots, float inv_n, float* out) {
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    for (int k = t; k < K; k += 256) {
        float sc, sh;
        if (MODE == 0) { sc = scale[k]; sh = shift[k]; }
        else {
            double s1 = 0, s2 = 0;
            if (MODE == 1) for (int s = 0; s < slots; ++s) { s1 += acc[(size_t)s * 2 * K + k]; s2 += acc[(size_t)s * 2 * K + K + k]; }
            else for (int s = 0; s < slots; ++s) {
                const long long* a = limbs + ((size_t)s * 2 * K + k) * 2;
                const long long* b = limbs + ((size_t)s * 2 * K + K + k) * 2;
                s1 += (double)a[1] + (double)a[0] * (1.0 / 1125899906842624.0);
                s2 += (double)b[1] + (double)b[0] * (1.0 / 1125899906842624.0);
            }
            const double mean = s1 * inv_n, var = s2 * inv_n - mean * mean;
            const float inv = rsqrtf((float)var + 1e-5f);
            sc = gamma[k] * inv; sh = beta[k] - (float)mean * sc;
        }
        lds[k] = sc; lds[K + k] = sh;
    }
    __syncthreads();
    float a = 0.f;
    for (int k = t; k < K; k += 256) a += lds[k] * 0.5f + lds[K + k];
    if (a == 12345.678f) out[blockIdx.x] = a;     // keep the work
}

__global__ void finalize_like(const float* rows, int G, int C, float* scale, float* shift) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r0 = threadIdx.x >> 6;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int r = r0; r < G; r += 4) { s1 += rows[(size_t)r * 2 * C + c]; s2 += rows[(size_t)r * 2 * C + C + c]; }
    if (r0 == 0) { scale[c] = (float)s1; shift[c] = (float)s2; }
}

template <class F>
static float time_us(F f, int iters = 50) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main() {
    const int maxG = 8192, maxC = 1280, maxS = 16;
    float *rows, *scale, *shift, *gamma, *beta, *out; double* acc; long long* limbs;
    CK(hipMalloc(&rows, (size_t)maxG * 2 * maxC * 4));
    CK(hipMalloc(&acc, (size_t)maxS * 2 * maxC * 8)); CK(hipMalloc(&limbs, (size_t)maxS * 2 * maxC * 16));
    CK(hipMalloc(&scale, maxC * 4)); CK(hipMalloc(&shift, maxC * 4)); CK(hipMalloc(&gamma, maxC * 4)); CK(hipMalloc(&beta, maxC * 4));
    CK(hipMalloc(&out, maxG * 4));
    CK(hipMemset(acc, 0, (size_t)maxS * 2 * maxC * 8)); CK(hipMemset(limbs, 0, (size_t)maxS * 2 * maxC * 16));
    CK(hipMemset(scale, 0, maxC * 4)); CK(hipMemset(shift, 0, maxC * 4)); CK(hipMemset(gamma, 0, maxC * 4)); CK(hipMemset(beta, 0, maxC * 4));
    hipStream_t st = 0;
    printf("== producers: us per launch (G workgroups, C channels x 2 statistics)\n");
    const int Gs[] = {64, 256, 512, 2048, 8192}, Cs[] = {96, 256, 1024};
    for (int G : Gs) for (int C : Cs) {
        if ((long)G * C > 8192L * 256) continue;
        const float rows_us = time_us([&] { hipLaunchKernelGGL(producer<0>, dim3(G), dim3(256), 0, st, rows, acc, limbs, C, 1, 1.f); });
        printf("G=%5d C=%5d rows %6.2f |", G, C, rows_us);
        for (int s : {1, 4, 16}) {
            const float a = time_us([&] { hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, st, rows, acc, limbs, C, s, 1.f); });
            const float l = time_us([&] { hipLaunchKernelGGL(producer<2>, dim3(G), dim3(256), 0, st, rows, acc, limbs, C, s, 1.f); });
            printf(" slots=%2d f64 %6.2f limbs %6.2f |", s, a, l);
        }
        printf("\n");
    }
    printf("== consumers: us per launch (W workgroups derive K coefficient pairs)\n");
    const int Ws[] = {256, 512, 2048, 8192}, Ks[] = {96, 256, 1024};
    for (int W : Ws) for (int K : Ks) {
        const size_t sm = 2 * K * 4;
        const float v = time_us([&] { hipLaunchKernelGGL(consumer<0>, dim3(W), dim3(256), sm, st, scale, shift, acc, limbs, gamma, beta, K, 1, 1e-3f, out); });
        printf("W=%5d K=%5d vectors %6.2f |", W, K, v);
        for (int s : {1, 4, 16}) {
            const float a = time_us([&] { hipLaunchKernelGGL(consumer<1>, dim3(W), dim3(256), sm, st, scale, shift, acc, limbs, gamma, beta, K, s, 1e-3f, out); });
            const float l = time_us([&] { hipLaunchKernelGGL(consumer<2>, dim3(W), dim3(256), sm, st, scale, shift, acc, limbs, gamma, beta, K, s, 1e-3f, out); });
            printf(" slots=%2d f64 %6.2f limbs %6.2f |", s, a, l);
        }
        printf("\n");
    }
    printf("== chains: us per (producer [, finalize], consumer) triple\n");
    for (int G : {64, 512, 8192}) for (int C : {96, 1024}) {
        if ((long)G * C > 8192L * 256) continue;
        const size_t sm = 2 * C * 4;
        const float three = time_us([&] {
            hipLaunchKernelGGL(producer<0>, dim3(G), dim3(256), 0, st, rows, acc, limbs, C, 1, 1.f);
            hipLaunchKernelGGL(finalize_like, dim3((C + 63) / 64), dim3(256), 0, st, rows, G, C, scale, shift);
            hipLaunchKernelGGL(consumer<0>, dim3(G), dim3(256), sm, st, scale, shift, acc, limbs, gamma, beta, C, 1, 1e-3f, out); });
        const int s = G >= 2048 ? 16 : (G >= 512 ? 4 : 1);
        const float two = time_us([&] {
            hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, st, rows, acc, limbs, C, s, 1.f);
            hipLaunchKernelGGL(consumer<1>, dim3(G), dim3(256), sm, st, scale, shift, acc, limbs, gamma, beta, C, s, 1e-3f, out); });
        const float twol = time_us([&] {
            hipLaunchKernelGGL(producer<2>, dim3(G), dim3(256), 0, st, rows, acc, limbs, C, s, 1.f);
            hipLaunchKernelGGL(consumer<2>, dim3(G), dim3(256), sm, st, scale, shift, acc, limbs, gamma, beta, C, s, 1e-3f, out); });
        printf("G=%5d C=%5d rows+finalize+vectors %6.2f | f64 atomics(slots %2d)+prologue %6.2f | limbs %6.2f\n", G, C, three, s, two, twol);
    }
    return 0;
}
