// Probe: what limits the fp32 MFMA GEMM inner loop on gfx950?  Variants share one kernel skeleton:
//   V0 pure MFMA (operands in registers)         V1 + ds_read_b128 fragment reads from LDS (no barriers)
//   V2 + barrier per k-tile (no LDS writes)      V3 + LDS writes of a fake tile per k-tile (full on-chip loop, no global)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDT = 36;
template <int V, int TI, int TJ>
__global__ __launch_bounds__(256) void probe(float* out, int nk) {
    __shared__ __attribute__((aligned(16))) float As[2 * 128 * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2 * 128 * LDT];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    for (int i = t; i < 2 * 128 * LDT; i += 256) { As[i] = 0.001f * (i & 7); Bs[i] = 0.002f * (i & 3); }
    __syncthreads();
    f32x16 acc[TI][TJ];
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = lane & 31, lh = lane >> 5;
    float4 af[TI], bf[TJ];
    for (int i = 0; i < TI; ++i) af[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int j = 0; j < TJ; ++j) bf[j] = make_float4(.1f, .2f, .3f, .4f);
    float4 stage[4];
    for (int i = 0; i < 4; ++i) stage[i] = make_float4(t, i, 1, 2);
    for (int kt = 0; kt < nk; ++kt) {
        const float* A = As + (kt & 1) * 128 * LDT;
        const float* B = Bs + (kt & 1) * 128 * LDT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (V >= 1) {
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float4*>(&A[(wm * 32 * TI + 32 * i + lrow) * LDT + ks * 8 + lh * 4]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const float4*>(&B[(wn * 32 * TJ + 32 * j + lrow) * LDT + ks * 8 + lh * 4]);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (V >= 3) {
            float* Aw = As + ((kt + 1) & 1) * 128 * LDT;
            float* Bw = Bs + ((kt + 1) & 1) * 128 * LDT;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(&Aw[((t >> 3) + 32 * i) * LDT + 4 * (t & 7)]) = stage[i];
                *reinterpret_cast<float4*>(&Bw[((t >> 3) + 32 * i) * LDT + 4 * (t & 7)]) = stage[i];
            }
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + t] = s;
}
template <int V, int TI, int TJ>
void run(const char* name, int blocks, int nk, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((probe<V, TI, TJ>), dim3(blocks), dim3(256), 0, 0, out, nk);
    hipEventRecord(e0);
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((probe<V, TI, TJ>), dim3(blocks), dim3(256), 0, 0, out, nk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double fl = (double)blocks * 4 * TI * TJ * 16.0 * nk * 4096.0;
    printf("%-34s blocks %5d nk %4d : %.3f ms  %.1f TF/s\n", name, blocks, nk, ms, fl / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0, 2, 2>("V0 pure mfma 2x2", blocks, 64, out);
        run<1, 2, 2>("V1 +ds_read 2x2", blocks, 64, out);
        run<2, 2, 2>("V2 +barrier 2x2", blocks, 64, out);
        run<3, 2, 2>("V3 +ds_write 2x2", blocks, 64, out);
        run<3, 1, 1>("V3 +ds_write 1x1 (64x64 tile)", blocks, 64, out);
        run<3, 2, 1>("V3 +ds_write 2x1 (128x64 tile)", blocks, 64, out);
    }
    run<3, 2, 2>("V3 2x2 short K (nk=8)", 512, 8, out);
    run<3, 2, 2>("V3 2x2 short K (nk=2)", 512, 2, out);
    run<3, 2, 2>("V3 2x2 short K (nk=8) 2048 blk", 2048, 8, out);
    return 0;
}
