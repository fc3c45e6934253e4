// Batched partial-sum reduction of the weight-gradient entries (pq_reduce_batch, include/pqdet_hip.h).
// Every weight-gradient kernel splits its contraction over workgroups and leaves one slab per split; the sum over the slabs of
// MANY layers is ONE launch here (table in device memory, like pq_transpose_batch) instead of one or two launches per layer on the
// weight-gradient stream (83 per mobilenetv2-fpn step: each took the dispatcher and a few CUs from the main chain for 4-8 us).
// The order of the additions is fixed by (lanes, splits) alone: replays of a step are bit-identical.
// Work mirrored: the accumulation torch.autograd does for nn.Conv2d weights behind the reference's trainer.py:231-233.
#include "common.h"

namespace pq {

// block = 256 threads = (256 / L) column units x L split lanes, L = 4 (a wave reads 256 / 1024 contiguous bytes of one slab row) or
// 16 (long lists: the per-workgroup rows of the walking depthwise backward and of the fused narrow 1x1 backward).  A column unit
// is a float4 where the item's geometry allows it (pad_ = 4: every pitch and count a multiple of four, 16-byte aligned slabs --
// all 1x1 / depthwise weight gradients of the nets), else one float.
template <typename V>
__device__ __forceinline__ V red_zero();
template <> __device__ __forceinline__ float red_zero<float>() { return 0.f; }
template <> __device__ __forceinline__ float4 red_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void red_add(float& a, const float b) { a += b; }
__device__ __forceinline__ void red_add(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <typename V>
__device__ __forceinline__ V red_column(const float* src, size_t stride, int splits, int lane, int L) {
    V s = red_zero<V>();
    int k = lane;
    for (; k + 7 * L < splits; k += 8 * L) {
        V v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const V*>(src + (size_t)(k + u * L) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) red_add(s, v[u]);
    }
    for (; k < splits; k += L) red_add(s, *reinterpret_cast<const V*>(src + (size_t)k * stride));
    return s;
}

__global__ __launch_bounds__(256) void reduce_batch_kernel(const pq_reduce_item* __restrict__ items, int n_items) {
    __shared__ float4 red[256];
    const int b = blockIdx.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {                      // last item with block0 <= b (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const pq_reduce_item it = items[lo];
    const int L = it.lanes, cols = 256 / L;
    const int t = threadIdx.x, c = t & (cols - 1), lane = t / cols;
    const int n = it.n0 + it.n1;
    if (it.pad_ == 4) {                    // float4 columns (one group, no bias column, no permutation)
        const int e = ((b - it.block0) * cols + c) * 4;
        float4 s = red_zero<float4>();
        if (e < n) s = red_column<float4>(it.slab + e, (size_t)it.stride, it.splits, lane, L);
        red[t] = s;
        __syncthreads();
        if (lane == 0 && e < n) {
            float4 tot = red[c];
            for (int q = 1; q < L; ++q) red_add(tot, red[q * cols + c]);
            *reinterpret_cast<float4*>(it.out0 + e) = tot;
        }
        return;
    }
    const int e = (b - it.block0) * cols + c;
    float s = 0.f;
    if (e < n) {
        const int g = e / it.ng, r = e - g * it.ng;
        s = red_column<float>(it.slab + (size_t)g * it.gstride + r, (size_t)it.stride, it.splits, lane, L);
    }
    float* redf = reinterpret_cast<float*>(red);
    redf[t] = s;
    __syncthreads();
    if (lane == 0 && e < n) {
        float tot = redf[c];
        for (int q = 1; q < L; ++q) tot += redf[q * cols + c];
        if (e >= it.n0) { it.out1[e - it.n0] = tot; return; }
        if (it.perm_gw > 0) {
            const int gw = it.perm_gw, K = 9 * gw;
            const int g = e / it.ng, r = e - g * it.ng, o = r / K, k = r - o * K, tap = k / gw, i = k - tap * gw;
            it.out0[((size_t)(g * gw + o) * gw + i) * 9 + tap] = tot;
        } else {
            it.out0[e] = tot;
        }
    }
}

}  // namespace pq

using namespace pq;

extern "C" int32_t pq_reduce_batch_plan(pq_reduce_item* items, int32_t n_items) {
    if (!items || n_items <= 0) return 0;
    long blocks = 0;
    for (int i = 0; i < n_items; ++i) {
        pq_reduce_item& it = items[i];
        if (!it.slab || !it.out0 || it.splits < 1 || it.n0 < 1 || it.n1 < 0 || (it.n1 > 0 && !it.out1) || it.stride < 1) {
            pq_set_error("pq_reduce_batch_plan: item %d is incomplete", i);
            return -1;
        }
        it.lanes = it.splits > 64 ? 16 : 4;
        if (it.ng <= 0) it.ng = it.n0 + it.n1;
        const bool v4 = it.n1 == 0 && it.perm_gw <= 0 && it.ng == it.n0 && (it.n0 & 3) == 0 && (it.stride & 3) == 0 &&
                        (((uintptr_t)it.slab | (uintptr_t)it.out0) & 15) == 0;
        it.pad_ = v4 ? 4 : 1;              // floats per column unit
        it.block0 = (int32_t)blocks;
        const int cols = (256 / it.lanes) * it.pad_;
        blocks += (it.n0 + it.n1 + cols - 1) / cols;
        if (blocks > 0x7fffffffL) return -1;
    }
    return (int32_t)blocks;
}

extern "C" int pq_reduce_batch(const pq_reduce_batch_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->items && a->n_items > 0 && a->n_blocks > 0, "empty table (plan it with pq_reduce_batch_plan)");
    PQ_LAUNCH(reduce_batch_kernel, dim3(a->n_blocks), dim3(256), 0, (hipStream_t)stream, a->items, a->n_items);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
