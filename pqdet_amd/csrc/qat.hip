// Quantisation-aware training (BASELINE config 5): the device side of what the reference gets from torch.ao's QAT modules
// (tools.py:256-290 fuse_model / prepare_qat('qnnpack'); model/parser.py:143-183 FloatFunctional add / cat;
// model/interpreter.py:27-53 QuantStub / DeQuantStub).  Algorithms restated in oracle/qat.py, which is pinned against the
// library op and the reference's own outputs (G9).
//
//   observer      moving-average min / max (constant 0.01, first batch initialises) -> ChooseQuantizationParams
//   fake-quant    (clamp(nearbyint(x / scale) + zp, qmin, qmax) - zp) * scale, straight-through gradient inside the range
//   activations   quint8 [0, 255] affine;  weights: qint8 [-128, 127] symmetric, per tensor
//
// Kernels: weight preparation for ALL conv layers in one launch (scale by gamma/sqrt(running_var+eps), observe,
// quantise), activation quantisation of one tensor (min/max partials -> observer/qparams -> apply), its backward
// (straight-through mask x activation derivative, with the BatchNorm-backward sums the ordinary gradient sinks produce),
// and the weight-gradient post-processing for all layers in one launch.  Compiled with -ffp-contract=off: rounding
// boundaries are part of the contract.
#include "common.h"
#include "qatmath.h"

namespace pq {

__device__ __forceinline__ float fq_value(float x, float scale, float inv, float zp, float qmin, float qmax, bool& inside) {
    const float q = rintf(x * inv) + zp;
    inside = q >= qmin && q <= qmax;
    return (fminf(fmaxf(q, qmin), qmax) - zp) * scale;
}

// block reduction of (min, max) over up to 1024 threads
__device__ __forceinline__ void block_minmax(float& lo, float& hi, float* sh /* 2*32 */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = lo; sh[32 + w] = hi; }
    __syncthreads();
    lo = sh[0]; hi = sh[32];
    for (int k = 1; k < nw; ++k) { lo = fminf(lo, sh[k]); hi = fmaxf(hi, sh[32 + k]); }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ weights
// one block per conv layer: ws = w * sf[o] (sf = gamma / sqrt(running_var + eps), 1 without BatchNorm); observe; quantise.
__global__ __launch_bounds__(1024) void qat_wprep_kernel(const pq_qat_weight_item* items, int observe, int quantise, float avg_const) {
    __shared__ float sh[64];
    const pq_qat_weight_item it = items[blockIdx.x];
    const long n = (long)it.O * it.K;
    // per-channel factor
    for (int o = threadIdx.x; o < it.O; o += blockDim.x)
        it.sf[o] = it.gamma ? it.gamma[o] / sqrtf(it.running_var[o] + it.eps) : 1.f;
    __syncthreads();
    if (observe) {
        float lo = INFINITY, hi = -INFINITY;
        for (long e = threadIdx.x; e < n; e += blockDim.x) {
            const float v = it.w[e] * it.sf[e / it.K];
            lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
        block_minmax(lo, hi, sh);
        if (threadIdx.x == 0) qat_observe(it.state, lo, hi, -128, 127, true, avg_const);
        __syncthreads();
        __threadfence_block();
    }
    const float scale = it.state[2], zp = it.state[3], inv = 1.f / scale;
    for (long e = threadIdx.x; e < n; e += blockDim.x) {
        const float v = it.w[e] * it.sf[e / it.K];
        bool inside;
        const float q = fq_value(v, scale, inv, zp, -128.f, 127.f, inside);
        it.wq[e] = quantise ? q : v;
    }
}

// dW = dWq * mask * sf;  dgamma += sum_k dWq * (mask * W - Wq / sf) / sqrt(running_var + eps)     (one block per layer)
__global__ __launch_bounds__(1024) void qat_wfix_kernel(const pq_qat_weight_item* items, int quantise) {
    __shared__ float red[1024];
    const pq_qat_weight_item it = items[blockIdx.x];
    const float scale = it.state[2], zp = it.state[3], inv = 1.f / scale;
    // threads walk channels in groups: a wave per output channel keeps the per-channel reduction local
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int o = wv; o < it.O; o += nw) {
        const float sf = it.sf[o];
        float acc = 0.f;
        for (int k = lane; k < it.K; k += 64) {
            const long e = (long)o * it.K + k;
            const float w = it.w[e], g = it.dwq[e];
            bool inside = true;
            float wq = w * sf;
            if (quantise) wq = fq_value(w * sf, scale, inv, zp, -128.f, 127.f, inside);
            const float m = inside ? 1.f : 0.f;
            it.dw[e] = g * m * sf;
            acc += g * (m * w - wq / sf);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0 && it.gamma) it.dgamma[o] += acc / sqrtf(it.running_var[o] + it.eps);
    }
    (void)red;
}

// ------------------------------------------------------------------------------------------------ activations
// v[r][c] = act(scale[c]*x[r*ld+c] + shift[c])  (pq_src);  flat != 0: x is a plain array of `rows` floats (the NCHW image)
__global__ __launch_bounds__(256) void qact_minmax_kernel(pq_qact_args a, float* partial) {
    __shared__ float sh[64];
    float lo = INFINITY, hi = -INFINITY;
    const ActParam act = act_param(a.in.act);
    if (a.flat) {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < a.rows; e += (long)gridDim.x * 256) {
            const float v = a.in.x[e];
            lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
    } else {
        const long n = (long)a.rows * a.C;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
            const long r = e / a.C;
            const int c = (int)(e - r * a.C);
            float u = a.in.x[r * a.in.ld + c];
            if (a.in.scale) u = fmaf(a.in.scale[c], u, a.in.shift[c]);
            const float v = act_fwd(u, act);
            lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
    }
    block_minmax(lo, hi, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = lo; partial[2 * blockIdx.x + 1] = hi; }
}

// float4 variants (C % 4 == 0, 16-byte aligned pitches): a thread owns 4 consecutive channels, no per-element division
__global__ __launch_bounds__(256) void qact_minmax4_kernel(pq_qact_args a, float* partial) {
    __shared__ float sh[64];
    float lo = INFINITY, hi = -INFINITY;
    const ActParam act = act_param(a.in.act);
    const int C4 = a.C >> 2;
    const long n4 = (long)a.rows * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const long r = e / C4;
        const int c = (int)(e - r * C4) * 4;
        const float4 x = ldg4(a.in.x + r * a.in.ld + c);
        float u[4] = {x.x, x.y, x.z, x.w};
        if (a.in.scale) {
            const float4 sc = ldg4(a.in.scale + c), sf = ldg4(a.in.shift + c);
            u[0] = fmaf(sc.x, u[0], sf.x); u[1] = fmaf(sc.y, u[1], sf.y); u[2] = fmaf(sc.z, u[2], sf.z); u[3] = fmaf(sc.w, u[3], sf.w);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float v = act_fwd(u[j], act); lo = fminf(lo, v); hi = fmaxf(hi, v); }
    }
    block_minmax(lo, hi, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = lo; partial[2 * blockIdx.x + 1] = hi; }
}

__global__ __launch_bounds__(256) void qact_apply4_kernel(pq_qact_args a) {
    const float scale = a.state[2], zp = a.state[3], inv = 1.f / scale;
    const ActParam act = act_param(a.in.act);
    const int C4 = a.C >> 2;
    const long n4 = (long)a.rows * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const long r = e / C4;
        const int c = (int)(e - r * C4) * 4;
        const float4 x = ldg4(a.in.x + r * a.in.ld + c);
        float u[4] = {x.x, x.y, x.z, x.w};
        if (a.in.scale) {
            const float4 sc = ldg4(a.in.scale + c), sf = ldg4(a.in.shift + c);
            u[0] = fmaf(sc.x, u[0], sf.x); u[1] = fmaf(sc.y, u[1], sf.y); u[2] = fmaf(sc.z, u[2], sf.z); u[3] = fmaf(sc.w, u[3], sf.w);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = act_fwd(u[j], act);
            bool inside;
            u[j] = a.quantise ? fq_value(v, scale, inv, zp, 0.f, 255.f, inside) : v;
        }
        stg4(a.out + r * a.ldo + c, make_float4(u[0], u[1], u[2], u[3]));
    }
}

__global__ __launch_bounds__(256) void qact_observe_kernel(const float* partial, int n, float* state, float avg_const) {
    __shared__ float sh[64];
    float lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < n; k += 256) { lo = fminf(lo, partial[2 * k]); hi = fmaxf(hi, partial[2 * k + 1]); }
    block_minmax(lo, hi, sh);
    if (threadIdx.x == 0) qat_observe(state, lo, hi, 0, 255, false, avg_const);
}

__global__ __launch_bounds__(256) void qact_apply_kernel(pq_qact_args a) {
    const float scale = a.state[2], zp = a.state[3], inv = 1.f / scale;
    const ActParam act = act_param(a.in.act);
    if (a.flat) {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < a.rows; e += (long)gridDim.x * 256) {
            bool inside;
            const float v = a.in.x[e];
            a.out[e] = a.quantise ? fq_value(v, scale, inv, zp, 0.f, 255.f, inside) : v;
        }
        return;
    }
    const long n = (long)a.rows * a.C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long r = e / a.C;
        const int c = (int)(e - r * a.C);
        float u = a.in.x[r * a.in.ld + c];
        if (a.in.scale) u = fmaf(a.in.scale[c], u, a.in.shift[c]);
        const float v = act_fwd(u, act);
        bool inside;
        a.out[r * a.ldo + c] = a.quantise ? fq_value(v, scale, inv, zp, 0.f, 255.f, inside) : v;
    }
}

// backward of fq(act(scale*y + shift)): g = gq * [inside] * act'(u), plus the BatchNorm-backward sums of an ordinary sink
__global__ __launch_bounds__(256) void qact_bwd_kernel(pq_qact_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + threadIdx.x, c0 = c4 * 4;
    const bool cok = c0 < a.C;
    const float scale = a.state[2], zp = a.state[3], inv = 1.f / scale;
    const ActParam act = act_param(a.dx.act);
    const bool affine = a.dx.scale != nullptr, stats = a.dx.mean != nullptr && a.dx.partial != nullptr;
    float ca[4], cb[4], cm[4], ci[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = c0 + j < a.C;
        ca[j] = (affine && ok) ? a.dx.scale[c0 + j] : 1.f;
        cb[j] = (affine && ok) ? a.dx.shift[c0 + j] : 0.f;
        cm[j] = (stats && ok) ? a.dx.mean[c0 + j] : 0.f;
        ci[j] = (stats && ok) ? a.dx.invstd[c0 + j] : 0.f;
    }
    float4 s1 = f4zero(), s2 = f4zero();
    for (int r = bx * blockDim.y + threadIdx.y; r < a.rows && cok; r += gridDim.x * blockDim.y) {
        const float4 g4 = ldg4(a.gq + (size_t)r * a.ldgq + c0);
        const float4 y4 = ldg4(a.dx.y + (size_t)r * a.dx.ldy + c0);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w}, y[4] = {y4.x, y4.y, y4.z, y4.w};
        float o[4], t1[4] = {s1.x, s1.y, s1.z, s1.w}, t2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = fmaf(ca[j], y[j], cb[j]);
            const float v = act_fwd(u, act);
            bool inside = true;
            if (a.quantise) (void)fq_value(v, scale, inv, zp, 0.f, 255.f, inside);
            o[j] = g[j] * (inside ? 1.f : 0.f) * act_bwd(u, act);
            t1[j] += o[j];
            t2[j] = fmaf(o[j], (y[j] - cm[j]) * ci[j], t2[j]);
        }
        stg4(a.dx.g + (size_t)r * a.dx.ldg + c0, make_float4(o[0], o[1], o[2], o[3]));
        s1 = make_float4(t1[0], t1[1], t1[2], t1[3]);
        s2 = make_float4(t2[0], t2[1], t2[2], t2[3]);
    }
    if (stats) block_reduce_store_stats<256>(s1, s2, a.dx.partial, a.dx.ldp, c0, a.C, lds, bx);
}

}  // namespace pq

using namespace pq;

extern "C" int pq_qat_weight_prep(const pq_qat_weight_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->items && a->n_items > 0, "bad args");
    PQ_LAUNCH(qat_wprep_kernel, dim3(a->n_items), dim3(1024), 0, (hipStream_t)stream, a->items, a->observe, a->quantise, a->avg_const);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_qat_weight_grad(const pq_qat_weight_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->items && a->n_items > 0, "bad args");
    PQ_LAUNCH(qat_wfix_kernel, dim3(a->n_items), dim3(1024), 0, (hipStream_t)stream, a->items, a->quantise);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int32_t pq_qact_blocks(int64_t elems) {
    int64_t nb = (elems + 256 * 8 - 1) / (256 * 8);
    return (int32_t)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

extern "C" int pq_qact_fwd(const pq_qact_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->state && a->partial, "null pointer");
    PQ_CHECK_ARG(a->out || a->observe, "nothing to do: no output and no observer update");
    PQ_CHECK_ARG(a->rows > 0 && (a->flat || a->C > 0), "empty");
    if (act_is_special(a->in.act)) { pq_set_error("pq_qact_fwd: logistic/tanh are not part of the QAT model (every activation is ReLU)"); return PQ_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t elems = a->flat ? a->rows : a->rows * (int64_t)a->C;
    const int nb = pq_qact_blocks(elems);
    const bool vec = !a->flat && (a->C & 3) == 0 && (a->in.ld & 3) == 0 && (!a->out || (a->ldo & 3) == 0) &&
                     (((uintptr_t)a->in.x | (uintptr_t)a->out) & 15) == 0;
    if (a->observe) {
        if (vec) PQ_LAUNCH(qact_minmax4_kernel, dim3(nb), dim3(256), 0, st, *a, a->partial);
        else PQ_LAUNCH(qact_minmax_kernel, dim3(nb), dim3(256), 0, st, *a, a->partial);
        PQ_CHECK_LAUNCH();
        PQ_LAUNCH(qact_observe_kernel, dim3(1), dim3(256), 0, st, (const float*)a->partial, nb, a->state, a->avg_const);
        PQ_CHECK_LAUNCH();
    }
    if (!a->out) return PQ_OK;                 // observer only: the consumers fake-quantise while loading (pq_src.qstate)
    if (vec) PQ_LAUNCH(qact_apply4_kernel, dim3(nb), dim3(256), 0, st, *a);
    else PQ_LAUNCH(qact_apply_kernel, dim3(nb), dim3(256), 0, st, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_qact_bwd(const pq_qact_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->gq && a->dx.g && a->dx.y && a->state, "null pointer");
    PQ_SINK_ACC_UNSUPPORTED(a->dx);
    PQ_CHECK_ARG(a->dx.addend == nullptr, "dx.addend is not supported by this entry");
    PQ_CHECK_ARG(a->rows > 0 && a->C > 0 && (a->C & 3) == 0 && (a->ldgq & 3) == 0 && (a->dx.ldg & 3) == 0 && (a->dx.ldy & 3) == 0,
                 "channels / pitches must be multiples of 4");
    if (act_is_special(a->dx.act)) { pq_set_error("pq_qact_bwd: logistic/tanh are not part of the QAT model"); return PQ_EUNSUPPORTED; }
    EwGeom g = ew_geom((long)a->rows, a->C);
    PQ_LAUNCH(qact_bwd_kernel, dim3(g.gx, g.gy), dim3(g.cx, g.py), (size_t)g.cx * g.py * 8 * sizeof(float), (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
