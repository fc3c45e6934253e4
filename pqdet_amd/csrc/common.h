// Shared device helpers for the gfx950 kernels of libpqdet_hip.so.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/pqdet_hip.h"

#define PQ_WAVE 64

namespace pq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- activations (reference model/parser.py:106-112) -------------------------------------------------------
// linear / relu / relu6 / leaky(0.1) are one branch-free family  act(u) = min(max(u,0) + slope*min(u,0), hi);
// logistic / tanh (never used between layers by the shipped cfgs) take a uniform slow branch.
struct ActParam { float slope, hi; int special; int code; };
__host__ __device__ __forceinline__ ActParam act_param(int act) {
    ActParam a;
    a.code = act;
    a.special = (act == PQ_ACT_LOGISTIC || act == PQ_ACT_TANH) ? 1 : 0;
    a.slope = (act == PQ_ACT_LINEAR) ? 1.f : (act == PQ_ACT_LEAKY ? 0.1f : 0.f);
    a.hi = (act == PQ_ACT_RELU6) ? 6.f : __builtin_huge_valf();
    return a;
}
__device__ __forceinline__ float act_fwd_slow(float u, int act) {
    return act == PQ_ACT_LOGISTIC ? 1.f / (1.f + __expf(-u)) : tanhf(u);
}
__device__ __forceinline__ float act_bwd_slow(float u, int act) {
    if (act == PQ_ACT_LOGISTIC) { float s = 1.f / (1.f + __expf(-u)); return s * (1.f - s); }
    float t = tanhf(u);
    return 1.f - t * t;
}
// piecewise-linear family only (callers reject logistic/tanh on virtual operands at the host)
__device__ __forceinline__ float act_fwd(float u, const ActParam& a) {
    return fminf(fmaf(a.slope, fminf(u, 0.f), fmaxf(u, 0.f)), a.hi);
}
// derivative w.r.t. the pre-activation u (ATen: threshold_backward / hardtanh_backward use strict bounds)
__device__ __forceinline__ float act_bwd(float u, const ActParam& a) {
    return (u > 0.f ? 1.f : a.slope) * (u < a.hi ? 1.f : 0.f);
}
inline bool act_is_special(int act) { return act == PQ_ACT_LOGISTIC || act == PQ_ACT_TANH; }

// ---- BatchNorm statistics hand-over without a finalize launch (pq_bn_stat) ---------------------------------------
// producer side: one fp32 partial sum -> two 64-bit integer atomics on its fixed-point image (unit 2^-50: low limb = the 50
// fraction bits, high limb = floor(v)).  Integer addition is associative, so the accumulated value does not depend on the
// arrival order; bits of v below 2^-50 are dropped (|error| < 1e-15 per partial, the same for every order).
constexpr float PQ_STAT_MAX = 5.0e14f;         // |v| < 2^49: 8192 partials of that size still fit the high limb
// The LOW limb takes up to 2^50 - 1 per partial and is read back as a signed 64-bit integer: at most PQ_STAT_MAX_ADDS partials may land
// in one slot (4096 x 2^50 = 2^62).  Every entry point that accepts accumulators checks its own row-block count against it
// (stat_adds_ok); the engine stays far below (<= 640 row blocks per layer).
constexpr int PQ_STAT_MAX_ADDS = 4096;
inline bool stat_adds_ok(long nblk, int slots) { return (nblk + (slots > 0 ? slots : 1) - 1) / (slots > 0 ? slots : 1) <= PQ_STAT_MAX_ADDS; }
__device__ __forceinline__ void bn_stat_add(int64_t* acc, uint32_t* bad, int slots, int ldp, int blk, int plane, int c, float v) {
    if (!(fabsf(v) < PQ_STAT_MAX)) { atomicOr(bad, 1u); return; }          // NaN / Inf / absurd: poison the layer (NaN statistics)
    unsigned long long* d = reinterpret_cast<unsigned long long*>(acc + (((size_t)(blk % slots) * 2 + plane) * ldp + c) * 2);
    const double sv = (double)v * 0x1p50;                                  // exact
    const double hi = floor((double)v);
    const long long h = (long long)hi, l = (long long)(sv - hi * 0x1p50);  // l in [0, 2^50)
    if (l) atomicAdd(d, (unsigned long long)l);
    if (h) atomicAdd(d + 1, (unsigned long long)h);
}
// consumer side: scale / shift (and mean / invstd) of channel c from the accumulators.  Called by ONE thread per channel and
// workgroup (the callers share the result through LDS): every workgroup of the consumer grid repeats this, so it is kept to two
// 16-byte loads per slot and a handful of fp64 operations -- mean and E[x^2] - mean^2 in fp64 (the cancellation), everything
// after that in fp32 (invstd = 1 / sqrt(var + eps) correctly rounded in fp32; pq_bn_fwd_finalize rounds the fp64 value: the two
// agree to an ulp, and every consumer of a layer sees the values the writer stored or derives exactly the same ones).
// writer: also store the vectors and update the running statistics (exactly one thread of the grid per channel does that).
struct BnCoef { float scale, shift, mean, inv; };
struct BnSums { long long lo1, hi1, lo2, hi2; };
__device__ __forceinline__ BnSums bn_acc_load(const int64_t* acc, int slots, int ldp, int c) {   // all loads of a channel, no use in between
    if (slots == 1) {                                                                     // uniform
        const longlong2 a = *reinterpret_cast<const longlong2*>(acc + ((size_t)0 * ldp + c) * 2);
        const longlong2 b = *reinterpret_cast<const longlong2*>(acc + ((size_t)1 * ldp + c) * 2);
        return BnSums{a.x, a.y, b.x, b.y};
    }
    longlong2 a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ss = s < slots ? s : 0;                                                 // slots <= 4 (pq_bn_stat_slots)
        a[s] = *reinterpret_cast<const longlong2*>(acc + (((size_t)ss * 2 + 0) * ldp + c) * 2);
        b[s] = *reinterpret_cast<const longlong2*>(acc + (((size_t)ss * 2 + 1) * ldp + c) * 2);
    }
    BnSums r{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < slots) { r.lo1 += a[s].x; r.hi1 += a[s].y; r.lo2 += b[s].x; r.hi2 += b[s].y; }
    return r;
}
__device__ __forceinline__ BnSums bn_stat_load(const pq_bn_stat& st, int c) { return bn_acc_load(st.acc, st.slots, st.ldp, c); }
// gamma / beta: the channel's affine parameters, loaded by the caller TOGETHER with the accumulators (loaded here, after the fp64
// chain, they were a second exposed round trip in every consumer's prologue)
__device__ __forceinline__ BnCoef bn_stat_finish(const pq_bn_stat& st, int c, const BnSums& u, bool bad, bool writer, float gamma, float beta) {
    double t1 = (double)u.hi1 + (double)u.lo1 * 0x1p-50, t2 = (double)u.hi2 + (double)u.lo2 * 0x1p-50;
    if (bad) t1 = t2 = __builtin_nan("");
    const double invn = (double)st.inv_count;
    const double mean = t1 * invn;
    double var = fma(-mean, mean, t2 * invn);
    if (var < 0.0) var = 0.0;
    BnCoef k;
    k.inv = 1.0f / sqrtf((float)var + st.eps);
    k.mean = (float)mean;
    k.scale = gamma * k.inv;
    k.shift = beta - k.mean * k.scale;
    if (writer) {
        st.mean[c] = k.mean; st.invstd[c] = k.inv; st.scale[c] = k.scale; st.shift[c] = k.shift;
        if (st.running_mean) {
            const double n = (double)st.count;
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            st.running_mean[c] = (1.f - st.momentum) * st.running_mean[c] + st.momentum * k.mean;
            st.running_var[c] = (1.f - st.momentum) * st.running_var[c] + st.momentum * (float)unb;
        }
        if (c == 0 && st.num_batches_tracked) *st.num_batches_tracked += 1;
    }
    return k;
}
// U channels c0, c0 + stride, ... of one thread with ALL their loads in flight before the first use (one memory round trip in the
// consumer's prologue instead of U); out(c, coef) receives the results
template <int U, class F>
__device__ __forceinline__ void bn_stat_coefs(const pq_bn_stat& st, int c0, int stride, int C, bool writer, F out) {
    BnSums u[U];
    float g[U], b[U];
    const unsigned bad = *st.bad;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int c = min(c0 + i * stride, C - 1);
        g[i] = st.gamma[c]; b[i] = st.beta[c];
        u[i] = bn_stat_load(st, c);
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int c = c0 + i * stride;
        if (c < C) out(c, bn_stat_finish(st, c, u[i], bad != 0, writer, g[i], b[i]));
    }
}

// ---- operand loaders -----------------------------------------------------------------------------------------
// Per-thread cached coefficient pack for 4 consecutive channels of a pq_src.
struct SrcCoef4 {
    float a[4], b[4];
    ActParam act;
    bool affine, q;
    float qs, qinv, qzp;      // QAT activation quantiser of the tensor: scale, 1/scale, zero point
};
// the activation fake-quantiser (quint8 [0, 255]); same expression order as qat.hip's fq_value
__device__ __forceinline__ float fq_act(float v, float qs, float qinv, float qzp) {
    return (fminf(fmaxf(rintf(v * qinv) + qzp, 0.f), 255.f) - qzp) * qs;
}
__device__ __forceinline__ SrcCoef4 src_coef4(const pq_src& s, int c0, int C) {
    SrcCoef4 k;
    k.act = act_param(s.act);
    k.affine = s.scale != nullptr;
    k.q = s.qstate != nullptr;
    k.qs = k.q ? s.qstate[2] : 1.f;
    k.qinv = k.q ? 1.f / k.qs : 1.f;
    k.qzp = k.q ? s.qstate[3] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = c0 + j;
        bool ok = k.affine && c < C;
        k.a[j] = ok ? s.scale[c] : 1.f;
        k.b[j] = ok ? s.shift[c] : 0.f;
    }
    return k;
}
// The same for an operand whose BatchNorm statistics were handed over by the producer (s.stat, pq_bn_stat): the workgroup's
// LEADER lane of a channel quad derives the 4 coefficient pairs from the accumulators and shares them through `sh8` (8 floats
// of LDS per quad); every thread of the workgroup must call this (it contains a barrier).  writer: this leader is the one of
// the whole GRID that stores the vectors of its quad.  Without s.stat it is src_coef4.
__device__ __forceinline__ SrcCoef4 src_coef4_shared(const pq_src& s, int c0, int C, bool leader, bool writer, float* sh8) {
    if (s.stat.acc == nullptr) return src_coef4(s, c0, C);          // uniform
    if (leader) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sh8[j] = 1.f; sh8[4 + j] = 0.f; }
        bn_stat_coefs<4>(s.stat, c0, 1, C, writer, [&](int c, const BnCoef& bc) { sh8[c - c0] = bc.scale; sh8[4 + c - c0] = bc.shift; });
    }
    __syncthreads();
    SrcCoef4 k;
    k.act = act_param(s.act);
    k.affine = true;
    k.q = false; k.qs = 1.f; k.qinv = 1.f; k.qzp = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { k.a[j] = sh8[j]; k.b[j] = sh8[4 + j]; }
    return k;
}
// The same for a 256-thread workgroup that owns a chunk of up to 32 channels starting at chunk0 (8 quads: the LDS-tiled depthwise
// kernels): thread (channel t % 32, part t / 32) issues ONE 16-byte load -- part = (slot, statistic) -- so the whole chunk costs
// one memory round trip and no extra registers; 32 threads then finish their channel.  lds: >= 2048 + 64 floats.  quad: this
// thread's quad inside the chunk; writer: this workgroup stores the vectors.  Every thread must call it (two barriers).
__device__ __forceinline__ SrcCoef4 src_coef4_chunk32(const pq_src& s, int chunk0, int quad, int C, bool writer, float* lds) {
    const pq_bn_stat& st = s.stat;                                         // callers check st.acc != NULL
    const int t = threadIdx.x, ch = t & 31, part = t >> 5;                 // part: slot = part >> 1, statistic = part & 1
    longlong2* stage = reinterpret_cast<longlong2*>(lds);                  // [8][32]
    float* coef = lds + 2048;                                              // [2][32]
    const int c = min(chunk0 + ch, C - 1);
    if ((part >> 1) < st.slots)
        stage[part * 32 + ch] = *reinterpret_cast<const longlong2*>(st.acc + (((size_t)(part >> 1) * 2 + (part & 1)) * st.ldp + c) * 2);
    const unsigned bad = *st.bad;
    const float ga = st.gamma[c], be = st.beta[c];                         // (same round trip as the accumulators)
    __syncthreads();
    if (t < 32) {
        BnSums u{0, 0, 0, 0};
        for (int sl = 0; sl < st.slots; ++sl) {
            const longlong2 a = stage[(2 * sl) * 32 + ch], b = stage[(2 * sl + 1) * 32 + ch];
            u.lo1 += a.x; u.hi1 += a.y; u.lo2 += b.x; u.hi2 += b.y;
        }
        float sc = 1.f, sh = 0.f;
        if (chunk0 + ch < C) { const BnCoef bc = bn_stat_finish(st, chunk0 + ch, u, bad != 0, writer, ga, be); sc = bc.scale; sh = bc.shift; }
        coef[ch] = sc; coef[32 + ch] = sh;
    }
    __syncthreads();
    SrcCoef4 k;
    k.act = act_param(s.act);
    k.affine = true;
    k.q = false; k.qs = 1.f; k.qinv = 1.f; k.qzp = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { k.a[j] = coef[4 * quad + j]; k.b[j] = coef[32 + 4 * quad + j]; }
    return k;
}
__device__ __forceinline__ float4 src_apply4(float4 v, const SrcCoef4& k) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = act_fwd(fmaf(k.a[j], r[j], k.b[j]), k.act);     // (a, b) default to (1, 0)
    if (k.q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = fq_act(r[j], k.qs, k.qinv, k.qzp);
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

struct GsrcCoef4 {
    float p[4], q[4], mu[4], r[4];
    bool two;
};
__device__ __forceinline__ GsrcCoef4 gsrc_coef4(const pq_gsrc& s, int c0, int C) {
    GsrcCoef4 k;
    k.two = s.y != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = c0 + j;
        bool ok = k.two && c < C;
        k.p[j] = ok ? s.p[c] : 1.f;
        k.q[j] = ok ? s.q[c] : 0.f;
        k.mu[j] = ok ? s.mu[c] : 0.f;
        k.r[j] = ok ? s.r[c] : 0.f;
    }
    return k;
}
// A gradient operand whose BatchNorm-backward sums were handed over by the sink kernel (pq_bn_bstat): p, q, r, mu of ONE channel
// are derived by the calling thread (arithmetic of bn_bwd_finalize_kernel on the exactly accumulated sums; a multiplication by
// 1 / count instead of its fp64 division); writer: this thread is the one of the whole grid that stores dgamma / dbeta and the
// channel's entries of the vectors.  c must be < C.
struct GsrcCoef1 { float p, q, mu, r; };
__device__ __forceinline__ GsrcCoef1 gsrc_coef1_bstat(const pq_gsrc& s, const pq_bn_bstat& st, int c, bool writer) {
    const unsigned bad = *st.bad;
    const float ga = st.gamma[c], iv = st.invstd[c];         // (in front of the accumulator loads and their uniform branch on the slot
    GsrcCoef1 k;                                             //  count: behind it the compiler issued them after the accumulators'
    k.mu = s.mu[c];                                          //  wait -- a second exposed round trip)
    const BnSums u = bn_acc_load(st.acc, st.slots, st.ldp, c);
    double t1 = (double)u.hi1 + (double)u.lo1 * 0x1p-50, t2 = (double)u.hi2 + (double)u.lo2 * 0x1p-50;
    if (bad) t1 = t2 = __builtin_nan("");
    const double invn = 1.0 / (double)st.count;
    k.p = ga * iv;
    k.q = st.train ? (float)(-(double)k.p * (double)iv * t2 * invn) : 0.f;
    k.r = st.train ? (float)(-(double)k.p * t1 * invn) : 0.f;
    if (writer) { st.dbeta[c] = (float)t1; st.dgamma[c] = (float)t2; st.p[c] = k.p; st.q[c] = k.q; st.r[c] = k.r; }
    return k;
}
// U channels c0, c0 + stride, ... of one thread with ALL their accumulator loads in flight before the first use (one memory round trip
// in the consumer's prologue); out(c, coef1) receives the results
template <int U, class F>
__device__ __forceinline__ void bn_bstat_coefs(const pq_bn_bstat& st, const float* mu, int c0, int stride, int C, bool writer, F out) {
    BnSums u[U];
    float ga[U], iv[U], m[U];
    const unsigned bad = *st.bad;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int c = min(c0 + i * stride, C - 1);
        u[i] = bn_acc_load(st.acc, st.slots, st.ldp, c);
        ga[i] = st.gamma[c]; iv[i] = st.invstd[c]; m[i] = mu[c];
    }
    const double invn = 1.0 / (double)st.count;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int c = c0 + i * stride;
        if (c < C) {
            double t1 = (double)u[i].hi1 + (double)u[i].lo1 * 0x1p-50, t2 = (double)u[i].hi2 + (double)u[i].lo2 * 0x1p-50;
            if (bad) t1 = t2 = __builtin_nan("");
            GsrcCoef1 k;
            k.mu = m[i];
            k.p = ga[i] * iv[i];
            k.q = st.train ? (float)(-(double)k.p * (double)iv[i] * t2 * invn) : 0.f;
            k.r = st.train ? (float)(-(double)k.p * t1 * invn) : 0.f;
            if (writer) { st.dbeta[c] = (float)t1; st.dgamma[c] = (float)t2; st.p[c] = k.p; st.q[c] = k.q; st.r[c] = k.r; }
            out(c, k);
        }
    }
}
// the same for 4 consecutive channels (kernels whose threads own a channel quad); writer: store the vectors too
__device__ __forceinline__ GsrcCoef4 gsrc_coef4_bstat(const pq_gsrc& s, const pq_bn_bstat& st, int c0, int C, bool writer) {
    GsrcCoef4 k;
    k.two = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        GsrcCoef1 k1{1.f, 0.f, 0.f, 0.f};
        if (c0 + j < C) k1 = gsrc_coef1_bstat(s, st, c0 + j, writer);
        k.p[j] = k1.p; k.q[j] = k1.q; k.mu[j] = k1.mu; k.r[j] = k1.r;
    }
    return k;
}
// a sink's statistics accumulators (pq_gsink.stat_acc) as block_reduce_store_stats / the GEMM epilogues take them
struct SinkAcc { int64_t* acc; uint32_t* bad; int slots; };
__device__ __forceinline__ SinkAcc sink_acc(const pq_gsink& s) {
    return SinkAcc{(s.finalize && s.mean != nullptr && s.partial != nullptr) ? s.stat_acc : nullptr, s.stat_bad, s.stat_slots > 0 ? s.stat_slots : 1};
}

__device__ __forceinline__ float4 gsrc_apply4(float4 g, float4 y, const GsrcCoef4& k) {
    if (!k.two) return g;
    float gg[4] = {g.x, g.y, g.z, g.w}, yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) gg[j] = fmaf(k.p[j], gg[j], fmaf(k.q[j], yy[j] - k.mu[j], k.r[j]));
    return make_float4(gg[0], gg[1], gg[2], gg[3]);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// ---- per-channel sink coefficients for 4 consecutive channels ------------------------------------------------
struct SinkCoef4 {
    float a[4], b[4], mean[4], inv[4];
    bool affine, stats, q;
    float qinv, qzp;          // QAT: 1/scale and zero point of the tensor's activation observer
    ActParam act;
};
__device__ __forceinline__ SinkCoef4 sink_coef4(const pq_gsink& s, int c0, int C) {
    SinkCoef4 k;
    k.affine = s.scale != nullptr;
    k.stats = s.mean != nullptr && s.partial != nullptr;
    k.act = act_param(s.act);
    k.q = s.qstate != nullptr;
    k.qinv = k.q ? 1.f / s.qstate[2] : 0.f;
    k.qzp = k.q ? s.qstate[3] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = c0 + j;
        bool ok = c < C;
        k.a[j] = (k.affine && ok) ? s.scale[c] : 1.f;
        k.b[j] = (k.affine && ok) ? s.shift[c] : 0.f;
        k.mean[j] = (k.stats && ok) ? s.mean[c] : 0.f;
        k.inv[j] = (k.stats && ok) ? s.invstd[c] : 0.f;
    }
    return k;
}
// does the sink read y when it finalises?
__device__ __forceinline__ bool sink_needs_y(const pq_gsink& s, const SinkCoef4& k) {
    return s.finalize && (k.act.code != PQ_ACT_LINEAR || k.stats || k.q);
}
// dz (accumulated already) -> gradient wrt y for 4 channels of one row; y4 is only used when sink_needs_y()
__device__ __forceinline__ float4 sink_apply4(const pq_gsink& s, const SinkCoef4& k, float4 dz, float4 y4, float4& s1, float4& s2) {
    if (sink_needs_y(s, k)) {
        float d[4] = {dz.x, dz.y, dz.z, dz.w}, y[4] = {y4.x, y4.y, y4.z, y4.w};
        float t1[4] = {s1.x, s1.y, s1.z, s1.w}, t2[4] = {s2.x, s2.y, s2.z, s2.w};
        if (k.q) {                // QAT only: straight-through mask of the activation fake-quantiser (arithmetic of qat.hip)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = k.affine ? fmaf(k.a[j], y[j], k.b[j]) : y[j];
                const float qv = rintf(act_fwd(u, k.act) * k.qinv) + k.qzp;
                d[j] = (qv >= 0.f && qv <= 255.f) ? d[j] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u = k.affine ? fmaf(k.a[j], y[j], k.b[j]) : y[j];
            float g = d[j] * act_bwd(u, k.act);
            d[j] = g;
            if (k.stats) {
                t1[j] += g;
                t2[j] = fmaf(g, (y[j] - k.mean[j]) * k.inv[j], t2[j]);
            }
        }
        dz = make_float4(d[0], d[1], d[2], d[3]);
        s1 = make_float4(t1[0], t1[1], t1[2], t1[3]);
        s2 = make_float4(t2[0], t2[1], t2[2], t2[3]);
    }
    return dz;
}
// where an accumulating sink reads the running gradient (pq_gsink.addend: the first contribution may live in another tensor)
__device__ __forceinline__ const float* sink_old(const pq_gsink& s) { return s.addend ? s.addend : s.g; }
// Emit 4 channels of one row into the sink; accumulates the two statistics.
__device__ __forceinline__ void sink_emit4(const pq_gsink& s, const SinkCoef4& k, size_t row, int c0, float4 dz,
                                           float4& s1, float4& s2) {
    float* gp = s.g + row * (size_t)s.ldg + c0;
    if (s.accumulate) dz = f4add(dz, ldg4(sink_old(s) + row * (size_t)s.ldg + c0));
    float4 y4 = f4zero();
    if (sink_needs_y(s, k)) y4 = ldg4(s.y + row * (size_t)s.ldy + c0);
    stg4(gp, sink_apply4(s, k, dz, y4, s1, s2));
}

// The same in two halves, so that a thread that emits several rows can have ALL their sink operands in flight before its first
// store (the compiler cannot hoist the y / g loads of row k+1 above the store of row k -- they may alias -- which made every
// emitted row its own exposed memory round trip in the streaming depthwise kernels).  `row` must be a valid (clamped) row.
struct SinkOps { float4 y, g; };
__device__ __forceinline__ SinkOps sink_load4(const pq_gsink& s, const SinkCoef4& k, size_t row, int c0) {
    SinkOps o;
    o.g = s.accumulate ? ldg4(sink_old(s) + row * (size_t)s.ldg + c0) : f4zero();
    o.y = sink_needs_y(s, k) ? ldg4(s.y + row * (size_t)s.ldy + c0) : f4zero();
    return o;
}
__device__ __forceinline__ void sink_store4(const pq_gsink& s, const SinkCoef4& k, size_t row, int c0, float4 dz, const SinkOps& o,
                                            float4& s1, float4& s2) {
    if (s.accumulate) dz = f4add(dz, o.g);
    stg4(s.g + row * (size_t)s.ldg + c0, sink_apply4(s, k, dz, o.y, s1, s2));
}

// ---- XCD-aware block numbering ---------------------------------------------------------------------------------
// MI355X has 8 XCDs with private L2s and deals workgroups to them round-robin by linear block id, so consecutive ids
// (which normally work on neighbouring rows / tiles and therefore share halo rows or operand panels) land in eight
// different L2s and every one of them fetches the shared data again.  Every kernel here derives its logical block
// coordinates from xcd_block(): the ids that one XCD receives (x, x+8, x+16, ...) are mapped onto one CONTIGUOUS slice of
// the logical index space, so neighbours share an L2.  Pure renumbering (a bijection): correctness never depends on
// the hardware's actual placement.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    constexpr int NX = 8;
    if (total < 2 * NX) return lin;
    const int per = total / NX, rem = total - per * NX;
    const int x = lin % NX, j = lin / NX;
    return x * per + min(x, rem) + j;
}
__device__ __forceinline__ void xcd_block(int& bx, int& by) {
    const int gx = gridDim.x;
    const int t = xcd_remap(blockIdx.y * gx + blockIdx.x, gx * (int)gridDim.y);
    by = t / gx;
    bx = t - by * gx;
}

// ---- 2-D thread layout used by the NHWC elementwise / depthwise kernels ----------------------------------------
// blockDim = (CX, PY): CX lanes walk channel quads (float4), PY walks rows / pixel groups.
// Reduce two float4 accumulators over threadIdx.y and store to partial[bx][0|1][ldp] (bx = logical row-block).
// planes = 2 (sum, sum of squares) or 4 (+ per-channel min, max of the stored values: QAT observers, written by
// block_reduce_store_minmax): the row pitch of `partial` is planes * ldp.
// sa.acc != nullptr (pq_gsink.stat_acc): the two sums are ADDED into the sink's exact accumulators instead (bn_stat_add).
template <int MAXT>
__device__ __forceinline__ void block_reduce_store_stats(float4 s1, float4 s2, float* partial, int ldp, int c0, int C,
                                                         float* lds /* >= MAXT*8 floats */, int bx, int planes = 2,
                                                         SinkAcc sa = SinkAcc{nullptr, nullptr, 1}) {
    const int tx = threadIdx.x, ty = threadIdx.y, nx = blockDim.x, ny = blockDim.y;
    float* slot = lds + (size_t)(ty * nx + tx) * 8;
    slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
    slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
    __syncthreads();
    if (ty == 0 && c0 < C) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int y = 0; y < ny; ++y) {
            const float* o = lds + (size_t)(y * nx + tx) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += o[j];
        }
        if (sa.acc != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < C) {
                    bn_stat_add(sa.acc, sa.bad, sa.slots, ldp, bx, 0, c0 + j, acc[j]);
                    bn_stat_add(sa.acc, sa.bad, sa.slots, ldp, bx, 1, c0 + j, acc[4 + j]);
                }
            return;
        }
        float* dst = partial + (size_t)bx * planes * ldp;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < C) { dst[c0 + j] = acc[j]; dst[ldp + c0 + j] = acc[4 + j]; }
    }
}
// same layout, planes 2 / 3 of a 4-plane row: per-channel minimum / maximum over the block (call after the sums: it reuses lds)
template <int MAXT>
__device__ __forceinline__ void block_reduce_store_minmax(float4 mn, float4 mx, float* partial, int ldp, int c0, int C,
                                                          float* lds /* >= MAXT*8 floats */, int bx) {
    const int tx = threadIdx.x, ty = threadIdx.y, nx = blockDim.x, ny = blockDim.y;
    __syncthreads();
    float* slot = lds + (size_t)(ty * nx + tx) * 8;
    slot[0] = mn.x; slot[1] = mn.y; slot[2] = mn.z; slot[3] = mn.w;
    slot[4] = mx.x; slot[5] = mx.y; slot[6] = mx.z; slot[7] = mx.w;
    __syncthreads();
    if (ty == 0 && c0 < C) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = __builtin_huge_valf(); acc[4 + j] = -__builtin_huge_valf(); }
        for (int y = 0; y < ny; ++y) {
            const float* o = lds + (size_t)(y * nx + tx) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] = fminf(acc[j], o[j]); acc[4 + j] = fmaxf(acc[4 + j], o[4 + j]); }
        }
        float* dst = partial + (size_t)bx * 4 * ldp;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < C) { dst[2 * ldp + c0 + j] = acc[j]; dst[3 * ldp + c0 + j] = acc[4 + j]; }
    }
}
__device__ __forceinline__ float4 f4min(float4 a, float4 b) { return make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)); }
__device__ __forceinline__ float4 f4max(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

// Largest divisor of c4 that is <= 64 (channel-quad lanes per block row).
inline int pick_cx(int c4) {
    int best = 1;
    for (int d = 1; d <= 64 && d <= c4; ++d)
        if (c4 % d == 0) best = d;
    return best;
}

// Launch geometry shared by every kernel that uses the (CX, PY) layout: gx row-blocks x gy channel chunks.
struct EwGeom { int cx, py, gy, gx; };
inline EwGeom ew_geom(long items, int C) {
    EwGeom g;
    const int c4 = (C + 3) / 4;
    g.cx = pick_cx(c4);
    g.py = 256 / g.cx;
    if (g.py < 1) g.py = 1;
    g.gy = (c4 + g.cx - 1) / g.cx;
    long want = (items + g.py - 1) / g.py;
    // at most ~512 workgroups (two per CU): fewer, fatter workgroups keep four rows per thread in flight and leave a quarter of the
    // statistics rows for the BatchNorm finalize that follows (2048 / 1024 / 512 / 256: mobilenetv2-fpn step 10.40 / 10.32 / 10.27 /
    // 10.30 ms, regnetx-600m-fpn 11.98 / 11.62 / 11.52 / 11.60)
    long cap = 512 / g.gy;
    if (cap < 64) cap = 64;
    g.gx = (int)(want < cap ? want : cap);
    if (g.gx < 1) g.gx = 1;
    return g;
}

}  // namespace pq

// host-side error plumbing (api.hip)
void pq_set_error(const char* fmt, ...);
// The two-stream executor (api.hip) can ask the next single-kernel op to BIND its fork event to the kernel dispatch itself
// (hipExtLaunchKernelGGL stopEvent) instead of recording it afterwards: a separate hipEventRecord is a barrier packet that
// delays the next kernel of the same queue by 5-8 us.  pq_take_stop_event() returns that event once (nullptr otherwise).
hipEvent_t pq_take_stop_event();
// EVERY kernel launch of the library goes through PQ_LAUNCH / PQ_LAUNCH_BINDABLE: with the launch log on (pq_launch_log, tests only)
// the kernel expression and the enclosing function (whose pretty name carries the template arguments of templated dispatchers) are
// noted, so the suite can prove that every kernel instantiation a plan of the BASELINE nets launches is also launched by a value test
// (tests/test_zz_instantiations_gpu.py).  Off: one predictable branch per launch.
extern bool pq_launch_log_on;
void pq_note_launch(const char* kernel, const char* where);
#define PQ_LAUNCH(kernel, grid, block, shmem, st, ...)                                                       \
    do {                                                                                                      \
        if (pq_launch_log_on) pq_note_launch(#kernel, __PRETTY_FUNCTION__);                                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                      \
    } while (0)
#define PQ_LAUNCH_BINDABLE(kernel, grid, block, shmem, st, ...)                                              \
    do {                                                                                                      \
        if (pq_launch_log_on) pq_note_launch(#kernel, __PRETTY_FUNCTION__);                                   \
        hipEvent_t stop_ = pq_take_stop_event();                                                              \
        if (stop_) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, nullptr, stop_, 0, __VA_ARGS__);     \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                 \
    } while (0)
#define PQ_CHECK_ARG(cond, msg)                         \
    do {                                                \
        if (!(cond)) {                                  \
            pq_set_error("%s: %s", __func__, msg);      \
            return PQ_EINVAL;                           \
        }                                               \
    } while (0)
// A sink that carries statistics accumulators (pq_gsink.stat_acc): what every entry point that HONOURS them checks -- nblk = the
// row blocks (workgroups) that will add into them.  Entry points whose kernels cannot honour them return PQ_EUNSUPPORTED through
// PQ_SINK_ACC_UNSUPPORTED instead of silently writing `partial` rows (a reader with dy_stat would derive p / q / r from zeros).
inline const char* sink_acc_check(const pq_gsink& s, int C, long nblk) {
    if (!s.stat_acc) return nullptr;
    if (!s.stat_bad) return "dx.stat_acc needs stat_bad";
    if (!(s.finalize && s.mean && s.partial)) return "dx.stat_acc needs a finalising sink with statistics (mean and partial set)";
    if (s.ldp < C) return "dx.stat_acc: statistics pitch smaller than the channel count";
    if (s.stat_slots > 4) return "dx.stat_acc: at most 4 accumulator slots";
    if (!pq::stat_adds_ok(nblk, s.stat_slots)) return "dx.stat_acc: more than 4096 row blocks per accumulator slot";
    return nullptr;
}
#define PQ_SINK_ACC_CHECK(sink, C, nblk)                                    \
    do {                                                                    \
        const char* m_ = sink_acc_check(sink, C, nblk);                     \
        if (m_) { pq_set_error("%s: %s", __func__, m_); return PQ_EINVAL; } \
    } while (0)
// pq_gsink.addend: only read by an accumulating sink, through float4 loads in most kernels
#define PQ_SINK_ADDEND_CHECK(sink)                                                                                              \
    do {                                                                                                                        \
        if ((sink).addend && (!(sink).accumulate || (((uintptr_t)(sink).addend) & 15))) {                                       \
            pq_set_error("%s: dx.addend needs accumulate = 1 and a 16-byte aligned address (pitch ldg)", __func__);             \
            return PQ_EINVAL;                                                                                                   \
        }                                                                                                                       \
    } while (0)
#define PQ_SINK_ACC_UNSUPPORTED(sink)                                                                              \
    do {                                                                                                            \
        if ((sink).stat_acc) {                                                                                      \
            pq_set_error("%s: this entry does not add into dx.stat_acc (ask pq_bn_bstat_ok, role 0)", __func__);    \
            return PQ_EUNSUPPORTED;                                                                                 \
        }                                                                                                           \
    } while (0)
#define PQ_CHECK_LAUNCH()                                                   \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            pq_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return PQ_ELAUNCH;                                              \
        }                                                                   \
    } while (0)
