// YOLO head: box decode (reference model/parser.py:206-235), fused per-cell loss forward / backward
// (reference model/loss.py:22-115) and the IoU family (reference tools.py:357-477), gfx950.
//
// One lane owns one (b, h, w, anchor) cell: it streams the cell's 5+C head values and 6+C label values once,
// evaluates box / objectness / class terms and -- in the backward kernel -- writes the gradient w.r.t. the raw head
// directly (the (B,H,W,A,M) IoU tensor, the focal / BCE temporaries and the decoded prediction of the eager
// reference are never materialised).  Gradients follow ATen's formulas exactly: BCE backward is
// (p - t) / max((1 - p) * p, 1e-12) (not the derivative of the clamped logs), focal weights are differentiated,
// the ignore mask is not, ciou's alpha is a constant.
#include "common.h"

namespace pq {

struct D4 { float v; float d[4]; };
__device__ __forceinline__ D4 dvar(float v, int k) { D4 r; r.v = v; r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0.f; r.d[k] = 1.f; return r; }
__device__ __forceinline__ D4 dcst(float v) { D4 r; r.v = v; r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0.f; return r; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
    D4 r; r.v = a.v / b.v;
    const float ib = 1.f / b.v;
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return (a.v >= b.v) ? a : b; }
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return (a.v <= b.v) ? a : b; }
__device__ __forceinline__ D4 datan(D4 a) { D4 r; r.v = atanf(a.v); const float g = 1.f / (1.f + a.v * a.v); for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g; return r; }

// IoU-family value (and derivative w.r.t. box 1) -- tools.py:357-477, no epsilon anywhere.
__device__ __forceinline__ D4 iou_family(const D4 (&b1)[4], const float (&b2)[4], int kind) {
    const D4 t0 = dcst(b2[0]), t1 = dcst(b2[1]), t2 = dcst(b2[2]), t3 = dcst(b2[3]);
    const D4 zero = dcst(0.f);
    const D4 w1 = b1[2] - b1[0], h1 = b1[3] - b1[1];
    const D4 a1 = w1 * h1;
    const float w2 = b2[2] - b2[0], h2 = b2[3] - b2[1];
    const D4 a2 = dcst(w2 * h2);
    const D4 iw = dmax(dmin(b1[2], t2) - dmax(b1[0], t0), zero);
    const D4 ih = dmax(dmin(b1[3], t3) - dmax(b1[1], t1), zero);
    const D4 inter = iw * ih;
    const D4 uni = a1 + a2 - inter;
    const D4 iou = inter / uni;
    if (kind == PQ_BOX_IOU) return iou;
    const D4 elx = dmin(b1[0], t0), ely = dmin(b1[1], t1), ehx = dmax(b1[2], t2), ehy = dmax(b1[3], t3);
    const D4 ew = dmax(ehx - elx, zero), eh = dmax(ehy - ely, zero);
    const D4 earea = ew * eh;
    const D4 g = iou - (earea - uni) / earea;
    if (kind == PQ_BOX_GIOU) return g;
    const D4 half = dcst(0.5f);                       // (a + b) / 2 == (a + b) * 0.5 exactly in binary fp
    const D4 cx = (b1[0] + b1[2]) * half - dcst((b2[0] + b2[2]) / 2.f);
    const D4 cy = (b1[1] + b1[3]) * half - dcst((b2[1] + b2[3]) / 2.f);
    const D4 dc = cx * cx + cy * cy;
    const D4 ex = elx - ehx, ey = ely - ehy;
    const D4 de = ex * ex + ey * ey;
    const D4 dterm = g + dc / de;                     // '+' as written in the reference (tools.py:437)
    if (kind == PQ_BOX_DIOU) return dterm;
    const D4 at = datan(w1 / h1) - dcst(atanf(w2 / h2));
    const D4 v = dcst(4.f / (3.14159265358979323846f * 3.14159265358979323846f)) * (at * at);
    const float alpha = v.v / ((1.f - iou.v) + v.v);  // no_grad in the reference (tools.py:473-475)
    return dterm + dcst(alpha) * v;
}

__device__ __forceinline__ float plain_iou(float x1, float y1, float x2, float y2, float a1, const float* bx) {
    const float a2 = (bx[2] - bx[0]) * (bx[3] - bx[1]);
    const float iw = fmaxf(fminf(x2, bx[2]) - fmaxf(x1, bx[0]), 0.f);
    const float ih = fmaxf(fminf(y2, bx[3]) - fmaxf(y1, bx[1]), 0.f);
    const float inter = iw * ih;
    return inter / (a1 + a2 - inter);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float bce_val(float p, float t) {
    return (t - 1.f) * fmaxf(logf(1.f - p), -100.f) - t * fmaxf(logf(p), -100.f);
}
__device__ __forceinline__ float bce_grad(float p, float t) { return (p - t) / fmaxf((1.f - p) * p, 1e-12f); }

// ------------------------------------------------------------------------------------------------ loss kernel
// One cell: IN(k) = k-th head value of the cell, LAB(k) = k-th label value, OUT(k, v) = store d loss / d head value k.
template <bool GRAD, class FIn, class FLab, class FOut>
__device__ __forceinline__ void yolo_cell(const pq_yolo_loss_args& a, int b, int hq, int wq, const float* sbox, bool boxes_in_lds,
                                          FIn IN, FLab LAB, FOut OUT, float& lbox, float& lconf, float& lcls) {
    const float fs = (float)a.stride;
    const float gx = (float)wq + 0.5f, gy = (float)hq + 0.5f;
    float pc[4], dpc[4];   // decoded corners and d(corner)/d(raw)
    float pconf, dconf;
    if (a.from_raw) {
        const float e0 = expf(IN(0)), e1 = expf(IN(1)), e2 = expf(IN(2)), e3 = expf(IN(3));
        pc[0] = (gx - e0) * fs; pc[1] = (gy - e1) * fs; pc[2] = (gx + e2) * fs; pc[3] = (gy + e3) * fs;
        dpc[0] = -e0 * fs; dpc[1] = -e1 * fs; dpc[2] = e2 * fs; dpc[3] = e3 * fs;
        pconf = sigmoidf_(IN(4));
        dconf = pconf * (1.f - pconf);
    } else {
        for (int k = 0; k < 4; ++k) { pc[k] = IN(k); dpc[k] = 1.f; }
        pconf = IN(4);
        dconf = 1.f;
    }
    const float t0 = LAB(0), t1 = LAB(1), t2 = LAB(2), t3 = LAB(3);
    const float respond = LAB(4);
    const float mixw = LAB(5 + a.C);
    const float in_area = (float)(a.stride * a.H) * (float)(a.stride * a.W);
    const float scale = 2.0f - 1.0f * (t2 - t0) * (t3 - t1) / in_area;
    const float invB = 1.f / (float)a.B;
    float gs_box = 0.f, gs_conf = 0.f, gs_cls = 0.f;
    if (GRAD) {
        if (a.gscale) { gs_box = a.gscale[0] * invB; gs_conf = a.gscale[1] * invB; gs_cls = a.gscale[2] * invB; }
        else {        // upstream gradient scalars of the step's totals (total, box, conf, cls, this head's branch)
            const float g0 = (a.gup[0] ? *a.gup[0] : 0.f) + (a.gup[4] ? *a.gup[4] : 0.f);
            gs_box = (g0 + (a.gup[1] ? *a.gup[1] : 0.f)) * invB;
            gs_conf = (g0 + (a.gup[2] ? *a.gup[2] : 0.f)) * invB;
            gs_cls = (g0 + (a.gup[3] ? *a.gup[3] : 0.f)) * invB;
        }
    }

    // ---- box term
    float gcorner[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.box_loss == PQ_BOX_L1) {
        const float tt[4] = {t0, t1, t2, t3};
        const float beta = 1.f / 9.f;
        float s = 0.f;
        for (int k = 0; k < 4; ++k) {
            const float d = pc[k] - tt[k];
            const float n = fabsf(d);
            const bool small = n < beta;
            s += small ? (0.5f * n * n / beta) : (n - 0.5f * beta);
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            gcorner[k] = (small ? (n / beta) * sg : sg) * 0.25f;
        }
        const float mean4 = s * 0.25f;
        lbox = respond * scale * mean4 * a.l1_gain;
        for (int k = 0; k < 4; ++k) gcorner[k] *= respond * scale * a.l1_gain;
    } else {
        D4 b1[4] = {dvar(pc[0], 0), dvar(pc[1], 1), dvar(pc[2], 2), dvar(pc[3], 3)};
        const float b2[4] = {t0, t1, t2, t3};
        const D4 x = iou_family(b1, b2, a.box_loss);
        lbox = respond * scale * (1.0f - x.v);
        for (int k = 0; k < 4; ++k) gcorner[k] = respond * scale * (-x.d[k]);
    }
    lbox *= mixw;

    // ---- objectness term with the ignore mask (max IoU against every ground-truth box of the image)
    const float a1 = (pc[2] - pc[0]) * (pc[3] - pc[1]);
    float best = -INFINITY;
    bool sawnan = false;
    for (int m = 0; m < a.M; ++m) {
        const float* bx = boxes_in_lds ? &sbox[m * 4] : &a.boxes[((size_t)b * a.M + m) * 4];
        const float v = plain_iou(pc[0], pc[1], pc[2], pc[3], a1, bx);
        sawnan |= (v != v);
        best = fmaxf(best, v);
    }
    if (sawnan) best = NAN;   // torch.max propagates NaN; (NaN < thr) is false
    const float bgd = (1.0f - respond) * ((best < a.ignore_thresh) ? 1.f : 0.f);
    {
        const float t = respond, p = pconf;
        const float alpha_t = 2.f * fabsf(t - 1.f + 0.75f);
        const float ad = fabsf(t - p);
        const float focal = alpha_t * (ad * ad);
        const float bce = bce_val(p, t);
        lconf = 1.f * focal * (respond * bce + bgd * bce) * mixw;
        if (GRAD) {
            const float dfocal = -2.f * alpha_t * (t - p);
            const float dl = (respond + bgd) * (dfocal * bce + focal * bce_grad(p, t));
            OUT(4, dl * dconf * mixw * gs_conf);
        }
    }
    if (GRAD)
        for (int k = 0; k < 4; ++k) OUT(k, gcorner[k] * dpc[k] * mixw * gs_box);

    // ---- class term.  Every product carries the factor `respond` (model/loss.py:103-106), which is exactly 0 for all but the
    // handful of cells that own a ground-truth box: there the term and its gradient are 0 whatever the class logits are --
    // unless one of them is NaN (0 * NaN = NaN reaches the loss and trips the reference's NaN trap), so the cheap path only
    // checks for that and skips the 3 transcendentals per class (this term was ~80 % of the kernel's instructions).
    if (respond == 0.f) {
        bool bad = false;
        for (int k = 0; k < a.C; ++k) {
            const float v = IN(5 + k);
            bad |= (v != v);
            if (GRAD) OUT(5 + k, (v != v) ? v : 0.f);
        }
        if (bad) lcls = NAN;
    } else {
        // eight classes per round, loads first (the loads of one round are the memory-level parallelism of the generic kernel)
        constexpr int CK = 8;
        for (int k0 = 0; k0 < a.C; k0 += CK) {
            float raw[CK], tt[CK];
    #pragma unroll
            for (int u = 0; u < CK; ++u) {
                const int k = min(k0 + u, a.C - 1);
                raw[u] = IN(5 + k);
                tt[u] = LAB(5 + k);
            }
    #pragma unroll
            for (int u = 0; u < CK; ++u) {
                const int k = k0 + u;
                if (k < a.C) {
                    const float q = a.from_raw ? sigmoidf_(raw[u]) : raw[u];
                    const float t = tt[u];
                    const float alpha_t = 2.f * fabsf(t - 1.f + 0.5f);
                    const float ad = fabsf(t - q);
                    const float focal = alpha_t * (ad * ad);
                    const float bce = bce_val(q, t);
                    lcls += 2.f * focal * respond * bce;
                    if (GRAD) {
                        const float dfocal = -2.f * alpha_t * (t - q);
                        const float dl = 2.f * respond * (dfocal * bce + focal * bce_grad(q, t));
                        const float dq = a.from_raw ? q * (1.f - q) : 1.f;
                        OUT(5 + k, dl * dq * mixw * gs_cls);
                    }
                }
            }
        }
    }
    lcls *= mixw;
}

__device__ __forceinline__ void yolo_block_sums(const pq_yolo_loss_args& a, int b, float lbox, float lconf, float lcls, float (*wred)[3]) {
    // block reduction of the three sums -> partial[b][blockIdx.x][3]
    for (int off = 32; off > 0; off >>= 1) {
        lbox += __shfl_xor(lbox, off);
        lconf += __shfl_xor(lconf, off);
        lcls += __shfl_xor(lcls, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { wred[wv][0] = lbox; wred[wv][1] = lconf; wred[wv][2] = lcls; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float s = wred[0][threadIdx.x] + wred[1][threadIdx.x] + wred[2][threadIdx.x] + wred[3][threadIdx.x];
        a.partial[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = s;
    }
}

// any addressing (NCHW views of the standalone API, strided heads): every lane walks its own record in global memory
template <bool GRAD>
__global__ __launch_bounds__(256) void yolo_loss_kernel(pq_yolo_loss_args a) {
    __shared__ float sbox[128 * 4];
    __shared__ float wred[4][3];
    const int b = blockIdx.y;
    const int cells = a.H * a.W * a.A;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int nv = 5 + a.C;
    const bool boxes_in_lds = a.M <= 128;
    if (boxes_in_lds)
        for (int i = threadIdx.x; i < a.M * 4; i += 256) sbox[i] = a.boxes[(size_t)b * a.M * 4 + i];
    __syncthreads();
    float lbox = 0.f, lconf = 0.f, lcls = 0.f;
    if (cell < cells) {
        const int an = cell % a.A;
        const int wq = (cell / a.A) % a.W;
        const int hq = cell / (a.A * a.W);
        const float* in = a.in.base + (size_t)b * a.in.sb + (size_t)hq * a.in.sh + (size_t)wq * a.in.sw + (size_t)(an * nv) * a.in.sc;
        const float* lab = a.label + ((size_t)b * cells + cell) * (6 + a.C);
        float* gout = nullptr;
        if (GRAD) gout = a.grad + (size_t)b * a.gsb + (size_t)hq * a.gsh + (size_t)wq * a.gsw + (size_t)(an * nv) * a.gsc;
        const size_t sc = (size_t)a.in.sc, gsc = (size_t)a.gsc;
        yolo_cell<GRAD>(a, b, hq, wq, sbox, boxes_in_lds, [&](int k) { return in[(size_t)k * sc]; }, [&](int k) { return lab[k]; },
                        [&](int k, float v) { gout[(size_t)k * gsc] = v; }, lbox, lconf, lcls);
    }
    if (!GRAD) yolo_block_sums(a, b, lbox, lconf, lcls, wred);
}

// cooperative copy of n floats global -> LDS with 16-byte loads; the LDS image starts at the source's offset inside its 16-byte
// line (dst must be 16-byte aligned; element j lands at dst[mis + j], mis = floats of src past the line start), so every
// vector store is aligned whatever the source address is
__device__ __forceinline__ int lds_stage(const float* src, float* dst, int n) {
    const int mis = (int)((((uintptr_t)src) & 15) >> 2);
    const float* line = src - mis;
    const int nq = (mis + n + 3) >> 2;
    for (int q = threadIdx.x; q < nq; q += 256) {
        const int lo = 4 * q - mis;                       // first element of this quad (may be < 0 or reach past n)
        if (lo >= 0 && lo + 3 < n) *reinterpret_cast<float4*>(&dst[4 * q]) = ldg4(line + 4 * q);
        else
            for (int u = 0; u < 4; ++u)
                if (lo + u >= 0 && lo + u < n) dst[4 * q + u] = src[lo + u];
    }
    return mis;
}
__device__ __forceinline__ void lds_unstage(float* dstg, const float* src_lds, int n) {      // the inverse (src_lds 16-byte aligned image)
    const int mis = (int)((((uintptr_t)dstg) & 15) >> 2);
    float* line = dstg - mis;
    const int nq = (mis + n + 3) >> 2;
    for (int q = threadIdx.x; q < nq; q += 256) {
        const int lo = 4 * q - mis;
        if (lo >= 0 && lo + 3 < n) stg4(line + 4 * q, *reinterpret_cast<const float4*>(&src_lds[4 * q]));
        else
            for (int u = 0; u < 4; ++u)
                if (lo + u >= 0 && lo + u < n) dstg[lo + u] = src_lds[4 * q + u];
    }
}

// The engine's head layout (NHWC rows of A * (5 + C) contiguous values per pixel, dense pixel rows): the records of the 256
// cells of a block are ONE contiguous span of the head and one of the label table.  Both are staged into LDS with coalesced
// 16-byte loads (each lane walking its own 25-float record in global memory was pure load latency: 35 us for the 64 x 64 head
// where the bytes take 8), the gradient records are written over the staged head values in LDS and leave in one coalesced sweep
// (the pad floats between pixels carry their zeros through).
// (bx: the block's index among the blocks of THIS head; the forward's block sums use blockIdx.x / gridDim.x themselves, i.e. the
// forward is always a launch of one head)
template <bool GRAD>
__device__ __forceinline__ void yolo_loss_staged_body(const pq_yolo_loss_args& a, int span_cap, int lab_cap, int bx, int b) {
    extern __shared__ __attribute__((aligned(16))) float ylds[];
    __shared__ float sbox[128 * 4];
    __shared__ float wred[4][3];
    float* s_in = ylds;                                   // span_cap floats (+4 for the alignment shift)
    float* s_lab = ylds + span_cap + 4;                   // lab_cap floats (+4)
    const int cells = a.H * a.W * a.A;
    const int c0 = bx * 256;
    const int c1 = min(c0 + 256, cells);                  // cells [c0, c1)
    const int nv = 5 + a.C, nl = 6 + a.C;
    const int ld = (int)a.in.sw;                          // floats per pixel (dense rows: sh == W * sw)
    const bool boxes_in_lds = a.M <= 128;
    if (boxes_in_lds)
        for (int i = threadIdx.x; i < a.M * 4; i += 256) sbox[i] = a.boxes[(size_t)b * a.M * 4 + i];
    const size_t off0 = (size_t)(c0 / a.A) * ld + (size_t)(c0 % a.A) * nv;
    const size_t off1 = (size_t)((c1 - 1) / a.A) * ld + (size_t)((c1 - 1) % a.A) * nv + nv;
    const int span = (int)(off1 - off0);
    const float* src = a.in.base + (size_t)b * a.in.sb + off0;
    const int mis_in = lds_stage(src, s_in, span);
    const int mis_lab = lds_stage(a.label + ((size_t)b * cells + c0) * nl, s_lab, (c1 - c0) * nl);
    __syncthreads();
    float lbox = 0.f, lconf = 0.f, lcls = 0.f;
    const int cell = c0 + threadIdx.x;
    if (cell < c1) {
        const int an = cell % a.A, pix = cell / a.A;
        const int wq = pix % a.W, hq = pix / a.W;
        float* rec = s_in + mis_in + (int)((size_t)pix * ld + (size_t)an * nv - off0);
        const float* lab = s_lab + mis_lab + threadIdx.x * nl;
        yolo_cell<GRAD>(a, b, hq, wq, sbox, boxes_in_lds, [&](int k) { return rec[k]; }, [&](int k) { return lab[k]; },
                        [&](int k, float v) { rec[k] = v; }, lbox, lconf, lcls);
    }
    if (GRAD) {
        __syncthreads();
        float* dst = a.grad + (size_t)b * a.gsb + off0;
        // the LDS image keeps the INPUT's line offset; shift it to the output's if they differ (never in the engine: same pitch,
        // both buffers 256-byte aligned)
        const int mis_out = (int)((((uintptr_t)dst) & 15) >> 2);
        if (mis_out == mis_in) lds_unstage(dst, s_in, span);
        else for (int j = threadIdx.x; j < span; j += 256) dst[j] = s_in[mis_in + j];
    } else {
        yolo_block_sums(a, b, lbox, lconf, lcls, wred);
    }
}
template <bool GRAD>
__global__ __launch_bounds__(256) void yolo_loss_staged_kernel(pq_yolo_loss_args a, int span_cap, int lab_cap) {
    yolo_loss_staged_body<GRAD>(a, span_cap, lab_cap, blockIdx.x, blockIdx.y);
}
// gradients of up to three heads from ONE launch (pq_yolo_loss_args.also): the blocks of the heads are laid end to end on
// grid.x; the argument structs are picked by an if-chain (indexing a by-value array of structs would put it in scratch memory)
struct LossMulti { pq_yolo_loss_args a0, a1, a2; int span_cap[3], lab_cap[3], blk1, blk2; };
__global__ __launch_bounds__(256) void yolo_loss_bwd_multi_kernel(LossMulti m) {
    const int bx = blockIdx.x, b = blockIdx.y;
    if (bx < m.blk1) yolo_loss_staged_body<true>(m.a0, m.span_cap[0], m.lab_cap[0], bx, b);
    else if (bx < m.blk2) yolo_loss_staged_body<true>(m.a1, m.span_cap[1], m.lab_cap[1], bx - m.blk1, b);
    else yolo_loss_staged_body<true>(m.a2, m.span_cap[2], m.lab_cap[2], bx - m.blk2, b);
}

__global__ __launch_bounds__(256) void yolo_loss_finalize_kernel(const float* partial, int B, int nblk, float* losses, int32_t* nan_flag,
                                                                 float* totals, int head_slot, int n_heads) {
    // 16 image lanes x 16 block lanes.  Per image: the block partials are summed in double (16 lanes + a butterfly, fixed
    // order) and rounded to float like the reference's per-image .sum(); images are then added in double and averaged.
    __shared__ double sh[3][16];
    const int jl = threadIdx.x & 15, bl = threadIdx.x >> 4;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b0 = 0; b0 < B; b0 += 16) {
        const int b = b0 + bl;
        double s[3] = {0.0, 0.0, 0.0};
        if (b < B)
            for (int j = jl; j < nblk; j += 16) {
                const float* pp = partial + ((size_t)b * nblk + j) * 3;
                s[0] += (double)pp[0]; s[1] += (double)pp[1]; s[2] += (double)pp[2];
            }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] += __shfl_xor(s[k], off);
        }
        if (b < B) { acc[0] += (double)(float)s[0]; acc[1] += (double)(float)s[1]; acc[2] += (double)(float)s[2]; }
    }
    if (jl == 0) { sh[0][bl] = acc[0]; sh[1][bl] = acc[1]; sh[2][bl] = acc[2]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float term[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double t = 0.0;
            for (int q = 0; q < 16; ++q) t += sh[k][q];
            term[k] = (float)(t / (double)B);
        }
        const float total = term[0] + term[1] + term[2];
        losses[0] = total; losses[1] = term[0]; losses[2] = term[1]; losses[3] = term[2];
        if (nan_flag && total != total) *nan_flag = 1;      // sticky: only ever SET here (the host clears it when it raises)
        if (totals) {            // the other heads' rows were written by earlier launches (stream order / PQ_OPF_JOIN)
            const float* table = losses - 4 * head_slot;
            float col[4] = {0.f, 0.f, 0.f, 0.f};
            for (int h = 0; h < n_heads; ++h) {                  // head order, like the reference's element-wise sum of 4-tuples
                const float row[4] = {h == head_slot ? total : table[4 * h], h == head_slot ? term[0] : table[4 * h + 1],
                                      h == head_slot ? term[1] : table[4 * h + 2], h == head_slot ? term[2] : table[4 * h + 3]};
                for (int k = 0; k < 4; ++k) col[k] += row[k];
                totals[4 + h] = row[0];
            }
            for (int k = 0; k < 4; ++k) totals[k] = col[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------ decode
__global__ __launch_bounds__(256) void decode_kernel(pq_decode_args a) {
    const int nv = 5 + a.C;
    const long per_img = (long)a.H * a.W * a.A * nv;
    const long total = per_img * a.B;
    const float fs = (float)a.stride;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img);
        long r = i % per_img;
        const int k = (int)(r % nv); r /= nv;
        const int an = (int)(r % a.A); r /= a.A;
        const int wq = (int)(r % a.W);
        const int hq = (int)(r / a.W);
        const float raw = a.raw.base[(size_t)b * a.raw.sb + (size_t)hq * a.raw.sh + (size_t)wq * a.raw.sw + (size_t)(an * nv + k) * a.raw.sc];
        float v;
        if (k < 4) {
            const float g = ((k & 1) ? (float)hq : (float)wq) + 0.5f;
            const float e = expf(raw);
            v = (k < 2 ? (g - e) : (g + e)) * fs;
        } else {
            v = sigmoidf_(raw);
        }
        const size_t row = (size_t)a.row0 + ((size_t)hq * a.W + wq) * a.A + an;
        a.out[((size_t)b * a.rows_total + row) * nv + k] = v;
    }
}

__global__ __launch_bounds__(256) void decode_bwd_kernel(pq_decode_bwd_args a) {
    const int nv = 5 + a.C;
    const long per_img = (long)a.H * a.W * a.A * nv;
    const long total = per_img * a.B;
    const float fs = (float)a.stride;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img);
        long r = i % per_img;
        const int k = (int)(r % nv); r /= nv;
        const int an = (int)(r % a.A); r /= a.A;
        const int wq = (int)(r % a.W);
        const int hq = (int)(r / a.W);
        const float p = a.pred[i], dp = a.dpred[i];
        float g;
        if (k < 4) {
            const float c = (((k & 1) ? (float)hq : (float)wq) + 0.5f) * fs;
            g = dp * (p - c);                 // d/draw of (c/s -+ e^raw) * s  =  -+ e^raw * s  =  pred - c
        } else {
            g = dp * p * (1.f - p);
        }
        a.draw[(size_t)b * a.view.sb + (size_t)hq * a.view.sh + (size_t)wq * a.view.sw + (size_t)(an * nv + k) * a.view.sc] = g;
    }
}

__global__ void iou_pairs_kernel(const float* b1, const float* b2, float* out, int n, int kind) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    D4 x[4] = {dvar(b1[4 * i], 0), dvar(b1[4 * i + 1], 1), dvar(b1[4 * i + 2], 2), dvar(b1[4 * i + 3], 3)};
    const float y[4] = {b2[4 * i], b2[4 * i + 1], b2[4 * i + 2], b2[4 * i + 3]};
    out[i] = iou_family(x, y, kind).v;
}

}  // namespace pq

using namespace pq;

extern "C" int32_t pq_yolo_loss_blocks(int32_t H, int32_t W, int32_t A) { return (H * W * A + 255) / 256; }

static int loss_check(const pq_yolo_loss_args* a, bool bwd) {
    if (!a || !a->in.base || !a->label || !a->boxes) return 0;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->A <= 0 || a->C <= 0 || a->M <= 0) return 0;
    if (a->box_loss < PQ_BOX_L1 || a->box_loss > PQ_BOX_IOU) return 0;
    if (bwd) return a->grad != nullptr && (a->gscale != nullptr || a->gup[0] || a->gup[1] || a->gup[2] || a->gup[3] || a->gup[4]);
    return a->partial != nullptr && a->losses != nullptr;
}

// staged kernels: engine layout only (see yolo_loss_staged_kernel); LDS need in bytes, 0 = not eligible
static size_t loss_staged_lds(const pq_yolo_loss_args* a, bool bwd, int* span_cap, int* lab_cap) {
    if (!a->from_raw || a->in.sc != 1 || a->in.sw < (int64_t)a->A * (5 + a->C) || a->in.sh != a->in.sw * a->W) return 0;
    if (bwd && (a->gsc != 1 || a->gsw != a->in.sw || a->gsh != a->in.sh)) return 0;
    // span of 256 cells: at most ceil(256 / A) + 1 pixels
    *span_cap = (int)(((256 + a->A - 1) / a->A + 1) * a->in.sw + 8);
    *lab_cap = 256 * (6 + a->C) + 8;
    const size_t bytes = (size_t)(*span_cap + *lab_cap + 8) * 4;
    return bytes <= 96 * 1024 ? bytes : 0;
}
template <bool GRAD>
static bool loss_launch_staged(const pq_yolo_loss_args* a, int nblk, hipStream_t st) {
    int span_cap = 0, lab_cap = 0;
    const size_t bytes = loss_staged_lds(a, GRAD, &span_cap, &lab_cap);
    if (!bytes) return false;
    static size_t raised = 0;
    if (bytes > raised) {
        if (hipFuncSetAttribute((const void*)yolo_loss_staged_kernel<GRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return false;
        raised = 96 * 1024;
    }
    PQ_LAUNCH((yolo_loss_staged_kernel<GRAD>), dim3(nblk, a->B), dim3(256), bytes, st, *a, span_cap, lab_cap);
    return true;
}

extern "C" int pq_yolo_loss_fwd(const pq_yolo_loss_args* a, void* stream) {
    PQ_CHECK_ARG(loss_check(a, false), "bad arguments");
    PQ_CHECK_ARG(a->totals == nullptr || (a->n_heads > 0 && a->head_slot >= 0 && a->head_slot < a->n_heads), "totals: bad head slot");
    const int nblk = pq_yolo_loss_blocks(a->H, a->W, a->A);
    hipStream_t st = (hipStream_t)stream;
    if (a->phase != 2) {
        if (!loss_launch_staged<false>(a, nblk, st)) PQ_LAUNCH((yolo_loss_kernel<false>), dim3(nblk, a->B), dim3(256), 0, st, *a);
        PQ_CHECK_LAUNCH();
    }
    if (a->phase == 1) return PQ_OK;
    PQ_LAUNCH(yolo_loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)a->partial, a->B, nblk, a->losses, a->nan_flag,
                       a->totals, a->head_slot, a->n_heads);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_yolo_loss_bwd(const pq_yolo_loss_args* a, void* stream) {
    PQ_CHECK_ARG(loss_check(a, true), "bad arguments");
    if (a->also[0]) {          // several heads: one launch when every head fits the staged kernel, else one launch per head
        const pq_yolo_loss_args* hd[3] = {a, a->also[0], a->also[1]};
        const int nh = a->also[1] ? 3 : 2;
        LossMulti m{};
        size_t bytes = 0;
        int blk[3] = {0, 0, 0};
        bool fused = true;
        for (int h = 0; h < nh; ++h) {
            PQ_CHECK_ARG(loss_check(hd[h], true) && hd[h]->B == a->B, "also[]: bad arguments / different batch");
            const size_t by = loss_staged_lds(hd[h], true, &m.span_cap[h], &m.lab_cap[h]);
            fused = fused && by != 0;
            bytes = by > bytes ? by : bytes;
            blk[h] = pq_yolo_loss_blocks(hd[h]->H, hd[h]->W, hd[h]->A);
        }
        if (fused) {
            static bool raised = false;
            if (!raised) {
                if (hipFuncSetAttribute((const void*)yolo_loss_bwd_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) fused = false;
                raised = fused;
            }
        }
        if (fused) {
            m.a0 = *hd[0]; m.a1 = *hd[1]; m.a2 = *hd[nh - 1];
            m.blk1 = blk[0]; m.blk2 = blk[0] + blk[1];
            const int total = blk[0] + blk[1] + (nh == 3 ? blk[2] : 0);
            PQ_LAUNCH(yolo_loss_bwd_multi_kernel, dim3(total, a->B), dim3(256), bytes, (hipStream_t)stream, m);
            PQ_CHECK_LAUNCH();
            return PQ_OK;
        }
        for (int h = 1; h < nh; ++h) {
            pq_yolo_loss_args one = *hd[h];
            one.also[0] = one.also[1] = nullptr;
            const int rc = pq_yolo_loss_bwd(&one, stream);
            if (rc != PQ_OK) return rc;
        }
    }
    const int nblk = pq_yolo_loss_blocks(a->H, a->W, a->A);
    if (!loss_launch_staged<true>(a, nblk, (hipStream_t)stream))
        PQ_LAUNCH((yolo_loss_kernel<true>), dim3(nblk, a->B), dim3(256), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_decode(const pq_decode_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->raw.base && a->out, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->H > 0 && a->W > 0 && a->A > 0 && a->C > 0, "bad shape");
    PQ_CHECK_ARG(a->row0 >= 0 && a->row0 + a->H * a->W * a->A <= a->rows_total, "rows out of range");
    long total = (long)a->B * a->H * a->W * a->A * (5 + a->C);
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    PQ_LAUNCH(decode_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_decode_bwd(const pq_decode_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->pred && a->dpred && a->draw, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->H > 0 && a->W > 0 && a->A > 0 && a->C > 0, "bad shape");
    long total = (long)a->B * a->H * a->W * a->A * (5 + a->C);
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    PQ_LAUNCH(decode_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_iou_pairs(const float* b1, const float* b2, float* out, int32_t n, int32_t kind, void* stream) {
    PQ_CHECK_ARG(b1 && b2 && out && n > 0, "bad arguments");
    PQ_CHECK_ARG(kind >= PQ_BOX_GIOU && kind <= PQ_BOX_IOU, "kind must be giou/diou/ciou/iou");
    PQ_LAUNCH(iou_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, b1, b2, out, n, kind);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
