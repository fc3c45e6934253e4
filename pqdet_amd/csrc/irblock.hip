// Fused kernels of a MobileNetV2 inverted-residual block (C -> 6C -> 6C -> C'): the EXPANDED tensor never travels through HBM.
//
// forward  (pq_xdw_fwd):  x (I channels) --1x1 expand, MFMA--> BatchNorm + activation --depthwise 3x3 (stride 1|2)--> raw y_dw (E channels)
//                         + its BatchNorm statistics.  Train-mode BatchNorm needs the sums of the expanded tensor over the whole
//                         batch before the first expanded value can be normalised: pq_xstat_fwd derives them from the first and
//                         second moments of the block INPUT (one pass over I channels), see xgram_kernel.
//
// Structure (forward): the walk of dwwalk.hip -- a workgroup walks down a strip of the plane for a chunk of 32 expanded
// channels with a ring of expanded rows in LDS -- but the rows that enter the ring are COMPUTED: the pixels of a step's 8 new
// input rows (16 / 32 owned columns + halo) are expanded by v_mfma_f32_16x16x4_f32 (A = the chunk's weight rows, held in
// registers for the whole walk; B = 16 pixels of x' = act(bn(x)), each lane's fragment loaded straight from global memory; the
// accumulator of a lane is ONE channel quad of ONE pixel: exactly the float4 the ring stores), normalised, activated, zeroed
// where the pixel is padding, and written to the ring.  The depthwise stage then runs as in the walking kernels.
// Bytes per pixel: I read (+ halo, once per channel chunk: L2), E written -- instead of E written + E read by the unfused pair.
//
// mirrors: conv -> BatchNorm -> ReLU6 triples of an inverted-residual block, model/parser.py:389-410 (mobilenetv2-fpn.cfg).
#include <type_traits>
#include "common.h"

namespace pq {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int XQ = 8;             // channel quads per workgroup (32 expanded channels), as WKQ of dwwalk.hip
constexpr int XPQ = 8;            // ring pixel pitch in quads

struct XdwP {
    pq_src in; const float* we; pq_src mid; float* yexp; int ldy;
    const float* wd; float* out; float* partial;
    int ldo, ldp;
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;
    int B, IH, IW, OH, OW, I, E, E4;
    int strips, segs, seg_rows;
};

// Geometry of a walker.  A step adds TH input rows; the MFMA tiles of a step are ROW ALIGNED -- 16 consecutive pixels of one row
// (row, ring slot, row pointer and the row's padding test are wave-uniform, the per-lane part of every address is a constant
// of the walk) -- plus ONE tile that gathers the halo columns of the step's rows.  Wave w expands the 16-channel half w & 1 of
// the tiles (w >> 1) + 2i: the tile numbers are compile-time constants inside the two instantiations of the expand stage.
template <int S> struct XGeom;
template <> struct XGeom<1> {     // 16 owned columns, ring columns 0 .. 17 (halo left / right); 8 rows per step, 4 outputs per lane
    static constexpr int SW = 16, TH = 8, RX = 2, R = 10, RT = 18, NO = 4, RPX = 18, WR = 6;
    static constexpr int TPR = 1;                               // row tiles per row
    static __device__ __forceinline__ int px(int slot, int rc) { return slot * RPX + rc; }
    // halo tile: lane -> (row l15 >> 1, ring column 0 | 17)
    static __device__ __forceinline__ int halo_row(int l15) { return l15 >> 1; }
    static __device__ __forceinline__ int halo_col(int l15) { return (l15 & 1) * 17; }
};
template <> struct XGeom<2> {     // 32 owned input columns (16 outputs), ring columns 0 .. 32 (halo left); 4 input rows per step (2 output
    static constexpr int SW = 32, TH = 4, RX = 1, R = 5, RT = 9, NO = 1, RPX = 34, WR = 3;      // rows), 1 output per lane; even / odd ring
    static constexpr int TPR = 2;                               // columns in two planes, so that lanes of adjacent outputs read adjacent pixels
    static __device__ __forceinline__ int px(int slot, int rc) { return slot * RPX + (rc & 1) * 17 + (rc >> 1); }
    static __device__ __forceinline__ int halo_row(int l15) { return l15 & 3; }
    static __device__ __forceinline__ int halo_col(int) { return 0; }
};

template <int KS> __device__ __forceinline__ void load_frag(const float* p, float (&f)[KS]) {
    if constexpr (KS % 4 == 0) {
#pragma unroll
        for (int k = 0; k < KS; k += 4) { const float4 v = ldg4(p + k); f[k] = v.x; f[k + 1] = v.y; f[k + 2] = v.z; f[k + 3] = v.w; }
    } else {
#pragma unroll
        for (int k = 0; k < KS; k += 2) { const float2 v = *reinterpret_cast<const float2*>(p + k); f[k] = v.x; f[k + 1] = v.y; }
    }
}
template <int N> using ic = std::integral_constant<int, N>;
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(ic<N - 1>{}); }
}

// KS = I / 4 (k-steps of the 16x16x4 MFMA; lane group g = lane >> 4 contracts over input channels g * KS .. g * KS + KS - 1: the
// B fragment of a lane is KS CONSECUTIVE floats of one pixel of x, loaded straight from global memory -- no staging of x).
// The input operand is LINEAR (every expand layer of MobileNetV2 reads a linear bottleneck or a materialised sum): x' = a x + b is
// one fma per fragment element; the backward (irbwd.hip) recomputes y_e with the same fma, the same MFMA and the same k order, so
// both passes see bit-identical expanded values (the ReLU6 masks of the two passes agree exactly).
// One barrier per step: while the depthwise stage consumes the ring rows of step k, the expand stage fills those of step k + 1
// (the ring holds R + TH rows), and the fragments of step k + 2 are in flight; expand and depthwise code of a step form ONE
// basic block (tile numbers are compile-time constants, every load is unconditional at a clamped address), so the MFMA chains
// of the five tiles and the depthwise stage's LDS reads / FMAs interleave.
// YE: also store the raw expanded tensor (a template flavour: the values to store stay live across the depthwise stage, 20 registers
// the plain kernel does not have to keep; the 16 -> 96 stride-2 instantiation then fits three workgroups per CU = ONE round)
template <int S, int KS, bool YE>
__global__ __launch_bounds__(256, (S == 2 && !YE) ? 3 : 2) void xdw_fwd_kernel(XdwP p) {
    using G = XGeom<S>;
    constexpr int I = 4 * KS, TH = G::TH;
    constexpr int NT = TH * G::TPR + 1;                         // tiles of a step (the last one gathers the halo columns)
    constexpr int NTW = (NT + 1) / 2;                           // ... per wave
    constexpr int NT0 = G::RX * G::TPR + 1, NTW0 = (NT0 + 1) / 2;   // the same for the RX rows above the first step
    constexpr int RING_F4 = G::RT * G::RPX * XPQ;
    static_assert(RING_F4 * 4 >= 2048 + 64, "ring doubles as the coefficient staging buffer");
    __shared__ float4 ring[RING_F4];
    __shared__ float4 wtab[XQ][9];
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;                           // depthwise stage: channel quad, pixel lane
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, g = lane >> 4, sub = w & 1, wh = w >> 1;      // expand stage: pixel of the tile, k group, 16-channel half, tile parity
    const int qm = 4 * sub + g;                                 // ... and the quad this lane's accumulator holds
    int tIdx, bcy;
    xcd_block(bcy, tIdx);                                       // grid = (channel chunks, walkers)
    const int strip = tIdx % p.strips, seg = (tIdx / p.strips) % p.segs, b = tIdx / (p.strips * p.segs);
    const int ys = seg * p.seg_rows, x0 = strip * G::SW;        // input rows / columns
    const int rows_here = min(p.seg_rows, p.IH - ys);
    const int nsteps = (rows_here + TH - 1) / TH;
    const int yend = ys + rows_here;
    const int cq0 = bcy * XQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.E4;
    const int cm = (cq0 + qm) * 4;                              // first channel of the expand-stage quad
    const bool cmok = cq0 + qm < p.E4;

    // ---- tables: depthwise taps (wave 1), the expand weights of this lane with the input operand's affine folded in (registers)
    float e[9][4];
    if (pl == 8) {
        const int cc = cok ? c0 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wp = p.wd + (size_t)min(cc + j, p.E - 1) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k][j] = wp[k];
        }
    }
    float wa[KS], xa[KS], xb[KS];
    {
        const int ch = min(cq0 * 4 + 16 * sub + l15, p.E - 1);
        load_frag<KS>(p.we + (size_t)ch * I + g * KS, wa);
    }
    const bool xaff = p.in.scale != nullptr;                    // uniform
#pragma unroll
    for (int k = 0; k < KS; ++k) { xa[k] = xaff ? p.in.scale[g * KS + k] : 1.f; xb[k] = xaff ? p.in.shift[g * KS + k] : 0.f; }
    const SrcCoef4 mk = p.mid.stat.acc ? src_coef4_chunk32(p.mid, cq0 * 4, qm, p.E, tIdx == 0, reinterpret_cast<float*>(ring))
                                       : src_coef4(p.mid, cmok ? cm : 0, p.E);
    if (pl == 8) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[k][j] = (cok && c0 + j < p.E) ? e[k][j] : 0.f;
            wtab[q][k] = make_float4(e[k][0], e[k][1], e[k][2], e[k][3]);
        }
    }
    // the expanded tensor's activation is linear / ReLU / ReLU6 (host-checked): a clamp; a padding pixel clamps to [0, 0]
    const float mlo = p.mid.act == PQ_ACT_LINEAR ? -__builtin_huge_valf() : 0.f, mhi = p.mid.act == PQ_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    const size_t img = (size_t)b * p.IH * p.IW;
    // per-lane constants of the walk: element offset of the lane's fragment inside a plane row and the column's padding test, for
    // the row tiles (one per tile of a row) and for the halo tile (whose row differs per lane)
    int offr[G::TPR]; bool colok[G::TPR];
#pragma unroll
    for (int j = 0; j < G::TPR; ++j) {
        const int ix = x0 + 16 * j + l15;
        offr[j] = min(ix, p.IW - 1) * p.in.ld + g * KS;
        colok[j] = ix < p.IW;
    }
    const int hrow = G::halo_row(l15), hcol = G::halo_col(l15);
    const int hix = x0 - 1 + hcol;
    const int offh = min(max(hix, 0), p.IW - 1) * p.in.ld + g * KS;
    const bool hcolok = (unsigned)hix < (unsigned)p.IW;
    __syncthreads();

    // ---- TILE (compile-time) of `NR` rows from ring-local row l0 (plane row ys - 1 + l0): fragment address (always valid) ...
    auto frag_ptr = [&](int l0, auto nr, auto tile) -> const float* {
        constexpr int NR = decltype(nr)::value, T = decltype(tile)::value;
        if constexpr (T < NR * G::TPR) {
            const int iy = min(max(ys - 1 + l0 + T / G::TPR, 0), p.IH - 1);                  // uniform
            return p.in.x + (img + (size_t)iy * p.IW) * p.in.ld + offr[T % G::TPR];
        } else {
            const int iy = min(max(ys - 1 + l0 + hrow, 0), p.IH - 1);
            return p.in.x + (img + (size_t)iy * p.IW) * p.in.ld + offh;
        }
    };
    // ---- ... and the tile itself: KS MFMAs of the fragment against the lane's weights -> BatchNorm + activation -> ring; returns the raw values
    auto expand_tile = [&](int l0, auto nr, auto tile, const float (&f)[KS]) -> f32x4 {
        constexpr int NR = decltype(nr)::value, T = decltype(tile)::value;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k) acc = mfma16(wa[k], fmaf(xa[k], f[k], xb[k]), acc);
        int lr, lc; bool on, in_img;
        if constexpr (T < NR * G::TPR) {
            lr = T / G::TPR; lc = 1 + 16 * (T % G::TPR) + l15; on = true;
            in_img = colok[T % G::TPR] && (unsigned)(ys - 1 + l0 + lr) < (unsigned)p.IH;
        } else {
            lr = hrow; lc = hcol; on = S == 1 ? hrow < NR : l15 < NR;
            in_img = on && hcolok && (unsigned)(ys - 1 + l0 + lr) < (unsigned)p.IH;
        }
        const float lo = in_img ? mlo : 0.f, hi = in_img ? mhi : 0.f;
        float4 o;
        o.x = fminf(fmaxf(fmaf(mk.a[0], acc[0], mk.b[0]), lo), hi);
        o.y = fminf(fmaxf(fmaf(mk.a[1], acc[1], mk.b[1]), lo), hi);
        o.z = fminf(fmaxf(fmaf(mk.a[2], acc[2], mk.b[2]), lo), hi);
        o.w = fminf(fmaxf(fmaf(mk.a[3], acc[3], mk.b[3]), lo), hi);
        int slot = l0 + lr;
        slot -= (slot / G::RT) * G::RT;
        if (on) ring[G::px(slot, lc) * XPQ + qm] = o;
        return acc;
    };
    // the raw expanded value of an OWNED pixel (tests; a backward that reads the tensor): each pixel has exactly one owner
    auto store_yexp = [&](int l0, auto nr, auto tile, f32x4 acc) {
        constexpr int NR = decltype(nr)::value, T = decltype(tile)::value;
        if constexpr (T < NR * G::TPR) {
            const int iy = ys - 1 + l0 + T / G::TPR, ix = x0 + 16 * (T % G::TPR) + l15;
            if (cmok && colok[T % G::TPR] && iy >= ys && iy < yend)
                stg4(p.yexp + (img + (size_t)iy * p.IW + ix) * p.ldy + cm, make_float4(acc[0], acc[1], acc[2], acc[3]));
        }
    };

    float xf[NTW][KS];
    // ---- prologue: the halo rows above the first step and the first step's rows
    auto prologue = [&](auto whc) {
        constexpr int WH = decltype(whc)::value;
        float x0f[NTW0][KS];
        f32x4 y0[NTW0], y1[NTW];
        static_for<NTW0>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
            if constexpr (T < NT0) load_frag<KS>(frag_ptr(0, ic<G::RX>{}, ic<T>{}), x0f[decltype(i)::value]); });
        static_for<NTW>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
            if constexpr (T < NT) load_frag<KS>(frag_ptr(G::RX, ic<TH>{}, ic<T>{}), xf[decltype(i)::value]); });
        static_for<NTW0>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
            if constexpr (T < NT0) y0[decltype(i)::value] = expand_tile(0, ic<G::RX>{}, ic<T>{}, x0f[decltype(i)::value]); });
        static_for<NTW>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
            if constexpr (T < NT) y1[decltype(i)::value] = expand_tile(G::RX, ic<TH>{}, ic<T>{}, xf[decltype(i)::value]); });
        static_for<NTW>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;       // (past the segment: clamped rows, never used)
            if constexpr (T < NT) load_frag<KS>(frag_ptr(G::RX + TH, ic<TH>{}, ic<T>{}), xf[decltype(i)::value]); });
        if constexpr (YE) {
            static_for<NTW0>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
                if constexpr (T < NT0) store_yexp(0, ic<G::RX>{}, ic<T>{}, y0[decltype(i)::value]); });
            static_for<NTW>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
                if constexpr (T < NT) store_yexp(G::RX, ic<TH>{}, ic<T>{}, y1[decltype(i)::value]); });
        }
    };
    float4 s1 = f4zero(), s2 = f4zero();
    const int col = pl & 15, rp = pl >> 4;
    const int oyend = yend / S;
    const size_t oimg = (size_t)b * p.OH * p.OW;
    // ---- one step: [the rows of step k + 1 (fragments loaded one iteration ago; those of step k + 2 are requested as they are
    // consumed)] + the depthwise stage of step k
    auto step = [&](auto whc, auto exc, int k) {
        constexpr int WH = decltype(whc)::value;
        constexpr bool EX = decltype(exc)::value != 0;
        f32x4 yv[NTW];
        const int l1 = G::RX + TH * (k + 1);
        if constexpr (EX) {
            static_for<NTW>([&](auto i) {
                constexpr int II = decltype(i)::value, T = WH + 2 * II;
                if constexpr (T < NT) {
                    float f[KS];
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) f[kk] = xf[II][kk];
                    load_frag<KS>(frag_ptr(l1 + TH, ic<TH>{}, ic<T>{}), xf[II]);
                    yv[II] = expand_tile(l1, ic<TH>{}, ic<T>{}, f);
                }
            });
        }
        // depthwise: lane -> output column col, output rows NO * rp + u of the step; window row wr = S * u + ky
        float4 acc[G::NO];
#pragma unroll
        for (int u = 0; u < G::NO; ++u) acc[u] = f4zero();
        const int kb = (TH * k) % G::RT;                         // uniform: ring slot of the step's first window row
        const int base = kb + S * G::NO * rp;                    // the lane's first window row: < 2 RT, one conditional subtraction
#pragma unroll
        for (int wr = 0; wr < G::WR; ++wr) {
            int slot = base + wr;
            slot = slot >= G::RT ? slot - G::RT : slot;
            float4 xr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xr[j] = ring[G::px(slot, S * col + j) * XPQ + q];
#pragma unroll
            for (int u = 0; u < G::NO; ++u) {
                const int ky = wr - S * u;
                if (ky >= 0 && ky <= 2) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[u] = f4fma(xr[j], wtab[q][ky * 3 + j], acc[u]);
                }
            }
        }
        const int ox = x0 / S + col;
#pragma unroll
        for (int u = 0; u < G::NO; ++u) {
            const int oy = (ys + TH * k) / S + G::NO * rp + u;
            if (cok && oy < oyend && ox < p.OW) {
                stg4(p.out + (oimg + (size_t)oy * p.OW + ox) * p.ldo + c0, acc[u]);
                s1 = f4add(s1, acc[u]);
                s2 = f4fma(acc[u], acc[u], s2);
            }
        }
        if constexpr (EX && YE) {
                static_for<NTW>([&](auto i) { constexpr int T = WH + 2 * decltype(i)::value;
                    if constexpr (T < NT) store_yexp(l1, ic<TH>{}, ic<T>{}, yv[decltype(i)::value]); });
        }
    };
    if (wh == 0) {
        prologue(ic<0>{});
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k + 1 < nsteps; ++k) { step(ic<0>{}, ic<1>{}, k); __syncthreads(); }
        step(ic<0>{}, ic<0>{}, nsteps - 1);
    } else {
        prologue(ic<1>{});
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k + 1 < nsteps; ++k) { step(ic<1>{}, ic<1>{}, k); __syncthreads(); }
        step(ic<1>{}, ic<0>{}, nsteps - 1);
    }
    if (p.partial || p.stat_acc) {      // uniform
        __syncthreads();
        float* red = reinterpret_cast<float*>(ring);
        float* slot = red + (pl * 8 + q) * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < 64) {     // thread -> (quad t >> 3, component t & 7)
            const int qq = t >> 3, comp = t & 7;
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) a += red[(l * 8 + qq) * 8 + comp];
            const int c = (cq0 + qq) * 4 + (comp & 3);
            if (c < p.E) {
                if (p.stat_acc) bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, p.ldp, tIdx, comp >> 2, c, a);
                else p.partial[(size_t)tIdx * 2 * p.ldp + (comp >> 2) * p.ldp + c] = a;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ statistics of the expanded tensor
// Train-mode BatchNorm of y = W x' needs sum y and sum y^2 per expanded channel over the whole batch BEFORE pq_xdw_fwd can
// normalise its first value.  Both follow from the first and second moments of the block INPUT:  sum y_c = w_c . s,
// sum y_c^2 = w_c^T G w_c  with  s = sum x', G = sum x' x'^T  (I x I, I <= 32): one pass over the I-channel tensor with the pixels
// as the contraction of an MFMA whose A and B operands are the SAME register (G = X^T X), instead of a pass of the expand GEMM
// (I x E multiply-adds per pixel and an E-wide statistics epilogue for values that are thrown away).  Workgroup partials are ADDED
// into exact fixed-point accumulators (bn_stat_add's image: integer atomics, order independent), XSLOTS replicas against
// same-address contention; the finalize kernel reads them, re-arms them (zero) and derives the vectors.
constexpr int XSLOTS = 16;
struct XgramP { pq_src in; int64_t* acc; uint32_t* bad; int M, rows_per_wg; };

template <int I>
__global__ __launch_bounds__(256) void xgram_kernel(XgramP p) {
    constexpr bool W16 = I <= 16;
    constexpr int NL = W16 ? 16 : 32;          // channel lanes
    constexpr int KP = 64 / NL;                // pixels per MFMA (the contraction)
    constexpr int U = 8;
    constexpr int NACC = W16 ? 4 : 16;
    constexpr int N = I * I + I;
    __shared__ float gl[4][NL * NL];
    __shared__ float sl[4][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & (NL - 1), kh = lane / NL;
    const bool iok = i < I;
    const int ic_ = iok ? i : 0;
    float ca = (p.in.scale && iok) ? p.in.scale[ic_] : 1.f, cb = (p.in.scale && iok) ? p.in.shift[ic_] : 0.f;
    if (p.in.stat.acc != nullptr) {            // uniform: the operand's BatchNorm handed over by its producer (pq_bn_stat): every lane derives
        ca = 1.f; cb = 0.f;                    // the coefficients of ITS channel; one lane per channel of workgroup 0 stores the vectors
        bn_stat_coefs<1>(p.in.stat, ic_, 1, I, blockIdx.x == 0 && w == 0 && kh == 0 && iok,
                         [&](int, const BnCoef& bc) { if (iok) { ca = bc.scale; cb = bc.shift; } });
    }
    const ActParam act = act_param(p.in.act);
    const int m0 = blockIdx.x * p.rows_per_wg, m1 = min(m0 + p.rows_per_wg, p.M);
    float acc[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] = 0.f;
    float sum = 0.f;
    for (int base = m0 + w * (KP * U); base < m1; base += 4 * KP * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = base + KP * u + kh;
            v[u] = p.in.x[(size_t)min(m, p.M - 1) * p.in.ld + ic_];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = base + KP * u + kh;
            float x = act_fwd(fmaf(ca, v[u], cb), act);
            x = (iok && m < m1) ? x : 0.f;
            sum += x;
            if constexpr (W16) {
                f32x4 a = {acc[0], acc[1], acc[2], acc[3]};
                a = mfma16(x, x, a);
                acc[0] = a[0]; acc[1] = a[1]; acc[2] = a[2]; acc[3] = a[3];
            } else {
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = acc[r];
                a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = a[r];
            }
        }
    }
    // accumulator layout: column = lane & (NL - 1); 16 lanes: row = 4 * (lane >> 4) + r; 32 lanes: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
        const int row = W16 ? 4 * kh + r : (r & 3) + 8 * (r >> 2) + 4 * kh;
        gl[w][row * NL + i] = acc[r];
    }
    sl[w][lane] = sum;
    __syncthreads();
    for (int el = t; el < N; el += 256) {
        float a = 0.f;
        if (el < I * I) {
            const int r = el / I, c = el - r * I;
            if (r > c) continue;                               // G is symmetric: the finalize kernel mirrors the upper triangle
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) a += gl[ww][r * NL + c];
        } else {
            const int c = el - I * I;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                for (int k = 0; k < KP; ++k) a += sl[ww][k * NL + c];
        }
        // element el of replica blockIdx.x % XSLOTS: the accumulator is a [XSLOTS][1][N][2] image for bn_stat_add (plane 0, pitch N)
        bn_stat_add(p.acc, p.bad, XSLOTS, N, blockIdx.x, 0, el, a);
    }
}

struct XstatFinP {
    int64_t* acc; uint32_t* bad; double* moments;
    const float* we; const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* nbt;
    float* mean; float* invstd; float* scale; float* shift;
    int I, E; float count, eps, momentum;
};
// ONE workgroup: sums the replicas (exact), re-arms them, then 4 threads per expanded channel evaluate w . s and w^T G w in fp64
__global__ __launch_bounds__(1024) void xstat_finalize_kernel(XstatFinP p) {
    __shared__ double G[32 * 32 + 32];
    __shared__ unsigned s_bad;
    const int t = threadIdx.x;
    const int I = p.I, n = I * I + I;
    if (t == 0) s_bad = *p.bad;
    for (int el = t; el < n; el += 1024) {
        longlong2 v[XSLOTS];
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) v[sl] = *reinterpret_cast<const longlong2*>(p.acc + (((size_t)sl * 2 + 0) * n + el) * 2);
        long long lo = 0, hi = 0;
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) { lo += v[sl].x; hi += v[sl].y; }
        G[el] = (double)hi + (double)lo * 0x1p-50;
    }
    __syncthreads();
    for (int el = t; el < I * I; el += 1024) {                  // mirror (the producers only add the upper triangle)
        const int r = el / I, c = el - r * I;
        if (r > c) G[el] = G[c * I + r];
    }
    __syncthreads();
    if (p.moments)
        for (int el = t; el < n; el += 1024) p.moments[el] = G[el];            // kept for the backward (pq_xdw_bwd)
    const bool bad = s_bad != 0;
    for (int el = t; el < n; el += 1024) {                      // re-arm: the next call finds zeros
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) *reinterpret_cast<longlong2*>(p.acc + (((size_t)sl * 2 + 0) * n + el) * 2) = longlong2{0, 0};
    }
    if (t == 0) *p.bad = 0u;
    for (int cb = 0; cb < p.E; cb += 256) {
        const int c = cb + (t >> 2), part = t & 3;
        const int cc = min(c, p.E - 1);
        const float* w = p.we + (size_t)cc * I;
        float wr[32];                                           // the channel's weight row, loaded once
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            const float4 v = j < I ? ldg4(w + j) : f4zero();
            wr[j] = v.x; wr[j + 1] = v.y; wr[j + 2] = v.z; wr[j + 3] = v.w;
        }
        double q = 0.0, m = 0.0;
#pragma unroll
        for (int i4 = 0; i4 < 8; ++i4) {
            const int i = part + 4 * i4;
            if (i < I) {
                double r = 0.0;
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (j < I) r = fma(G[i * I + j], (double)wr[j], r);
                const double wi = (double)w[i];
                q = fma(wi, r, q);
                m = fma(wi, G[I * I + i], m);
            }
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
        m += __shfl_xor(m, 1); m += __shfl_xor(m, 2);
        if (part == 0 && c < p.E) {
            const double cnt = (double)p.count;
            double mean = m / cnt;
            double var = q / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            if (bad) mean = var = __builtin_nan("");
            const float inv = (float)(1.0 / sqrt(var + (double)p.eps));
            const float sc = p.gamma[c] * inv;
            p.mean[c] = (float)mean; p.invstd[c] = inv; p.scale[c] = sc; p.shift[c] = p.beta[c] - (float)mean * sc;
            if (p.running_mean) {
                const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
                p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unb;
            }
            if (c == 0 && p.nbt) *p.nbt += 1;
        }
    }
}

}  // namespace pq

using namespace pq;

struct XdwPlan { int strips, segs, seg_rows, nblk, chunks; };
static XdwPlan xdw_plan(int B, int IH, int IW, int E, int stride) {
    XdwPlan w{};
    const int sw = stride == 1 ? 16 : 32;
    w.strips = (IW + sw - 1) / sw;
    w.chunks = ((E + 3) / 4 + XQ - 1) / XQ;
    const int th = stride == 1 ? 8 : 4;
    const int steps_total = (IH + th - 1) / th;
    const long base = (long)B * w.strips * w.chunks;
    // row segments per image column: the choice that costs the fewest (rounds of resident workgroups) x (steps per workgroup + ~2
    // steps of prologue); 3 workgroups per CU at stride 2 (registers), 2 at stride 1
    const long slots = 256L * (stride == 2 ? 3 : 2);
    const long maxsegs = steps_total / 2 > 0 ? steps_total / 2 : 1;
    long segs = 1, best = -1;
    for (long sgs = 1; sgs <= 4 && sgs <= maxsegs; ++sgs) {
        const long ss = (steps_total + sgs - 1) / sgs, real = (steps_total + ss - 1) / ss;
        const long cost = ((base * real + slots - 1) / slots) * (ss + 2);
        if (best < 0 || cost < best) { best = cost; segs = sgs; }
    }
    const int seg_steps = (int)((steps_total + segs - 1) / segs);
    w.segs = (steps_total + seg_steps - 1) / seg_steps;
    w.seg_rows = seg_steps * th;
    w.nblk = B * w.strips * w.segs;
    return w;
}

static bool xdw_shape_ok(int I, int E, int stride, int IH, int IW) {
    if (!(I == 16 || I == 24) || E < 4) return false;       // (32 -> 192 on the 64 x 64 planes measured at break-even: not built)
    if (stride == 2) return (IH % 2 == 0) && (IW % 2 == 0);
    return stride == 1;
}

extern "C" int32_t pq_xdw_fwd_blocks(int32_t B, int32_t IH, int32_t IW, int32_t I, int32_t E, int32_t stride) {
    if (!xdw_shape_ok(I, E, stride, IH, IW)) return 0;
    return xdw_plan(B, IH, IW, E, stride).nblk;
}

extern "C" int pq_xdw_fwd(const pq_xdw_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->we && a->wd && a->out, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->IH > 0 && a->IW > 0 && a->I > 0 && a->E > 0, "empty shape");
    PQ_CHECK_ARG(a->stride == 1 || a->stride == 2, "stride must be 1 or 2");
    PQ_CHECK_ARG(a->OH == (a->IH - 1) / a->stride + 1 && a->OW == (a->IW - 1) / a->stride + 1, "output size does not match pad 1");
    PQ_CHECK_ARG((a->in.ld & 3) == 0 && a->in.ld >= a->I && (((uintptr_t)a->in.x) & 15) == 0, "input pitch / pointer must be 16B aligned");
    PQ_CHECK_ARG((a->ldo & 3) == 0 && a->ldo >= a->E && (((uintptr_t)a->out) & 15) == 0, "output pitch / pointer must be 16B aligned");
    PQ_CHECK_ARG(a->yexp == nullptr || ((a->ldy & 3) == 0 && a->ldy >= a->E && (((uintptr_t)a->yexp) & 15) == 0), "yexp pitch / pointer");
    PQ_CHECK_ARG(a->partial == nullptr || a->ldp >= a->E, "stats pitch too small");
    PQ_CHECK_ARG(a->stat_acc == nullptr || (a->stat_bad != nullptr && a->ldp >= a->E), "stat_acc needs stat_bad");
    PQ_CHECK_ARG(a->mid.stat.acc != nullptr || a->mid.scale != nullptr, "mid needs scale / shift or statistics accumulators");
    if (a->in.act != PQ_ACT_LINEAR || act_is_special(a->mid.act) || a->mid.act == PQ_ACT_LEAKY || a->in.qstate || a->mid.qstate || a->in.stat.acc) {
        pq_set_error("pq_xdw_fwd: the input operand must be linear (BatchNorm affine only); leaky / logistic / tanh on the expanded tensor, QAT "
                     "observers and in.stat are not supported");
        return PQ_EUNSUPPORTED;
    }
    if (!xdw_shape_ok(a->I, a->E, a->stride, a->IH, a->IW)) {
        pq_set_error("pq_xdw_fwd: I must be 16 / 24 (stride 2: even plane sizes); ask pq_xdw_fwd_blocks first");
        return PQ_EUNSUPPORTED;
    }
    const XdwPlan w = xdw_plan(a->B, a->IH, a->IW, a->E, a->stride);
    if (a->stat_acc && !stat_adds_ok(w.nblk, a->stat_slots)) {
        pq_set_error("pq_xdw_fwd: more than %d row blocks per accumulator slot", PQ_STAT_MAX_ADDS);
        return PQ_EUNSUPPORTED;
    }
    PQ_CHECK_ARG(w.nblk <= 65535, "too many walkers for one launch");
    XdwP p{};
    p.in = a->in; p.we = a->we; p.mid = a->mid; p.yexp = a->yexp; p.ldy = a->ldy; p.wd = a->wd; p.out = a->out; p.partial = a->partial;
    p.ldo = a->ldo; p.ldp = a->ldp;
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.I = a->I; p.E = a->E; p.E4 = (a->E + 3) / 4;
    p.strips = w.strips; p.segs = w.segs; p.seg_rows = w.seg_rows;
    const dim3 grid(w.chunks, w.nblk), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const int key = a->stride * 100 + a->I;
    switch (key) {
        case 116: if (a->yexp) PQ_LAUNCH((xdw_fwd_kernel<1, 4, true>), grid, blk, 0, st, p); else PQ_LAUNCH((xdw_fwd_kernel<1, 4, false>), grid, blk, 0, st, p); break;
        case 124: if (a->yexp) PQ_LAUNCH((xdw_fwd_kernel<1, 6, true>), grid, blk, 0, st, p); else PQ_LAUNCH((xdw_fwd_kernel<1, 6, false>), grid, blk, 0, st, p); break;
        case 216: if (a->yexp) PQ_LAUNCH((xdw_fwd_kernel<2, 4, true>), grid, blk, 0, st, p); else PQ_LAUNCH((xdw_fwd_kernel<2, 4, false>), grid, blk, 0, st, p); break;
        default: if (a->yexp) PQ_LAUNCH((xdw_fwd_kernel<2, 6, true>), grid, blk, 0, st, p); else PQ_LAUNCH((xdw_fwd_kernel<2, 6, false>), grid, blk, 0, st, p); break;
    }
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

// ---- statistics of the expanded tensor from the moments of the block input
static void xstat_plan(int M, int I, int& rows, int& nwg) {
    const int unit = I <= 16 ? 128 : 64;                       // pixels one workgroup iteration covers (4 waves x 8 MFMAs)
    rows = ((M + 255) / 256 + unit - 1) / unit * unit;          // at most 256 workgroups add into the accumulators
    if (rows < unit) rows = unit;
    nwg = (M + rows - 1) / rows;
}
extern "C" int64_t pq_xstat_fwd_scratch_bytes(int32_t M, int32_t I) {
    if (M <= 0 || !(I == 16 || I == 24)) return 0;
    return (int64_t)XSLOTS * 2 * (I * I + I) * 2 * 8 + 64;      // [XSLOTS][2][N][2] int64 (plane 1 unused) + the `bad` word
}
extern "C" int pq_xstat_fwd(const pq_xstat_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->we && a->gamma && a->beta && a->mean && a->invstd && a->scale && a->shift && a->scratch, "null pointer");
    PQ_CHECK_ARG(a->M > 0 && a->E > 0 && a->in.ld >= a->I && (((uintptr_t)a->scratch) & 15) == 0, "bad shape / scratch alignment");
    if (!(a->I == 16 || a->I == 24) || act_is_special(a->in.act) || a->in.qstate) {
        pq_set_error("pq_xstat_fwd: I must be 16 / 24; logistic / tanh operands and QAT observers are not supported");
        return PQ_EUNSUPPORTED;
    }
    int rows, nwg;
    xstat_plan(a->M, a->I, rows, nwg);
    if (!stat_adds_ok(nwg, XSLOTS)) { pq_set_error("pq_xstat_fwd: too many partials per accumulator slot"); return PQ_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const int n = a->I * a->I + a->I;
    int64_t* acc = reinterpret_cast<int64_t*>(a->scratch);
    uint32_t* bad = reinterpret_cast<uint32_t*>(acc + (size_t)XSLOTS * 2 * n * 2);
    XgramP g{a->in, acc, bad, a->M, rows};
    if (a->I == 16) PQ_LAUNCH((xgram_kernel<16>), dim3(nwg), dim3(256), 0, st, g);
    else PQ_LAUNCH((xgram_kernel<24>), dim3(nwg), dim3(256), 0, st, g);
    XstatFinP f{acc, bad, a->moments, a->we, a->gamma, a->beta, a->running_mean, a->running_var, a->num_batches_tracked,
                a->mean, a->invstd, a->scale, a->shift, a->I, a->E, (float)a->M, a->eps, a->momentum};
    PQ_LAUNCH(xstat_finalize_kernel, dim3(1), dim3(1024), 0, st, f);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
