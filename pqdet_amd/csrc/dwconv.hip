// Depthwise 3x3 (pad 1, stride 1|2) and the dense 3x3 network stem, NHWC fp32, gfx950.
//
// HBM-bound kernels.  Thread layout: blockDim = (CX, PY); CX lanes own one channel quad (float4) each so that a
// wave touches CX*16 contiguous bytes per pixel and consecutive pixels are contiguous too; PY walks pixel groups.
// A thread keeps its 9 (x4) weights in registers and slides a 3 x (TW*s+2) input window along W, producing TW
// outputs per step; vertical reuse is left to L1/L2.  Input operands are "virtual": the producer's BatchNorm affine
// and activation are applied while loading (zero padding is applied AFTER the transform, like the reference's
// conv on the already-activated tensor).  BatchNorm statistics / BatchNorm-backward sums are reduced per block.
#include "common.h"

namespace pq {

struct DwP {
    pq_src in;        // forward operand (fwd / wgrad)
    pq_gsrc dy;       // gradient operand (dgrad / wgrad)
    const float* w;   // [C][9]
    float* out; float* partial; int ldo, ldp;
    pq_gsink sink;
    int B, IH, IW, OH, OW, C, C4;
    float* wpartial;  // wgrad: [gridDim.x][C][9]
    int minmax;       // forward: `partial` rows have 4 planes (+ per-channel min, max of the outputs)
};

__device__ __forceinline__ void load_w9(const float* w, int c0, int C, float4 (&wk)[9], bool flip) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (c0 + j < C) ? w[(size_t)(c0 + j) * 9 + (flip ? 8 - t : t)] : 0.f;
        wk[t] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// ---- forward (S = stride).  Also serves as dgrad for stride 1 (GRAD=true: gradient operand, flipped taps, sink).
template <int S, int TW, bool GRAD>
__global__ __launch_bounds__(256) void dw_fwd_kernel(DwP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx;
    const int c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const int OWG = (p.OW + TW - 1) / TW;
    const int items = p.B * p.OH * OWG;
    constexpr int NCOL = (TW - 1) * S + 3;

    float4 wk[9];
    load_w9(p.w, c0, p.C, wk, GRAD);
    const SrcCoef4 xk = src_coef4(p.in, c0, p.C);
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.C);
    const SinkCoef4 sk = sink_coef4(p.sink, c0, p.C);
    float4 s1 = f4zero(), s2 = f4zero();
    const float inf = __builtin_huge_valf();
    float4 mn = make_float4(inf, inf, inf, inf), mx = make_float4(-inf, -inf, -inf, -inf);
    const bool mm = !GRAD && p.minmax != 0;

    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int oxg = (int)(it % OWG);
        const int t2 = it / OWG;
        const int oy = (int)(t2 % p.OH);
        const int b = (int)(t2 / p.OH);
        const int ox0 = oxg * TW;
        float4 acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = f4zero();
        const size_t orow = ((size_t)b * p.OH + oy) * p.OW + ox0;
        SinkOps so[GRAD ? TW : 1];
        if (GRAD) {        // the sink's operands of all TW outputs travel with the first window row
#pragma unroll
            for (int t = 0; t < TW; ++t) so[t] = sink_load4(p.sink, sk, orow + min(t, p.OW - 1 - ox0), c0);
        }
        // Loads are issued unconditionally at clamped coordinates and masked afterwards (branching around each load would serialise
        // them behind their own waits).  Forward: ALL THREE window rows are requested before the first one is used (18 float4 in
        // flight per thread, one exposed round trip per item).  Gradient operand (two tensors per element): one row at a time --
        // 36 float4 in flight cost more in occupancy than the two extra round trips (measured: 1.03 -> 1.05 ms per step).
        constexpr int RB = GRAD ? 1 : 3;                 // window rows per load batch
#pragma unroll
        for (int kb = 0; kb < 3; kb += RB) {
            float4 ra[RB][NCOL], rb[RB][NCOL];
            bool yok[RB];
#pragma unroll
            for (int kk = 0; kk < RB; ++kk) {
                const int iy = oy * S - 1 + kb + kk;
                yok[kk] = iy >= 0 && iy < p.IH;
                const int iyc = min(max(iy, 0), p.IH - 1);
                const size_t rowbase = ((size_t)b * p.IH + iyc) * p.IW;
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    const int ixc = min(max(ox0 * S - 1 + j, 0), p.IW - 1);
                    if (GRAD) {
                        ra[kk][j] = ldg4(p.dy.g + (rowbase + ixc) * p.dy.ldg + c0);
                        rb[kk][j] = gk.two ? ldg4(p.dy.y + (rowbase + ixc) * p.dy.ldy + c0) : f4zero();
                    } else {
                        ra[kk][j] = ldg4(p.in.x + (rowbase + ixc) * p.in.ld + c0);
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < RB; ++kk) {
                const int ky = kb + kk;
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    const int ix = ox0 * S - 1 + j;
                    const bool ok = yok[kk] && ix >= 0 && ix < p.IW;
                    float4 v = GRAD ? gsrc_apply4(ra[kk][j], rb[kk][j], gk) : src_apply4(ra[kk][j], xk);
                    v = ok ? v : f4zero();
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        const int kx = j - t * S;
                        if (kx >= 0 && kx < 3) acc[t] = f4fma(v, wk[ky * 3 + kx], acc[t]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            if (ox0 + t >= p.OW) break;
            if (GRAD) {
                sink_store4(p.sink, sk, orow + t, c0, acc[t], so[t], s1, s2);
            } else {
                stg4(p.out + (orow + t) * p.ldo + c0, acc[t]);
                s1 = f4add(s1, acc[t]);
                s2 = f4fma(acc[t], acc[t], s2);
                if (mm) { mn = f4min(mn, acc[t]); mx = f4max(mx, acc[t]); }
            }
        }
    }
    float* part = GRAD ? ((p.sink.finalize && p.sink.mean) ? p.sink.partial : nullptr) : p.partial;
    if (part) block_reduce_store_stats<256>(s1, s2, part, GRAD ? p.sink.ldp : p.ldp, c0, p.C, lds, bx, mm ? 4 : 2);
    if (part && mm) block_reduce_store_minmax<256>(mn, mx, part, p.ldp, c0, p.C, lds, bx);
}

// ---- dgrad, generic stride (used for stride 2): one input pixel per thread step, predicated taps.
template <int S>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(DwP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx;
    const int c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const int items = p.B * p.IH * p.IW;
    float4 wk[9];
    load_w9(p.w, c0, p.C, wk, false);
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.C);
    const SinkCoef4 sk = sink_coef4(p.sink, c0, p.C);
    float4 s1 = f4zero(), s2 = f4zero();
    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int ix = (int)(it % p.IW);
        const int t2 = it / p.IW;
        const int iy = (int)(t2 % p.IH);
        const int b = (int)(t2 / p.IH);
        // stride 2: input pixel (iy, ix) receives at most 2 x 2 taps.  Tap "a" is output floor((i+1)/2) with kernel
        // index k = (i+1) & 1, tap "b" (only when k == 0) is the output before it with kernel index 2.
        static_assert(S == 2, "generic-tap dgrad is written for stride 2");
        const int oya = (iy + 1) >> 1, kya = (iy + 1) & 1, oxa = (ix + 1) >> 1, kxa = (ix + 1) & 1;
        const bool ya_ok = oya < p.OH, yb_ok = kya == 0 && oya >= 1;
        const bool xa_ok = oxa < p.OW, xb_ok = kxa == 0 && oxa >= 1;
        const int oy_[2] = {min(oya, p.OH - 1), max(oya - 1, 0)};
        const int ox_[2] = {min(oxa, p.OW - 1), max(oxa - 1, 0)};
        const bool yv[2] = {ya_ok, yb_ok}, xv[2] = {xa_ok, xb_ok};
        float4 rg[4], ry[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t r = ((size_t)b * p.OH + oy_[q >> 1]) * p.OW + ox_[q & 1];
            rg[q] = ldg4(p.dy.g + r * p.dy.ldg + c0);
            ry[q] = gk.two ? ldg4(p.dy.y + r * p.dy.ldy + c0) : f4zero();
        }
        float4 acc = f4zero();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = yv[q >> 1] && xv[q & 1];
            // kernel row: tap a -> kya (0|1), tap b -> 2; same for columns
            float4 w0, w1, w2;   // the three columns of the selected kernel row
            if ((q >> 1) == 0) {
                w0 = kya ? wk[3] : wk[0]; w1 = kya ? wk[4] : wk[1]; w2 = kya ? wk[5] : wk[2];
            } else {
                w0 = wk[6]; w1 = wk[7]; w2 = wk[8];
            }
            const float4 wsel = ((q & 1) == 0) ? (kxa ? w1 : w0) : w2;
            float4 v = gsrc_apply4(rg[q], ry[q], gk);
            v = ok ? v : f4zero();
            acc = f4fma(v, wsel, acc);
        }
        sink_emit4(p.sink, sk, (size_t)it, c0, acc, s1, s2);
    }
    float* part = (p.sink.finalize && p.sink.mean) ? p.sink.partial : nullptr;
    if (part) block_reduce_store_stats<256>(s1, s2, part, p.sink.ldp, c0, p.C, lds, bx);
}

// ---- dgrad, stride 2, even input sizes: one thread owns a 2x2 block of input pixels, which is fed by exactly the 2x2
// block of outputs (a..a+1, b..b+1): 4 gradient operands are loaded for 4 results (the per-pixel kernel above loads 4 each).
//   dx[2a  ][2b  ] = d00*w11
//   dx[2a  ][2b+1] = d01*w10 + d00*w12
//   dx[2a+1][2b  ] = d10*w01 + d00*w21
//   dx[2a+1][2b+1] = d11*w00 + d10*w02 + d01*w20 + d00*w22          (dAB = dy'[a+A][b+B], zero outside the plane)
__global__ __launch_bounds__(256) void dw_dgrad_s2_quad_kernel(DwP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx;
    const int c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const int QH = p.IH >> 1, QW = p.IW >> 1;
    const int items = p.B * QH * QW;
    float4 wk[9];
    load_w9(p.w, c0, p.C, wk, false);
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.C);
    const SinkCoef4 sk = sink_coef4(p.sink, c0, p.C);
    float4 s1 = f4zero(), s2 = f4zero();
    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int qb = (int)(it % QW);
        const int t2 = it / QW;
        const int qa = (int)(t2 % QH);
        const int b = (int)(t2 / QH);
        const bool y1ok = qa + 1 < p.OH, x1ok = qb + 1 < p.OW;
        const int oy1 = min(qa + 1, p.OH - 1), ox1 = min(qb + 1, p.OW - 1);
        const size_t r00 = ((size_t)b * p.OH + qa) * p.OW + qb, r01 = ((size_t)b * p.OH + qa) * p.OW + ox1;
        const size_t r10 = ((size_t)b * p.OH + oy1) * p.OW + qb, r11 = ((size_t)b * p.OH + oy1) * p.OW + ox1;
        float4 g00 = ldg4(p.dy.g + r00 * p.dy.ldg + c0), g01 = ldg4(p.dy.g + r01 * p.dy.ldg + c0);
        float4 g10 = ldg4(p.dy.g + r10 * p.dy.ldg + c0), g11 = ldg4(p.dy.g + r11 * p.dy.ldg + c0);
        float4 y00 = f4zero(), y01 = f4zero(), y10 = f4zero(), y11 = f4zero();
        if (gk.two) {
            y00 = ldg4(p.dy.y + r00 * p.dy.ldy + c0); y01 = ldg4(p.dy.y + r01 * p.dy.ldy + c0);
            y10 = ldg4(p.dy.y + r10 * p.dy.ldy + c0); y11 = ldg4(p.dy.y + r11 * p.dy.ldy + c0);
        }
        const size_t i00 = ((size_t)b * p.IH + 2 * qa) * p.IW + 2 * qb;
        const SinkOps o00 = sink_load4(p.sink, sk, i00, c0), o01 = sink_load4(p.sink, sk, i00 + 1, c0);
        const SinkOps o10 = sink_load4(p.sink, sk, i00 + p.IW, c0), o11 = sink_load4(p.sink, sk, i00 + p.IW + 1, c0);
        const float4 d00 = gsrc_apply4(g00, y00, gk);
        const float4 d01 = x1ok ? gsrc_apply4(g01, y01, gk) : f4zero();
        const float4 d10 = y1ok ? gsrc_apply4(g10, y10, gk) : f4zero();
        const float4 d11 = (x1ok && y1ok) ? gsrc_apply4(g11, y11, gk) : f4zero();
        const float4 e00 = f4fma(d00, wk[4], f4zero());
        const float4 e01 = f4fma(d01, wk[3], f4fma(d00, wk[5], f4zero()));
        const float4 e10 = f4fma(d10, wk[1], f4fma(d00, wk[7], f4zero()));
        const float4 e11 = f4fma(d11, wk[0], f4fma(d10, wk[2], f4fma(d01, wk[6], f4fma(d00, wk[8], f4zero()))));
        sink_store4(p.sink, sk, i00, c0, e00, o00, s1, s2);
        sink_store4(p.sink, sk, i00 + 1, c0, e01, o01, s1, s2);
        sink_store4(p.sink, sk, i00 + p.IW, c0, e10, o10, s1, s2);
        sink_store4(p.sink, sk, i00 + p.IW + 1, c0, e11, o11, s1, s2);
    }
    float* part = (p.sink.finalize && p.sink.mean) ? p.sink.partial : nullptr;
    if (part) block_reduce_store_stats<256>(s1, s2, part, p.sink.ldp, c0, p.C, lds, bx);
}

// ---- dgrad, stride 1, column walk: a thread owns TW columns x RW rows of the (same-size) plane and walks down the rows
// with a rolling window of three TRANSFORMED gradient rows in registers, so every gradient operand is loaded once per
// thread instead of three times (vertical reuse no longer goes through L1/L2).
template <int TW, int RW>
__global__ __launch_bounds__(256) void dw_dgrad_s1_walk_kernel(DwP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx;
    const int c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const int H = p.OH, W = p.OW;                       // dy plane == dx plane (stride 1, pad 1)
    const int WG = (W + TW - 1) / TW, HG = (H + RW - 1) / RW;
    const int items = p.B * HG * WG;
    constexpr int NCOL = TW + 2;
    float4 wk[9];
    load_w9(p.w, c0, p.C, wk, true);                    // flipped taps: dx = corr(dy', flip(w))
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.C);
    const SinkCoef4 sk = sink_coef4(p.sink, c0, p.C);
    float4 s1 = f4zero(), s2 = f4zero();
    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int xg = (int)(it % WG);
        const int t2 = it / WG;
        const int yg = (int)(t2 % HG);
        const int b = (int)(t2 / HG);
        const int x0 = xg * TW, y0 = yg * RW;
        float4 win[3][NCOL];                            // rows (y-1, y, y+1) of dy', rotated by renaming
        auto load_row = [&](int y, float4 (&dst)[NCOL]) {
            const bool yok = y >= 0 && y < H;
            const int yc = min(max(y, 0), H - 1);
            const size_t rowbase = ((size_t)b * H + yc) * W;
            float4 ra[NCOL], rb[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int xc = min(max(x0 - 1 + j, 0), W - 1);
                ra[j] = ldg4(p.dy.g + (rowbase + xc) * p.dy.ldg + c0);
                rb[j] = gk.two ? ldg4(p.dy.y + (rowbase + xc) * p.dy.ldy + c0) : f4zero();
            }
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int x = x0 - 1 + j;
                const float4 v = gsrc_apply4(ra[j], rb[j], gk);
                dst[j] = (yok && x >= 0 && x < W) ? v : f4zero();
            }
        };
        load_row(y0 - 1, win[0]);
        load_row(y0, win[1]);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int y = y0 + r;
            SinkOps so[TW];                              // this row's sink operands travel with the next gradient row
            {
                const size_t srow = ((size_t)b * H + min(y, H - 1)) * W + x0;
#pragma unroll
                for (int t = 0; t < TW; ++t) so[t] = sink_load4(p.sink, sk, srow + min(t, W - 1 - x0), c0);
            }
            load_row(y + 1, win[(r + 2) % 3]);          // r is a compile-time constant: plain register renaming
            if (y < H) {
                float4 acc[TW];
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = f4zero();
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float4 (&row)[NCOL] = win[(r + ky) % 3];
#pragma unroll
                    for (int t = 0; t < TW; ++t)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[t] = f4fma(row[t + kx], wk[ky * 3 + kx], acc[t]);
                }
                const size_t orow = ((size_t)b * H + y) * W + x0;
#pragma unroll
                for (int t = 0; t < TW; ++t)
                    if (x0 + t < W) sink_store4(p.sink, sk, orow + t, c0, acc[t], so[t], s1, s2);
            }
        }
    }
    float* part = (p.sink.finalize && p.sink.mean) ? p.sink.partial : nullptr;
    if (part) block_reduce_store_stats<256>(s1, s2, part, p.sink.ldp, c0, p.C, lds, bx);
}

// ---- wgrad: dw[c][ky][kx] = sum dy'[b,oy,ox,c] * x'[b,oy*S+ky-1,ox*S+kx-1,c]
template <int S, int TW>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y, nx = blockDim.x, ny = blockDim.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx;
    const int c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const int OWG = (p.OW + TW - 1) / TW;
    const int items = p.B * p.OH * OWG;
    constexpr int NCOL = (TW - 1) * S + 3;
    const SrcCoef4 xk = src_coef4(p.in, c0, p.C);
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.C);
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f4zero();
    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int oxg = (int)(it % OWG);
        const int t2 = it / OWG;
        const int oy = (int)(t2 % p.OH);
        const int b = (int)(t2 / p.OH);
        const int ox0 = oxg * TW;
        float4 dy[TW];
        const size_t orow = ((size_t)b * p.OH + oy) * p.OW;
        {
            float4 rg[TW], ry[TW];
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const int oxc = min(ox0 + t, p.OW - 1);
                rg[t] = ldg4(p.dy.g + (orow + oxc) * p.dy.ldg + c0);
                ry[t] = gk.two ? ldg4(p.dy.y + (orow + oxc) * p.dy.ldy + c0) : f4zero();
            }
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const float4 v = gsrc_apply4(rg[t], ry[t], gk);
                dy[t] = (ox0 + t < p.OW) ? v : f4zero();
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * S - 1 + ky;
            const bool yok = iy >= 0 && iy < p.IH;
            const int iyc = min(max(iy, 0), p.IH - 1);
            const size_t rowbase = ((size_t)b * p.IH + iyc) * p.IW;
            float4 rx[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int ixc = min(max(ox0 * S - 1 + j, 0), p.IW - 1);
                rx[j] = ldg4(p.in.x + (rowbase + ixc) * p.in.ld + c0);
            }
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int ix = ox0 * S - 1 + j;
                const bool ok = yok && ix >= 0 && ix < p.IW;
                float4 v = src_apply4(rx[j], xk);
                v = ok ? v : f4zero();
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    const int kx = j - t * S;
                    if (kx >= 0 && kx < 3) acc[ky * 3 + kx] = f4fma(dy[t], v, acc[ky * 3 + kx]);
                }
            }
        }
    }
    // reduce the 9 float4 accumulators over ty, one tap at a time (LDS: ny*nx float4)
    float4* l4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        l4[ty * nx + tx] = acc[t];
        __syncthreads();
        if (ty == 0 && cok) {
            float4 s = f4zero();
            for (int y = 0; y < ny; ++y) s = f4add(s, l4[y * nx + tx]);
            float* dst = p.wpartial + ((size_t)bx * p.C + c0) * 9 + t;
            if (c0 + 0 < p.C) dst[0] = s.x;
            if (c0 + 1 < p.C) dst[9] = s.y;
            if (c0 + 2 < p.C) dst[18] = s.z;
            if (c0 + 3 < p.C) dst[27] = s.w;
        }
    }
}

// out[j] = sum_k part[k*n + j]; block = 32 columns x 32 row lanes (deterministic tree)
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const float* part, int nblk, int n, float* out) {
    __shared__ float red[32][33];
    const int j = blockIdx.x * 32 + threadIdx.x;
    float s = 0.f;
    if (j < n) {
        int k = threadIdx.y;
        for (; k + 7 * 32 < nblk; k += 8 * 32) {       // 8 independent loads in flight (the loop is pure load latency)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + 32 * u) * n + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nblk; k += 32) s += part[(size_t)k * n + j];
    }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && j < n) {
        float tt = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tt += red[k][threadIdx.x];
        out[j] = tt;
    }
}

// ------------------------------------------------------------------------------------------------ stem
// Dense 3x3 on the NCHW image (CI input planes, CI<=4).  8.. lanes per pixel, each lane owns 4 output channels.
struct StemP {
    const float* img; const float* w; float* out; float* partial;
    pq_gsrc dy; float* wpartial;
    int B, CI, IH, IW, OH, OW, O, O4, S, ldo, ldp;
    int minmax;       // forward: `partial` rows have 4 planes (+ per-channel min, max of the outputs)
};

template <int CI>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx, c0 = c4 * 4;
    const bool cok = c4 < p.O4;
    float4 wk[CI * 9];
#pragma unroll
    for (int t = 0; t < CI * 9; ++t) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (c0 + j < p.O) ? p.w[(size_t)(c0 + j) * CI * 9 + t] : 0.f;
        wk[t] = make_float4(e[0], e[1], e[2], e[3]);
    }
    const int items = p.B * p.OH * p.OW;
    const size_t plane = (size_t)p.IH * p.IW;
    float4 s1 = f4zero(), s2 = f4zero();
    const float inf = __builtin_huge_valf();
    float4 mn = make_float4(inf, inf, inf, inf), mx = make_float4(-inf, -inf, -inf, -inf);
    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int ox = (int)(it % p.OW);
        const int t2 = it / p.OW;
        const int oy = (int)(t2 % p.OH);
        const int b = (int)(t2 / p.OH);
        float4 acc = f4zero();
        float px[CI * 9];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const float* pl = p.img + ((size_t)b * CI + ci) * plane;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iyc = min(max(oy * p.S - 1 + ky, 0), p.IH - 1);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ixc = min(max(ox * p.S - 1 + kx, 0), p.IW - 1);
                    px[ci * 9 + ky * 3 + kx] = pl[(size_t)iyc * p.IW + ixc];      // unconditional, masked below
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * p.S - 1 + ky;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * p.S - 1 + kx;
                    const bool ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    const float v = ok ? px[ci * 9 + ky * 3 + kx] : 0.f;
                    acc = f4fma(make_float4(v, v, v, v), wk[ci * 9 + ky * 3 + kx], acc);
                }
            }
        stg4(p.out + (size_t)it * p.ldo + c0, acc);
        s1 = f4add(s1, acc);
        s2 = f4fma(acc, acc, s2);
        if (p.minmax) { mn = f4min(mn, acc); mx = f4max(mx, acc); }
    }
    if (p.partial) block_reduce_store_stats<256>(s1, s2, p.partial, p.ldp, c0, p.O, lds, bx, p.minmax ? 4 : 2);
    if (p.partial && p.minmax) block_reduce_store_minmax<256>(mn, mx, p.partial, p.ldp, c0, p.O, lds, bx);
}

// Image patch [CI][PH][PW] of an LDS-tiled stem kernel: thread `tid` stages elements e = tid + u * nthr, u < NE.  Which (channel,
// row, column) of the patch an element is does not depend on the tile, so it is decoded once (three integer divisions per element
// were 40 % of the VALU instructions of these kernels); per tile only the clamped address and the bounds test remain.
template <int NE> struct StemPatch {
    int dst[NE];          // LDS offset in the padded patch; -1 beyond the patch
    int pyx[NE];          // row << 16 | column inside the patch
    int cio[NE];          // channel * IH * IW
    __device__ __forceinline__ void init(int tid, int nthr, int CI, int PH, int PW, int PWp, int IH, int IW) {
        const int npatch = CI * PH * PW;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + u * nthr, ec = min(e, npatch - 1);
            const int ci = ec / (PH * PW), rem = ec - ci * PH * PW;
            const int py = rem / PW, px = rem - py * PW;
            dst[u] = e < npatch ? (ci * PH + py) * PWp + px : -1;
            pyx[u] = (py << 16) | px;
            cio[u] = ci * IH * IW;
        }
    }
    __device__ __forceinline__ void stage(const float* img, int iy0, int ix0, int IH, int IW, float* patch) const {
        float v[NE];
        bool ok[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int iy = iy0 + (pyx[u] >> 16), ix = ix0 + (pyx[u] & 0xffff);
            ok[u] = iy >= 0 && iy < IH && ix >= 0 && ix < IW;
            v[u] = img[cio[u] + min(max(iy, 0), IH - 1) * IW + min(max(ix, 0), IW - 1)];
        }
#pragma unroll
        for (int u = 0; u < NE; ++u)
            if (dst[u] >= 0) patch[dst[u]] = ok[u] ? v[u] : 0.f;
    }
};

// LDS-tiled forward: a workgroup walks 4x32-pixel output tiles (tile = bx, bx + gridDim.x, ...; one statistics row per
// workgroup as before), stages the CI x PH x PW image patch once with coalesced row loads and every lane reads its 27
// taps from LDS (the 8 channel-quad lanes of a pixel hit the same word: broadcast).  The streaming version above issued 27
// scalar global loads per output pixel and lane.
template <int CI>
__global__ __launch_bounds__(256) void stem_fwd_tiled_kernel(StemP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TH = 4, TW = 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx, c0 = c4 * 4;
    const bool cok = c4 < p.O4;
    const int S = p.S, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PWp = PW + 1;
    float* red = lds;                                   // [256][8] for the statistics reduction
    float* patch = lds + 256 * 8;                       // [CI][PH][PWp]
    float4 wk[CI * 9];
#pragma unroll
    for (int t = 0; t < CI * 9; ++t) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (c0 + j < p.O) ? p.w[(size_t)(c0 + j) * CI * 9 + t] : 0.f;
        wk[t] = make_float4(e[0], e[1], e[2], e[3]);
    }
    const int tilesH = (p.OH + TH - 1) / TH, tilesW = (p.OW + TW - 1) / TW, ntiles = p.B * tilesH * tilesW;
    const int tid = ty * blockDim.x + tx, nthr = blockDim.x * blockDim.y;
    StemPatch<8> sp;
    sp.init(tid, nthr, CI, PH, PW, PWp, p.IH, p.IW);
    float4 s1 = f4zero(), s2 = f4zero();
    const float inf = __builtin_huge_valf();
    float4 mn = make_float4(inf, inf, inf, inf), mx = make_float4(-inf, -inf, -inf, -inf);
    for (int tile = bx; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tilesW, th = (tile / tilesW) % tilesH, b = tile / (tilesW * tilesH);
        const int oy0 = th * TH, ox0 = tw * TW;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        __syncthreads();                                // the previous tile's taps have been read
        sp.stage(p.img + (size_t)b * CI * p.IH * p.IW, iy0, ix0, p.IH, p.IW, patch);
        __syncthreads();
        if (cok) {
            for (int pi = ty; pi < TH * TW; pi += blockDim.y) {
                const int oyl = pi / TW, oxl = pi - oyl * TW;
                const int oy = oy0 + oyl, ox = ox0 + oxl;
                if (oy < p.OH && ox < p.OW) {
                    float4 acc = f4zero();
                    const float* pp = patch + (oyl * S) * PWp + oxl * S;
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float v = pp[(ci * PH + ky) * PWp + kx];
                                acc = f4fma(make_float4(v, v, v, v), wk[ci * 9 + ky * 3 + kx], acc);
                            }
                    stg4(p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * p.ldo + c0, acc);
                    s1 = f4add(s1, acc);
                    s2 = f4fma(acc, acc, s2);
                    if (p.minmax) { mn = f4min(mn, acc); mx = f4max(mx, acc); }
                }
            }
        }
    }
    if (p.partial) {
        __syncthreads();
        block_reduce_store_stats<256>(s1, s2, p.partial, p.ldp, c0, p.O, red, bx, p.minmax ? 4 : 2);
        if (p.minmax) block_reduce_store_minmax<256>(mn, mx, p.partial, p.ldp, c0, p.O, red, bx);
    }
}

template <int CI>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x, ty = threadIdx.y, nx = blockDim.x, ny = blockDim.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx, c0 = c4 * 4;
    const bool cok = c4 < p.O4;
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.O);
    float4 acc[CI * 9];
#pragma unroll
    for (int t = 0; t < CI * 9; ++t) acc[t] = f4zero();
    const int items = p.B * p.OH * p.OW;
    const size_t plane = (size_t)p.IH * p.IW;
    for (int it = bx * blockDim.y + ty; it < items && cok; it += gridDim.x * blockDim.y) {
        const int ox = (int)(it % p.OW);
        const int t2 = it / p.OW;
        const int oy = (int)(t2 % p.OH);
        const int b = (int)(t2 / p.OH);
        float4 g = ldg4(p.dy.g + (size_t)it * p.dy.ldg + c0);
        float4 y = gk.two ? ldg4(p.dy.y + (size_t)it * p.dy.ldy + c0) : f4zero();
        const float4 dy = gsrc_apply4(g, y, gk);
        float px[CI * 9];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const float* pl = p.img + ((size_t)b * CI + ci) * plane;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iyc = min(max(oy * p.S - 1 + ky, 0), p.IH - 1);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ixc = min(max(ox * p.S - 1 + kx, 0), p.IW - 1);
                    px[ci * 9 + ky * 3 + kx] = pl[(size_t)iyc * p.IW + ixc];
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * p.S - 1 + ky;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * p.S - 1 + kx;
                    const bool ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    const float v = ok ? px[ci * 9 + ky * 3 + kx] : 0.f;
                    acc[ci * 9 + ky * 3 + kx] = f4fma(dy, make_float4(v, v, v, v), acc[ci * 9 + ky * 3 + kx]);
                }
            }
    }
    float4* l4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int t = 0; t < CI * 9; ++t) {
        __syncthreads();
        l4[ty * nx + tx] = acc[t];
        __syncthreads();
        if (ty == 0 && cok) {
            float4 s = f4zero();
            for (int y = 0; y < ny; ++y) s = f4add(s, l4[y * nx + tx]);
            float* dst = p.wpartial + ((size_t)bx * p.O + c0) * (CI * 9) + t;
            if (c0 + 0 < p.O) dst[0] = s.x;
            if (c0 + 1 < p.O) dst[CI * 9] = s.y;
            if (c0 + 2 < p.O) dst[2 * CI * 9] = s.z;
            if (c0 + 3 < p.O) dst[3 * CI * 9] = s.w;
        }
    }
}

// LDS-tiled weight gradient of the stem: same tile walk and image patch as stem_fwd_tiled_kernel; a lane keeps its 27 x 4
// accumulators over all tiles of the workgroup, then the usual reduction over the pixel lanes.
template <int CI>
__global__ __launch_bounds__(256) void stem_wgrad_tiled_kernel(StemP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TH = 4, TW = 32;
    const int tx = threadIdx.x, ty = threadIdx.y, nx = blockDim.x, ny = blockDim.y;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + tx, c0 = c4 * 4;
    const bool cok = c4 < p.O4;
    const int S = p.S, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PWp = PW + 1;
    float* patch = lds + 256 * 4;                       // after the [ny*nx] float4 reduction area
    const GsrcCoef4 gk = gsrc_coef4(p.dy, c0, p.O);
    float4 acc[CI * 9];
#pragma unroll
    for (int t = 0; t < CI * 9; ++t) acc[t] = f4zero();
    const int tilesH = (p.OH + TH - 1) / TH, tilesW = (p.OW + TW - 1) / TW, ntiles = p.B * tilesH * tilesW;
    const int tid = ty * nx + tx, nthr = nx * ny;
    StemPatch<8> sp;
    sp.init(tid, nthr, CI, PH, PW, PWp, p.IH, p.IW);
    for (int tile = bx; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tilesW, th = (tile / tilesW) % tilesH, b = tile / (tilesW * tilesH);
        const int oy0 = th * TH, ox0 = tw * TW;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        // the gradient operands of this lane's first NPX pixels are requested BEFORE the patch is staged: their latency then
        // overlaps the patch loads (the kernel was one exposed load latency per pixel: 136 us for 318 MB)
        constexpr int NPX = 4;
        float4 gq[NPX], yq[NPX];
        auto fetch = [&](int pi0) {
#pragma unroll
            for (int u = 0; u < NPX; ++u) {
                const int pi = min(pi0 + u * ny, TH * TW - 1);
                const int oyl = pi / TW, oxl = pi - oyl * TW;
                const size_t row = ((size_t)b * p.OH + min(oy0 + oyl, p.OH - 1)) * p.OW + min(ox0 + oxl, p.OW - 1);
                gq[u] = ldg4(p.dy.g + row * p.dy.ldg + (cok ? c0 : 0));
                yq[u] = gk.two ? ldg4(p.dy.y + row * p.dy.ldy + (cok ? c0 : 0)) : f4zero();
            }
        };
        fetch(ty);
        __syncthreads();
        sp.stage(p.img + (size_t)b * CI * p.IH * p.IW, iy0, ix0, p.IH, p.IW, patch);
        __syncthreads();
        for (int pi0 = ty; pi0 < TH * TW; pi0 += NPX * ny) {
            if (pi0 != ty) fetch(pi0);                  // wide outputs (O > 32): fewer pixel lanes, more rounds
#pragma unroll
            for (int u = 0; u < NPX; ++u) {
                const int pi = pi0 + u * ny;
                const int oyl = pi / TW, oxl = pi - oyl * TW;
                const int oy = oy0 + oyl, ox = ox0 + oxl;
                if (cok && pi < TH * TW && oy < p.OH && ox < p.OW) {
                    const float4 dy = gsrc_apply4(gq[u], yq[u], gk);
                    const float* pp = patch + (oyl * S) * PWp + oxl * S;
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float v = pp[(ci * PH + ky) * PWp + kx];
                                acc[ci * 9 + ky * 3 + kx] = f4fma(dy, make_float4(v, v, v, v), acc[ci * 9 + ky * 3 + kx]);
                            }
                }
            }
        }
    }
    float4* l4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int t = 0; t < CI * 9; ++t) {
        __syncthreads();
        l4[ty * nx + tx] = acc[t];
        __syncthreads();
        if (ty == 0 && cok) {
            float4 sum = f4zero();
            for (int y = 0; y < ny; ++y) sum = f4add(sum, l4[y * nx + tx]);
            float* dst = p.wpartial + ((size_t)bx * p.O + c0) * (CI * 9) + t;
            if (c0 + 0 < p.O) dst[0] = sum.x;
            if (c0 + 1 < p.O) dst[CI * 9] = sum.y;
            if (c0 + 2 < p.O) dst[2 * CI * 9] = sum.z;
            if (c0 + 3 < p.O) dst[3 * CI * 9] = sum.w;
        }
    }
}

}  // namespace pq

using namespace pq;

// ---------------------------------------------------------------------------------------------- launch geometry
extern "C" int32_t pq_ew_blocks(int32_t rows, int32_t C) { return ew_geom(rows, C).gx; }
extern "C" int32_t pq_dw3x3_blocks(int32_t B, int32_t H, int32_t W, int32_t C) {
    // the forward kernel walks pixel groups of TW outputs; the geometry is a function of (rows/TW-ish, C) --
    // to keep ONE answer for every kernel writing (B,H,W,C) statistics we size by rows and cap the grid.
    return ew_geom((long)B * H * W, C).gx;
}

static int check_dw_common(int B, int IH, int IW, int OH, int OW, int C, int stride) {
    if (B <= 0 || IH <= 0 || IW <= 0 || C <= 0) return 0;
    if (stride != 1 && stride != 2) return 0;
    if (OH != (IH - 1) / stride + 1 || OW != (IW - 1) / stride + 1) return 0;
    return 1;      // any channel count: kernels walk ceil(C/4) quads and mask by C; pitches (multiples of 4) are checked by callers
}

// streaming (register-window) forward; the LDS-tiled variant lives in dwtile.hip, pq_dw3x3_fwd picks one per shape
int pq_dw_fwd_stream(const pq_dw3x3_fwd_args* a, void* stream) {
    DwP p{};
    p.in = a->in; p.w = a->w; p.out = a->out; p.partial = a->partial; p.ldo = a->ldo; p.ldp = a->ldp;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    p.minmax = (a->partial && a->minmax) ? 1 : 0;
    EwGeom g = ew_geom((long)a->B * a->OH * a->OW, a->C);
    dim3 grid(g.gx, g.gy), blk(g.cx, g.py);
    size_t sh = (size_t)g.cx * g.py * 8 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (a->stride == 1) { pq_set_error("pq_dw3x3_fwd: internal -- stride 1 runs on the walking kernel"); return PQ_EINVAL; }
    PQ_LAUNCH((dw_fwd_kernel<2, 2, false>), grid, blk, sh, st, p);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_dw3x3_dgrad(const pq_dw3x3_dgrad_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->w && a->dx.g, "null pointer");
    PQ_SINK_ACC_UNSUPPORTED(a->dx);
    PQ_SINK_ADDEND_CHECK(a->dx);
    PQ_CHECK_ARG(check_dw_common(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->stride), "bad shape (need stride 1|2, pad 1, pitches that are multiples of 4)");
    PQ_CHECK_ARG(!(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    if (act_is_special(a->dx.act)) { pq_set_error("pq_dw3x3_dgrad: logistic/tanh sinks are not supported"); return PQ_EUNSUPPORTED; }
    DwP p{};
    p.dy = a->dy; p.w = a->w; p.sink = a->dx;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    EwGeom g = ew_geom((long)a->B * a->IH * a->IW, a->C);
    dim3 grid(g.gx, g.gy), blk(g.cx, g.py);
    size_t sh = (size_t)g.cx * g.py * 8 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (a->stride == 1) {
        DwP q = p;
        q.IH = a->OH; q.IW = a->OW; q.OH = a->IH; q.OW = a->IW;   // "input" = dy plane, "output" = dx plane
        // the column-walk kernel only pays on the largest planes (256x256: 0.167 -> 0.144 ms); below that the per-row kernel
        // has more parallelism and is as fast
        if ((long)a->IH * a->IW >= 256L * 256L) PQ_LAUNCH((dw_dgrad_s1_walk_kernel<4, 4>), grid, blk, sh, st, q);
        else PQ_LAUNCH((dw_fwd_kernel<1, 4, true>), grid, blk, sh, st, q);
    } else if ((a->IH & 1) == 0 && (a->IW & 1) == 0) {
        PQ_LAUNCH(dw_dgrad_s2_quad_kernel, grid, blk, sh, st, p);
    } else {
        PQ_LAUNCH((dw_dgrad_kernel<2>), grid, blk, sh, st, p);
    }
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

static int dw_wgrad_blocks(int B, int OH, int OW, int C, EwGeom* out) {
    EwGeom g = ew_geom((long)B * OH * OW / 4 + 1, C);
    if (g.gx > 512) g.gx = 512;
    if (out) *out = g;
    return g.gx;
}
extern "C" int64_t pq_dw3x3_wgrad_scratch_bytes(int32_t B, int32_t OH, int32_t OW, int32_t C) {
    return (int64_t)dw_wgrad_blocks(B, OH, OW, C, nullptr) * C * 9 * 4;
}

extern "C" int pq_dw3x3_wgrad(const pq_dw3x3_wgrad_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->in.x && a->dw && a->scratch, "null pointer");
    PQ_CHECK_ARG(check_dw_common(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->stride), "bad shape (need stride 1|2, pad 1, pitches that are multiples of 4)");
    DwP p{};
    p.dy = a->dy; p.in = a->in; p.wpartial = a->scratch;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    EwGeom g;
    dw_wgrad_blocks(a->B, a->OH, a->OW, a->C, &g);
    dim3 grid(g.gx, g.gy), blk(g.cx, g.py);
    size_t sh = (size_t)g.cx * g.py * 4 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (a->stride == 1) PQ_LAUNCH((dw_wgrad_kernel<1, 4>), grid, blk, sh, st, p);
    else PQ_LAUNCH((dw_wgrad_kernel<2, 2>), grid, blk, sh, st, p);
    PQ_CHECK_LAUNCH();
    const int n = a->C * 9;
    PQ_LAUNCH(partial_reduce_kernel, dim3((n + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, g.gx, n, a->dw);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}


// ------------------------------------------------------------------------------------------------ stem host
static int stem_check(int B, int CI, int IH, int IW, int OH, int OW, int O, int stride) {
    if (B <= 0 || CI != 3 || O <= 0) return 0;
    if (stride != 1 && stride != 2) return 0;
    return OH == (IH - 1) / stride + 1 && OW == (IW - 1) / stride + 1;
}

extern "C" int pq_stem3x3_fwd(const pq_stem3x3_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->image && a->w && a->out, "null pointer");
    PQ_CHECK_ARG(stem_check(a->B, a->CI, a->IH, a->IW, a->OH, a->OW, a->O, a->stride), "unsupported stem shape (CI==3)");
    StemP p{};
    p.img = a->image; p.w = a->w; p.out = a->out; p.partial = a->partial;
    p.B = a->B; p.CI = a->CI; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.O = a->O; p.O4 = (a->O + 3) / 4;
    p.S = a->stride; p.ldo = a->ldo; p.ldp = a->ldp; p.minmax = (a->partial && a->minmax) ? 1 : 0;
    EwGeom g = ew_geom((long)a->B * a->OH * a->OW, a->O);
    dim3 grid(g.gx, g.gy), blk(g.cx, g.py);
    const int npatch_f = 3 * (3 * a->stride + 3) * (31 * a->stride + 3);
    if (npatch_f > 8 * g.cx * g.py) {           // patch too large for the LDS-tiled kernel: per-pixel global loads
        PQ_LAUNCH((stem_fwd_kernel<3>), grid, blk, (size_t)g.cx * g.py * 8 * sizeof(float), (hipStream_t)stream, p);
    } else {
        const int PH = 3 * a->stride + 3, PW = 31 * a->stride + 3;
        const size_t sh = ((size_t)256 * 8 + (size_t)3 * PH * (PW + 1) + 16) * sizeof(float);
        PQ_LAUNCH((stem_fwd_tiled_kernel<3>), grid, blk, sh, (hipStream_t)stream, p);
    }
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

static int stem_wgrad_blocks(int B, int OH, int OW, int O, EwGeom* out) {
    EwGeom g = ew_geom((long)B * OH * OW / 8 + 1, O);
    if (g.gx > 512) g.gx = 512;
    if (out) *out = g;
    return g.gx;
}
extern "C" int64_t pq_stem3x3_wgrad_scratch_bytes(int32_t B, int32_t OH, int32_t OW, int32_t O, int32_t CI) {
    return (int64_t)stem_wgrad_blocks(B, OH, OW, O, nullptr) * O * CI * 9 * 4;
}

extern "C" int pq_stem3x3_wgrad(const pq_stem3x3_wgrad_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->image && a->dw && a->scratch, "null pointer");
    PQ_CHECK_ARG(stem_check(a->B, a->CI, a->IH, a->IW, a->OH, a->OW, a->O, a->stride), "unsupported stem shape (CI==3)");
    StemP p{};
    p.img = a->image; p.dy = a->dy; p.wpartial = a->scratch;
    p.B = a->B; p.CI = a->CI; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.O = a->O; p.O4 = (a->O + 3) / 4; p.S = a->stride;
    EwGeom g;
    stem_wgrad_blocks(a->B, a->OH, a->OW, a->O, &g);
    dim3 grid(g.gx, g.gy), blk(g.cx, g.py);
    hipStream_t st = (hipStream_t)stream;
    const int npatch_w = 3 * (3 * a->stride + 3) * (31 * a->stride + 3);
    if (npatch_w > 8 * g.cx * g.py) {
        PQ_LAUNCH((stem_wgrad_kernel<3>), grid, blk, (size_t)g.cx * g.py * 4 * sizeof(float), st, p);
    } else {
        const int PH = 3 * a->stride + 3, PW = 31 * a->stride + 3;
        const size_t sh = ((size_t)256 * 4 + (size_t)3 * PH * (PW + 1) + 16) * sizeof(float);
        PQ_LAUNCH((stem_wgrad_tiled_kernel<3>), grid, blk, sh, st, p);
    }
    PQ_CHECK_LAUNCH();
    const int n = a->O * a->CI * 9;
    PQ_LAUNCH(partial_reduce_kernel, dim3((n + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, g.gx, n, a->dw);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
