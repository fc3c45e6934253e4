// Depthwise 3x3 backward (pad 1, stride 1|2), both gradients from ONE pass over the tensors: a workgroup WALKS down a 16-column
// strip of the input plane for a chunk of 32 channels (8 float4 quads x 32 pixel lanes, the layout of dwtile.hip).
//
//   * the dy' = p*g + q*(y - mu) + r tile and the RAW layer input live in two LDS row RINGS; a step of TH (4 / 8) input rows only
//     fetches the rows the previous step did not have (no vertical halo re-read), and those loads are issued BEFORE the
//     step's arithmetic and written to the rings after it (register-staged prefetch: the memory round trip hides behind the
//     compute of the current step, the ring needs no second buffer because the rows it overwrites are the ones just consumed);
//   * the raw input tile serves the weight gradient (x' = act(scale * y + shift) applied at the fragment read, zero padding
//     masked there) AND the gradient sink (activation derivative, BatchNorm-backward sums) -- the engine's sink IS the conv's
//     input tensor, so (g, y_out, y_in) are each read once and the input gradient written once: 4 tensor passes instead of
//     the 7 of separate data- and weight-gradient kernels;
//   * the 9 weight-gradient taps and the two BatchNorm-backward sums stay in registers for the whole walk and leave through
//     one LDS reduction per workgroup (one row of `scratch` / of the sink's statistics per workgroup, far fewer than one per
//     8 x 16 tile).
//
// mirrors: backward of nn.Conv2d(groups=C, k=3) inside the conv blocks of the reference, model/parser.py:389-410.
#include "common.h"

namespace pq {

constexpr int WKQ = 8;            // channel quads per workgroup (32 channels)
constexpr int WPQ = 8;            // LDS pixel pitch in quads: a 16-lane ds_read_b128 group = two adjacent pixels = 256 contiguous bytes = every bank once

struct WalkGeom2 {                                    // stride 2: a step = 8 input rows x 16 columns (4 x 8 outputs)
    static constexpr int TH = 8;
    static constexpr int DYR = 5, DYC = 9;            // 4 owned output rows + the one below (kernel row 0 of the last odd input row)
    static constexpr int XR = 9, XC = 17;             // input rows y0-1 .. y0+7
    static constexpr int NDY = 4, NX = 8;
};

struct DwWP {
    pq_src in; pq_gsrc dy; const float* w; pq_gsink sink; float* wpartial;
    pq_bn_bstat dys;                                   // dy's BatchNorm-backward sums (hand-over: no finalize launch in front)
    int B, IH, IW, OH, OW, C, C4;
    int strips, segs, seg_rows;
};

// ------------------------------------------------------------------------------------------------ stride 2
// A step is small on purpose (8 x 16 input pixels: four per lane): the prefetch of the next step's rows (9 float4 per lane)
// stays in registers across the step's arithmetic next to the 9 weight-gradient accumulators -- 16-row steps spilled.
__global__ __launch_bounds__(256, 2) void dw_bwd_walk_s2_kernel(DwWP p) {
    using G = WalkGeom2;
    constexpr int S = 2;
    constexpr int XPQ = 12;            // input ring pixel pitch: its readers are two pixels apart (2 x 192 bytes = 128 mod 256: no bank conflicts)
    constexpr int DY_F4 = G::DYR * G::DYC * WPQ, X_F4 = G::XR * G::XC * XPQ;
    constexpr int LDS_F4 = (DY_F4 + X_F4) > 9 * 32 * 8 ? (DY_F4 + X_F4) : 9 * 32 * 8;      // the tap reduction reuses the rings
    __shared__ float4 lds4[LDS_F4];
    // per-quad constants that are only needed in one phase of a step live in LDS, not in registers across the whole walk (the
    // step loop's barriers keep the compiler from hoisting the reads back out): the 9 weight taps, the BatchNorm-backward
    // coefficients of dy' (staging phase only) and mean / invstd of the sink (store phase only)
    __shared__ float4 wtab[WKQ][9];
    __shared__ float4 ctab[WKQ][6];                   // p, q, mu, r | mean, invstd
    float4* dyt = lds4;
    float4* xt = lds4 + DY_F4;
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;
    int tIdx, bcy;
    xcd_block(bcy, tIdx);        // grid = (channel chunks, walkers): see dw_walk_launch
    const int strip = tIdx % p.strips, seg = (tIdx / p.strips) % p.segs, b = tIdx / (p.strips * p.segs);
    const int ys = seg * p.seg_rows, x0 = strip * 16;
    const int rows_here = min(p.seg_rows, p.IH - ys);
    const int nsteps = (rows_here + G::TH - 1) / G::TH;
    const int cq0 = bcy * WKQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;
    const SrcCoef4 xk = src_coef4(p.in, cc, p.C);
    const bool two = p.dy.y != nullptr;
    const bool dy_stat = p.dys.acc != nullptr;         // uniform: dy's coefficients from the sink kernel's accumulators (no finalize launch)
    if (dy_stat && t < 4 * WKQ) {                      // one thread per channel of the chunk: one memory round trip, one fp64 chain
        const int c = cq0 * 4 + t;
        GsrcCoef1 k1{1.f, 0.f, 0.f, 0.f};
        if (c < p.C) k1 = gsrc_coef1_bstat(p.dy, p.dys, c, tIdx == 0);
        float* ct = reinterpret_cast<float*>(&ctab[t >> 2][0]) + (t & 3);
        ct[0] = k1.p; ct[4] = k1.q; ct[8] = k1.mu; ct[12] = k1.r;
    }
    if (pl == 8) {             // (wave 1: beside the hand-over derivation above, which runs in wave 0)
        const SinkCoef4 k0 = sink_coef4(p.sink, cc, p.C);
        if (!dy_stat) {
            const GsrcCoef4 g0 = gsrc_coef4(p.dy, cc, p.C);
            ctab[q][0] = make_float4(g0.p[0], g0.p[1], g0.p[2], g0.p[3]);
            ctab[q][1] = make_float4(g0.q[0], g0.q[1], g0.q[2], g0.q[3]);
            ctab[q][2] = make_float4(g0.mu[0], g0.mu[1], g0.mu[2], g0.mu[3]);
            ctab[q][3] = make_float4(g0.r[0], g0.r[1], g0.r[2], g0.r[3]);
        }
        ctab[q][4] = make_float4(k0.mean[0], k0.mean[1], k0.mean[2], k0.mean[3]);
        ctab[q][5] = make_float4(k0.inv[0], k0.inv[1], k0.inv[2], k0.inv[3]);
        {
            // all 36 taps of the quad requested before the first one is used (unconditional loads at clamped channels: with the loads
            // inside per-element conditions the compiler waited for every tap's four loads before it issued the next four -- nine
            // serialised round trips at the head of every workgroup)
            float e[9][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wp = p.w + (size_t)min(cc + j, p.C - 1) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) e[k][j] = wp[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) e[k][j] = (cok && c0 + j < p.C) ? e[k][j] : 0.f;
                wtab[q][k] = make_float4(e[k][0], e[k][1], e[k][2], e[k][3]);
            }
        }
    }
    __syncthreads();
    auto gcoef = [&]() {                               // dy' coefficients of this lane's quad (staging phases)
        GsrcCoef4 g;
        g.two = two;
        const float4 a = ctab[q][0], b2 = ctab[q][1], c2 = ctab[q][2], d = ctab[q][3];
        g.p[0] = a.x; g.p[1] = a.y; g.p[2] = a.z; g.p[3] = a.w;
        g.q[0] = b2.x; g.q[1] = b2.y; g.q[2] = b2.z; g.q[3] = b2.w;
        g.mu[0] = c2.x; g.mu[1] = c2.y; g.mu[2] = c2.z; g.mu[3] = c2.w;
        g.r[0] = d.x; g.r[1] = d.y; g.r[2] = d.z; g.r[3] = d.w;
        return g;
    };
    auto scoef = [&]() {                               // the sink's coefficients: scale / shift are the input operand's (same tensor)
        SinkCoef4 k;
        k.affine = p.sink.scale != nullptr;
        k.stats = p.sink.mean != nullptr && p.sink.partial != nullptr;
        k.act = act_param(p.sink.act);
        k.q = p.sink.qstate != nullptr;
        k.qinv = k.q ? 1.f / p.sink.qstate[2] : 0.f;
        k.qzp = k.q ? p.sink.qstate[3] : 0.f;
        const float4 m = ctab[q][4], iv = ctab[q][5];
        k.mean[0] = m.x; k.mean[1] = m.y; k.mean[2] = m.z; k.mean[3] = m.w;
        k.inv[0] = iv.x; k.inv[1] = iv.y; k.inv[2] = iv.z; k.inv[3] = iv.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.a[j] = k.affine ? xk.a[j] : 1.f; k.b[j] = k.affine ? xk.b[j] : 0.f; }
        return k;
    };
    // ring origins in plane coordinates (local row 0 / column 0 of each ring at the start of the segment)
    const int dy_r0 = ys / S, dy_c0 = x0 / S;
    const int x_r0 = ys - 1, x_c0 = x0 - 1;
    const size_t dyimg = (size_t)b * p.OH * p.OW, ximg = (size_t)b * p.IH * p.IW;

    // ---- staging: `npx` pixels = whole rows l0.. of a ring, pixel pl + 32u of them to this lane; every load before any use
    auto load_dy = [&](auto& rg, auto& ry, int l0, int npx, int pl) {
        constexpr int N = sizeof(rg) / sizeof(rg[0]);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int pix = min(pl + 32 * u, npx - 1);
            const int lr = pix / G::DYC, lc = pix - lr * G::DYC;
            const int oy = min(max(dy_r0 + l0 + lr, 0), p.OH - 1), ox = min(max(dy_c0 + lc, 0), p.OW - 1);
            const size_t r = dyimg + (size_t)oy * p.OW + ox;
            rg[u] = ldg4(p.dy.g + r * p.dy.ldg + cc);
            ry[u] = two ? ldg4(p.dy.y + r * p.dy.ldy + cc) : f4zero();
        }
    };
    auto store_dy = [&](const auto& rg, const auto& ry, int l0, int slot0, int npx, int pl) {
        constexpr int N = sizeof(rg) / sizeof(rg[0]);
        const GsrcCoef4 gk = gcoef();
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int pix = pl + 32 * u;
            if (pix < npx) {
                const int lr = pix / G::DYC, lc = pix - lr * G::DYC;
                const int oy = dy_r0 + l0 + lr, ox = dy_c0 + lc;
                const bool ok = cok && oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW;
                int slot = slot0 + lr;
                if (slot >= G::DYR) slot -= G::DYR;
                dyt[(slot * G::DYC + lc) * WPQ + q] = ok ? gsrc_apply4(rg[u], ry[u], gk) : f4zero();
            }
        }
    };
    auto load_x = [&](auto& rx, int l0, int npx, int pl) {
        constexpr int N = sizeof(rx) / sizeof(rx[0]);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int pix = min(pl + 32 * u, npx - 1);
            const int lr = pix / G::XC, lc = pix - lr * G::XC;
            const int iy = min(max(x_r0 + l0 + lr, 0), p.IH - 1), ix = min(max(x_c0 + lc, 0), p.IW - 1);
            rx[u] = ldg4(p.in.x + (ximg + (size_t)iy * p.IW + ix) * p.in.ld + cc);
        }
    };
    auto store_x = [&](const auto& rx, int slot0, int npx, int pl) {       // raw: masked / transformed where it is read
        constexpr int N = sizeof(rx) / sizeof(rx[0]);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int pix = pl + 32 * u;
            if (pix < npx) {
                const int lr = pix / G::XC, lc = pix - lr * G::XC;
                int slot = slot0 + lr;
                if (slot >= G::XR) slot -= G::XR;
                xt[(slot * G::XC + lc) * XPQ + q] = rx[u];
            }
        }
    };
    auto ring = [&](int slot, int R) { return slot >= R ? slot - R : slot; };

    // ---- the rows of step 0 (rings filled from local row 0)
    {
        constexpr int PDY = G::DYR * G::DYC, PX = G::XR * G::XC;
        float4 rg[(PDY + 31) / 32], ry[(PDY + 31) / 32], rx[(PX + 31) / 32];
        load_dy(rg, ry, 0, PDY, pl);
        load_x(rx, 0, PX, pl);
        store_dy(rg, ry, 0, 0, PDY, pl);
        store_x(rx, 0, PX, pl);
    }
    float4 dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4zero();
    float4 s1 = f4zero(), s2 = f4zero();
    __syncthreads();

    constexpr int NPDY = G::NDY * G::DYC, NPX = G::NX * G::XC;          // new pixels per step
    constexpr int UDY = (NPDY + 31) / 32, UX = (NPX + 31) / 32;
    int sdy = 0, sx = 0;                                                 // ring slot of the window's first row (local row k * NDY / k * NX)
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        const bool more = k + 1 < nsteps;                               // uniform
        // the lane -> pixel mapping is the same in every step, and left alone the compiler hoists every clamped coordinate and
        // 64-bit address derived from it out of the loop (~70 VGPRs held across the walk: spills); an opaque copy of the lane
        // index per iteration keeps them temporaries
        int plv = t >> 3;
        asm volatile("" : "+v"(plv));
        float4 rg[UDY], ry[UDY], rx[UX];
        if (more) {                                                     // the rows step k + 1 adds: local rows DYR + k*NDY .. / XR + k*NX ..
            load_dy(rg, ry, G::DYR + k * G::NDY, NPDY, plv);
            load_x(rx, G::XR + k * G::NX, NPX, plv);
        }
        const int y0 = ys + k * G::TH;                                   // first input row of this step
        const int yend = ys + rows_here;
        // ---- input gradient: lane pl -> input rows 2*(pl/8) + {0,1}, input columns 2*(pl%8) + {0,1}.  An even input row
        // meets kernel row 1 of output row rp, the odd one kernel row 0 of output row rp + 1 and kernel row 2 of rp; an even
        // input column kernel column 1 of output column cg, the odd one kernel column 0 of cg + 1 and kernel column 2 of cg
        const int rp = plv >> 3, cg = plv & 7;
        float4 da[2], db[2];
        {
            const int sa = ring(sdy + rp, G::DYR), sb = ring(sdy + rp + 1, G::DYR);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                da[j] = dyt[(sa * G::DYC + cg + j) * WPQ + q];
                db[j] = dyt[(sb * G::DYC + cg + j) * WPQ + q];
            }
        }
        float4 wk[9];
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) wk[k9] = wtab[q][k9];
        float4 dxv[2][2];                                                // [row parity][column parity]
        dxv[0][0] = f4fma(da[0], wk[1 * 3 + 1], f4zero());
        dxv[0][1] = f4fma(da[1], wk[1 * 3 + 0], f4fma(da[0], wk[1 * 3 + 2], f4zero()));
        dxv[1][0] = f4fma(db[0], wk[0 * 3 + 1], f4fma(da[0], wk[2 * 3 + 1], f4zero()));
        dxv[1][1] = f4fma(db[1], wk[0 * 3 + 0], f4fma(db[0], wk[0 * 3 + 2], f4fma(da[1], wk[2 * 3 + 0], f4fma(da[0], wk[2 * 3 + 2], f4zero()))));
        const SinkCoef4 sk = scoef();
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy) {
            const int y = y0 + 2 * rp + dyy;
            const int slx = ring(sx + 2 * rp + dyy + 1, G::XR);          // the input ring starts one row above y0
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int x = x0 + 2 * cg + u;
                const float4 raw = xt[(slx * G::XC + 2 * cg + u + 1) * XPQ + q];
                if (cok && y < yend && x < p.IW) {
                    const size_t r = ximg + (size_t)y * p.IW + x;
                    float4 d = dxv[dyy][u];
                    if (p.sink.accumulate) d = f4add(d, ldg4(sink_old(p.sink) + r * p.sink.ldg + c0));
                    stg4(p.sink.g + r * p.sink.ldg + c0, sink_apply4(p.sink, sk, d, raw, s1, s2));
                }
            }
        }
        // ---- weight gradient: lane pl -> owned output (row pl/8, column pl%8); beyond the segment's rows the dy' ring only
        // holds the halo of the next segment
        const bool own = 2 * rp < yend - y0;
        const float4 dyo = own ? da[0] : f4zero();                        // = dy'(rp, cg)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int slx = ring(sx + 2 * rp + ky, G::XR);
            const bool rok = (unsigned)(y0 + 2 * rp + ky - 1) < (unsigned)p.IH;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 raw = xt[(slx * G::XC + 2 * cg + kx) * XPQ + q];
                const bool ok = rok && (unsigned)(x0 + 2 * cg + kx - 1) < (unsigned)p.IW;
                const float4 xv = ok ? src_apply4(raw, xk) : f4zero();
                dwacc[ky * 3 + kx] = f4fma(dyo, xv, dwacc[ky * 3 + kx]);
            }
        }
        __syncthreads();                       // every lane is done reading the rows the next step replaces
        if (more) {
            // the new rows take the slots of the window's oldest NDY / NX rows (the ring keeps the last DYR - NDY / XR - NX)
            store_dy(rg, ry, G::DYR + k * G::NDY, sdy, NPDY, plv);
            store_x(rx, sx, NPX, plv);
            sdy = ring(sdy + G::NDY, G::DYR);
            sx = ring(sx + G::NX, G::XR);
            __syncthreads();
        }
    }

    // ---- the workgroup's 9 weight-gradient taps: [tap][pixel lane][quad] float4 over the rings, summed over the 32 lanes
#pragma unroll
    for (int k = 0; k < 9; ++k) lds4[(k * 32 + pl) * 8 + q] = dwacc[k];
    __syncthreads();
    for (int o = t; o < 9 * 32; o += 256) {            // output o -> (tap o / 32, quad (o % 32) / 4, component o % 4)
        const int k = o >> 5, qc = o & 31;
        const float* base = reinterpret_cast<const float*>(lds4 + k * 32 * 8) + qc;
        float a = 0.f;
#pragma unroll
        for (int l = 0; l < 32; ++l) a += base[l * 32];
        const int c = (cq0 + (qc >> 2)) * 4 + (qc & 3);
        if (c < p.C) p.wpartial[((size_t)tIdx * p.C + c) * 9 + k] = a;
    }
    if (p.sink.finalize && p.sink.mean && p.sink.partial) {     // uniform
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds4);
        float* slot = red + (pl * 8 + q) * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < 64) {
            const int qq = t >> 3, comp = t & 7;
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) a += red[(l * 8 + qq) * 8 + comp];
            const int c = (cq0 + qq) * 4 + (comp & 3);
            if (c < p.C) {
                const SinkAcc sa = sink_acc(p.sink);
                if (sa.acc) bn_stat_add(sa.acc, sa.bad, sa.slots, p.sink.ldp, tIdx, comp >> 2, c, a);       // no finalize launch follows
                else p.sink.partial[(size_t)tIdx * 2 * p.sink.ldp + (comp >> 2) * p.sink.ldp + c] = a;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ stride 1
// A step = 4 input rows x 16 columns (two pixels per lane).  Rings of 6 rows x 18 columns: dy', the RAW input (the sink's y) and
// the TRANSFORMED input x' (zero padding applied) -- transforming at the fragment read costs 6 transforms per element (every
// element sits in the window of ~6 lanes) and made the kernel issue-bound (1345 instructions per step, 1.6-2.2 TB/s).  Staging
// is column-per-lane: pixel lane c < 18 owns ring column c and fetches one element of each new ROW per load, so the row
// coordinate, its clamp / padding mask and the ring slot are wave-uniform and the per-lane part of every address is a
// constant of the walk (no divisions, no 64-bit lane arithmetic in the loop).
__global__ __launch_bounds__(256, 2) void dw_bwd_walk_s1_kernel(DwWP p) {
    constexpr int R = 6, CW = 18, TH = 4;
    constexpr int RING_F4 = R * CW * WPQ;
    constexpr int LDS_F4 = 3 * RING_F4 > 9 * 32 * 8 ? 3 * RING_F4 : 9 * 32 * 8;           // the tap reduction reuses the rings
    __shared__ float4 lds4[LDS_F4];
    __shared__ float4 wtab[WKQ][9];
    __shared__ float4 ctab[WKQ][6];                   // p, q, mu, r | mean, invstd
    float4* dyt = lds4;
    float4* xraw = lds4 + RING_F4;
    float4* xtr = lds4 + 2 * RING_F4;
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;
    int tIdx, bcy;
    xcd_block(bcy, tIdx);        // grid = (channel chunks, walkers): see dw_walk_launch
    const int strip = tIdx % p.strips, seg = (tIdx / p.strips) % p.segs, b = tIdx / (p.strips * p.segs);
    const int ys = seg * p.seg_rows, x0 = strip * 16;
    const int rows_here = min(p.seg_rows, p.IH - ys);
    const int nsteps = (rows_here + TH - 1) / TH;
    const int yend = ys + rows_here;
    const int cq0 = bcy * WKQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;
    const SrcCoef4 xk = src_coef4(p.in, cc, p.C);
    const bool two = p.dy.y != nullptr;
    const bool dy_stat = p.dys.acc != nullptr;         // uniform: dy's coefficients from the sink kernel's accumulators (no finalize launch)
    if (dy_stat && t < 4 * WKQ) {                      // one thread per channel of the chunk: one memory round trip, one fp64 chain
        const int c = cq0 * 4 + t;
        GsrcCoef1 k1{1.f, 0.f, 0.f, 0.f};
        if (c < p.C) k1 = gsrc_coef1_bstat(p.dy, p.dys, c, tIdx == 0);
        float* ct = reinterpret_cast<float*>(&ctab[t >> 2][0]) + (t & 3);
        ct[0] = k1.p; ct[4] = k1.q; ct[8] = k1.mu; ct[12] = k1.r;
    }
    if (pl == 8) {             // (wave 1: beside the hand-over derivation above, which runs in wave 0)
        const SinkCoef4 k0 = sink_coef4(p.sink, cc, p.C);
        if (!dy_stat) {
            const GsrcCoef4 g0 = gsrc_coef4(p.dy, cc, p.C);
            ctab[q][0] = make_float4(g0.p[0], g0.p[1], g0.p[2], g0.p[3]);
            ctab[q][1] = make_float4(g0.q[0], g0.q[1], g0.q[2], g0.q[3]);
            ctab[q][2] = make_float4(g0.mu[0], g0.mu[1], g0.mu[2], g0.mu[3]);
            ctab[q][3] = make_float4(g0.r[0], g0.r[1], g0.r[2], g0.r[3]);
        }
        ctab[q][4] = make_float4(k0.mean[0], k0.mean[1], k0.mean[2], k0.mean[3]);
        ctab[q][5] = make_float4(k0.inv[0], k0.inv[1], k0.inv[2], k0.inv[3]);
        {
            // all 36 taps of the quad requested before the first one is used (unconditional loads at clamped channels: with the loads
            // inside per-element conditions the compiler waited for every tap's four loads before it issued the next four -- nine
            // serialised round trips at the head of every workgroup)
            float e[9][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wp = p.w + (size_t)min(cc + j, p.C - 1) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) e[k][j] = wp[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) e[k][j] = (cok && c0 + j < p.C) ? e[k][j] : 0.f;
                wtab[q][k] = make_float4(e[k][0], e[k][1], e[k][2], e[k][3]);
            }
        }
    }
    auto ring = [&](int slot) { return slot >= R ? slot - R : slot; };
    // ---- staging lane: ring column pl (plane column x0 - 1 + pl), lanes 18..31 idle
    const bool lane_on = pl < CW;
    const int lc = min(pl, CW - 1);
    const int gx = x0 - 1 + lc;
    const bool cin = cok && (unsigned)gx < (unsigned)p.IW;                 // inside the image (zero padding otherwise)
    const int gxc = min(max(gx, 0), p.IW - 1);
    const int og = gxc * p.dy.ldg + cc, oy_ = gxc * p.dy.ldy + cc, ox_ = gxc * p.in.ld + cc;      // element offsets inside a plane row
    const int lcol = lc * WPQ + q;
    const size_t img = (size_t)b * p.IH * p.IW;                            // stride 1: one plane geometry for dy and the input
    // N rows from local row l0 (plane row ys - 1 + l0 + u): all loads first, then transform + ring stores
    auto load_rows = [&](auto& rg, auto& ry, auto& rx, int l0) {
        constexpr int N = sizeof(rx) / sizeof(rx[0]);
        if (lane_on) {
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const size_t rowb = img + (size_t)min(max(ys - 1 + l0 + u, 0), p.IH - 1) * p.IW;      // uniform
                rg[u] = ldg4(p.dy.g + rowb * p.dy.ldg + og);
                rx[u] = ldg4(p.in.x + rowb * p.in.ld + ox_);
            }
            if (two) {
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const size_t rowb = img + (size_t)min(max(ys - 1 + l0 + u, 0), p.IH - 1) * p.IW;
                    ry[u] = ldg4(p.dy.y + rowb * p.dy.ldy + oy_);
                }
            }
        }
    };
    auto store_rows = [&](const auto& rg, const auto& ry, const auto& rx, int l0, int slot0) {
        constexpr int N = sizeof(rx) / sizeof(rx[0]);
        if (lane_on) {
            GsrcCoef4 gk;
            gk.two = two;
            {
                const float4 a = ctab[q][0], b2 = ctab[q][1], c2 = ctab[q][2], d = ctab[q][3];
                gk.p[0] = a.x; gk.p[1] = a.y; gk.p[2] = a.z; gk.p[3] = a.w;
                gk.q[0] = b2.x; gk.q[1] = b2.y; gk.q[2] = b2.z; gk.q[3] = b2.w;
                gk.mu[0] = c2.x; gk.mu[1] = c2.y; gk.mu[2] = c2.z; gk.mu[3] = c2.w;
                gk.r[0] = d.x; gk.r[1] = d.y; gk.r[2] = d.z; gk.r[3] = d.w;
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const bool ok = cin && (unsigned)(ys - 1 + l0 + u) < (unsigned)p.IH;
                int slot = slot0 + u;
                while (slot >= R) slot -= R;                                // uniform
                const int idx = slot * (CW * WPQ) + lcol;
                dyt[idx] = ok ? gsrc_apply4(rg[u], two ? ry[u] : f4zero(), gk) : f4zero();
                xraw[idx] = rx[u];
                xtr[idx] = ok ? src_apply4(rx[u], xk) : f4zero();
            }
        }
    };
    __syncthreads();                                   // the tables
    {
        float4 rg[R], ry[R], rx[R];
        load_rows(rg, ry, rx, 0);
        store_rows(rg, ry, rx, 0, 0);
    }
    float4 dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4zero();
    float4 s1 = f4zero(), s2 = f4zero();
    __syncthreads();

    // compute lane: input column pl % 16, rows 2 * (pl / 16) + {0, 1}: adjacent lanes read adjacent ring pixels (144 bytes apart:
    // conflict-free 16-lane ds_read_b128 groups; lanes two pixels apart -- 288 bytes -- made 47 % of the LDS cycles bank conflicts)
    const int col = pl & 15, rp = pl >> 4;
    const int wcol = col * WPQ + q;
    int s0 = 0;                                        // ring slot of the window's first row (local row 4k)
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        const bool more = k + 1 < nsteps;              // uniform
        float4 rg[TH], ry[TH], rx[TH];
        if (more) load_rows(rg, ry, rx, R + k * TH);   // the rows step k + 1 adds: local rows 6 + 4k ..
        const int y0 = ys + k * TH;
        float4 dx[2], dyc[2], yc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int sl = ring(s0 + 2 * rp + u + 1) * (CW * WPQ) + wcol + WPQ;
            dx[u] = f4zero(); dyc[u] = dyt[sl]; yc[u] = xraw[sl];
        }
        // window rows 2rp .. 2rp+3 of the step (plane rows y0 + 2rp - 1 ..), one at a time (left to the scheduler, the LDS reads
        // of all rows are issued first and their values are live at once): row wr meets kernel row wr - u of owned row u
#pragma unroll
        for (int wr = 0; wr < 4; ++wr) {               // input gradient: dx[u] += dy'[r-1+ky][c-1+j] * w[2-ky][2-j], ky = wr - u
            const int sl = ring(s0 + 2 * rp + wr) * (CW * WPQ) + wcol;
            float4 dr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dr[j] = dyt[sl + j * WPQ];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ky = wr - u;
                if (ky >= 0 && ky <= 2) {
                    const float4 w3[3] = {wtab[q][(2 - ky) * 3 + 2], wtab[q][(2 - ky) * 3 + 1], wtab[q][(2 - ky) * 3 + 0]};
#pragma unroll
                    for (int j = 0; j < 3; ++j) dx[u] = f4fma(dr[j], w3[j], dx[u]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int wr = 0; wr < 4; ++wr) {               // weight gradient: dw[ky][j] += dy'[r][c] * x'[r-1+ky][c-1+j]
            const int sl = ring(s0 + 2 * rp + wr) * (CW * WPQ) + wcol;
            float4 xr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xr[j] = xtr[sl + j * WPQ];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ky = wr - u;
                if (ky >= 0 && ky <= 2) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) dwacc[ky * 3 + j] = f4fma(dyc[u], xr[j], dwacc[ky * 3 + j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.sink.accumulate) {                        // uniform; rare (a tensor with several consumers): the running gradient
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t r = img + (size_t)min(y0 + 2 * rp + u, p.IH - 1) * p.IW + min(x0 + col, p.IW - 1);
                dx[u] = f4add(dx[u], ldg4(sink_old(p.sink) + r * p.sink.ldg + cc));
            }
        }
        {
            SinkCoef4 sk;
            sk.affine = p.sink.scale != nullptr;
            sk.stats = p.sink.mean != nullptr && p.sink.partial != nullptr;
            sk.act = act_param(p.sink.act);
            sk.q = p.sink.qstate != nullptr;
            sk.qinv = sk.q ? 1.f / p.sink.qstate[2] : 0.f;
            sk.qzp = sk.q ? p.sink.qstate[3] : 0.f;
            const float4 m = ctab[q][4], iv = ctab[q][5];
            sk.mean[0] = m.x; sk.mean[1] = m.y; sk.mean[2] = m.z; sk.mean[3] = m.w;
            sk.inv[0] = iv.x; sk.inv[1] = iv.y; sk.inv[2] = iv.z; sk.inv[3] = iv.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) { sk.a[j] = sk.affine ? xk.a[j] : 1.f; sk.b[j] = sk.affine ? xk.b[j] : 0.f; }   // same tensor as `in`
            const int ox = x0 + col;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int oy = y0 + 2 * rp + u;
                if (cok && oy < yend && ox < p.IW) {
                    const size_t r = img + (size_t)oy * p.IW + ox;
                    stg4(p.sink.g + r * p.sink.ldg + c0, sink_apply4(p.sink, sk, dx[u], yc[u], s1, s2));
                }
            }
        }
        __syncthreads();                               // every lane is done reading the rows the next step replaces
        if (more) {
            store_rows(rg, ry, rx, R + k * TH, s0);    // the new rows take the slots of the window's oldest 4 rows
            s0 = ring(s0 + TH);
        }
        if (more) __syncthreads();
    }

    // ---- the workgroup's 9 weight-gradient taps: [tap][pixel lane][quad] float4 over the rings, summed over the 32 lanes
#pragma unroll
    for (int k = 0; k < 9; ++k) lds4[(k * 32 + pl) * 8 + q] = dwacc[k];
    __syncthreads();
    for (int o = t; o < 9 * 32; o += 256) {            // output o -> (tap o / 32, quad (o % 32) / 4, component o % 4)
        const int k = o >> 5, qc = o & 31;
        const float* base = reinterpret_cast<const float*>(lds4 + k * 32 * 8) + qc;
        float a = 0.f;
#pragma unroll
        for (int l = 0; l < 32; ++l) a += base[l * 32];
        const int c = (cq0 + (qc >> 2)) * 4 + (qc & 3);
        if (c < p.C) p.wpartial[((size_t)tIdx * p.C + c) * 9 + k] = a;
    }
    if (p.sink.finalize && p.sink.mean && p.sink.partial) {     // uniform
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds4);
        float* slot = red + (pl * 8 + q) * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < 64) {
            const int qq = t >> 3, comp = t & 7;
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) a += red[(l * 8 + qq) * 8 + comp];
            const int c = (cq0 + qq) * 4 + (comp & 3);
            if (c < p.C) {
                const SinkAcc sa = sink_acc(p.sink);
                if (sa.acc) bn_stat_add(sa.acc, sa.bad, sa.slots, p.sink.ldp, tIdx, comp >> 2, c, a);       // no finalize launch follows
                else p.sink.partial[(size_t)tIdx * 2 * p.sink.ldp + (comp >> 2) * p.sink.ldp + c] = a;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, stride 1
// The same walk for the forward pass: ONE ring (the transformed input x' = act(scale * y + shift), zero padding applied at
// staging), a step = 4 output rows x 16 columns, two outputs per lane (column pl % 16, rows 2 * (pl / 16) + {0, 1}; taps of an
// output in (ky, kx) order as in every other depthwise kernel: bit-identical outputs).  Every input element is fetched once
// (+ the 2 / 16 column halo), the next step's rows are in flight during the arithmetic, and the workgroup leaves ONE
// statistics row (or one set of accumulator atomics: pq_bn_stat hand-over, both roles, any plane size).
struct DwFWP {
    pq_src in; const float* w; float* out; float* partial;
    int ldo, ldp, B, H, W, C, C4, strips, segs, seg_rows, minmax;
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;
};

__global__ __launch_bounds__(256, 3) void dw_fwd_walk_s1_kernel(DwFWP p) {
    constexpr int R = 6, CW = 18, TH = 4;
    constexpr int RING_F4 = R * CW * WPQ;
    constexpr int LDS_F4 = RING_F4 > 32 * 8 * 2 + 16 ? RING_F4 : 32 * 8 * 2 + 16;      // ring | statistics reduction | coefficient staging
    __shared__ float4 xtr[LDS_F4 < 544 ? 544 : LDS_F4];                                  // >= 2048 + 64 floats for src_coef4_chunk32
    __shared__ float4 wtab[WKQ][9];
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;
    int tIdx, bcy;
    xcd_block(bcy, tIdx);        // grid = (channel chunks, walkers): see dw_walk_launch
    const int strip = tIdx % p.strips, seg = (tIdx / p.strips) % p.segs, b = tIdx / (p.strips * p.segs);
    const int ys = seg * p.seg_rows, x0 = strip * 16;
    const int rows_here = min(p.seg_rows, p.H - ys);
    const int nsteps = (rows_here + TH - 1) / TH;
    const int yend = ys + rows_here;
    const int cq0 = bcy * WKQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;
    // (with p.in.stat: the workgroup derives its 32 channels' coefficients from the producer's accumulators; workgroup 0 of the
    // channel chunk also stores the vectors for every later consumer)
    // the 36 taps of the quad are requested BEFORE the derivation (unconditional loads at clamped channels, all in flight at once; they
    // share the round trip of the accumulator loads: the derivation's first barrier drains both) and written to LDS after it
    float e[9][4];
    if (pl == 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wp = p.w + (size_t)min(cc + j, p.C - 1) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k][j] = wp[k];
        }
    }
    const SrcCoef4 xk = p.in.stat.acc ? src_coef4_chunk32(p.in, cq0 * 4, q, p.C, tIdx == 0, reinterpret_cast<float*>(xtr))
                                      : src_coef4(p.in, cc, p.C);
    if (pl == 8) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[k][j] = (cok && c0 + j < p.C) ? e[k][j] : 0.f;
            wtab[q][k] = make_float4(e[k][0], e[k][1], e[k][2], e[k][3]);
        }
    }
    auto ring = [&](int slot) { return slot >= R ? slot - R : slot; };
    const bool lane_on = pl < CW;
    const int lc = min(pl, CW - 1);
    const int gx = x0 - 1 + lc;
    const bool cin = cok && (unsigned)gx < (unsigned)p.W;
    const int ox_ = min(max(gx, 0), p.W - 1) * p.in.ld + cc;
    const int lcol = lc * WPQ + q;
    const size_t img = (size_t)b * p.H * p.W;
    auto load_rows = [&](auto& rx, int l0) {
        constexpr int N = sizeof(rx) / sizeof(rx[0]);
        if (lane_on) {
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const size_t rowb = img + (size_t)min(max(ys - 1 + l0 + u, 0), p.H - 1) * p.W;      // uniform
                rx[u] = ldg4(p.in.x + rowb * p.in.ld + ox_);
            }
        }
    };
    auto store_rows = [&](const auto& rx, int l0, int slot0) {
        constexpr int N = sizeof(rx) / sizeof(rx[0]);
        if (lane_on) {
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const bool ok = cin && (unsigned)(ys - 1 + l0 + u) < (unsigned)p.H;
                int slot = slot0 + u;
                while (slot >= R) slot -= R;                                // uniform
                xtr[slot * (CW * WPQ) + lcol] = ok ? src_apply4(rx[u], xk) : f4zero();
            }
        }
    };
    __syncthreads();                                   // the coefficient staging of src_coef4_chunk32 is done with the buffer
    {
        float4 rx[R];
        load_rows(rx, 0);
        store_rows(rx, 0, 0);
    }
    float4 s1 = f4zero(), s2 = f4zero();
    const float inf = __builtin_huge_valf();
    float4 mn = make_float4(inf, inf, inf, inf), mx = make_float4(-inf, -inf, -inf, -inf);
    __syncthreads();

    const int col = pl & 15, rp = pl >> 4;
    const int wcol = col * WPQ + q;
    int s0 = 0;
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        const bool more = k + 1 < nsteps;              // uniform
        float4 rx[TH];
        if (more) load_rows(rx, R + k * TH);
        const int y0 = ys + k * TH;
        float4 acc[2] = {f4zero(), f4zero()};
#pragma unroll
        for (int wr = 0; wr < 4; ++wr) {               // window row wr (plane row y0 + 2rp - 1 + wr) meets kernel row wr - u of output u
            const int sl = ring(s0 + 2 * rp + wr) * (CW * WPQ) + wcol;
            float4 xr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xr[j] = xtr[sl + j * WPQ];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ky = wr - u;
                if (ky >= 0 && ky <= 2) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[u] = f4fma(xr[j], wtab[q][ky * 3 + j], acc[u]);
                }
            }
        }
        const int ox = x0 + col;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int oy = y0 + 2 * rp + u;
            if (cok && oy < yend && ox < p.W) {
                stg4(p.out + (img + (size_t)oy * p.W + ox) * p.ldo + c0, acc[u]);
                s1 = f4add(s1, acc[u]);
                s2 = f4fma(acc[u], acc[u], s2);
                mn = f4min(mn, acc[u]); mx = f4max(mx, acc[u]);
            }
        }
        __syncthreads();                               // every lane is done reading the rows the next step replaces
        if (more) {
            store_rows(rx, R + k * TH, s0);
            s0 = ring(s0 + TH);
            __syncthreads();
        }
    }
    if (p.partial || p.stat_acc) {      // uniform
        const int planes = p.minmax ? 4 : 2;
        float* red = reinterpret_cast<float*>(xtr);
        float* slot = red + (pl * 8 + q) * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < 64) {     // thread -> (quad t>>3, component t&7)
            const int qq = t >> 3, comp = t & 7;
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) a += red[(l * 8 + qq) * 8 + comp];
            const int c = (cq0 + qq) * 4 + (comp & 3);
            if (c < p.C) {
                if (p.stat_acc) bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, p.ldp, tIdx, comp >> 2, c, a);
                else p.partial[(size_t)tIdx * planes * p.ldp + (comp >> 2) * p.ldp + c] = a;
            }
        }
        if (p.minmax) {   // planes 2 / 3: per-channel extremes of the workgroup's outputs (QAT observers)
            __syncthreads();
            slot[0] = mn.x; slot[1] = mn.y; slot[2] = mn.z; slot[3] = mn.w;
            slot[4] = mx.x; slot[5] = mx.y; slot[6] = mx.z; slot[7] = mx.w;
            __syncthreads();
            if (t < 64) {
                const int qq = t >> 3, comp = t & 7;
                float a = comp < 4 ? inf : -inf;
#pragma unroll
                for (int l = 0; l < 32; ++l) { const float v = red[(l * 8 + qq) * 8 + comp]; a = comp < 4 ? fminf(a, v) : fmaxf(a, v); }
                const int c = (cq0 + qq) * 4 + (comp & 3);
                if (c < p.C) p.partial[(size_t)tIdx * 4 * p.ldp + (2 + (comp >> 2)) * p.ldp + c] = a;
            }
        }
    }
}

// out[j] = sum over the workgroups' rows (few: one per strip segment); 32 columns x 32 row lanes per block, fixed order
__global__ __launch_bounds__(1024) void dww_partial_reduce_kernel(const float* part, int nblk, int n, float* out) {
    __shared__ float red[32][33];
    const int j = blockIdx.x * 32 + threadIdx.x;
    float s = 0.f;
    if (j < n) {
        int k = threadIdx.y;
        for (; k + 7 * 32 < nblk; k += 8 * 32) {       // 8 independent loads in flight (the loop is pure load latency: one at a
            float v[8];                                // time, the 1024-row reduction of the first layer took 270 us at the step's tail)
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

// Long partial lists (the 256 x 256 planes leave 512 rows): rows r = s, s + ns, ... are first summed INTO row s by (column tile, s)
// workgroups spread over the chip (fixed order; a row is only touched by the workgroups of its own slice, so in place is safe), and
// dww_partial_reduce_kernel then sums ns rows.  One launch over 9 workgroups walking 512 rows took 79 us at the tail of the step.
__global__ __launch_bounds__(256) void dww_slice_rows_kernel(float* part, int nblk, int n, int ns) {
    const int e = blockIdx.x * 256 + threadIdx.x, s0 = blockIdx.y;
    if (e >= n) return;
    float a = 0.f;
    int r = s0;
    for (; r + 7 * ns < nblk; r += 8 * ns) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + u * ns) * n + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; r < nblk; r += ns) a += part[(size_t)r * n + e];
    part[(size_t)s0 * n + e] = a;
}

}  // namespace pq

using namespace pq;

// Walk geometry: 16-column strips; the rows of an image are cut into at most two segments of whole steps.
struct WalkPlan { int strips, segs, seg_rows, nblk, chunks; };
static WalkPlan walk_plan(int B, int IH, int IW, int C, int stride) {
    WalkPlan w{};
    const int th = stride == 1 ? 4 : 8;
    w.strips = (IW + 15) / 16;
    w.chunks = ((C + 3) / 4 + WKQ - 1) / WKQ;
    const int steps_total = (IH + th - 1) / th;
    const long base = (long)B * w.strips * w.chunks;
    // measured (scripts/dw_bench.py, every layer shape of the nets, 1 .. 16 segments): long walks win -- two segments once the
    // launch would otherwise leave workgroup slots empty (512 fit the chip) or the plane has >= 16 steps, one otherwise; 3+ lose
    long segs = (512 + base - 1) / base;
    if (segs < 2 && steps_total >= 16) segs = 2;
    const long maxsegs = steps_total / 4 > 0 ? steps_total / 4 : 1;
    if (segs > maxsegs) segs = maxsegs;
    if (segs > 2) segs = 2;
    if (segs < 1) segs = 1;
    const int seg_steps = (int)((steps_total + segs - 1) / segs);
    w.segs = (steps_total + seg_steps - 1) / seg_steps;
    w.seg_rows = seg_steps * th;
    w.nblk = B * w.strips * w.segs;
    return w;
}
int dw_walk_blocks(int B, int IH, int IW, int C, int stride) { return walk_plan(B, IH, IW, C, stride).nblk; }

extern "C" int pq_dw3x3_bwd_reduce_item(const pq_dw3x3_bwd_args* a, pq_reduce_item* out) {
    PQ_CHECK_ARG(a && out && a->dw && a->scratch && a->part == 6, "part-6 arguments expected");
    PQ_CHECK_ARG(a->B > 0 && a->IH > 0 && a->IW > 0 && a->C > 0 && (a->stride == 1 || a->stride == 2), "bad shape");
    const WalkPlan w = walk_plan(a->B, a->IH, a->IW, a->C, a->stride);
    const int n = a->C * 9;
    *out = pq_reduce_item{a->scratch, a->dw, nullptr, (int64_t)n, 0, w.nblk, n, 0, n, 0, 0, 0, 0};
    return PQ_OK;
}

// part 5 of pq_dw3x3_bwd (kernel; rows left in scratch) / part 6 (their reduction into dw); dwtile.hip checks the arguments
int dw_walk_launch(const pq_dw3x3_bwd_args* a, void* stream) {
    const WalkPlan w = walk_plan(a->B, a->IH, a->IW, a->C, a->stride);
    hipStream_t st = (hipStream_t)stream;
    if (a->part == 6) {
        const int n = a->C * 9;
        int rows = w.nblk;
        if (rows > 32) {
            constexpr int NS = 16;
            PQ_LAUNCH(dww_slice_rows_kernel, dim3((n + 255) / 256, NS), dim3(256), 0, st, a->scratch, rows, n, NS);
            rows = NS;
        }
        PQ_LAUNCH(dww_partial_reduce_kernel, dim3((n + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, rows, n, a->dw);
        return PQ_OK;
    }
    DwWP p{};
    p.in = a->in; p.dy = a->dy; p.w = a->w; p.sink = a->dx; p.wpartial = a->scratch;
    p.dys = a->dy_stat;
    if (p.dys.slots < 1) p.dys.slots = 1;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    p.strips = w.strips; p.segs = w.segs; p.seg_rows = w.seg_rows;
    // grid = (channel chunks, walkers): the chunk index is the FASTEST one in the (XCD-aware) block numbering, so the workgroups that
    // walk the same strip for neighbouring channel chunks start together on one XCD.  Where a pixel is not a whole number of
    // 128-byte lines (144 channels = 576 bytes) every second 128-byte chunk straddles two lines, and the neighbour's half then comes
    // from L2 instead of a second HBM fetch: 128 x 128 x 144 backward 0.213 -> 0.161 ms (2.8 -> 3.7 TB/s), forward 0.106 -> 0.082;
    // neutral on the line-aligned shapes (scripts/dw_bench.py).
    dim3 grid(w.chunks, w.nblk), blk(256);
    if (a->stride == 1) PQ_LAUNCH_BINDABLE(dw_bwd_walk_s1_kernel, grid, blk, 0, st, p);
    else PQ_LAUNCH_BINDABLE(dw_bwd_walk_s2_kernel, grid, blk, 0, st, p);
    return PQ_OK;
}

// pq_bn_bstat_ok for PQ_OP_DW_BWD (consumer role): the walking kernels derive dy's coefficients per channel quad
int dw_bwd_bstat_ok(const pq_dw3x3_bwd_args* a) { return a->part == 5 && a->dy.y != nullptr && a->dy.mu != nullptr; }

// forward, stride 1 (pq_dw3x3_fwd in dwtile.hip checks the arguments and decides)
int dw_fwd_walk_blocks(int B, int H, int W, int C) { return walk_plan(B, H, W, C, 1).nblk; }
int dw_fwd_walk_launch(const pq_dw3x3_fwd_args* a, void* stream) {
    const WalkPlan w = walk_plan(a->B, a->IH, a->IW, a->C, 1);
    DwFWP p{};
    p.in = a->in; p.w = a->w; p.out = a->out; p.partial = a->partial; p.ldo = a->ldo; p.ldp = a->ldp;
    p.B = a->B; p.H = a->IH; p.W = a->IW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    p.strips = w.strips; p.segs = w.segs; p.seg_rows = w.seg_rows;
    p.minmax = (a->partial && a->minmax) ? 1 : 0;
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    if (w.nblk > 65535) { pq_set_error("pq_dw3x3_fwd: more than 65535 walkers (grid.y): split the batch"); return PQ_EUNSUPPORTED; }
    PQ_LAUNCH(dw_fwd_walk_s1_kernel, dim3(w.chunks, w.nblk), dim3(256), 0, (hipStream_t)stream, p);     // (see dw_walk_launch)
    return PQ_OK;
}
