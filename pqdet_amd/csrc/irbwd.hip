// Backward of the fused inverted-residual block (forward: irblock.hip): depthwise-3x3 backward + expand-1x1 backward in ONE pass,
// with the expanded tensor y_e = W_e x' RECOMPUTED from the block input (it was never stored) and its gradient g_e never stored either.
//
//   inputs : (g_d, y_d) the gradient pair of the depthwise output (dy' = p g + q (y - mu) + r, closed-form BatchNorm backward),
//            the block input x' = a x + b (I channels, linear), W_e [E][I], the BatchNorm of the expanded tensor (scale, shift, mean,
//            invstd, gamma), the depthwise taps.
//   per pixel of the INPUT plane:  y_e = W_e x'  (MFMA, bit-identical to the forward's value: same instruction, same k order),
//            u = scale y_e + shift, h = relu6(u);  t = sum_taps w_d dy'  (depthwise data gradient),  g_e = t * [0 < u < 6];
//            S1 += g_e, S2 += g_e xhat  (BatchNorm-backward sums of the expanded tensor),  dW_d[tap] += dy' * h.
//   The BatchNorm backward of the expanded tensor, dy_e = p_e g_e + q_e (y_e - mu_e) + r_e, needs S1 / S2 of the WHOLE batch -- but it is
//   linear in (g_e, y_e, 1) and y_e = W_e x', so both products of the expand layer's backward split into a part this pass can compute
//   (p_e = gamma_e invstd_e is known) and a part that only needs the I x I moments of x' (kept from the forward's pq_xstat_fwd):
//       dx' = W_e^T (p_e g_e)  +  A x' + b'        A = W_e^T diag(q_e) W_e,  b' = W_e^T (r_e - q_e mu_e)
//       dW_e = G1 + diag(q_e) (W_e Gxx - mu_e sx^T) + r_e sx^T       G1 = sum_px (p_e g_e) x'^T,  Gxx = sum x' x'^T,  sx = sum x'
//   This kernel emits Z1 = W_e^T (p_e g_e) per pixel (I floats) and per-workgroup partial rows of G1, dW_d, S1, S2; pq_xfold_bwd
//   (one small launch) reduces them and derives p_e / q_e / r_e, the parameter gradients, A and b'; pq_xfix_bwd adds A x' + b' while it
//   finalises the block input's gradient sink (activation derivative, BatchNorm-backward sums of the layer below).
//   Bytes per input pixel: 2 E / s^2 read (g_d, y_d) + I read + I written, instead of 4 E / s^2 + 2 E read, E written (walking depthwise
//   backward) and 2 E + I read, I written (fused narrow 1x1 backward).
//
// Structure: ONE 512-thread workgroup per (image, 16-column strip, row segment) owns ALL E channels.  Per step of TH input rows:
//   [B] expand: 16x16x4 MFMA, A = W_e rows from LDS, B = x' tile -> y_e tile in LDS                              | barrier
//   [C] depthwise stage: thread = (channel quad, pixel lane), owns its quad for the WHOLE walk (9 tap accumulators + 2 sums in
//       registers, as in dwwalk.hip); overwrites y_e by p_e g_e in place                                          | barrier
//   [A] global loads of the next step's dy' rows and x rows (registers, in flight across [D])
//   [D] MFMA stage 2: Z1 tiles (contraction over E) -> global; G1 tiles (contraction over the step's pixels) accumulate in
//       registers over the whole walk; the staged rows of [A] go to the dy' ring / the other x' tile                | barrier
//
// mirrors: backward of conv(1x1) -> BatchNorm -> ReLU6 -> conv(3x3, groups = C) of an inverted-residual block (model/parser.py:389-410).
#include <type_traits>
#include "common.h"

#ifndef XB_SKIP
#define XB_SKIP 0      // diagnostic builds only (scripts/irbwd_phases.sh): bit mask of phases to leave out -- results are then WRONG
#endif

namespace pq {

__device__ __forceinline__ f32x4 mfma16b(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct XbwdP {
    pq_src in; const float* we;
    const float* escale; const float* eshift; const float* emean; const float* einv; const float* egamma;
    pq_gsrc dy; const float* wd;
    float* z1; int ldz; float* part; float* gexp; int ldge;
    int B, IH, IW, OH, OW;
    int strips, segs, seg_rows, row_floats;
};

// a step = TH input rows x 16 columns; dy' ring: stride 1: rows y0 - 1 .. y0 + TH, columns x0 - 1 .. x0 + 16; stride 2 (output plane):
// rows y0 / 2 .. y0 / 2 + TH / 2, columns x0 / 2 .. x0 / 2 + 8.  DN new rows per step.  TH is chosen so that all of E fits the LDS.
template <int S, int TH_> struct XBG;
template <int TH_> struct XBG<1, TH_> { static constexpr int TH = TH_, DR = TH_ + 2, DC = 18, DN = TH_; };
template <int TH_> struct XBG<2, TH_> { static constexpr int TH = TH_, DR = TH_ / 2 + 1, DC = 9, DN = TH_ / 2; };

// S: stride; KS = I / 4; NQ = E / 4 (channel quads).  512 threads.  GE: also store g_e (tests).
// Between the request of a step's global loads [A] and their use at the end of the step there is NO vector-memory instruction inside
// a loop or a condition (the Z1 stores are straight-line code, the optional g_e store is a template flavour): the compiler can
// then COUNT the outstanding loads (s_waitcnt vmcnt(N)) instead of draining them in front of the first loop -- with the Z1 store
// inside a run-time loop it waited for the whole batch right after issuing it, every step.
template <int S, int KS, int NQ, int TH_, bool GE>
__global__ __launch_bounds__(512) void xdwb_kernel(XbwdP p) {
    using G = XBG<S, TH_>;
    static_assert(G::DR <= 2 * G::DN, "the first step's ring is filled in two batches");
    constexpr int I = 4 * KS, E = 4 * NQ, TH = G::TH;
    constexpr int NPXS = TH * 16, NPT = TH;                     // owned pixels / 16-pixel tiles (= rows) of a step
    constexpr int NI = (I + 15) / 16, NCT = E / 16;             // 16-wide tiles of the input / expanded channels
    constexpr int EP = E + 4, XP = I + 4, WP = I + 4;           // LDS pitches (floats)
    constexpr int NPL = 512 / NQ;                               // pixel lanes of the depthwise stage
    constexpr int KE = E / 4;                                   // k-steps of the Z1 contraction (lane group g: channels g * KE ..)
    constexpr int RING_F4 = G::DR * G::DC * NQ;
    constexpr int UD = (G::DN * G::DC + NPL - 1) / NPL;         // dy' pixels (float4 pairs of its quad) a thread stages per step
    constexpr int UX = (NPXS * KS + 511) / 512;                 // x float4 a thread stages per step
    constexpr int NUE = NPT * NCT, NUZ = NPT * NI, NUG = NCT * NI;       // work units of the three MFMA stages
    constexpr int MUG = (NUG + 7) / 8;
    static_assert(E % 16 == 0 && NPL >= 1, "E must be a multiple of 16");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* ring = smem;                                        // [DR][DC][NQ]            dy'
    float* yt = reinterpret_cast<float*>(ring + RING_F4);       // [NPXS][EP]              y_e, then p_e g_e
    float* xt = yt + NPXS * EP;                                 // [2][NPXS][XP]           x'
    float* wx = xt + 2 * NPXS * XP;                             // [E][WP]                 W_e
    float4* wtab = reinterpret_cast<float4*>(wx + E * WP);      // [NQ][9]                 depthwise taps
    float4* cd = wtab + NQ * 9;                                 // [NQ][4]                 p, q, mu, r of dy'
    float4* ce = cd + NQ * 4;                                   // [NQ][4]                 scale, shift, mean, invstd of the expanded tensor
    float4* cp = ce + NQ * 4;                                   // [NQ]                    p_e = gamma * invstd
    float4* cxa = cp + NQ;                                      // [KS][2]                 scale, shift of the block input
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int qd = t % NQ, pl = t / NQ;                         // depthwise stage: channel quad (fixed for the walk), pixel lane
    const bool dact = pl < NPL;
    int tIdx = blockIdx.x;
    tIdx = xcd_remap(tIdx, gridDim.x);
    const int strip = tIdx % p.strips, seg = (tIdx / p.strips) % p.segs, b = tIdx / (p.strips * p.segs);
    const int ys = seg * p.seg_rows, x0 = strip * 16;
    const int rows_here = min(p.seg_rows, p.IH - ys);
    const int nsteps = (rows_here + TH - 1) / TH;
    const int yend = ys + rows_here;
    const size_t img = (size_t)b * p.IH * p.IW, oimg = (size_t)b * p.OH * p.OW;
    const bool two = p.dy.y != nullptr;
    // every load of a staging batch is unconditional (a load inside a condition is waited for before the next one is issued: one
    // exposed memory round trip per element instead of one per batch)
    const float* ysrc = two ? p.dy.y : p.dy.g;
    const int yld = two ? p.dy.ldy : p.dy.ldg;

    // ---- tables
    for (int e = t; e < E * KS; e += 512) {                     // W_e rows, pitch WP
        const int c = e / KS, k4 = e - c * KS;
        *reinterpret_cast<float4*>(wx + c * WP + 4 * k4) = ldg4(p.we + (size_t)c * I + 4 * k4);
    }
    for (int e = t; e < E; e += 512) {                          // pad columns of W_e (read by the Z1 stage when I is not a multiple of 16)
#pragma unroll
        for (int j = 0; j < 4; ++j) wx[e * WP + I + j] = 0.f;
    }
    for (int e = t; e < NQ * 9; e += 512) {
        const int qq = e / 9, k = e - qq * 9;
        const float* wp = p.wd + (size_t)(4 * qq) * 9 + k;
        wtab[e] = make_float4(wp[0], wp[9], wp[18], wp[27]);
    }
    for (int e = t; e < NQ; e += 512) {
        const int c = 4 * e;
        cd[4 * e + 0] = two ? ldg4(p.dy.p + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        cd[4 * e + 1] = two ? ldg4(p.dy.q + c) : f4zero();
        cd[4 * e + 2] = two ? ldg4(p.dy.mu + c) : f4zero();
        cd[4 * e + 3] = two ? ldg4(p.dy.r + c) : f4zero();
        ce[4 * e + 0] = ldg4(p.escale + c); ce[4 * e + 1] = ldg4(p.eshift + c);
        ce[4 * e + 2] = ldg4(p.emean + c); ce[4 * e + 3] = ldg4(p.einv + c);
        const float4 ga = ldg4(p.egamma + c), iv = ce[4 * e + 3];
        cp[e] = make_float4(ga.x * iv.x, ga.y * iv.y, ga.z * iv.z, ga.w * iv.w);
    }
    if (t < KS) {
        const SrcCoef4 k0 = src_coef4(p.in, 4 * t, I);
        cxa[2 * t] = make_float4(k0.a[0], k0.a[1], k0.a[2], k0.a[3]);
        cxa[2 * t + 1] = make_float4(k0.b[0], k0.b[1], k0.b[2], k0.b[3]);
    }
    __syncthreads();

    // ---- staging: dy' rows into the ring -- thread (quad qd, pixel lane pl) takes the pixels pl, pl + NPL, ... of a batch of whole rows
    // for ITS quad (the mapping of the depthwise stage: the four coefficient quads of dy' are read once per batch, there is no
    // per-element table lookup or quad arithmetic) -- and x rows into an x' tile
    // (the pixel -> (row, column) arithmetic of a thread is the same in every step: left alone the compiler hoists all of it out
    // of the step loop -- VGPRs held across the walk, spilled; an opaque copy of the lane index per call keeps it temporary)
    auto load_dy = [&](float4 (&rg)[UD], float4 (&ry)[UD], int l0, int nrows, int ua, int ub) {
        const int npx = nrows * G::DC;
        int plv = pl;
        asm volatile("" : "+v"(plv));
#pragma unroll
        for (int u = 0; u < UD; ++u) {
            if (u < ua || u >= ub) continue;
            const int pxi = min(plv + NPL * u, npx - 1);
            const int lr = pxi / G::DC, lc = pxi - lr * G::DC;
            const int oy = S == 1 ? ys - 1 + l0 + lr : ys / 2 + l0 + lr, ox = S == 1 ? x0 - 1 + lc : x0 / 2 + lc;
            const size_t r = oimg + (size_t)min(max(oy, 0), p.OH - 1) * p.OW + min(max(ox, 0), p.OW - 1);
            rg[u] = ldg4(p.dy.g + r * p.dy.ldg + 4 * qd);
            ry[u] = ldg4(ysrc + r * yld + 4 * qd);              // (one-tensor operand: g again, its coefficients are q = 0)
        }
    };
    auto store_dy = [&](const float4 (&rg)[UD], const float4 (&ry)[UD], int l0, int nrows) {
        const int npx = nrows * G::DC;
        const float4 cdp = cd[4 * qd], cdq = cd[4 * qd + 1], cdm = cd[4 * qd + 2], cdr = cd[4 * qd + 3];     // once per batch
        int plv = pl;
        asm volatile("" : "+v"(plv));
#pragma unroll
        for (int u = 0; u < UD; ++u) {
            const int pxi = plv + NPL * u;
            if (pxi < npx && dact) {
                const int lr = pxi / G::DC, lc = pxi - lr * G::DC;
                const int oy = S == 1 ? ys - 1 + l0 + lr : ys / 2 + l0 + lr, ox = S == 1 ? x0 - 1 + lc : x0 / 2 + lc;
                const bool ok = (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
                float4 o;
                o.x = ok ? fmaf(cdp.x, rg[u].x, fmaf(cdq.x, ry[u].x - cdm.x, cdr.x)) : 0.f;
                o.y = ok ? fmaf(cdp.y, rg[u].y, fmaf(cdq.y, ry[u].y - cdm.y, cdr.y)) : 0.f;
                o.z = ok ? fmaf(cdp.z, rg[u].z, fmaf(cdq.z, ry[u].z - cdm.z, cdr.z)) : 0.f;
                o.w = ok ? fmaf(cdp.w, rg[u].w, fmaf(cdq.w, ry[u].w - cdm.w, cdr.w)) : 0.f;
                int slot = l0 + lr;
                slot -= (slot / G::DR) * G::DR;
                ring[(slot * G::DC + lc) * NQ + qd] = o;
            }
        }
    };
    auto load_x = [&](float4 (&rx)[UX], int k) {                // the owned pixels of step k: rows ys + TH k .., columns x0 ..
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int el = min(t + 512 * u, NPXS * KS - 1);
            const int pxi = el / KS, c4 = el - pxi * KS;
            const int iy = min(ys + TH * k + (pxi >> 4), p.IH - 1), ix = min(x0 + (pxi & 15), p.IW - 1);
            rx[u] = ldg4(p.in.x + (img + (size_t)iy * p.IW + ix) * p.in.ld + 4 * c4);
        }
    };
    auto store_x = [&](const float4 (&rx)[UX], int buf) {
        float* xd = xt + buf * (NPXS * XP);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int el = t + 512 * u;
            if (el < NPXS * KS) {
                const int pxi = el / KS, c4 = el - pxi * KS;
                const float4 a = cxa[2 * c4], bb = cxa[2 * c4 + 1];
                *reinterpret_cast<float4*>(xd + pxi * XP + 4 * c4) =
                    make_float4(fmaf(a.x, rx[u].x, bb.x), fmaf(a.y, rx[u].y, bb.y), fmaf(a.z, rx[u].z, bb.z), fmaf(a.w, rx[u].w, bb.w));
            }
        }
    };
    {
        // pad columns of the two x' tiles (the G1 stage reads 16-wide column tiles): zero once
        for (int e = t; e < 2 * NPXS; e += 512) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xt[e * XP + I + j] = 0.f;
        }
        float4 rg[UD], ry[UD], rx[UX];
        // first step: ring rows 0 .. DR - 1 (DR <= 2 DN: two batches)
        load_dy(rg, ry, 0, G::DN, 0, UD);
        load_x(rx, 0);
        store_dy(rg, ry, 0, G::DN);
        store_x(rx, 0);
        load_dy(rg, ry, G::DN, G::DR - G::DN, 0, UD);
        store_dy(rg, ry, G::DN, G::DR - G::DN);
    }
    float4 dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4zero();
    float4 s1 = f4zero(), s2 = f4zero();
    f32x4 g1[MUG];
#pragma unroll
    for (int j = 0; j < MUG; ++j) g1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* wq_l = wtab + (dact ? qd : 0) * 9;            // the quad's taps: LDS, re-read per pixel (the registers go to the accumulators) ...
    constexpr bool WREG = NQ <= 24;                             // ... or registers where the budget allows (E = 96)
    float4 wq_r[WREG ? 9 : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wq_r[k] = wq_l[k];
    }
    auto wq = [&](int k) -> float4 { if constexpr (WREG) return wq_r[k]; else return wq_l[k]; };
    const float4* ceq = ce + 4 * (dact ? qd : 0);
    __syncthreads();

#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        const bool more = k + 1 < nsteps;                       // uniform
        const int cur = k & 1;
        const float* xc = xt + cur * (NPXS * XP);
        const int y0 = ys + TH * k;
        // ---- [A] the next step's dy' rows and x rows: in flight across the whole step, written to LDS at its end
        // (UNCONDITIONAL -- the last step requests clamped rows it never uses: inside `if (more)` the loaded registers become phi
        // copies at the end of the block, and every copy waits for its load: the whole batch was drained right after it was issued)
        float4 rg[UD], ry[UD], rx[UX];
        if constexpr (!(XB_SKIP & 16)) {
            load_dy(rg, ry, G::DR + G::DN * k, G::DN, 0, UD);
            load_x(rx, k + 1);
        }
        // ---- [B] expand: unit u = (pixel tile pt, 16-channel tile ct), wave w takes u = w, w + 8, ...; two chains at a time
        if constexpr (!(XB_SKIP & 1))
        for (int u0 = w; u0 < NUE; u0 += 16) {
            const int u1 = u0 + 8;
            const int pa = u0 / NCT, ca = u0 - pa * NCT;
            const int pb = min(u1, NUE - 1) / NCT, cb = min(u1, NUE - 1) - pb * NCT;
            const float* aa = wx + (ca * 16 + l15) * WP + g * KS;
            const float* ab = wx + (cb * 16 + l15) * WP + g * KS;
            const float* ba = xc + (pa * 16 + l15) * XP + g * KS;
            const float* bb = xc + (pb * 16 + l15) * XP + g * KS;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; kk += 2) {
                const float2 wa = *reinterpret_cast<const float2*>(aa + kk), wb = *reinterpret_cast<const float2*>(ab + kk);
                const float2 xa = *reinterpret_cast<const float2*>(ba + kk), xb = *reinterpret_cast<const float2*>(bb + kk);
                acc0 = mfma16b(wa.x, xa.x, acc0); acc1 = mfma16b(wb.x, xb.x, acc1);
                acc0 = mfma16b(wa.y, xa.y, acc0); acc1 = mfma16b(wb.y, xb.y, acc1);
            }
            *reinterpret_cast<float4*>(yt + (pa * 16 + l15) * EP + ca * 16 + 4 * g) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
            if (u1 < NUE) *reinterpret_cast<float4*>(yt + (pb * 16 + l15) * EP + cb * 16 + 4 * g) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        }
        __syncthreads();
        // ---- [C] depthwise stage
        if (dact && !(XB_SKIP & 2)) {
            if constexpr (S == 1) {
                for (int px = pl; px < NPXS; px += NPL) {
                    const int r = px >> 4, c = px & 15;
                    const bool valid = y0 + r < yend && x0 + c < p.IW;
                    float* yp = yt + px * EP + 4 * qd;
                    const float4 y = *reinterpret_cast<const float4*>(yp);
                    const float4 ces = ceq[0], cet = ceq[1];
                    const float u0 = fmaf(ces.x, y.x, cet.x), u1 = fmaf(ces.y, y.y, cet.y), u2 = fmaf(ces.z, y.z, cet.z), u3 = fmaf(ces.w, y.w, cet.w);
                    const float4 h4 = valid ? make_float4(fminf(fmaxf(u0, 0.f), 6.f), fminf(fmaxf(u1, 0.f), 6.f), fminf(fmaxf(u2, 0.f), 6.f),
                                                          fminf(fmaxf(u3, 0.f), 6.f)) : f4zero();
                    float4 tt = f4zero();
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        int slot = G::DN * k + r + 2 - ky;
                        slot -= (slot / G::DR) * G::DR;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 d = ring[(slot * G::DC + c + 2 - kx) * NQ + qd];
                            tt = f4fma(d, wq(ky * 3 + kx), tt);
                            dwacc[ky * 3 + kx] = f4fma(d, h4, dwacc[ky * 3 + kx]);
                        }
                    }
                    float4 ge;
                    ge.x = (valid && u0 > 0.f && u0 < 6.f) ? tt.x : 0.f; ge.y = (valid && u1 > 0.f && u1 < 6.f) ? tt.y : 0.f;
                    ge.z = (valid && u2 > 0.f && u2 < 6.f) ? tt.z : 0.f; ge.w = (valid && u3 > 0.f && u3 < 6.f) ? tt.w : 0.f;
                    const float4 cem = ceq[2], cei = ceq[3], cpe = cp[qd];
                    s1 = f4add(s1, ge);
                    s2 = f4fma(ge, make_float4((y.x - cem.x) * cei.x, (y.y - cem.y) * cei.y, (y.z - cem.z) * cei.z, (y.w - cem.w) * cei.w), s2);
                    *reinterpret_cast<float4*>(yp) = make_float4(ge.x * cpe.x, ge.y * cpe.y, ge.z * cpe.z, ge.w * cpe.w);
                    if constexpr (GE) { if (valid) stg4(p.gexp + (img + (size_t)(y0 + r) * p.IW + x0 + c) * p.ldge + 4 * qd, ge); }
                }
            } else {
                for (int bq = pl; bq < NPXS / 4; bq += NPL) {       // 2 x 2 input pixels <-> one output neighbourhood
                    const int by = bq >> 3, bx = bq & 7;
                    int sa = G::DN * k + by, sb = sa + 1;
                    sa -= (sa / G::DR) * G::DR; sb -= (sb / G::DR) * G::DR;
                    const float4 da0 = ring[(sa * G::DC + bx) * NQ + qd], da1 = ring[(sa * G::DC + bx + 1) * NQ + qd];
                    const float4 db0 = ring[(sb * G::DC + bx) * NQ + qd], db1 = ring[(sb * G::DC + bx + 1) * NQ + qd];
                    float4 tv4[4];
                    tv4[0] = f4fma(da0, wq(4), f4zero());
                    tv4[1] = f4fma(da1, wq(3), f4fma(da0, wq(5), f4zero()));
                    tv4[2] = f4fma(db0, wq(1), f4fma(da0, wq(7), f4zero()));
                    tv4[3] = f4fma(db1, wq(0), f4fma(db0, wq(2), f4fma(da1, wq(6), f4fma(da0, wq(8), f4zero()))));
                    const float4 ces = ceq[0], cet = ceq[1], cem = ceq[2], cei = ceq[3], cpe = cp[qd];
                    float4 h4[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {                   // v = 2 * (row parity) + column parity
                        const int r = 2 * by + (v >> 1), c = 2 * bx + (v & 1);
                        const bool valid = y0 + r < yend && x0 + c < p.IW;
                        float* yp = yt + (r * 16 + c) * EP + 4 * qd;
                        const float4 y = *reinterpret_cast<const float4*>(yp);
                        const float u0 = fmaf(ces.x, y.x, cet.x), u1 = fmaf(ces.y, y.y, cet.y), u2 = fmaf(ces.z, y.z, cet.z), u3 = fmaf(ces.w, y.w, cet.w);
                        h4[v] = valid ? make_float4(fminf(fmaxf(u0, 0.f), 6.f), fminf(fmaxf(u1, 0.f), 6.f), fminf(fmaxf(u2, 0.f), 6.f),
                                                    fminf(fmaxf(u3, 0.f), 6.f)) : f4zero();
                        float4 ge;
                        ge.x = (valid && u0 > 0.f && u0 < 6.f) ? tv4[v].x : 0.f; ge.y = (valid && u1 > 0.f && u1 < 6.f) ? tv4[v].y : 0.f;
                        ge.z = (valid && u2 > 0.f && u2 < 6.f) ? tv4[v].z : 0.f; ge.w = (valid && u3 > 0.f && u3 < 6.f) ? tv4[v].w : 0.f;
                        s1 = f4add(s1, ge);
                        s2 = f4fma(ge, make_float4((y.x - cem.x) * cei.x, (y.y - cem.y) * cei.y, (y.z - cem.z) * cei.z, (y.w - cem.w) * cei.w), s2);
                        *reinterpret_cast<float4*>(yp) = make_float4(ge.x * cpe.x, ge.y * cpe.y, ge.z * cpe.z, ge.w * cpe.w);
                        if constexpr (GE) { if (valid) stg4(p.gexp + (img + (size_t)(y0 + r) * p.IW + x0 + c) * p.ldge + 4 * qd, ge); }
                    }
                    dwacc[4] = f4fma(da0, h4[0], dwacc[4]);
                    dwacc[3] = f4fma(da1, h4[1], dwacc[3]); dwacc[5] = f4fma(da0, h4[1], dwacc[5]);
                    dwacc[1] = f4fma(db0, h4[2], dwacc[1]); dwacc[7] = f4fma(da0, h4[2], dwacc[7]);
                    dwacc[0] = f4fma(db1, h4[3], dwacc[0]); dwacc[2] = f4fma(db0, h4[3], dwacc[2]);
                    dwacc[6] = f4fma(da1, h4[3], dwacc[6]); dwacc[8] = f4fma(da0, h4[3], dwacc[8]);
                }
            }
        }
        __syncthreads();
        // ---- [D] Z1 = W_e^T (p_e g_e): unit (pixel tile pt, input-channel tile it); A[i][k] = W_e[c(k)][i], B[k][px] = (p_e g_e)[px][c(k)],
        // c(g, kk) = g * KE + kk; two chains over the even / odd k-steps
        if constexpr (!(XB_SKIP & 4))
#pragma unroll
        for (int uj = 0; uj < (NUZ + 7) / 8; ++uj) {
            const bool uon = w + 8 * uj < NUZ;                  // uniform (NUZ < 8: the upper waves repeat the last unit and store nothing)
            const int u = min(w + 8 * uj, NUZ - 1);
            const int pt = u / NI, it = u - pt * NI;
            const float* ap = wx + (g * KE) * WP + it * 16 + l15;
            const float* bp = yt + (pt * 16 + l15) * EP + g * KE;
            f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
            for (int kk = 0; kk < KE; kk += 4) {
                const float4 bv = *reinterpret_cast<const float4*>(bp + kk);
                z0 = mfma16b(ap[(kk + 0) * WP], bv.x, z0); z1v = mfma16b(ap[(kk + 1) * WP], bv.y, z1v);
                z0 = mfma16b(ap[(kk + 2) * WP], bv.z, z0); z1v = mfma16b(ap[(kk + 3) * WP], bv.w, z1v);
            }
            const int r = pt, c = l15;                          // rows of D: input channels it * 16 + 4 g + j; column: pixel
            const int i0 = it * 16 + 4 * g;
            if (uon && i0 < I && y0 + r < yend && x0 + c < p.IW)
                stg4(p.z1 + (img + (size_t)(y0 + r) * p.IW + x0 + c) * p.ldz + i0,
                     make_float4(z0[0] + z1v[0], z0[1] + z1v[1], z0[2] + z1v[2], z0[3] + z1v[3]));
        }
        // ---- G1 += (p_e g_e)^T x': unit (channel tile ct, input tile it); contraction over the step's pixels px(g, kk) = 4 kk + g
#pragma unroll
        for (int j = 0; j < MUG; ++j) {
            const int u = w + 8 * j;
            if (u < NUG && !(XB_SKIP & 8)) {                    // uniform
                const int ct = u / NI, it = u - ct * NI;
                const float* ap = yt + g * EP + ct * 16 + l15;
                const float* bp = xc + g * XP + it * 16 + l15;
                f32x4 a0 = g1[j], a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NPXS / 4; kk += 2) {
                    a0 = mfma16b(ap[(4 * kk) * EP], bp[(4 * kk) * XP], a0);
                    a1 = mfma16b(ap[(4 * kk + 4) * EP], bp[(4 * kk + 4) * XP], a1);
                }
                g1[j] = f32x4{a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]};
            }
        }
        if constexpr ((XB_SKIP & 48) == 32) {                   // diagnostic: the loads are waited for, nothing is written
#pragma unroll
            for (int u = 0; u < UD; ++u) asm volatile("" :: "v"(rg[u].x), "v"(rg[u].w), "v"(ry[u].x), "v"(ry[u].w));
        } else
        if (more && !(XB_SKIP & 16)) {
            store_dy(rg, ry, G::DR + G::DN * k, G::DN);         // the new rows take the slots of the window's oldest rows
            store_x(rx, cur ^ 1);
        }
        __syncthreads();
    }

    // ---- the workgroup's partial row: G1 [E][I] | dW_d [E][9] | S1 [E] | S2 [E]
    float* row = p.part + (size_t)tIdx * p.row_floats;
#pragma unroll
    for (int j = 0; j < MUG; ++j) {
        const int u = w + 8 * j;
        if (u < NUG) {
            const int ct = u / NI, it = u - ct * NI;
            const int i = it * 16 + l15;                        // column of D; rows: channels ct * 16 + 4 g + r
            if (i < I) {
#pragma unroll
                for (int r = 0; r < 4; ++r) row[(ct * 16 + 4 * g + r) * I + i] = g1[j][r];
            }
        }
    }
    {
        float* red = reinterpret_cast<float*>(smem);            // [pixel lane][quad][44]: 512 x 44 floats <= ring + y tile
        if (dact) {
            float* slot = red + (size_t)(pl * NQ + qd) * 44;
#pragma unroll
            for (int j = 0; j < 9; ++j) { slot[4 * j] = dwacc[j].x; slot[4 * j + 1] = dwacc[j].y; slot[4 * j + 2] = dwacc[j].z; slot[4 * j + 3] = dwacc[j].w; }
            slot[36] = s1.x; slot[37] = s1.y; slot[38] = s1.z; slot[39] = s1.w;
            slot[40] = s2.x; slot[41] = s2.y; slot[42] = s2.z; slot[43] = s2.w;
        }
        __syncthreads();
        for (int o = t; o < NQ * 44; o += 512) {
            const int qq = o / 44, comp = o - qq * 44;
            float a = 0.f;
            for (int l = 0; l < NPL; ++l) a += red[(size_t)(l * NQ + qq) * 44 + comp];
            const int c = 4 * qq + (comp & 3);
            if (comp < 36) row[E * I + c * 9 + (comp >> 2)] = a;
            else row[E * I + E * 9 + ((comp - 36) >> 2) * E + c] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------ fold
// One workgroup per 16 expanded channels: sums the walkers' partial rows (fixed order), then
//   p_e = gamma invstd, q_e = -p_e invstd S2 / n, r_e = -p_e S1 / n;  dgamma = S2, dbeta = S1;  dW_d = sum of the rows;
//   dW_e[c][i] = G1[c][i] + q_e[c] (sum_j W_e[c][j] Gxx[j][i] - mu_e[c] sx[i]) + r_e[c] sx[i]
// and its share of  A = W_e^T diag(q_e) W_e,  b' = W_e^T (r_e - q_e mu_e)  as a partial row (the fix-up kernel sums the E / 16 rows).
// rows r = s, s + NS, ... of the walkers' partial rows -> slice s (fixed order): the 256 x 20 K-float reduction spread over the chip
__global__ __launch_bounds__(256) void xslice_kernel(const float* part, int nblk, int rowf, int ns, float* out) {
    const int e = blockIdx.x * 256 + threadIdx.x, s0 = blockIdx.y;
    if (e >= rowf) return;
    float a = 0.f;
    int r = s0;
    for (; r + 7 * ns < nblk; r += 8 * ns) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + u * ns) * rowf + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; r < nblk; r += ns) a += part[(size_t)r * rowf + e];
    out[(size_t)s0 * rowf + e] = a;
}

struct XfoldP {
    const float* part; int nblk, row_floats;
    const double* mom;                       // [I * I + I]: Gxx, sx of x' (pq_xstat_fwd)
    const float* we; const float* egamma; const float* einv; const float* emean;
    float* dwe; float* dwd; float* dgamma; float* dbeta;
    float* ab;                               // [E / 16][I * I + I] partial rows of A and b'
    int I, E; float count;
};
__global__ __launch_bounds__(256) void xfold_kernel(XfoldP p) {
    __shared__ float sums[16 * (32 + 11)];   // [16 channels][I + 9 + 2]
    __shared__ float qv[16], rv[16], mu[16];
    __shared__ float wl[16][33];             // the group's weight rows (the loops below re-read them I times)
    const int t = threadIdx.x, c0 = blockIdx.x * 16;
    const int I = p.I, E = p.E, W = I + 11;
    for (int e = t; e < 16 * I; e += 256) wl[e / I][e % I] = p.we[(size_t)c0 * I + e];
    // element e of the channel group: (channel cl, column j): j < I: G1, j < I + 9: dW_d tap, then S1, S2
    for (int e = t; e < 16 * W; e += 256) {
        const int cl = e / W, j = e - cl * W, c = c0 + cl;
        size_t off;
        if (j < I) off = (size_t)c * I + j;
        else if (j < I + 9) off = (size_t)E * I + (size_t)c * 9 + (j - I);
        else off = (size_t)E * I + (size_t)E * 9 + (size_t)(j - I - 9) * E + c;
        float a = 0.f;
        int k = 0;
        for (; k + 8 <= p.nblk; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p.part[(size_t)(k + u) * p.row_floats + off];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; k < p.nblk; ++k) a += p.part[(size_t)k * p.row_floats + off];
        sums[e] = a;
    }
    __syncthreads();
    if (t < 16) {
        const int c = c0 + t;
        const double s1 = sums[t * W + I + 9], s2 = sums[t * W + I + 10];
        const double pe = (double)p.egamma[c] * (double)p.einv[c];
        qv[t] = (float)(-pe * (double)p.einv[c] * s2 / (double)p.count);
        rv[t] = (float)(-pe * s1 / (double)p.count);
        mu[t] = p.emean[c];
        p.dgamma[c] = (float)s2; p.dbeta[c] = (float)s1;
    }
    for (int e = t; e < 16 * 9; e += 256) {
        const int cl = e / 9, k = e - cl * 9;
        p.dwd[(size_t)(c0 + cl) * 9 + k] = sums[cl * W + I + k];
    }
    __syncthreads();
    for (int e = t; e < 16 * I; e += 256) {
        const int cl = e / I, i = e - cl * I, c = c0 + cl;
        double acc = 0.0;
        for (int j = 0; j < I; ++j) acc = fma((double)wl[cl][j], p.mom[j * I + i], acc);
        const double sx = p.mom[I * I + i];
        p.dwe[(size_t)c * I + i] = (float)((double)sums[cl * W + i] + (double)qv[cl] * (acc - (double)mu[cl] * sx) + (double)rv[cl] * sx);
    }
    // A[i][j] += sum_c W[c][i] q[c] W[c][j] over this group's channels; b'[i] += sum_c W[c][i] (r[c] - q[c] mu[c])
    float* ab = p.ab + (size_t)blockIdx.x * (I * I + I);
    for (int e = t; e < I * I + I; e += 256) {
        double acc = 0.0;
        if (e < I * I) {
            const int i = e / I, j = e - i * I;
            for (int cl = 0; cl < 16; ++cl) acc = fma((double)wl[cl][i] * (double)qv[cl], (double)wl[cl][j], acc);
        } else {
            const int i = e - I * I;
            for (int cl = 0; cl < 16; ++cl) acc = fma((double)wl[cl][i], (double)rv[cl] - (double)qv[cl] * (double)mu[cl], acc);
        }
        ab[e] = (float)acc;
    }
}

// ------------------------------------------------------------------------------------------------ fix-up
// dz = Z1 + A x' + b' (+ the running gradient) -> the block input's gradient sink (activation derivative, BatchNorm-backward sums): an
// element-wise pass over the I-channel tensor with an I x I matrix-vector product per pixel.  Thread = (pixel, channel quad).
struct XfixP {
    const float* z1; int ldz; pq_src in; pq_gsink dx; const float* ab; int nab; int M, I;
};
// The operand's affine is folded in once per workgroup:  A (a x + b) + b' = (A diag(a)) x + (A b + b'); a thread keeps the four rows
// of that matrix its channel quad needs in registers for all its pixels.
template <int KS>
__global__ __launch_bounds__(256) void xfix_kernel(XfixP p) {
    constexpr int I = 4 * KS;
    constexpr int PY = 256 / KS;                                // pixels per block pass (KS quads x PY pixel lanes; 256 % KS threads idle)
    __shared__ float A[I * I + I];
    __shared__ float red[256 * 8];
    const int t = threadIdx.x;
    for (int e = t; e < I * I + I; e += 256) {
        float a = 0.f;
        for (int k = 0; k < p.nab; ++k) a += p.ab[(size_t)k * (I * I + I) + e];
        A[e] = a;
    }
    __syncthreads();
    const int qx = t % KS, py = t / KS;
    const bool act_t = py < PY;
    const int c0 = 4 * qx;
    float ar[4][I], br[4];
    {
        const bool aff = p.in.scale != nullptr;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            float bsum = A[I * I + c0 + ii];
#pragma unroll
            for (int j = 0; j < I; ++j) {
                const float av = A[(c0 + ii) * I + j];
                ar[ii][j] = aff ? av * p.in.scale[j] : av;
                bsum = aff ? fmaf(av, p.in.shift[j], bsum) : bsum;
            }
            br[ii] = bsum;
        }
    }
    const SinkCoef4 sk = sink_coef4(p.dx, c0, I);
    float4 s1 = f4zero(), s2 = f4zero();
    const int rows_per = (p.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * rows_per, m1 = min(m0 + rows_per, p.M);
    // two pixels per iteration, all loads of both requested before the first use: with one pixel per iteration the loop was a chain
    // of dependent memory round trips (12 iterations x ~2.5 us at 262144 x 24: 1.9 TB/s)
    if (act_t)
    for (int m = m0 + py; m < m1; m += 2 * PY) {
        const int mb = min(m + PY, m1 - 1);                       // second pixel (clamped: never stored when past the range)
        const bool bon = m + PY < m1;
        const float* xra = p.in.x + (size_t)m * p.in.ld;          // the KS threads of a pixel read the same cache lines
        const float* xrb = p.in.x + (size_t)mb * p.in.ld;
        float4 xa[KS], xb[KS];
#pragma unroll
        for (int j4 = 0; j4 < KS; ++j4) { xa[j4] = ldg4(xra + 4 * j4); xb[j4] = ldg4(xrb + 4 * j4); }
        const float4 za = ldg4(p.z1 + (size_t)m * p.ldz + c0), zb = ldg4(p.z1 + (size_t)mb * p.ldz + c0);
        const SinkOps soa = sink_load4(p.dx, sk, (size_t)m, c0), sob = sink_load4(p.dx, sk, (size_t)mb, c0);
        float acca[4] = {br[0], br[1], br[2], br[3]}, accb[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
        for (int j4 = 0; j4 < KS; ++j4) {
            const float sa[4] = {xa[j4].x, xa[j4].y, xa[j4].z, xa[j4].w}, sb[4] = {xb[j4].x, xb[j4].y, xb[j4].z, xb[j4].w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    acca[ii] = fmaf(ar[ii][4 * j4 + jj], sa[jj], acca[ii]);
                    accb[ii] = fmaf(ar[ii][4 * j4 + jj], sb[jj], accb[ii]);
                }
        }
        sink_store4(p.dx, sk, (size_t)m, c0, make_float4(za.x + acca[0], za.y + acca[1], za.z + acca[2], za.w + acca[3]), soa, s1, s2);
        if (bon) sink_store4(p.dx, sk, (size_t)mb, c0, make_float4(zb.x + accb[0], zb.y + accb[1], zb.z + accb[2], zb.w + accb[3]), sob, s1, s2);
    }
    if (p.dx.finalize && p.dx.mean && p.dx.partial) {          // uniform
        float* slot = red + t * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < KS * 8) {
            const int qq = t >> 3, comp = t & 7;
            float a = 0.f;
            for (int l = 0; l < PY; ++l) a += red[(l * KS + qq) * 8 + comp];
            const int c = 4 * qq + (comp & 3);
            p.dx.partial[(size_t)blockIdx.x * 2 * p.dx.ldp + (comp >> 2) * p.dx.ldp + c] = a;
        }
    }
}

}  // namespace pq

using namespace pq;

struct XbPlan { int strips, segs, seg_rows, nblk, th; size_t lds; };
static bool xb_shape_ok(int I, int E, int stride, int IH, int IW) {
    if (stride == 2 && ((IH & 1) || (IW & 1))) return false;
    if (stride != 1 && stride != 2) return false;
    // (32 -> 192 on the 64 x 64 planes was built with two-row steps and measured at break-even against the unfused kernels: 112 / 86 us
    //  against 104 / 93 alone, 8.96 against 8.94 ms per step -- not shipped)
    return (I == 16 && E == 96 && stride == 2) || (I == 24 && E == 144);
}
static XbPlan xb_plan(int B, int IH, int IW, int I, int E, int stride) {
    XbPlan w{};
    w.th = stride == 1 ? 4 : 8;                                // input rows per step: all E channels of a step must fit the LDS
    w.strips = (IW + 15) / 16;
    const int steps_total = (IH + w.th - 1) / w.th;
    const long base = (long)B * w.strips;
    long segs = (256 + base - 1) / base;                       // one workgroup per CU
    const long maxsegs = steps_total / 2 > 0 ? steps_total / 2 : 1;
    if (segs > maxsegs) segs = maxsegs;
    if (segs < 1) segs = 1;
    const int seg_steps = (int)((steps_total + segs - 1) / segs);
    w.segs = (steps_total + seg_steps - 1) / seg_steps;
    w.seg_rows = seg_steps * w.th;
    w.nblk = B * w.strips * w.segs;
    const int nq = E / 4, npxs = w.th * 16;
    const int dr = stride == 1 ? w.th + 2 : w.th / 2 + 1, dc = stride == 1 ? 18 : 9;
    w.lds = (size_t)dr * dc * nq * 16 + (size_t)npxs * (E + 4) * 4 + (size_t)2 * npxs * (I + 4) * 4 + (size_t)E * (I + 4) * 4 +
            (size_t)nq * 9 * 16 + (size_t)nq * 9 * 16 + (size_t)(I / 4) * 2 * 16;
    const size_t red = (size_t)512 * 44 * 4;
    if (w.lds < red) w.lds = red;
    return w;
}
static int xb_row_floats(int I, int E) { return E * I + E * 9 + 2 * E; }
constexpr int XB_SLICES = 16;

extern "C" int32_t pq_xdw_bwd_blocks(int32_t B, int32_t IH, int32_t IW, int32_t I, int32_t E, int32_t stride) {
    if (!xb_shape_ok(I, E, stride, IH, IW)) return 0;
    return xb_plan(B, IH, IW, I, E, stride).nblk;
}
extern "C" int64_t pq_xdw_bwd_scratch_bytes(int32_t B, int32_t IH, int32_t IW, int32_t I, int32_t E, int32_t stride) {
    if (!xb_shape_ok(I, E, stride, IH, IW)) return 0;
    const XbPlan w = xb_plan(B, IH, IW, I, E, stride);
    // partial rows | their slices | A / b' rows of the fold | Z1
    return ((int64_t)(w.nblk + XB_SLICES) * xb_row_floats(I, E) + (int64_t)(E / 16) * (I * I + I) + (int64_t)B * IH * IW * I) * 4 + 256;
}

template <int S, int KS, int NQ, int TH_, bool GE>
static int xb_launch1(const XbwdP& p, const XbPlan& w, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)xdwb_kernel<S, KS, NQ, TH_, GE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            pq_set_error("pq_xdw_bwd: cannot raise the dynamic LDS limit");
            return PQ_ELAUNCH;
        }
        attr_set = true;
    }
    PQ_LAUNCH((xdwb_kernel<S, KS, NQ, TH_, GE>), dim3(w.nblk), dim3(512), w.lds, st, p);
    return PQ_OK;
}
template <int S, int KS, int NQ, int TH_>
static int xb_launch(const XbwdP& p, const XbPlan& w, hipStream_t st) {
    return p.gexp ? xb_launch1<S, KS, NQ, TH_, true>(p, w, st) : xb_launch1<S, KS, NQ, TH_, false>(p, w, st);
}

extern "C" int pq_xdw_bwd(const pq_xdw_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->we && a->wd && a->dy.g && a->dx.g && a->scratch && a->moments, "null pointer");
    PQ_SINK_ADDEND_CHECK(a->dx);
    PQ_CHECK_ARG(a->mid_scale && a->mid_shift && a->mid_mean && a->mid_invstd && a->mid_gamma && a->dwe && a->dwd && a->dgamma && a->dbeta,
                 "null pointer (BatchNorm of the expanded tensor / parameter gradients)");
    PQ_CHECK_ARG(a->B > 0 && a->IH > 0 && a->IW > 0, "empty shape");
    PQ_CHECK_ARG(a->OH == (a->IH - 1) / a->stride + 1 && a->OW == (a->IW - 1) / a->stride + 1, "output size does not match pad 1");
    PQ_CHECK_ARG((a->in.ld & 3) == 0 && a->in.ld >= a->I && (a->dy.ldg & 3) == 0 && a->dy.ldg >= a->E && (a->dx.ldg & 3) == 0 && a->dx.ldg >= a->I,
                 "pitches must be multiples of 4 and cover the channels");
    PQ_CHECK_ARG(a->dy.y == nullptr || ((a->dy.ldy & 3) == 0 && a->dy.p && a->dy.q && a->dy.mu && a->dy.r), "gradient pair: incomplete");
    PQ_CHECK_ARG(!(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    PQ_CHECK_ARG((((uintptr_t)a->scratch) & 15) == 0, "scratch must be 16-byte aligned");
    if (a->in.act != PQ_ACT_LINEAR || a->mid_act != PQ_ACT_RELU6 || a->in.qstate || a->in.stat.acc || a->dx.qstate || a->dx.stat_acc ||
        act_is_special(a->dx.act)) {
        pq_set_error("pq_xdw_bwd: linear input operand, ReLU6 on the expanded tensor, no QAT observers / statistics hand-over");
        return PQ_EUNSUPPORTED;
    }
    if (!xb_shape_ok(a->I, a->E, a->stride, a->IH, a->IW)) {
        pq_set_error("pq_xdw_bwd: shape outside the kernel's range (pq_xdw_bwd_blocks() == 0)");
        return PQ_EUNSUPPORTED;
    }
    const XbPlan w = xb_plan(a->B, a->IH, a->IW, a->I, a->E, a->stride);
    const int M = a->B * a->IH * a->IW;
    const int rowf = xb_row_floats(a->I, a->E);
    float* part = a->scratch;
    float* slices = part + (size_t)w.nblk * rowf;
    float* ab = slices + (size_t)XB_SLICES * rowf;
    float* z1 = ab + (size_t)(a->E / 16) * (a->I * a->I + a->I);
    z1 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(z1) + 63) & ~(uintptr_t)63);
    hipStream_t st = (hipStream_t)stream;
    XbwdP p{};
    p.in = a->in; p.we = a->we;
    p.escale = a->mid_scale; p.eshift = a->mid_shift; p.emean = a->mid_mean; p.einv = a->mid_invstd; p.egamma = a->mid_gamma;
    p.dy = a->dy; p.wd = a->wd; p.z1 = z1; p.ldz = a->I; p.part = part; p.gexp = a->gexp; p.ldge = a->ldge;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW;
    p.strips = w.strips; p.segs = w.segs; p.seg_rows = w.seg_rows; p.row_floats = rowf;
    int rc;
    if (a->I == 16) rc = xb_launch<2, 4, 24, 8>(p, w, st);
    else rc = a->stride == 1 ? xb_launch<1, 6, 36, 4>(p, w, st) : xb_launch<2, 6, 36, 8>(p, w, st);
    if (rc != PQ_OK) return rc;
    const int ns = w.nblk < XB_SLICES ? w.nblk : XB_SLICES;
    PQ_LAUNCH(xslice_kernel, dim3((rowf + 255) / 256, ns), dim3(256), 0, st, (const float*)part, w.nblk, rowf, ns, slices);
    XfoldP f{slices, ns, rowf, a->moments, a->we, a->mid_gamma, a->mid_invstd, a->mid_mean, a->dwe, a->dwd, a->dgamma, a->dbeta, ab,
             a->I, a->E, (float)((double)a->B * a->IH * a->IW)};
    PQ_LAUNCH(xfold_kernel, dim3(a->E / 16), dim3(256), 0, st, f);
    XfixP x{z1, a->I, a->in, a->dx, ab, a->E / 16, M, a->I};
    const int nfix = pq_xdw_bwd_fix_blocks(M, a->I);
    if (a->I == 16) PQ_LAUNCH((xfix_kernel<4>), dim3(nfix), dim3(256), 0, st, x);
    else PQ_LAUNCH((xfix_kernel<6>), dim3(nfix), dim3(256), 0, st, x);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
extern "C" int32_t pq_xdw_bwd_fix_blocks(int32_t M, int32_t I) {
    if (M <= 0 || I <= 0) return 0;
    const int py = 256 / (I / 4);
    long want = ((long)M + py * 8 - 1) / (py * 8);             // >= 8 pixels per thread
    if (want > 512) want = 512;                                // two workgroups per CU are resident (register budget): ONE round
    if (want < 1) want = 1;
    return (int32_t)want;
}
