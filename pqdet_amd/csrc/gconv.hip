// Grouped / dense 3x3 convolution (pad 1, stride 1|2, group width a multiple of 4 and <= 32) as an implicit GEMM on the
// fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.  RegNetX: group width 24 (backbone) and 16 (neck).
//
// A workgroup owns an 8x16 tile of output pixels (4 waves x 32 pixels) of ONE group.  The input tile with halo is staged
// once into LDS ([pixel][channel quad], operand transform applied at staging time, zero padding after the transform); no
// im2col buffer exists: for tap (ky,kx) and channel quad the MFMA A fragment of a lane is one ds_read_b128 at
// ((py*S+ky)*ITW + px*S+kx) * pitch + quad.  The group's weights are gathered once per workgroup into a [32][9*gw] LDS image
// with k = tap*gw + i.  K = 9*gw is consumed in 8-wide steps (gw is a multiple of 8 or 4 -> steps never straddle a tap when
// gw % 8 == 0; for gw % 8 == 4 the kernel uses 4-wide steps).
//
// The same kernel is the data gradient: A = dy' (closed-form BN backward applied at staging), weights gathered transposed
// and flipped, output through the gradient sink.  For stride 2 the dy' plane is read as its zero-inserted upsampling
// (u even ? dy'[u/2] : 0), which turns the transposed convolution into the stride-1 form.
//
// wgrad: dw[o][i][tap] = sum_p dy'[p][o] * x'[p*S+tap-1][i] -- persistent workgroups walk pixel tiles of one group, the
// four waves split the pixels of a tile (split-K), 32 x (9*gw) accumulators per wave, deterministic two-level reduction.
#include "common.h"

namespace pq {

struct GcP {
    // operand (forward: src; backward: gsrc) living on an AH x AW plane with C channels
    pq_src in; pq_gsrc dy; int grad;      // grad = 1: A operand is dy'
    int up;                               // dgrad of a stride-2 conv: A(u,v) = dy'(u/2, v/2) if both even else 0 (plane DH x DW)
    int DH, DW;                           // real dy plane when up
    const float* w; int wmode;            // 0: forward B[o][tap*gw+i] = w[go+o][i][tap]; 1: dgrad B[i][tap*gw+o] = w[go+o][i][8-tap]
    float* out; float* partial; int ldo, ldp;
    pq_gsink sink;
    int B, AH, AW, OH, OW, C, gw, S, tilesH, tilesW;
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;      // forward: statistics into pq_bn_stat accumulators instead of rows
};

constexpr int GTH = 8, GTW = 16;          // output tile

__device__ __forceinline__ f32x16 gmfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---- operand tile staging.  A thread owns ONE channel quad (coefficients live in registers) and walks pixels; the loads
// of four pixels are issued back to back at clamped (always valid) coordinates and masked afterwards -- a load inside an
// `if` per element made the staging loop one memory latency per iteration, which dominated the stride-2 layers.
constexpr int GUB = 4;            // weight-image staging: loads in flight per thread (14 -- two batches instead of seven for K = 216 -- measured: no change in the step, the co-resident workgroups hide it)
// operand-tile staging: pixels a thread keeps in flight per batch.  A batch = one exposed memory round trip (its LDS stores sit
// between it and the next batch's loads), so the 10 x 18 tile of the stride-1 kernels (at most 6 pixels per thread with >= 32
// pixel lanes) goes in ONE batch; the 17 x 33 tile of the stride-2 kernels in batches of 8
template <int NPIX> struct GcStageBatch { static constexpr int U = NPIX <= 6 * 32 ? 6 : 8; };
// lcoef != NULL: the group's BatchNorm scale / shift ([2][32] floats in LDS) were derived from the producer's statistic
// accumulators by the workgroup itself (pq_bn_stat hand-over: no finalize launch in front of this kernel)
template <int ITW, int NPIX>
__device__ __forceinline__ void gc_stage_src(float4* At, int pitch, int gq, int cbase, int b, int ay0, int ax0, int AH, int AW,
                                             const pq_src& in, int t, const float* lcoef = nullptr) {
    const int qd = t % gq, pl = t / gq, PL = 256 / gq;
    if (pl >= PL) return;
    const int c = cbase + 4 * qd;
    const ActParam aact = act_param(in.act);
    float sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (lcoef) { sc[j] = lcoef[4 * qd + j]; sh[j] = lcoef[32 + 4 * qd + j]; }
        else { sc[j] = in.scale ? in.scale[c + j] : 1.f; sh[j] = in.scale ? in.shift[c + j] : 0.f; }
    }
    const bool fq = in.qstate != nullptr;
    const float qs = fq ? in.qstate[2] : 1.f, qinv = fq ? 1.f / qs : 1.f, qzp = fq ? in.qstate[3] : 0.f;
    const float* base = in.x + (size_t)b * AH * AW * in.ld + c;
    constexpr int UB = GcStageBatch<NPIX>::U;
    for (int pix0 = pl; pix0 < NPIX; pix0 += UB * PL) {
        float4 r[UB];
        bool ok[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int pix = pix0 + u * PL, pc = min(pix, NPIX - 1);
            const int py = pc / ITW, px = pc - py * ITW;
            const int ay = ay0 + py, ax = ax0 + px;
            ok[u] = pix < NPIX && ay >= 0 && ay < AH && ax >= 0 && ax < AW;
            const int yy = min(max(ay, 0), AH - 1), xx = min(max(ax, 0), AW - 1);
            r[u] = ldg4(base + ((size_t)yy * AW + xx) * in.ld);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int pix = pix0 + u * PL;
            if (pix < NPIX) {
                float e4[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e4[j] = act_fwd(fmaf(sc[j], e4[j], sh[j]), aact);
                    if (fq) e4[j] = fq_act(e4[j], qs, qinv, qzp);
                }
                At[pix * pitch + qd] = ok[u] ? make_float4(e4[0], e4[1], e4[2], e4[3]) : f4zero();
            }
        }
    }
}
// gradient operand dy' = p*g + q*(y-mu) + r on a PH x PW plane; up: the tile lives on the zero-inserted (2x) grid
template <int ITW, int NPIX>
__device__ __forceinline__ void gc_stage_gsrc(float4* At, int pitch, int gq, int cbase, int b, int ay0, int ax0, int AH, int AW,
                                              const pq_gsrc& dy, int up, int DH, int DW, int t) {
    const int qd = t % gq, pl = t / gq, PL = 256 / gq;
    if (pl >= PL) return;
    const int c = cbase + 4 * qd;
    const bool two = dy.y != nullptr;
    float cp[4], cq[4], cm[4], cr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cp[j] = two ? dy.p[c + j] : 1.f; cq[j] = two ? dy.q[c + j] : 0.f; cm[j] = two ? dy.mu[c + j] : 0.f; cr[j] = two ? dy.r[c + j] : 0.f;
    }
    const int PH = up ? DH : AH, PW = up ? DW : AW;
    const float* gb = dy.g + (size_t)b * PH * PW * dy.ldg + c;
    const float* yb = two ? dy.y + (size_t)b * PH * PW * dy.ldy + c : nullptr;
    constexpr int UB = GcStageBatch<NPIX>::U;
    for (int pix0 = pl; pix0 < NPIX; pix0 += UB * PL) {
        float4 rg[UB], ry[UB];
        bool ok[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int pix = pix0 + u * PL, pc = min(pix, NPIX - 1);
            const int py = pc / ITW, px = pc - py * ITW;
            const int ay = ay0 + py, ax = ax0 + px;
            bool in = pix < NPIX && ay >= 0 && ay < AH && ax >= 0 && ax < AW;
            int yy = min(max(ay, 0), AH - 1), xx = min(max(ax, 0), AW - 1);
            if (up) { in = in && !(ay & 1) && !(ax & 1); yy >>= 1; xx >>= 1; in = in && yy < DH && xx < DW; yy = min(yy, DH - 1); xx = min(xx, DW - 1); }
            ok[u] = in;
            const size_t row = (size_t)yy * PW + xx;
            rg[u] = ldg4(gb + row * dy.ldg);
            ry[u] = two ? ldg4(yb + row * dy.ldy) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int pix = pix0 + u * PL;
            if (pix < NPIX) {
                float e4[4] = {rg[u].x, rg[u].y, rg[u].z, rg[u].w};
                const float y4[4] = {ry[u].x, ry[u].y, ry[u].z, ry[u].w};
                if (two) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) e4[j] = fmaf(cp[j], e4[j], fmaf(cq[j], y4[j] - cm[j], cr[j]));
                }
                At[pix * pitch + qd] = ok[u] ? make_float4(e4[0], e4[1], e4[2], e4[3]) : f4zero();
            }
        }
    }
}

// PERSIST: the workgroup keeps its group's weight image in LDS and walks tiles bx, bx + gridDim.x, ... -- used for the big
// stride-2 forward layers, where one workgroup fills a CU's LDS (nothing overlaps the image gather) and has >= 8 tiles to walk;
// elsewhere the loop costs registers (occupancy) and loses.
template <int EMODE, int S, bool PERSIST = false>     // EMODE 1: raw store + BN statistics, 2: gradient sink; S: stride of the tile walk (dgrad: 1)
__global__ __launch_bounds__(256) void gconv_kernel(GcP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int gw = p.gw, gq = gw >> 2, pitch = gq + 1;          // quads per pixel (+1 pad)
    const int K = 9 * gw, KP = K + 4;
    constexpr int ITH = (GTH - 1) * S + 3, ITW = (GTW - 1) * S + 3;
    float4* At = reinterpret_cast<float4*>(smem);                       // [ITH*ITW][pitch]
    float* Bt = smem + (size_t)ITH * ITW * pitch * 4;                   // [32][KP]
    float* red = Bt + 32 * KP;                                          // [2][4][32]
    float* lco = red + 2 * 4 * 32;                                      // [2][32]: scale, shift of the group (statistics hand-over)

    int tIdx, g;
    xcd_block(tIdx, g);
    const int cbase = g * gw;
    const int ntiles = p.B * p.tilesH * p.tilesW;
    const bool from_stat = EMODE == 1 && !p.grad && p.in.stat.acc != nullptr;       // uniform
    // one thread per channel quad derives the 4 coefficient pairs from the producer's accumulators: the loads are issued HERE, the
    // arithmetic follows the weight staging below (the round trip hides behind it; every workgroup of the grid pays this
    // prologue); the workgroup of logical tile 0 of every group also stores the vectors / running statistics
    BnSums su[4];
    float sga[4] = {1.f, 1.f, 1.f, 1.f}, sbe[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned sbad = 0;
    if (from_stat && t < gq) {
        sbad = *p.in.stat.bad;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = min(cbase + 4 * t + i, p.C - 1);
            su[i] = bn_stat_load(p.in.stat, c);
            sga[i] = p.in.stat.gamma[c]; sbe[i] = p.in.stat.beta[c];
        }
    }
    // ---- stage B (this group's weights, k = tap*gw + i): row n = t/8, eight lanes walk k; four loads in flight
    {
        const int n = t >> 3, kk = t & 7;
        const int nc = min(n, gw - 1);
        int tap = 0, j = kk;
        while (j >= gw) { j -= gw; ++tap; }
        for (int k0 = kk; k0 < K; k0 += 8 * GUB) {
            float v[GUB];
#pragma unroll
            for (int u = 0; u < GUB; ++u) {
                const int tc = min(tap, 8);
                const size_t idx = p.wmode == 0 ? ((size_t)(cbase + nc) * gw + j) * 9 + tc : ((size_t)(cbase + j) * gw + nc) * 9 + (8 - tc);
                v[u] = p.w[idx];
                j += 8;
                while (j >= gw) { j -= gw; ++tap; }
            }
#pragma unroll
            for (int u = 0; u < GUB; ++u) {
                const int k = k0 + 8 * u;
                if (k < K) Bt[n * KP + k] = n < gw ? v[u] : 0.f;
            }
        }
    }
    const int lrow = lane & 31, lh = lane >> 5;
    if (from_stat) {
        if (t < gq) {
            const bool writer = tIdx == 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cbase + 4 * t + i;
                if (c < p.C) { const BnCoef bc = bn_stat_finish(p.in.stat, c, su[i], sbad != 0, writer, sga[i], sbe[i]); lco[4 * t + i] = bc.scale; lco[32 + 4 * t + i] = bc.shift; }
            }
        }
        __syncthreads();               // the coefficient table
    }
    do {                               // one pass unless PERSIST
    const int tw = tIdx % p.tilesW, th = (tIdx / p.tilesW) % p.tilesH, b = tIdx / (p.tilesW * p.tilesH);
    const int oy0 = th * GTH, ox0 = tw * GTW;
    const int ay0 = oy0 * S - 1, ax0 = ox0 * S - 1;
    if (PERSIST) __syncthreads();      // the previous tile's reads of At / red are done
    // ---- stage A (operand tile with halo)
    if (!p.grad) gc_stage_src<ITW, ITH * ITW>(At, pitch, gq, cbase, b, ay0, ax0, p.AH, p.AW, p.in, t, from_stat ? lco : nullptr);
    else gc_stage_gsrc<ITW, ITH * ITW>(At, pitch, gq, cbase, b, ay0, ax0, p.AH, p.AW, p.dy, p.up, p.DH, p.DW, t);
    __syncthreads();

    // ---- MFMA: wave wv owns pixels 32*wv .. +31 of the tile (two tile rows), all 32 (padded) output channels
    const int pp = 32 * wv + lrow, py = pp >> 4, px = pp & 15;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float4* bq = reinterpret_cast<const float4*>(Bt);
    const int KPq = KP >> 2;
    if ((gw & 7) == 0) {
        const int steps = gw >> 3;                  // 8-wide k steps per tap
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const float4* arow = At + ((py * S + ky) * ITW + px * S + kx) * pitch;
            const float4* brow = bq + lrow * KPq + (tap * gw >> 2);
            for (int s = 0; s < steps; ++s) {
                const float4 af = arow[2 * s + lh];
                const float4 bf = brow[2 * s + lh];
                acc = gmfma(af.x, bf.x, acc);
                acc = gmfma(af.y, bf.y, acc);
                acc = gmfma(af.z, bf.z, acc);
                acc = gmfma(af.w, bf.w, acc);
            }
        }
    } else {
        // gw % 8 == 4: 4-wide steps, lane half h takes k pair (2h, 2h+1) of the quad
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const float* arow = reinterpret_cast<const float*>(At + ((py * S + ky) * ITW + px * S + kx) * pitch);
            const float* brow = Bt + lrow * KP + tap * gw;
            for (int s = 0; s < gq; ++s) {
                const float a0 = arow[4 * s + 2 * lh], a1 = arow[4 * s + 2 * lh + 1];
                const float b0 = brow[4 * s + 2 * lh], b1 = brow[4 * s + 2 * lh + 1];
                acc = gmfma(a0, b0, acc);
                acc = gmfma(a1, b1, acc);
            }
        }
    }

    // ---- epilogue: C row = pixel (r&3) + 8*(r>>2) + 4*lh of this wave's 32, column = channel lrow (< gw valid)
    const int n = lrow;
    const bool nok = n < gw;
    const int ch = cbase + (nok ? n : 0);
    float s1 = 0.f, s2 = 0.f;
    bool want_stats = false;
    if (EMODE == 1) {
        want_stats = p.partial != nullptr || p.stat_acc != nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int oy = oy0 + (q >> 4), ox = ox0 + (q & 15);
            if (nok && oy < p.OH && ox < p.OW) {
                const float v = acc[r];
                p.out[(((size_t)b * p.OH + oy) * p.OW + ox) * p.ldo + ch] = v;
                s1 += v; s2 = fmaf(v, v, s2);
            }
        }
    } else {
        const pq_gsink& sk = p.sink;
        const bool affine = sk.scale != nullptr;
        const bool stats = sk.mean != nullptr && sk.partial != nullptr;
        const bool qmask = sk.finalize && sk.qstate != nullptr;
        const float qinv = qmask ? 1.f / sk.qstate[2] : 0.f, qzp = qmask ? sk.qstate[3] : 0.f;
        const bool need_y = sk.finalize && (sk.act != PQ_ACT_LINEAR || stats || qmask);
        want_stats = sk.finalize && stats;
        const ActParam sact = act_param(sk.act);
        const float ca = affine ? sk.scale[ch] : 1.f, cb = affine ? sk.shift[ch] : 0.f;
        const float cm = stats ? sk.mean[ch] : 0.f, ci = stats ? sk.invstd[ch] : 0.f;
        float yv[16], gv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int oy = min(oy0 + (q >> 4), p.OH - 1), ox = min(ox0 + (q & 15), p.OW - 1);
            const size_t row = ((size_t)b * p.OH + oy) * p.OW + ox;
            yv[r] = need_y ? sk.y[row * sk.ldy + ch] : 0.f;
            gv[r] = sk.accumulate ? sink_old(sk)[row * sk.ldg + ch] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int oy = oy0 + (q >> 4), ox = ox0 + (q & 15);
            float dz = acc[r] + gv[r];
            if (need_y) {
                const float uu = fmaf(ca, yv[r], cb);
                dz *= act_bwd(uu, sact);
                if (qmask) { const float qv = rintf(act_fwd(uu, sact) * qinv) + qzp; dz = (qv >= 0.f && qv <= 255.f) ? dz : 0.f; }
            }
            if (nok && oy < p.OH && ox < p.OW) {
                sk.g[(((size_t)b * p.OH + oy) * p.OW + ox) * sk.ldg + ch] = dz;
                if (stats) { s1 += dz; s2 = fmaf(dz, (yv[r] - cm) * ci, s2); }
            }
        }
    }
    if (want_stats) {       // uniform
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lh == 0) { red[(0 * 4 + wv) * 32 + lrow] = s1; red[(1 * 4 + wv) * 32 + lrow] = s2; }
        __syncthreads();
        if (t < 32 && t < gw) {
            float a = 0.f, bsum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { a += red[(0 * 4 + q) * 32 + t]; bsum += red[(1 * 4 + q) * 32 + t]; }
            float* part = (EMODE == 2) ? p.sink.partial : p.partial;
            const int ldp = (EMODE == 2) ? p.sink.ldp : p.ldp;
            if (EMODE == 1 && p.stat_acc != nullptr) {       // no finalize launch: the first consumer derives the coefficients
                bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, ldp, tIdx, 0, cbase + t, a);
                bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, ldp, tIdx, 1, cbase + t, bsum);
            } else {
                float* dst = part + (size_t)tIdx * 2 * ldp;
                dst[cbase + t] = a;
                dst[ldp + cbase + t] = bsum;
            }
        }
    }
    if (!PERSIST) break;
    tIdx += gridDim.x;
    } while (tIdx < ntiles);
}

// ------------------------------------------------------------------------------------------------ stride-2 data gradient
// dx[2a+pi, 2b+pj] only receives the taps with ky = pi+1 (mod 2), kx = pj+1 (mod 2): 1 / 2 / 2 / 4 taps for the four pixel
// parities instead of 9 taps on a zero-inserted plane (4x the MFMA work, 4x the LDS tile, 3/4 of the staged pixels zero).
// A workgroup owns a 16x32 tile of dx of one group = four 8x16 parity classes; it stages the 9x17 dy' tile once (no halo on
// top / left: odd rows also use dy row a+1), wave wv owns pixels 32*wv..+31 of every class.
//   class (pi, pj), pixel (a, b):  dx = sum over (ky, dr) in {pi=0: (1,0); pi=1: (0,+1), (2,0)} x (kx, dc) likewise of
//   dy'[a+dr][b+dc][o] * w[o][i][ky][kx]
__global__ __launch_bounds__(256) void gconv_dgrad_s2_kernel(GcP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int DTH = 9, DTW = 17, NPIX = DTH * DTW;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int gw = p.gw, gq = gw >> 2, pitch = gq + 1;
    const int K = 9 * gw, KP = K + 4;
    float4* At = reinterpret_cast<float4*>(smem);                       // [9*17][pitch]   dy' tile
    float* Bt = smem + (size_t)NPIX * pitch * 4;                        // [32][KP]        B[i][tap*gw + o] = w[o][i][tap]
    float* red = Bt + 32 * KP;                                          // [2][4][32]
    int tIdx, g;
    xcd_block(tIdx, g);
    const int tw = tIdx % p.tilesW, th = (tIdx / p.tilesW) % p.tilesH, b = tIdx / (p.tilesW * p.tilesH);
    const int iy0 = th * 16, ix0 = tw * 32;                             // dx tile origin (p.OH x p.OW is the dx plane here)
    const int cbase = g * gw;
    gc_stage_gsrc<DTW, NPIX>(At, pitch, gq, cbase, b, iy0 >> 1, ix0 >> 1, p.DH, p.DW, p.dy, 0, 0, 0, t);
    {
        const int n = t >> 3, kk = t & 7;
        const int nc = min(n, gw - 1);
        int tap = 0, j = kk;
        while (j >= gw) { j -= gw; ++tap; }
        for (int k0 = kk; k0 < K; k0 += 8 * GUB) {
            float v[GUB];
#pragma unroll
            for (int u = 0; u < GUB; ++u) {
                v[u] = p.w[((size_t)(cbase + j) * gw + nc) * 9 + min(tap, 8)];      // n = i, j = o
                j += 8;
                while (j >= gw) { j -= gw; ++tap; }
            }
#pragma unroll
            for (int u = 0; u < GUB; ++u) {
                const int k = k0 + 8 * u;
                if (k < K) Bt[n * KP + k] = n < gw ? v[u] : 0.f;
            }
        }
    }
    __syncthreads();

    const int lrow = lane & 31, lh = lane >> 5;
    const int pp = 32 * wv + lrow, pa = pp >> 4, pb = pp & 15;
    const int n = lrow;
    const bool nok = n < gw;
    const int ch = cbase + (nok ? n : 0);
    const pq_gsink& sk = p.sink;
    const bool affine = sk.scale != nullptr;
    const bool stats = sk.mean != nullptr && sk.partial != nullptr;
    const bool qmask = sk.finalize && sk.qstate != nullptr;
    const float qinv = qmask ? 1.f / sk.qstate[2] : 0.f, qzp = qmask ? sk.qstate[3] : 0.f;
    const bool need_y = sk.finalize && (sk.act != PQ_ACT_LINEAR || stats || qmask);
    const bool want_stats = sk.finalize && stats;
    const ActParam sact = act_param(sk.act);
    const float ca = affine ? sk.scale[ch] : 1.f, cb = affine ? sk.shift[ch] : 0.f;
    const float cm = stats ? sk.mean[ch] : 0.f, ci = stats ? sk.invstd[ch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const float4* bq = reinterpret_cast<const float4*>(Bt);
    const int KPq = KP >> 2;

#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int pi = cls >> 1, pj = cls & 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
            if (ty > pi) continue;                                  // pi = 0: one row tap, pi = 1: two
            const int ky = pi ? 2 * ty : 1, dr = pi ? 1 - ty : 0;   // (ky=0, dr=+1), (ky=2, dr=0) | (ky=1, dr=0)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                if (tx > pj) continue;
                const int kx = pj ? 2 * tx : 1, dc = pj ? 1 - tx : 0;
                const int tap = ky * 3 + kx;
                const float4* arow = At + ((pa + dr) * DTW + pb + dc) * pitch;
                if ((gw & 7) == 0) {
                    const float4* brow = bq + lrow * KPq + (tap * gw >> 2);
                    for (int s_ = 0; s_ < (gw >> 3); ++s_) {
                        const float4 af = arow[2 * s_ + lh];
                        const float4 bf = brow[2 * s_ + lh];
                        acc = gmfma(af.x, bf.x, acc);
                        acc = gmfma(af.y, bf.y, acc);
                        acc = gmfma(af.z, bf.z, acc);
                        acc = gmfma(af.w, bf.w, acc);
                    }
                } else {
                    const float* ar = reinterpret_cast<const float*>(arow);
                    const float* br = Bt + lrow * KP + tap * gw;
                    for (int s_ = 0; s_ < gq; ++s_) {
                        acc = gmfma(ar[4 * s_ + 2 * lh], br[4 * s_ + 2 * lh], acc);
                        acc = gmfma(ar[4 * s_ + 2 * lh + 1], br[4 * s_ + 2 * lh + 1], acc);
                    }
                }
            }
        }
        // sink epilogue of this parity class: C row r = pixel q of the wave's 32, column = channel lrow
        float yv[16], gv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int iy = min(iy0 + 2 * (q >> 4) + pi, p.OH - 1), ix = min(ix0 + 2 * (q & 15) + pj, p.OW - 1);
            const size_t row = ((size_t)b * p.OH + iy) * p.OW + ix;
            yv[r] = need_y ? sk.y[row * sk.ldy + ch] : 0.f;
            gv[r] = sk.accumulate ? sink_old(sk)[row * sk.ldg + ch] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int iy = iy0 + 2 * (q >> 4) + pi, ix = ix0 + 2 * (q & 15) + pj;
            float dz = acc[r] + gv[r];
            if (need_y) {
                const float uu = fmaf(ca, yv[r], cb);
                dz *= act_bwd(uu, sact);
                if (qmask) { const float qv = rintf(act_fwd(uu, sact) * qinv) + qzp; dz = (qv >= 0.f && qv <= 255.f) ? dz : 0.f; }
            }
            if (nok && iy < p.OH && ix < p.OW) {
                sk.g[(((size_t)b * p.OH + iy) * p.OW + ix) * sk.ldg + ch] = dz;
                if (stats) { s1 += dz; s2 = fmaf(dz, (yv[r] - cm) * ci, s2); }
            }
        }
    }
    if (want_stats) {       // uniform.  The statistics rows are those of the 8x16 tiling (pq_gconv3x3_blocks): this workgroup
        s1 += __shfl_xor(s1, 32);       // covers up to four of them -- sums go to the first, zeros to the others
        s2 += __shfl_xor(s2, 32);
        if (lh == 0) { red[(0 * 4 + wv) * 32 + lrow] = s1; red[(1 * 4 + wv) * 32 + lrow] = s2; }
        __syncthreads();
        if (t < 32 && t < gw) {
            float a = 0.f, bsum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { a += red[(0 * 4 + q) * 32 + t]; bsum += red[(1 * 4 + q) * 32 + t]; }
            const int tH8 = (p.OH + GTH - 1) / GTH, tW16 = (p.OW + GTW - 1) / GTW;
            for (int sy = 0; sy < 2; ++sy)
                for (int sx = 0; sx < 2; ++sx) {
                    const int r8 = 2 * th + sy, c16 = 2 * tw + sx;
                    if (r8 < tH8 && c16 < tW16) {
                        float* dst = sk.partial + ((size_t)(b * tH8 + r8) * tW16 + c16) * 2 * sk.ldp;
                        const bool first = sy == 0 && sx == 0;
                        dst[cbase + t] = first ? a : 0.f;
                        dst[sk.ldp + cbase + t] = first ? bsum : 0.f;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct GcWP {
    pq_gsrc dy; pq_src in;
    float* partial;          // [G][nsplit][gw][9*gw]   (k = tap*gw + i)
    int B, IH, IW, OH, OW, C, gw, S, tilesH, tilesW, nsplit;
};

template <int NC, int S>      // NC: 32-wide column chunks of the (tap, i) axis, ceil(9*gw/32) <= NC; S: stride
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(GcWP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int gw = p.gw, gq = gw >> 2, pitch = gq + 1;
    const int K = 9 * gw, NCH = (K + 31) / 32;
    constexpr int ITH = (GTH - 1) * S + 3, ITW = (GTW - 1) * S + 3;
    // x' tile [ITH*ITW][pitch*4] (at least 4096 floats: the region doubles as the [4][32*32] scratch of the final reduction over the
    // four waves -- a weight-gradient workgroup lives for the whole launch on the side stream, and every KB of LDS it holds is a KB the
    // main stream's workgroups on that CU cannot have), then dy' [128][32] (o padded to 32)
    float* Xt = smem;
    const int xfl = ITH * ITW * pitch * 4;
    float* Yt = Xt + (xfl > 4096 ? xfl : 4096);
    float* red = smem;
    const int g = blockIdx.x, split = blockIdx.y;
    const int cbase = g * gw;
    const int ntiles = p.B * p.tilesH * p.tilesW;
    const int lrow = lane & 31, lh = lane >> 5;

    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // per-lane decode of its B column j = 32*c + lrow -> (tap, i) -> LDS offset relative to the pixel's tile origin
    int boff[NC];
    bool bok[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = 32 * c + lrow;
        bok[c] = c < NCH && k < K;
        const int kk = bok[c] ? k : 0;
        const int tap = kk / gw, i = kk - tap * gw;
        const int ky = tap / 3, kx = tap - 3 * ky;
        boff[c] = (ky * ITW + kx) * pitch * 4 + i;
    }

    float yc[4][4];          // p, q, mu, r of this thread's dy' channel quad (t & 7)
    {
        const int qd = t & 7;
        const int c = cbase + 4 * (qd < gq ? qd : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            yc[0][j] = p.dy.y ? p.dy.p[c + j] : 1.f; yc[1][j] = p.dy.y ? p.dy.q[c + j] : 0.f;
            yc[2][j] = p.dy.y ? p.dy.mu[c + j] : 0.f; yc[3][j] = p.dy.y ? p.dy.r[c + j] : 0.f;
        }
    }
    for (int tile = split; tile < ntiles; tile += p.nsplit) {
        const int tw = tile % p.tilesW, th = (tile / p.tilesW) % p.tilesH, b = tile / (p.tilesW * p.tilesH);
        const int oy0 = th * GTH, ox0 = tw * GTW;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        __syncthreads();
        // stage x' tile
        gc_stage_src<ITW, ITH * ITW>(reinterpret_cast<float4*>(Xt), pitch, gq, cbase, b, iy0, ix0, p.IH, p.IW, p.in, t);
        // stage dy' tile: [pixel 0..127][o 0..31]; thread = (quad t&7, pixel lane t>>3), four pixels in flight
        {
            const int qd = t & 7, pl = t >> 3;
            const bool qok = qd < gq;
            const int c = cbase + 4 * (qok ? qd : 0);
            float4 rg[4], ry[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pix = pl + 32 * u;
                const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
                ok[u] = qok && oy < p.OH && ox < p.OW;
                const size_t r = ((size_t)b * p.OH + min(oy, p.OH - 1)) * p.OW + min(ox, p.OW - 1);
                rg[u] = ldg4(p.dy.g + r * p.dy.ldg + c);
                ry[u] = p.dy.y ? ldg4(p.dy.y + r * p.dy.ldy + c) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pix = pl + 32 * u;
                float e4[4] = {rg[u].x, rg[u].y, rg[u].z, rg[u].w};
                if (p.dy.y) {
                    const float y4[4] = {ry[u].x, ry[u].y, ry[u].z, ry[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) e4[j] = fmaf(yc[0][j], e4[j], fmaf(yc[1][j], y4[j] - yc[2][j], yc[3][j]));
                }
                *reinterpret_cast<float4*>(&Yt[pix * 32 + 4 * qd]) = ok[u] ? make_float4(e4[0], e4[1], e4[2], e4[3]) : f4zero();
            }
        }
        __syncthreads();
        // wave wv: pixels 32*wv .. +31 (k dimension), 16 steps of 2 pixels.  All NC chunks are multiplied (columns past
        // 9*gw read a valid offset and are masked to zero): the operands of a step are fetched from LDS first and the MFMAs
        // follow back to back -- per-chunk branches made every MFMA wait for its own LDS read.
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
            const int pp = 32 * wv + 2 * ks + lh;
            const int py = pp >> 4, px = pp & 15;
            const float a = Yt[pp * 32 + lrow];
            const float* xb = Xt + ((py * S) * ITW + px * S) * pitch * 4;
            float bv[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) bv[c] = xb[boff[c]];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = gmfma(a, bok[c] ? bv[c] : 0.f, acc[c]);
        }
    }
    // reduce the four waves' accumulators chunk by chunk; D[o][j]: row o = (r&3)+8*(r>>2)+4*lh, column j = lrow
    float* dst = p.partial + ((size_t)(g * p.nsplit + split) * gw) * K;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < NCH) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wv * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lrow] = acc[c][r];
            __syncthreads();
            for (int e = t; e < 1024; e += 256) {
                const int o = e >> 5, j = e & 31;
                const int k = 32 * c + j;
                if (o < gw && k < K) dst[(size_t)o * K + k] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            }
        }
    }
}

// dw[(cbase+o)][i][tap] = sum_split partial[g][split][o][tap*gw+i]; block = 32 elements x 32 split lanes (a thread per
// element walking all splits was one memory latency per split: 60 us of a 110 us weight gradient)
__global__ __launch_bounds__(1024) void gconv_wgrad_reduce_kernel(const float* partial, int G, int nsplit, int gw, float* dw) {
    __shared__ float red[32][33];
    const int K = 9 * gw;
    const int total = G * gw * K;
    const int e = blockIdx.x * 32 + threadIdx.x;
    float s = 0.f;
    int k = 0, o = 0, g = 0;
    if (e < total) {
        k = e % K; o = (e / K) % gw; g = e / (K * gw);
        const float* src = partial + ((size_t)g * nsplit * gw + o) * K + k;
        const size_t stride = (size_t)gw * K;
        int sp = threadIdx.y;
        for (; sp + 3 * 32 < nsplit; sp += 4 * 32) {
            const float v0 = src[(size_t)sp * stride], v1 = src[(size_t)(sp + 32) * stride];
            const float v2 = src[(size_t)(sp + 64) * stride], v3 = src[(size_t)(sp + 96) * stride];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; sp < nsplit; sp += 32) s += src[(size_t)sp * stride];
    }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && e < total) {
        float tsum = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tsum += red[q][threadIdx.x];
        const int tap = k / gw, i = k - tap * gw;
        dw[((size_t)(g * gw + o) * gw + i) * 9 + tap] = tsum;
    }
}

}  // namespace pq

using namespace pq;

static int gc_check(int B, int IH, int IW, int OH, int OW, int C, int gw, int stride) {
    if (B <= 0 || IH <= 0 || IW <= 0 || C <= 0 || gw <= 0 || gw > 32 || (gw & 3) || C % gw) return 0;
    if (stride != 1 && stride != 2) return 0;
    return OH == (IH - 1) / stride + 1 && OW == (IW - 1) / stride + 1;
}
static size_t gc_lds_bytes(int gw, int S) {
    const int ITH = (GTH - 1) * S + 3, ITW = (GTW - 1) * S + 3, pitch = gw / 4 + 1;
    return ((size_t)ITH * ITW * pitch * 4 + 32 * (9 * gw + 4) + 2 * 4 * 32 + 2 * 32) * sizeof(float);
}
extern "C" int32_t pq_gconv3x3_blocks(int32_t B, int32_t H, int32_t W) { return B * ((H + GTH - 1) / GTH) * ((W + GTW - 1) / GTW); }

extern "C" int pq_gconv3x3_fwd(const pq_gconv3x3_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->w && a->out, "null pointer");
    PQ_CHECK_ARG(gc_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->gw, a->stride), "bad shape (group width multiple of 4 and <= 32, stride 1|2, pad 1)");
    PQ_CHECK_ARG((a->in.ld & 3) == 0, "input pitch must be a multiple of 4");
    if (act_is_special(a->in.act)) { pq_set_error("pq_gconv3x3_fwd: logistic/tanh operands must be materialised first"); return PQ_EUNSUPPORTED; }
    PQ_CHECK_ARG(a->stat_acc == nullptr || (a->stat_bad != nullptr && a->ldp >= a->C), "stat_acc needs stat_bad");
    PQ_CHECK_ARG(a->stat_acc == nullptr || stat_adds_ok(pq_gconv3x3_blocks(a->B, a->OH, a->OW), a->stat_slots), "too many row blocks per accumulator slot");
    PQ_CHECK_ARG(a->in.stat.acc == nullptr || !a->in.qstate, "the statistics hand-over has no QAT path");
    GcP p{};
    p.in = a->in; p.grad = 0; p.up = 0; p.w = a->w; p.wmode = 0; p.out = a->out; p.partial = a->partial; p.ldo = a->ldo; p.ldp = a->ldp;
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    p.B = a->B; p.AH = a->IH; p.AW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.gw = a->gw; p.S = a->stride;
    p.tilesH = (a->OH + GTH - 1) / GTH; p.tilesW = (a->OW + GTW - 1) / GTW;
    const size_t lds = gc_lds_bytes(a->gw, a->stride);
    const dim3 grid(a->B * p.tilesH * p.tilesW, a->C / a->gw);
    if (a->stride == 1) {
        (void)hipFuncSetAttribute((const void*)gconv_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        PQ_LAUNCH((gconv_kernel<1, 1>), grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        // big stride-2 layers: persistent workgroups (one resident wave of them), at least 8 tiles each
        const int G = a->C / a->gw, ntiles = a->B * p.tilesH * p.tilesW;
        int occ = (int)((160 * 1024) / lds); if (occ > 3) occ = 3; if (occ < 1) occ = 1;
        int nb = 256 * occ / G; if (nb < 1) nb = 1;
        const int per = (ntiles + nb - 1) / nb;
        if (nb < ntiles && per >= 8) {
            (void)hipFuncSetAttribute((const void*)gconv_kernel<1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            PQ_LAUNCH((gconv_kernel<1, 2, true>), dim3((ntiles + per - 1) / per, G), dim3(256), lds, (hipStream_t)stream, p);
        } else {
            (void)hipFuncSetAttribute((const void*)gconv_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            PQ_LAUNCH((gconv_kernel<1, 2>), grid, dim3(256), lds, (hipStream_t)stream, p);
        }
    }
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_gconv3x3_dgrad(const pq_gconv3x3_dgrad_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->w && a->dx.g, "null pointer");
    PQ_SINK_ACC_UNSUPPORTED(a->dx);
    PQ_SINK_ADDEND_CHECK(a->dx);
    PQ_CHECK_ARG(gc_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->gw, a->stride), "bad shape (group width multiple of 4 and <= 32, stride 1|2, pad 1)");
    PQ_CHECK_ARG(!(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    if (act_is_special(a->dx.act)) { pq_set_error("pq_gconv3x3_dgrad: logistic/tanh sinks are not supported"); return PQ_EUNSUPPORTED; }
    if (a->stride == 2) {
        GcP q{};
        q.dy = a->dy; q.grad = 1; q.w = a->w; q.sink = a->dx; q.DH = a->OH; q.DW = a->OW;
        q.B = a->B; q.OH = a->IH; q.OW = a->IW; q.C = a->C; q.gw = a->gw; q.S = 2;
        q.tilesH = (a->IH + 15) / 16; q.tilesW = (a->IW + 31) / 32;
        const int pitch = a->gw / 4 + 1;
        const size_t lds2 = ((size_t)9 * 17 * pitch * 4 + 32 * (9 * a->gw + 4) + 2 * 4 * 32) * sizeof(float);
        (void)hipFuncSetAttribute((const void*)gconv_dgrad_s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        PQ_LAUNCH(gconv_dgrad_s2_kernel, dim3(a->B * q.tilesH * q.tilesW, a->C / a->gw), dim3(256), lds2, (hipStream_t)stream, q);
        PQ_CHECK_LAUNCH();
        return PQ_OK;
    }
    GcP p{};
    p.dy = a->dy; p.grad = 1; p.w = a->w; p.wmode = 1; p.sink = a->dx;
    p.up = a->stride == 2; p.DH = a->OH; p.DW = a->OW;
    // the transposed conv runs as a stride-1 conv over the (zero-inserted) dy plane laid out on the INPUT grid
    p.B = a->B; p.AH = a->IH; p.AW = a->IW; p.OH = a->IH; p.OW = a->IW; p.C = a->C; p.gw = a->gw; p.S = 1;
    if (a->stride == 1) { p.AH = a->OH; p.AW = a->OW; }
    p.tilesH = (a->IH + GTH - 1) / GTH; p.tilesW = (a->IW + GTW - 1) / GTW;
    const size_t lds = gc_lds_bytes(a->gw, 1);
    (void)hipFuncSetAttribute((const void*)gconv_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PQ_LAUNCH((gconv_kernel<2, 1>), dim3(a->B * p.tilesH * p.tilesW, a->C / a->gw), dim3(256), lds, (hipStream_t)stream, p);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

// pq_bn_stat_ok for PQ_OP_GCONV_FWD: both roles (statistics added per 8 x 16 tile; coefficients derived per group)
int gconv_fwd_stat_ok(const pq_gconv3x3_fwd_args* a, int role) {
    if (!a) return 0;
    return role == 0 ? 1 : (a->in.qstate == nullptr);
}

static int gc_nsplit(int ntiles, int G) {
    int want = (384 + G - 1) / G;          // workgroups per launch (256 / 384 / 512 / 768 / 1024: regnetx-600m-fpn step 11.42 / 11.40 / 11.45 / 11.51 / 11.61 ms)
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    return want;
}
extern "C" int64_t pq_gconv3x3_wgrad_scratch_bytes(int32_t B, int32_t OH, int32_t OW, int32_t C, int32_t gw) {
    const int G = C / gw, ntiles = pq_gconv3x3_blocks(B, OH, OW);
    return (int64_t)G * gc_nsplit(ntiles, G) * gw * 9 * gw * 4;
}

extern "C" int pq_gconv3x3_wgrad(const pq_gconv3x3_wgrad_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->in.x && a->scratch, "null pointer");          // dw == NULL: deferred reduction
    PQ_CHECK_ARG(gc_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->gw, a->stride), "bad shape (group width multiple of 4 and <= 32, stride 1|2, pad 1)");
    if (act_is_special(a->in.act)) { pq_set_error("pq_gconv3x3_wgrad: logistic/tanh operands must be materialised first"); return PQ_EUNSUPPORTED; }
    GcWP p{};
    p.dy = a->dy; p.in = a->in; p.partial = a->scratch;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.gw = a->gw; p.S = a->stride;
    p.tilesH = (a->OH + GTH - 1) / GTH; p.tilesW = (a->OW + GTW - 1) / GTW;
    const int G = a->C / a->gw, ntiles = a->B * p.tilesH * p.tilesW;
    p.nsplit = gc_nsplit(ntiles, G);
    const int S = a->stride, ITH = (GTH - 1) * S + 3, ITW = (GTW - 1) * S + 3, pitch = a->gw / 4 + 1;
    const size_t xfl = (size_t)ITH * ITW * pitch * 4;
    const size_t lds = ((xfl > 4096 ? xfl : 4096) + 128 * 32) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int nch = (9 * a->gw + 31) / 32;
    const dim3 wgrid(G, p.nsplit);
#define PQ_GCW_LAUNCH(NCH_, S_)                                                                                              \
    do {                                                                                                                     \
        (void)hipFuncSetAttribute((const void*)gconv_wgrad_kernel<NCH_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        PQ_LAUNCH((gconv_wgrad_kernel<NCH_, S_>), wgrid, dim3(256), lds, st, p);                                     \
    } while (0)
    if (S == 1) {
        if (nch <= 3) PQ_GCW_LAUNCH(3, 1); else if (nch <= 5) PQ_GCW_LAUNCH(5, 1); else if (nch <= 7) PQ_GCW_LAUNCH(7, 1); else PQ_GCW_LAUNCH(9, 1);
    } else {
        if (nch <= 3) PQ_GCW_LAUNCH(3, 2); else if (nch <= 5) PQ_GCW_LAUNCH(5, 2); else if (nch <= 7) PQ_GCW_LAUNCH(7, 2); else PQ_GCW_LAUNCH(9, 2);
    }
#undef PQ_GCW_LAUNCH
    PQ_CHECK_LAUNCH();
    if (!a->dw) return PQ_OK;          // the splits' slabs stay in `scratch` (pq_gconv3x3_wgrad_reduce_item / pq_reduce_batch)
    const int total = G * a->gw * 9 * a->gw;
    PQ_LAUNCH(gconv_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, G, p.nsplit, a->gw, a->dw);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_gconv3x3_wgrad_reduce_item(const pq_gconv3x3_wgrad_args* a, pq_reduce_item* out) {
    PQ_CHECK_ARG(a && out && a->dw && a->scratch, "null pointer");
    PQ_CHECK_ARG(gc_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->gw, a->stride), "bad shape");
    const int G = a->C / a->gw, ntiles = a->B * ((a->OH + GTH - 1) / GTH) * ((a->OW + GTW - 1) / GTW);
    const int nsplit = gc_nsplit(ntiles, G), ng = a->gw * 9 * a->gw;
    // partial[g][split][o][tap * gw + i] -> dw[g * gw + o][i][tap]
    *out = pq_reduce_item{a->scratch, a->dw, nullptr, (int64_t)ng, (int64_t)nsplit * ng, nsplit, G * ng, 0, ng, a->gw, 0, 0, 0};
    return PQ_OK;
}
