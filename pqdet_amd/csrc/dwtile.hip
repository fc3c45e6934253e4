// Depthwise 3x3 (pad 1, stride 1|2) as LDS-staged NHWC tiles, gfx950.
//
// A workgroup (256 threads) owns an OTH x OTW output tile of one image for a chunk of 32 channels (8 float4 quads).
// The input tile with its halo is staged ONCE into LDS with coalesced 128-byte-per-pixel loads; the producer's
// BatchNorm affine + activation (or, in the backward kernel, the closed-form BN backward  dy = p*g + q*(y-mu) + r) is
// applied at staging time, so every element is loaded from HBM and transformed exactly once.  Threads then slide a
// 3 x (TW*S+2) register window over the LDS tile (lane = 8 quads x 32 pixel lanes).  Pixel pitch in LDS is 9 quads
// (144 B) to spread the 16-lane ds_read_b128 groups over the banks.
//
// The backward kernel is fused: one staging of (g_out, y_out) and y_in yields BOTH the input gradient (written through
// the gradient sink: accumulate / act' / BN-backward sums) and the per-block weight-gradient partials -- 3 tensor reads +
// 1 write instead of the 7 passes of separate dgrad + wgrad kernels.
#include <cstdlib>
#include "common.h"

namespace pq {

constexpr int CBQ = 8;            // channel quads per block (32 channels)
constexpr int PQP = 9;            // LDS pixel pitch in quads (8 + 1 pad)

template <int S> struct DwTile {
    static constexpr int OTH = 8;
    static constexpr int OTW = (S == 1) ? 16 : 8;
    static constexpr int TW = OTW / 4;                       // outputs per thread along W (32 pixel lanes = 8 rows x 4)
    static constexpr int ITH = (OTH - 1) * S + 3;
    static constexpr int ITW = (OTW - 1) * S + 3;
    static constexpr int NCOL = (TW - 1) * S + 3;
};

struct DwTP {
    pq_src in; pq_gsrc dy; const float* w;
    float* out; float* partial; int ldo, ldp;
    pq_gsink sink; float* wpartial;
    int B, IH, IW, OH, OW, C, C4, tilesH, tilesW;
    int minmax;       // forward: `partial` rows have 4 planes (+ per-channel min, max of the outputs)
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;     // forward: statistics into pq_bn_stat accumulators instead of rows
};

__device__ __forceinline__ float4 lds_px(const float4* tile, int pix, int q) { return tile[pix * PQP + q]; }

// ------------------------------------------------------------------------------------------------ forward
template <int S>
__global__ __launch_bounds__(256) void dwt_fwd_kernel(DwTP p) {
    using T = DwTile<S>;
    __shared__ float4 tile[T::ITH * T::ITW * PQP];
    __shared__ __attribute__((aligned(16))) float red[32 * 8 * 8 + 64];
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;                       // quad, pixel lane
    int tIdx, bcy;
    xcd_block(tIdx, bcy);
    const int tw = tIdx % p.tilesW, th = (tIdx / p.tilesW) % p.tilesH, b = tIdx / (p.tilesW * p.tilesH);
    const int oy0 = th * T::OTH, ox0 = tw * T::OTW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int cq0 = bcy * CBQ;                       // first quad of this block's channel chunk
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;                            // clamped channel offset for address generation
    // (with p.in.stat: the workgroup derives its 32 channels' coefficients from the producer's accumulators -- one 16-byte load
    // per thread -- and tile 0 also stores the vectors for every later consumer)
    const SrcCoef4 xk = p.in.stat.acc ? src_coef4_chunk32(p.in, cq0 * 4, q, p.C, tIdx == 0, red) : src_coef4(p.in, cc, p.C);

    // ---- stage the input tile (transform once, zero outside the image); every load of the lane is issued before the first LDS
    // write: one exposed memory round trip for the whole tile
    constexpr int NPIX = T::ITH * T::ITW;
    constexpr int NITS = (NPIX + 31) / 32;
    {
        float4 r[NITS];
#pragma unroll
        for (int u = 0; u < NITS; ++u) {
            const int pix = min(pl + 32 * u, NPIX - 1);
            const int py = pix / T::ITW, px = pix % T::ITW;
            const int iy = min(max(iy0 + py, 0), p.IH - 1), ix = min(max(ix0 + px, 0), p.IW - 1);
            r[u] = ldg4(p.in.x + (((size_t)b * p.IH + iy) * p.IW + ix) * p.in.ld + cc);
        }
#pragma unroll
        for (int u = 0; u < NITS; ++u) {
            const int pix = pl + 32 * u;
            if (pix < NPIX) {
                const int py = pix / T::ITW, px = pix % T::ITW;
                const int iy = iy0 + py, ix = ix0 + px;
                const bool ok = cok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                tile[pix * PQP + q] = ok ? src_apply4(r[u], xk) : f4zero();
            }
        }
    }
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (cok && c0 + j < p.C) ? p.w[(size_t)(c0 + j) * 9 + k] : 0.f;
        wk[k] = make_float4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();

    // ---- compute: lane pl -> output row pl/4, columns (pl%4)*TW .. +TW-1
    const int orow = pl >> 2, ocol0 = (pl & 3) * T::TW;
    float4 acc[T::TW];
#pragma unroll
    for (int u = 0; u < T::TW; ++u) acc[u] = f4zero();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float4 row[T::NCOL];
#pragma unroll
        for (int j = 0; j < T::NCOL; ++j) row[j] = lds_px(tile, (orow * S + ky) * T::ITW + ocol0 * S + j, q);
#pragma unroll
        for (int j = 0; j < T::NCOL; ++j)
#pragma unroll
            for (int u = 0; u < T::TW; ++u) {
                const int kx = j - u * S;
                if (kx >= 0 && kx < 3) acc[u] = f4fma(row[j], wk[ky * 3 + kx], acc[u]);
            }
    }
    float4 s1 = f4zero(), s2 = f4zero();
    const float inf = __builtin_huge_valf();
    float4 mn = make_float4(inf, inf, inf, inf), mx = make_float4(-inf, -inf, -inf, -inf);
    const int oy = oy0 + orow;
#pragma unroll
    for (int u = 0; u < T::TW; ++u) {
        const int ox = ox0 + ocol0 + u;
        if (cok && oy < p.OH && ox < p.OW) {
            stg4(p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * p.ldo + c0, acc[u]);
            s1 = f4add(s1, acc[u]);
            s2 = f4fma(acc[u], acc[u], s2);
            mn = f4min(mn, acc[u]); mx = f4max(mx, acc[u]);
        }
    }
    if (p.partial || p.stat_acc) {      // uniform
        const int planes = p.minmax ? 4 : 2;
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
        if (p.minmax) {   // planes 2 / 3: per-channel extremes of the tile (QAT observers)
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

// ------------------------------------------------------------------------------------------------ fused backward
// Tile over the INPUT plane (dx).  S=1: 8x16 inputs, dy' tile 10x18, x' tile 10x18.
//                                  S=2: 16x16 inputs (outputs 8x8 owned), dy' tile 9x9, x' tile 17x17.
template <int S> struct DwBTile {
    static constexpr int DXH = (S == 1) ? 8 : 16;
    static constexpr int DXW = 16;
    static constexpr int OWH = DXH / S, OWW = DXW / S;                 // outputs owned by the tile (for the weight gradient)
    static constexpr int DYH = (S == 1) ? DXH + 2 : OWH + 1;          // dy' tile (with halo)
    static constexpr int DYW = (S == 1) ? DXW + 2 : OWW + 1;
    static constexpr int XH = (OWH - 1) * S + 3, XW = (OWW - 1) * S + 3;   // x' tile (with halo)
};

// stride-2 dgrad helper: add the contribution of dy' tile row `drow` through kernel row KROW to 4 consecutive input columns
// starting at the (multiple-of-4) local column cx0.  Input column u: tap a = output (u+1)>>1 with kernel column (u+1)&1,
// tap b (odd u only) = the output before it with kernel column 2.
template <int KROW, int DYW>
__device__ __forceinline__ void s2_dgrad_row(const float4* dyt, int drow, int cx0, int q, const float4 (&wk)[9], float4 (&dxr)[4]) {
    float4 d[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j] = lds_px(dyt, drow * DYW + (cx0 >> 1) + j, q);
    dxr[0] = f4fma(d[0], wk[KROW * 3 + 1], dxr[0]);
    dxr[1] = f4fma(d[1], wk[KROW * 3 + 0], dxr[1]);
    dxr[1] = f4fma(d[0], wk[KROW * 3 + 2], dxr[1]);
    dxr[2] = f4fma(d[1], wk[KROW * 3 + 1], dxr[2]);
    dxr[3] = f4fma(d[2], wk[KROW * 3 + 0], dxr[3]);
    dxr[3] = f4fma(d[1], wk[KROW * 3 + 2], dxr[3]);
}

// Stride 1, two phases through ONE LDS tile (34 KB -> 4 workgroups per CU, ~100 VGPRs):
//   phase 1: stage dy' (10x18), input gradient from LDS with the flipped taps -> sink; the owned dy' values stay in registers
//   phase 2: stage x'  (10x18) into the same buffer, weight-gradient taps
// PART: 0 = both gradients, 1 = input gradient only (phase 1), 2 = weight gradient only (phase 2; the owned dy' values are
// read straight from global memory) -- the engine runs 1 on the main stream and 2 on the weight-gradient stream.
template <int PART>
__global__ __launch_bounds__(256, 3) void dwt_bwd_s1_kernel(DwTP p) {
    constexpr int TH = 8, TWD = 16, PH = TH + 2, PW = TWD + 2, NPX = PH * PW;
    __shared__ float4 tile[NPX * PQP];
    __shared__ float red[32 * 8 * 8];
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;
    int tIdx, bcy;
    xcd_block(tIdx, bcy);
    const int tw = tIdx % p.tilesW, th = (tIdx / p.tilesW) % p.tilesH, b = tIdx / (p.tilesW * p.tilesH);
    const int y0 = th * TH, x0 = tw * TWD;
    const int cq0 = bcy * CBQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;
    const int row = pl >> 2, col0 = (pl & 3) * 4;
    float4 s1 = f4zero(), s2 = f4zero();
    float4 dyc[4];
    if (PART == 2) {   // only the dy' values this thread owns
        const GsrcCoef4 gk = gsrc_coef4(p.dy, cc, p.C);
        const int oy = y0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ox = x0 + col0 + u;
            const bool ok = cok && oy < p.OH && ox < p.OW;
            const size_t r = ((size_t)b * p.OH + min(oy, p.OH - 1)) * p.OW + min(ox, p.OW - 1);
            const float4 g4 = ldg4(p.dy.g + r * p.dy.ldg + cc);
            const float4 y4 = gk.two ? ldg4(p.dy.y + r * p.dy.ldy + cc) : f4zero();
            dyc[u] = ok ? gsrc_apply4(g4, y4, gk) : f4zero();
        }
    } else {   // ---------------- phase 1
        const GsrcCoef4 gk = gsrc_coef4(p.dy, cc, p.C);
        {   // every (g, y) load of the lane before the first LDS write
            constexpr int NITS = (NPX + 31) / 32;
            float4 rg[NITS], ry[NITS];
#pragma unroll
            for (int u = 0; u < NITS; ++u) {
                const int pix = min(pl + 32 * u, NPX - 1);
                const int py = pix / PW, px = pix % PW;
                const int oy = min(max(y0 - 1 + py, 0), p.OH - 1), ox = min(max(x0 - 1 + px, 0), p.OW - 1);
                const size_t r = ((size_t)b * p.OH + oy) * p.OW + ox;
                rg[u] = ldg4(p.dy.g + r * p.dy.ldg + cc);
                ry[u] = gk.two ? ldg4(p.dy.y + r * p.dy.ldy + cc) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < NITS; ++u) {
                const int pix = pl + 32 * u;
                if (pix < NPX) {
                    const int py = pix / PW, px = pix % PW;
                    const int oy = y0 - 1 + py, ox = x0 - 1 + px;
                    const bool ok = cok && oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW;
                    tile[pix * PQP + q] = ok ? gsrc_apply4(rg[u], ry[u], gk) : f4zero();
                }
            }
        }
        float4 wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (cok && c0 + j < p.C) ? p.w[(size_t)(c0 + j) * 9 + k] : 0.f;
            wk[k] = make_float4(e[0], e[1], e[2], e[3]);
        }
        const SinkCoef4 sk = sink_coef4(p.sink, cc, p.C);
        SinkOps so[4];                                 // the sink's operands of the 4 owned pixels: in flight across the barrier
        {
            const size_t srow = ((size_t)b * p.IH + min(y0 + row, p.IH - 1)) * p.IW;
#pragma unroll
            for (int u = 0; u < 4; ++u) so[u] = sink_load4(p.sink, sk, srow + min(x0 + col0 + u, p.IW - 1), cc);
        }
        __syncthreads();
        float4 dx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dx[u] = f4zero();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float4 dr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) dr[j] = lds_px(tile, (row + ky) * PW + col0 + j, q);
            if (ky == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) dyc[u] = dr[u + 1];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kx2 = u + 2 - j;          // dx[u] += dy'[row-1+ky][col-1+j] * w[2-ky][u+2-j]
                    if (kx2 >= 0 && kx2 < 3) dx[u] = f4fma(dr[j], wk[(2 - ky) * 3 + kx2], dx[u]);
                }
        }
        const int y = y0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = x0 + col0 + u;
            if (cok && y < p.IH && x < p.IW) sink_store4(p.sink, sk, ((size_t)b * p.IH + y) * p.IW + x, c0, dx[u], so[u], s1, s2);
        }
    }
    if (PART == 1) {      // input gradient only: statistics rows of the sink, then done
        if (p.sink.finalize && p.sink.mean && p.sink.partial) {
            __syncthreads();
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
                if (c < p.C) p.sink.partial[(size_t)tIdx * 2 * p.sink.ldp + (comp >> 2) * p.sink.ldp + c] = a;
            }
        }
        return;
    }
    __syncthreads();      // every wave is done reading the dy' tile
    float4 dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4zero();
    {   // ---------------- phase 2
        const SrcCoef4 xk = src_coef4(p.in, cc, p.C);
        for (int e0 = 0; e0 < NPX; e0 += 32 * 4) {
            float4 r[4];
            int pix[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pix[u] = e0 + pl + 32 * u;
                const int py = min(pix[u], NPX - 1) / PW, px = min(pix[u], NPX - 1) % PW;
                const int iy = min(max(y0 - 1 + py, 0), p.IH - 1), ix = min(max(x0 - 1 + px, 0), p.IW - 1);
                r[u] = ldg4(p.in.x + (((size_t)b * p.IH + iy) * p.IW + ix) * p.in.ld + cc);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (pix[u] < NPX) {
                    const int py = pix[u] / PW, px = pix[u] % PW;
                    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                    const bool ok = cok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    const float4 v = src_apply4(r[u], xk);
                    tile[pix[u] * PQP + q] = ok ? v : f4zero();
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float4 xr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) xr[j] = lds_px(tile, (row + ky) * PW + col0 + j, q);
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kx = j - u;               // dw[ky][kx] += dy'[row][col+u] * x'[row-1+ky][col+u-1+kx]
                    if (kx >= 0 && kx < 3) dwacc[ky * 3 + kx] = f4fma(dyc[u], xr[j], dwacc[ky * 3 + kx]);
                }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
        float* slot = red + (pl * 8 + q) * 4;
        slot[0] = dwacc[k].x; slot[1] = dwacc[k].y; slot[2] = dwacc[k].z; slot[3] = dwacc[k].w;
        __syncthreads();
        if (t < 32) {
            const int qq = t >> 2, comp = t & 3;
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) a += red[(l * 8 + qq) * 4 + comp];
            const int c = (cq0 + qq) * 4 + comp;
            if (c < p.C) p.wpartial[((size_t)tIdx * p.C + c) * 9 + k] = a;
        }
    }
    if (PART == 0 && p.sink.finalize && p.sink.mean && p.sink.partial) {     // uniform
        __syncthreads();
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
            if (c < p.C) p.sink.partial[(size_t)tIdx * 2 * p.sink.ldp + (comp >> 2) * p.sink.ldp + c] = a;
        }
    }
}

// Stride 1, ONE staging for both gradients (PART 3 of pq_dw3x3_bwd): the dy' tile (10 x 18, transformed while it is staged)
// and the RAW layer input y (10 x 18) -- which is both the weight gradient's operand (x' = act(scale * y + shift), applied at
// the fragment read; pixels outside the image are zero padding and masked there) and the gradient sink's y (activation
// derivative, BatchNorm-backward sums), so the input tensor is read once instead of three times (tile + halo in the
// weight-gradient launch, centre pixels in the sink of the input-gradient launch) and (g, y_out) once instead of twice.
// Requires in.x == dx.y (the engine's case: the sink IS the conv's input tensor).  All loads of a lane are issued before the
// first LDS write.  The 9 weight-gradient taps of the block leave in one LDS round (the tile buffers are reused).
__global__ __launch_bounds__(256, 2) void dwt_bwd_s1_fused_kernel(DwTP p) {
    constexpr int TH = 8, TWD = 16, PH = TH + 2, PW = TWD + 2, NPX = PH * PW;       // 180 staged pixels per tensor
    constexpr int NIT = (NPX + 31) / 32;                                             // 6 pixels per pixel lane
    __shared__ float4 dyt[NPX * PQP];
    __shared__ float4 xt[NPX * PQP];
    __shared__ float red[32 * 8 * 8];
    static_assert(NPX * PQP >= 5 * 32 * 8, "tap reduction reuses the tile buffers (5 + 4 taps)");
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;
    int tIdx, bcy;
    xcd_block(tIdx, bcy);
    const int tw = tIdx % p.tilesW, th = (tIdx / p.tilesW) % p.tilesH, b = tIdx / (p.tilesW * p.tilesH);
    const int y0 = th * TH, x0 = tw * TWD;
    const int cq0 = bcy * CBQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;
    const int row = pl >> 2, col0 = (pl & 3) * 4;
    const GsrcCoef4 gk = gsrc_coef4(p.dy, cc, p.C);
    const SrcCoef4 xk = src_coef4(p.in, cc, p.C);
    const SinkCoef4 sk = sink_coef4(p.sink, cc, p.C);

    // ---- every load of this lane first: (g, y_out, y_in) of its 6 staged pixels, the running gradient of its 4 owned pixels
    float4 rg[NIT], ry[NIT], rx[NIT], gacc[4];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int pix = min(pl + 32 * u, NPX - 1);
        const int py = pix / PW, px = pix % PW;
        const int yy = min(max(y0 - 1 + py, 0), p.IH - 1), xx = min(max(x0 - 1 + px, 0), p.IW - 1);     // stride 1: OH == IH, OW == IW
        const size_t r = ((size_t)b * p.IH + yy) * p.IW + xx;
        rg[u] = ldg4(p.dy.g + r * p.dy.ldg + cc);
        ry[u] = gk.two ? ldg4(p.dy.y + r * p.dy.ldy + cc) : f4zero();
        rx[u] = ldg4(p.in.x + r * p.in.ld + cc);
    }
    const int oy = y0 + row;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ox = x0 + col0 + u;
        const size_t r = ((size_t)b * p.IH + min(oy, p.IH - 1)) * p.IW + min(ox, p.IW - 1);
        gacc[u] = p.sink.accumulate ? ldg4(p.sink.g + r * p.sink.ldg + cc) : f4zero();
    }
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (cok && c0 + j < p.C) ? p.w[(size_t)(c0 + j) * 9 + k] : 0.f;
        wk[k] = make_float4(e[0], e[1], e[2], e[3]);
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int pix = pl + 32 * u;
        if (pix < NPX) {
            const int py = pix / PW, px = pix % PW;
            const int yy = y0 - 1 + py, xx = x0 - 1 + px;
            const bool ok = cok && yy >= 0 && yy < p.IH && xx >= 0 && xx < p.IW;
            dyt[pix * PQP + q] = ok ? gsrc_apply4(rg[u], ry[u], gk) : f4zero();
            xt[pix * PQP + q] = rx[u];                      // raw; masked / transformed where it is read
        }
    }
    __syncthreads();

    float4 dx[4], dyc[4], dwacc[9];
#pragma unroll
    for (int u = 0; u < 4; ++u) dx[u] = f4zero();
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4zero();
#pragma unroll
    for (int u = 0; u < 4; ++u) dyc[u] = lds_px(dyt, (row + 1) * PW + col0 + 1 + u, q);       // dy' at the owned pixels
    float4 yc[4];                                                                          // raw y at the owned pixels
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float4 dr[6], xr[6];
        const bool rok = (unsigned)(y0 + row + ky - 1) < (unsigned)p.IH;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            dr[j] = lds_px(dyt, (row + ky) * PW + col0 + j, q);
            const float4 raw = lds_px(xt, (row + ky) * PW + col0 + j, q);
            if (ky == 1 && j >= 1 && j <= 4) yc[j - 1] = raw;
            const bool ok = rok && (unsigned)(x0 + col0 + j - 1) < (unsigned)p.IW;
            xr[j] = ok ? src_apply4(raw, xk) : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kx2 = u + 2 - j;          // dx[u] += dy'[row-1+ky][col-1+j] * w[2-ky][u+2-j]
                if (kx2 >= 0 && kx2 < 3) dx[u] = f4fma(dr[j], wk[(2 - ky) * 3 + kx2], dx[u]);
                const int kx = j - u;               // dw[ky][kx] += dy'[row][col+u] * x'[row-1+ky][col+u-1+kx]
                if (kx >= 0 && kx < 3) dwacc[ky * 3 + kx] = f4fma(dyc[u], xr[j], dwacc[ky * 3 + kx]);
            }
    }
    float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ox = x0 + col0 + u;
        if (cok && oy < p.IH && ox < p.IW) {
            const size_t r = ((size_t)b * p.IH + oy) * p.IW + ox;
            stg4(p.sink.g + r * p.sink.ldg + c0, sink_apply4(p.sink, sk, f4add(dx[u], gacc[u]), yc[u], s1, s2));
        }
    }
    // ---- the block's 9 weight-gradient taps: [tap][pixel lane][quad] float4 over both tile buffers, summed over the 32 lanes
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float4* dstp = (k < 5) ? (dyt + (k * 32 + pl) * 8 + q) : (xt + ((k - 5) * 32 + pl) * 8 + q);
        *dstp = dwacc[k];
    }
    __syncthreads();
    for (int o = t; o < 9 * 32; o += 256) {            // output o -> (tap o / 32, quad (o % 32) / 4, component o % 4)
        const int k = o >> 5, qc = o & 31;
        const float* base = reinterpret_cast<const float*>(k < 5 ? dyt + k * 32 * 8 : xt + (k - 5) * 32 * 8) + qc;
        float a = 0.f;
#pragma unroll
        for (int l = 0; l < 32; ++l) a += base[l * 32];
        const int c = (cq0 + (qc >> 2)) * 4 + (qc & 3);
        if (c < p.C) p.wpartial[((size_t)tIdx * p.C + c) * 9 + k] = a;
    }
    if (p.sink.finalize && p.sink.mean && p.sink.partial) {     // uniform
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
            if (c < p.C) p.sink.partial[(size_t)tIdx * 2 * p.sink.ldp + (comp >> 2) * p.sink.ldp + c] = a;
        }
    }
}

template <int S>
__global__ __launch_bounds__(256) void dwt_bwd_kernel(DwTP p) {
    using T = DwBTile<S>;
    __shared__ float4 dyt[T::DYH * T::DYW * PQP];
    __shared__ float4 xt[T::XH * T::XW * PQP];
    __shared__ float red[32 * 8 * 8];
    const int t = threadIdx.x;
    const int q = t & 7, pl = t >> 3;
    int tIdx, bcy;
    xcd_block(tIdx, bcy);
    const int tw = tIdx % p.tilesW, th = (tIdx / p.tilesW) % p.tilesH, b = tIdx / (p.tilesW * p.tilesH);
    const int y0 = th * T::DXH, x0 = tw * T::DXW;                        // first input pixel of the dx tile
    const int oy0 = (S == 1) ? y0 - 1 : y0 / 2, ox0 = (S == 1) ? x0 - 1 : x0 / 2;     // first output pixel of the dy' tile
    const int xy0 = y0 - 1, xx0 = x0 - 1;                                  // first input pixel of the x' tile
    const int cq0 = bcy * CBQ;
    const int c0 = (cq0 + q) * 4;
    const bool cok = cq0 + q < p.C4;
    const int cc = cok ? c0 : 0;
    const SrcCoef4 xk = src_coef4(p.in, cc, p.C);
    const GsrcCoef4 gk = gsrc_coef4(p.dy, cc, p.C);
    const SinkCoef4 sk = sink_coef4(p.sink, cc, p.C);

    // ---- stage dy' (outputs) and x' (inputs)
    constexpr int NDY = T::DYH * T::DYW, NX = T::XH * T::XW;
    for (int e0 = 0; e0 < NDY; e0 += 32 * 2) {
        float4 rg[2], ry[2];
        int pix[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pix[u] = e0 + pl + 32 * u;
            const int py = min(pix[u], NDY - 1) / T::DYW, px = min(pix[u], NDY - 1) % T::DYW;
            const int oy = min(max(oy0 + py, 0), p.OH - 1), ox = min(max(ox0 + px, 0), p.OW - 1);
            const size_t r = ((size_t)b * p.OH + oy) * p.OW + ox;
            rg[u] = ldg4(p.dy.g + r * p.dy.ldg + cc);
            ry[u] = gk.two ? ldg4(p.dy.y + r * p.dy.ldy + cc) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (pix[u] < NDY) {
                const int py = pix[u] / T::DYW, px = pix[u] % T::DYW;
                const int oy = oy0 + py, ox = ox0 + px;
                const bool ok = cok && oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW;
                const float4 v = gsrc_apply4(rg[u], ry[u], gk);
                dyt[pix[u] * PQP + q] = ok ? v : f4zero();
            }
        }
    }
    for (int e0 = 0; e0 < NX; e0 += 32 * 4) {
        float4 r[4];
        int pix[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pix[u] = e0 + pl + 32 * u;
            const int py = min(pix[u], NX - 1) / T::XW, px = min(pix[u], NX - 1) % T::XW;
            const int iy = min(max(xy0 + py, 0), p.IH - 1), ix = min(max(xx0 + px, 0), p.IW - 1);
            r[u] = ldg4(p.in.x + (((size_t)b * p.IH + iy) * p.IW + ix) * p.in.ld + cc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pix[u] < NX) {
                const int py = pix[u] / T::XW, px = pix[u] % T::XW;
                const int iy = xy0 + py, ix = xx0 + px;
                const bool ok = cok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                const float4 v = src_apply4(r[u], xk);
                xt[pix[u] * PQP + q] = ok ? v : f4zero();
            }
        }
    }
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (cok && c0 + j < p.C) ? p.w[(size_t)(c0 + j) * 9 + k] : 0.f;
        wk[k] = make_float4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();

    float4 s1 = f4zero(), s2 = f4zero();
    float4 dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4zero();

    if (S == 1) {
        // lane pl -> row pl/4, 4 consecutive columns (pl%4)*4.. ; window rows r = 0..2 of both tiles
        const int row = pl >> 2, col0 = (pl & 3) * 4;
        float4 dx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dx[u] = f4zero();
        float4 dyc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dyc[u] = lds_px(dyt, (row + 1) * T::DYW + col0 + 1 + u, q);     // dy' at the owned pixels
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float4 dr[6], xr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                dr[j] = lds_px(dyt, (row + ky) * T::DYW + col0 + j, q);
                xr[j] = lds_px(xt, (row + ky) * T::XW + col0 + j, q);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kx2 = u + 2 - j;          // dx[u] += dy'[row-1+ky][col-1+j] * w[2-ky][u+2-j]
                    if (kx2 >= 0 && kx2 < 3) dx[u] = f4fma(dr[j], wk[(2 - ky) * 3 + kx2], dx[u]);
                    const int kx = j - u;               // dw[ky][kx] += dy'[row][col+u] * x'[row-1+ky][col+u-1+kx]
                    if (kx >= 0 && kx < 3) dwacc[ky * 3 + kx] = f4fma(dyc[u], xr[j], dwacc[ky * 3 + kx]);
                }
        }
        const int y = y0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = x0 + col0 + u;
            if (cok && y < p.IH && x < p.IW) sink_emit4(p.sink, sk, ((size_t)b * p.IH + y) * p.IW + x, c0, dx[u], s1, s2);
        }
    } else {
        // ---- dgrad: lane pl -> input rows 2*(pl/4) + {0,1}, input columns 4*(pl%4) .. +3 (8 pixels)
        const int ry0 = (pl >> 2) * 2, cx0 = (pl & 3) * 4;
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy) {
            const int iyl = ry0 + dyy;                       // local input row (y0 and ry0 are even)
            const int oal = (ry0 >> 1) + dyy;                // dy' tile row of tap a
            float4 dxr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) dxr[u] = f4zero();
            if (dyy == 0) {                                  // even input row: kernel row 1 only
                s2_dgrad_row<1, T::DYW>(dyt, oal, cx0, q, wk, dxr);
            } else {                                         // odd input row: kernel row 0 at oal, kernel row 2 at oal-1
                s2_dgrad_row<0, T::DYW>(dyt, oal, cx0, q, wk, dxr);
                s2_dgrad_row<2, T::DYW>(dyt, oal - 1, cx0, q, wk, dxr);
            }
            const int y = y0 + iyl;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + cx0 + u;
                if (cok && y < p.IH && x < p.IW) sink_emit4(p.sink, sk, ((size_t)b * p.IH + y) * p.IW + x, c0, dxr[u], s1, s2);
            }
        }
        // ---- wgrad: lane pl -> owned output row pl/4, output columns (pl%4)*2 .. +1
        const int orow = pl >> 2, ocol0 = (pl & 3) * 2;
        float4 dyc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) dyc[u] = lds_px(dyt, orow * T::DYW + ocol0 + u, q);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float4 xr[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) xr[j] = lds_px(xt, (orow * 2 + ky) * T::XW + ocol0 * 2 + j, q);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kx = j - 2 * u;
                    if (kx >= 0 && kx < 3) dwacc[ky * 3 + kx] = f4fma(dyc[u], xr[j], dwacc[ky * 3 + kx]);
                }
        }
    }

    // ---- block reductions: 9 weight-gradient taps, then the two BN-backward sums
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
        float* slot = red + (pl * 8 + q) * 4;
        slot[0] = dwacc[k].x; slot[1] = dwacc[k].y; slot[2] = dwacc[k].z; slot[3] = dwacc[k].w;
        __syncthreads();
        if (t < 32) {
            const int qq = t >> 2, comp = t & 3;
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) a += red[(l * 8 + qq) * 4 + comp];
            const int c = (cq0 + qq) * 4 + comp;
            if (c < p.C) p.wpartial[((size_t)tIdx * p.C + c) * 9 + k] = a;
        }
    }
    if (p.sink.finalize && p.sink.mean && p.sink.partial) {     // uniform
        __syncthreads();
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
            if (c < p.C) p.sink.partial[(size_t)tIdx * 2 * p.sink.ldp + (comp >> 2) * p.sink.ldp + c] = a;
        }
    }
}

__global__ __launch_bounds__(1024) void dwt_partial_reduce_kernel(const float* part, int nblk, int n, float* out) {
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

}  // namespace pq

using namespace pq;

static int dwt_check(int B, int IH, int IW, int OH, int OW, int C, int stride) {
    if (B <= 0 || IH <= 0 || IW <= 0 || C <= 0) return 0;      // any C: quads are ceil(C/4), masked by C
    if (stride != 1 && stride != 2) return 0;
    return OH == (IH - 1) / stride + 1 && OW == (IW - 1) / stride + 1;
}
static void fwd_tiles(int OH, int OW, int stride, int& th, int& tw) {
    th = (OH + 7) / 8;
    tw = stride == 1 ? (OW + 15) / 16 : (OW + 7) / 8;
}
static void bwd_tiles(int IH, int IW, int stride, int& th, int& tw) {
    th = stride == 1 ? (IH + 7) / 8 : (IH + 15) / 16;
    tw = (IW + 15) / 16;
}

int pq_dw_fwd_stream(const pq_dw3x3_fwd_args* a, void* stream);     // dwconv.hip
int dw_walk_blocks(int B, int IH, int IW, int C, int stride);       // dwwalk.hip
int dw_walk_launch(const pq_dw3x3_bwd_args* a, void* stream);
int dw_fwd_walk_blocks(int B, int H, int W, int C);
int dw_fwd_walk_launch(const pq_dw3x3_fwd_args* a, void* stream);

// Which forward / backward variant pays: the LDS-tiled kernels win while the planes are small (<= 64x64 at batch 16:
// few, fat workgroups per image), the streaming register-window kernels win on the big early planes (measured with
// scripts/dw_bench.py).  PQ_DW_MODE=tiled|stream overrides (benchmarking only).
static int dw_mode_override() {
    static int mode = -2;
    if (mode == -2) {
        const char* e = getenv("PQ_DW_MODE");
        mode = !e ? -1 : (e[0] == 't' ? 1 : 0);
    }
    return mode;
}
// stride-1 forward: the walking kernel (dwwalk.hip) at every plane size; PQ_DW_FWD_WALK=0 restores the tiled / streaming pair (A/B)
static bool dw_fwd_use_walk(int stride) {
    static const char* e = getenv("PQ_DW_FWD_WALK");
    return stride == 1 && !(e && e[0] == '0') && dw_mode_override() < 0;
}
static bool dw_fwd_use_tiled(int OH, int OW, int stride) {
    const int o = dw_mode_override();
    if (o >= 0) return o == 1;
    (void)stride;
    return (long)OH * OW <= 64 * 64;
}
extern "C" int32_t pq_dw3x3_bwd_fused_pays(int32_t IH, int32_t IW, int32_t stride) {
    const int o = dw_mode_override();
    if (o >= 0) return o;
    return stride == 1 && (long)IH * IW <= 64 * 64;
}

extern "C" int32_t pq_dw3x3_fwd_blocks(int32_t B, int32_t OH, int32_t OW, int32_t C, int32_t stride) {
    if (dw_fwd_use_walk(stride)) return dw_fwd_walk_blocks(B, OH, OW, C);
    if (!dw_fwd_use_tiled(OH, OW, stride)) return ew_geom((long)B * OH * OW, C).gx;
    int th, tw; fwd_tiles(OH, OW, stride, th, tw);
    return B * th * tw;
}
extern "C" int32_t pq_dw3x3_bwd_blocks(int32_t B, int32_t IH, int32_t IW, int32_t stride) {
    int th, tw; bwd_tiles(IH, IW, stride, th, tw);
    return B * th * tw;
}

extern "C" int pq_dw3x3_fwd(const pq_dw3x3_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->w && a->out, "null pointer");
    PQ_CHECK_ARG(dwt_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->stride), "bad shape (need stride 1|2, pad 1)");
    PQ_CHECK_ARG((a->in.ld & 3) == 0 && (a->ldo & 3) == 0, "pitches must be multiples of 4");
    if (act_is_special(a->in.act)) { pq_set_error("pq_dw3x3_fwd: logistic/tanh operands must be materialised first"); return PQ_EUNSUPPORTED; }
    if (dw_fwd_use_walk(a->stride)) {
        PQ_CHECK_ARG(a->stat_acc == nullptr || (a->stat_bad != nullptr && a->minmax == 0 && a->ldp >= a->C), "stat_acc needs stat_bad, no minmax");
        const int rc = dw_fwd_walk_launch(a, stream);
        if (rc != PQ_OK) return rc;
        PQ_CHECK_LAUNCH();
        return PQ_OK;
    }
    if (!dw_fwd_use_tiled(a->OH, a->OW, a->stride)) {
        if (a->stat_acc || a->in.stat.acc) { pq_set_error("pq_dw3x3_fwd: the statistics hand-over needs the LDS-tiled kernel (ask pq_bn_stat_ok first)"); return PQ_EUNSUPPORTED; }
        return pq_dw_fwd_stream(a, stream);
    }
    DwTP p{};
    p.in = a->in; p.w = a->w; p.out = a->out; p.partial = a->partial; p.ldo = a->ldo; p.ldp = a->ldp;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    p.minmax = (a->partial && a->minmax) ? 1 : 0;
    PQ_CHECK_ARG(a->stat_acc == nullptr || (a->stat_bad != nullptr && a->minmax == 0 && a->ldp >= a->C), "stat_acc needs stat_bad, no minmax");
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    fwd_tiles(a->OH, a->OW, a->stride, p.tilesH, p.tilesW);
    dim3 grid(a->B * p.tilesH * p.tilesW, (p.C4 + CBQ - 1) / CBQ), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (a->stride == 1) hipLaunchKernelGGL((dwt_fwd_kernel<1>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((dwt_fwd_kernel<2>), grid, blk, 0, st, p);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

// pq_bn_stat_ok for PQ_OP_DW_FWD: producer and consumer roles both need the LDS-tiled kernel (planes up to 64 x 64)
int dw_fwd_stat_ok(const pq_dw3x3_fwd_args* a, int role) {
    (void)role;
    return a && !a->minmax && !a->in.qstate && (dw_fwd_use_walk(a->stride) || dw_fwd_use_tiled(a->OH, a->OW, a->stride));
}

extern "C" int64_t pq_dw3x3_bwd_scratch_bytes(int32_t B, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t C, int32_t stride) {
    (void)OH; (void)OW;
    return (int64_t)pq_dw3x3_bwd_blocks(B, IH, IW, stride) * C * 9 * 4;
}

extern "C" int32_t pq_dw3x3_bwd_walk_blocks(int32_t B, int32_t IH, int32_t IW, int32_t C, int32_t stride) {
    if (B <= 0 || IH <= 0 || IW <= 0 || C <= 0 || (stride != 1 && stride != 2)) return 0;
    return dw_walk_blocks(B, IH, IW, C, stride);
}

extern "C" int pq_dw3x3_bwd(const pq_dw3x3_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->part >= 0 && a->part <= 6, "null pointer / bad part");
    if (a->dy_stat.acc) {     // BatchNorm-backward sums handed over by the sink kernel: only the walking kernel derives from them
        PQ_CHECK_ARG(a->part == 5 && a->dy.y && a->dy.mu, "dy_stat: part 5 with a two-tensor gradient operand only");
        PQ_CHECK_ARG(a->dy_stat.bad && a->dy_stat.gamma && a->dy_stat.invstd && a->dy_stat.dgamma && a->dy_stat.dbeta && a->dy_stat.p &&
                     a->dy_stat.q && a->dy_stat.r && a->dy_stat.count > 0.f && a->dy_stat.ldp >= a->C && a->dy_stat.slots <= 4, "dy_stat: incomplete");
    }
    if (a->part >= 5) {       // the walking kernel (dwwalk.hip): 5 = both gradients, rows left in scratch; 6 = their reduction
        PQ_CHECK_ARG(dwt_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->stride), "bad shape (need stride 1|2, pad 1)");
        PQ_CHECK_ARG(a->scratch && (a->part == 5 || a->dw), "part 5 / 6 need scratch (6: and dw)");
        if (a->part == 5) {
            PQ_CHECK_ARG(a->dy.g && a->w && a->dx.g && a->in.x, "null pointer");
            PQ_CHECK_ARG(a->in.x == a->dx.y && a->in.ld == a->dx.ldy, "part 5: the sink must be the conv's input tensor");
            PQ_CHECK_ARG(a->in.scale == a->dx.scale && a->in.shift == a->dx.shift && a->in.act == a->dx.act,
                         "part 5: the sink's scale / shift / activation must be the input operand's (same tensor)");
            PQ_CHECK_ARG(a->stride == 1 || ((a->IH & 1) == 0 && (a->IW & 1) == 0), "stride-2 backward needs even input sizes");
            PQ_CHECK_ARG((a->dy.ldg & 3) == 0 && (a->in.ld & 3) == 0 && (a->dx.ldg & 3) == 0 && (!a->dy.y || (a->dy.ldy & 3) == 0), "pitches must be multiples of 4");
            if (act_is_special(a->dx.act) || act_is_special(a->in.act)) { pq_set_error("pq_dw3x3_bwd: logistic/tanh are not supported"); return PQ_EUNSUPPORTED; }
        }
        const int rc = dw_walk_launch(a, stream);
        if (rc != PQ_OK) return rc;
        PQ_CHECK_LAUNCH();
        return PQ_OK;
    }
    if (a->part == 4) {       // only the reduction of the per-block weight-gradient rows a part-3 launch left in `scratch`
        PQ_CHECK_ARG(a->dw && a->scratch && a->stride == 1 && a->B > 0 && a->C > 0, "part 4 needs dw, scratch, stride 1");
        int th4, tw4; bwd_tiles(a->IH, a->IW, a->stride, th4, tw4);
        const int n4 = a->C * 9;
        hipLaunchKernelGGL(dwt_partial_reduce_kernel, dim3((n4 + 31) / 32), dim3(32, 32), 0, (hipStream_t)stream, (const float*)a->scratch,
                           a->B * th4 * tw4, n4, a->dw);
        PQ_CHECK_LAUNCH();
        return PQ_OK;
    }
    PQ_CHECK_ARG(a->dy.g, "null pointer");
    PQ_CHECK_ARG(a->part == 2 || (a->w && a->dx.g), "input gradient needs w and the sink");
    PQ_CHECK_ARG(a->part == 1 || (a->in.x && a->scratch && (a->dw || a->part == 3)), "weight gradient needs in, dw and scratch");
    PQ_CHECK_ARG(a->part == 0 || a->stride == 1, "the split backward exists for stride 1 only");
    PQ_CHECK_ARG(a->part != 3 || (a->in.x == a->dx.y && a->in.ld == a->dx.ldy), "part 3: the sink must be the conv's input tensor");
    PQ_CHECK_ARG(dwt_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->stride), "bad shape (need stride 1|2, pad 1)");
    PQ_CHECK_ARG(a->stride == 1 || ((a->IH & 1) == 0 && (a->IW & 1) == 0), "stride-2 backward needs even input sizes");
    PQ_CHECK_ARG(a->part == 2 || !(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    if ((a->part != 2 && act_is_special(a->dx.act)) || (a->part != 1 && act_is_special(a->in.act))) { pq_set_error("pq_dw3x3_bwd: logistic/tanh are not supported"); return PQ_EUNSUPPORTED; }
    DwTP p{};
    p.dy = a->dy; p.in = a->in; p.w = a->w; p.sink = a->dx; p.wpartial = a->scratch;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.C = a->C; p.C4 = (a->C + 3) / 4;
    bwd_tiles(a->IH, a->IW, a->stride, p.tilesH, p.tilesW);
    const int nblk = a->B * p.tilesH * p.tilesW;
    dim3 grid(nblk, (p.C4 + CBQ - 1) / CBQ), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (a->stride != 1) hipLaunchKernelGGL((dwt_bwd_kernel<2>), grid, blk, 0, st, p);
    else if (a->part == 1) hipLaunchKernelGGL((dwt_bwd_s1_kernel<1>), grid, blk, 0, st, p);
    else if (a->part == 2) hipLaunchKernelGGL((dwt_bwd_s1_kernel<2>), grid, blk, 0, st, p);
    else if (a->part == 3) PQ_LAUNCH_BINDABLE(dwt_bwd_s1_fused_kernel, grid, blk, 0, st, p);
    else hipLaunchKernelGGL((dwt_bwd_s1_kernel<0>), grid, blk, 0, st, p);
    PQ_CHECK_LAUNCH();
    if (a->part == 1 || a->part == 3) return PQ_OK;
    const int n = a->C * 9;
    hipLaunchKernelGGL(dwt_partial_reduce_kernel, dim3((n + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, nblk, n, a->dw);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
