// NOT BUILT, NOT SHIPPED: round-2 experiment kept for the record.  LDS-DMA, software-pipelined variant of
// pqdet_amd/csrc/pwbwd.hip (drop-in for that file: same entry points, passes tests/test_kernels_gpu.py -k conv1x1_bwd).
// With all arithmetic switched off its two-stage ring streams 6.2 TB/s; with it on, one 8-wave workgroup per CU is bound by
// the serial phases of a tile (44 % of wave cycles parked in s_waitcnt / barrier, MFMA pipe 19 % busy) and by LDS traffic
// (230 KB per 64-row tile against 128 KB for register staging): 282 / 181 / 174 / 149 / 142 / 111 us on the six narrow layers
// of mobilenetv2-fpn 512x512 bs16 against 277 / 142 / 151 / 154 / 148 / 97 us for the shipped kernel.
// Fused backward of a narrow 1x1 convolution: input gradient AND weight gradient from ONE pass over dy.
//
// The pointwise layers at the high-resolution end of a MobileNet / RegNet body have a handful of channels on one side
// (16..32) and up to 192 on the other, over 10^5..10^6 rows: both backward GEMMs are HBM-bound, and both read the same
// gradient operand dy' = p*g + q*(y-mu) + r (two tensors).  Run as two kernels (pq_conv1x1_dgrad + pq_conv1x1_wgrad) the
// widest tensors of the network are read twice per step.  Here a persistent workgroup walks R-row tiles: the raw tiles of
// g, y and the layer input travel by LDS-DMA, dy' and x' are formed once in LDS, the input gradient is contracted over the
// output channels (-> pq_gsink epilogue with the BatchNorm-backward sums) and the weight gradient over the R rows; it stays
// in registers across all tiles of the workgroup; one slab per workgroup, summed by a small second kernel (deterministic).
//   mirrors: nn.Conv2d(k=1).backward inside the conv blocks of model/parser.py:389-410 (autograd in the reference).
#include "common.h"
#include "ldsdma.h"
#include <type_traits>

namespace pq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct PwP {
    pq_gsrc dy; pq_src x; pq_gsink dx;
    const float* w;      // [O][I]
    float* slab;         // [gridDim.x][O*I]
    int M, I, O, LDZ, LDX, tiles;
    int sg, sx, stage, sep_y;   // floats per padded gradient image, per padded input image, per stage; sink y staged separately
};

// The tile loop is software-pipelined over a two-stage ring so that HBM streams continuously behind the MFMAs (the first
// version of this kernel ran load -> barrier -> MFMA -> barrier -> sink strictly one after the other: 150 us = 88 us of loads
// + 50 of MFMA + 20 of sink for M=262144, I=24, O=144).  Iteration t of a workgroup:
//     wait vmcnt(0), barrier        tile t has landed (its DMA was issued one whole iteration ago)
//     issue the DMA of tile t+1     into the other stage (last read before the mid-iteration barrier of t-1)
//     sink of tile t-1              dx tile (LDS) + raw y_in / running gradient (registers) -> global stores, BN-backward sums
//     transform tile t              raw (g, y) -> dy' and raw x -> x' into padded LDS tiles; the sink operands of tile t (raw
//                                   y_in, and the running gradient when the sink accumulates) -> registers
//     barrier
//     MFMA                          input gradient -> LDS dx tile;  weight gradient -> persistent accumulators
// Two barriers per tile; the stores of tile t-1 have a whole iteration to drain before the next vmcnt(0).  The sink's y is
// normally the layer input itself (dx.y == in.x in every engine plan) and is then taken from the staged tile instead of being
// read again; a different tensor is staged as one more image.
// R rows per tile (128 / 64 / 32 / 16): the host takes the largest whose images are <= 2048 chunks of 16 B, whose sink rows
// per thread are <= 8 and whose two stages fit LDS.
constexpr int PWD_T = 512;     // threads: 8 waves, two per SIMD (one wave per SIMD leaves every LDS round trip exposed)
constexpr int PWD_NP = 4;      // 1 KiB DMA pieces per wave per image, at most

template <int R, int NT>
__global__ __launch_bounds__(PWD_T) void pw_bwd_dma_kernel(PwP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l16 = lane & 15, lq = lane >> 4;
    const int I = p.I, O = p.O, LDZ = p.LDZ, LDX = p.LDX, M = p.M;
    const int I16 = (I + 15) >> 4, O16 = (O + 15) >> 4, O4 = O >> 2, I4 = I >> 2;
    float* Wt = lds;                              // [16*I16][LDZ]  w transposed: Wt[i][o] (contraction index contiguous)
    float* dzs = Wt + 16 * I16 * LDZ + 16;        // [R][LDZ]   dy'
    float* xs = dzs + R * LDZ + 16;               // [R][LDX]   x'
    float* dxs = xs + R * LDX + 16;               // [R][LDX]   input gradient of the tile
    float* cg = dxs + R * LDX + 16;               // [4][O]     p, q, mu, r
    float* cx = cg + 4 * O;                       // [2][I]     scale, shift of the layer input
    float* cs = cx + 2 * I;                       // [4][I]     scale, shift, mean, invstd of the sink
    float* ring = cs + 4 * I;                     // [2][stage]: g | y | x | go, each padded to whole DMA rounds
    const bool two = p.dy.y != nullptr, xaff = p.x.scale != nullptr, saff = p.dx.scale != nullptr;
    const bool sstats = p.dx.mean != nullptr && p.dx.partial != nullptr, acc = p.dx.accumulate != 0;
    const bool sepy = p.sep_y != 0;
    const int off_y = p.sg, off_x = (two ? 2 : 1) * p.sg, off_go = off_x + p.sx, off_sy = off_go + (acc ? p.sx : 0);

    // ---- DMA bookkeeping: piece q of this wave covers chunks ((8q + w)*64 + lane) of an image; (row, column) of that chunk
    const int ncg = R * O4, ncx = R * I4;         // chunks per image (images are padded to whole pieces of 64 chunks)
    // Per-lane byte offsets of this wave's pieces inside a FULL tile (row * pitch + column): the DMA address of a tile is then a
    // wave-uniform base (SGPRs) + that 32-bit offset, i.e. no per-lane address arithmetic in the loop (the kernel is VALU-issue
    // bound: every wave64 VALU instruction holds its SIMD for 4 cycles).  The ragged last tile recomputes clamped addresses.
    // (fast path: y has the pitch of g, the running gradient / separate sink y have the pitch of the input -- else the slow one)
    unsigned vg[PWD_NP], vx[PWD_NP];
#pragma unroll
    for (int q = 0; q < PWD_NP; ++q) {
        const int c = (8 * q + w) * 64 + lane;
        const int cgk = min(c, ncg - 1), cxk = min(c, ncx - 1);
        const int rg = cgk / O4, rx = cxk / I4;
        vg[q] = (unsigned)(rg * p.dy.ldg + 4 * (cgk - rg * O4)) * 4u;
        vx[q] = (unsigned)(rx * p.x.ld + 4 * (cxk - rx * I4)) * 4u;
    }
    const bool fastp = (!two || p.dy.ldy == p.dy.ldg) && (!acc || p.dx.ldg == p.x.ld) && (!sepy || p.dx.ldy == p.x.ld);
    const unsigned ring_b = lds_addr(ring);
    auto issue = [&](int tile, int s) {
        const int m0 = tile * R;
        const unsigned sb = ring_b + (unsigned)(s * p.stage) * 4u;
        if (m0 + R <= M && fastp) {
            const float* bg = p.dy.g + (size_t)m0 * p.dy.ldg;
            const float* by = two ? p.dy.y + (size_t)m0 * p.dy.ldy : nullptr;
            const float* bx = p.x.x + (size_t)m0 * p.x.ld;
            const float* bgo = p.dx.g + (size_t)m0 * p.dx.ldg;
            const float* bsy = sepy ? p.dx.y + (size_t)m0 * p.dx.ldy : nullptr;
#pragma unroll
            for (int q = 0; q < PWD_NP; ++q)
                if ((8 * q + w) * 64 < ncg) {
                    const unsigned d = (unsigned)((8 * q + w) * 256) * 4u;
                    glds16s(bg, vg[q], __builtin_amdgcn_readfirstlane(sb + d));
                    if (two) glds16s(by, vg[q], __builtin_amdgcn_readfirstlane(sb + (unsigned)off_y * 4u + d));
                }
#pragma unroll
            for (int q = 0; q < PWD_NP; ++q)
                if ((8 * q + w) * 64 < ncx) {
                    const unsigned d = (unsigned)((8 * q + w) * 256) * 4u;
                    glds16s(bx, vx[q], __builtin_amdgcn_readfirstlane(sb + (unsigned)off_x * 4u + d));
                    if (acc) glds16s(bgo, vx[q], __builtin_amdgcn_readfirstlane(sb + (unsigned)off_go * 4u + d));
                    if (sepy) glds16s(bsy, vx[q], __builtin_amdgcn_readfirstlane(sb + (unsigned)off_sy * 4u + d));
                }
            return;
        }
#pragma unroll
        for (int q = 0; q < PWD_NP; ++q)
            if ((8 * q + w) * 64 < ncg) {
                const int ck = min((8 * q + w) * 64 + lane, ncg - 1), row = ck / O4;
                const size_t m = (size_t)min(m0 + row, M - 1);
                const int c = 4 * (ck - row * O4);
                const unsigned d = (unsigned)((8 * q + w) * 256) * 4u;
                glds16(p.dy.g + m * p.dy.ldg + c, __builtin_amdgcn_readfirstlane(sb + d));
                if (two) glds16(p.dy.y + m * p.dy.ldy + c, __builtin_amdgcn_readfirstlane(sb + (unsigned)off_y * 4u + d));
            }
#pragma unroll
        for (int q = 0; q < PWD_NP; ++q)
            if ((8 * q + w) * 64 < ncx) {
                const int ck = min((8 * q + w) * 64 + lane, ncx - 1), row = ck / I4;
                const size_t m = (size_t)min(m0 + row, M - 1);
                const int c = 4 * (ck - row * I4);
                const unsigned d = (unsigned)((8 * q + w) * 256) * 4u;
                glds16(p.x.x + m * p.x.ld + c, __builtin_amdgcn_readfirstlane(sb + (unsigned)off_x * 4u + d));
                if (acc) glds16(p.dx.g + m * p.dx.ldg + c, __builtin_amdgcn_readfirstlane(sb + (unsigned)off_go * 4u + d));
                if (sepy) glds16(p.dx.y + m * p.dx.ldy + c, __builtin_amdgcn_readfirstlane(sb + (unsigned)off_sy * 4u + d));
            }
    };
    int tile = blockIdx.x;
    if (tile < p.tiles) issue(tile, 0);           // in flight behind the prologue

    // ---- per-channel coefficients and the weights
    for (int c = t; c < O; c += PWD_T) {
        cg[c] = two ? p.dy.p[c] : 1.f; cg[O + c] = two ? p.dy.q[c] : 0.f;
        cg[2 * O + c] = two ? p.dy.mu[c] : 0.f; cg[3 * O + c] = two ? p.dy.r[c] : 0.f;
    }
    for (int c = t; c < I; c += PWD_T) {
        cx[c] = xaff ? p.x.scale[c] : 1.f; cx[I + c] = xaff ? p.x.shift[c] : 0.f;
        cs[c] = saff ? p.dx.scale[c] : 1.f; cs[I + c] = saff ? p.dx.shift[c] : 0.f;
        cs[2 * I + c] = sstats ? p.dx.mean[c] : 0.f; cs[3 * I + c] = sstats ? p.dx.invstd[c] : 0.f;
    }
    for (int idx = t; idx < O * I; idx += PWD_T) {
        const int o = idx / I, i = idx - o * I;
        Wt[i * LDZ + o] = p.w[idx];
    }
    __syncthreads();
    const int zx = t % O4, zy = t / O4, PYz = PWD_T / O4;
    const int ix = t % I4, iy = t / I4, PYi = PWD_T / I4;
    const bool zact = zy < PYz, iact = iy < PYi;
    // QAT observer scalars: read once (a global load inside the tile loop would stall on the DMA in flight)
    const bool xq = p.x.qstate != nullptr, sq = p.dx.qstate != nullptr;
    const float xqs = xq ? p.x.qstate[2] : 1.f, xqzp = xq ? p.x.qstate[3] : 0.f;
    const float sqinv = sq ? 1.f / p.dx.qstate[2] : 0.f, sqzp = sq ? p.dx.qstate[3] : 0.f;
    // per-channel coefficients are re-read from LDS in the phase that uses them (registers are the scarce resource here)
    auto gcoef = [&]() {
        GsrcCoef4 k;
        k.two = two;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.p[j] = cg[4 * zx + j]; k.q[j] = cg[O + 4 * zx + j]; k.mu[j] = cg[2 * O + 4 * zx + j]; k.r[j] = cg[3 * O + 4 * zx + j]; }
        return k;
    };
    auto xcoef = [&]() {
        SrcCoef4 k;
        k.act = act_param(p.x.act); k.affine = xaff; k.q = xq;
        k.qs = xqs; k.qinv = 1.f / xqs; k.qzp = xqzp;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.a[j] = cx[4 * ix + j]; k.b[j] = cx[I + 4 * ix + j]; }
        return k;
    };
    auto scoef = [&]() {
        SinkCoef4 k;
        k.affine = saff; k.stats = sstats; k.act = act_param(p.dx.act); k.q = sq;
        k.qinv = sqinv; k.qzp = sqzp;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.a[j] = cs[4 * ix + j]; k.b[j] = cs[I + 4 * ix + j]; k.mean[j] = cs[2 * I + 4 * ix + j]; k.inv[j] = cs[3 * I + 4 * ix + j]; }
        return k;
    };
    constexpr int NSK = 4;                        // rows of the sink a thread owns per tile, at most (host: R <= 4 * (512 / I4))
    float4 yin[NSK], go[NSK];
#pragma unroll
    for (int u = 0; u < NSK; ++u) { yin[u] = f4zero(); go[u] = f4zero(); }
    float4 s1 = f4zero(), s2 = f4zero();

    // weight-gradient tiles (16 x 16) of this wave: e = w + 8j, the NARROW side (<= 2 tiles) as the fast index, so that a wave
    // always works on the same narrow tile and fetches that operand once per k-step
    const int NW = O16 * I16;
    const bool ashare = O16 <= I16;               // dy' (A) is the narrow operand
    const int Nn = ashare ? O16 : I16;
    f32x4 aw[NT];
    int wo[NT];                                   // column offset of the wide operand per tile
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        wo[j] = 16 * (min(w + 8 * j, NW - 1) / Nn) + l16;
    }
    const int no = 16 * (w % Nn) + l16;           // column offset of the narrow operand
    const int nt = w < NW ? (NW - w + 7) >> 3 : 0;
    const int ND = (R / 16) * I16;                // input-gradient tiles of 16 x 16

    auto sink_tile = [&](int m0p) {
        if (iact) {
            const SinkCoef4 sk = scoef();
#pragma unroll
            for (int u = 0; u < NSK; ++u) {
                const int r = iy + u * PYi, m = m0p + r;
                if (r < R && m < M) {
                    const float4 dz = f4add(*reinterpret_cast<const float4*>(&dxs[r * LDX + 4 * ix]), go[u]);
                    stg4(p.dx.g + (size_t)m * p.dx.ldg + 4 * ix, sink_apply4(p.dx, sk, dz, yin[u], s1, s2));
                }
            }
        }
    };

    int s = 0, m0p = -1;
    for (; tile < p.tiles; tile += gridDim.x, s ^= 1) {
        const int m0 = tile * R;
        wait_vmcnt<0>();
        __syncthreads();                          // tile `tile` is in stage s; the previous dx tile is complete
        if (tile + (int)gridDim.x < p.tiles) issue(tile + gridDim.x, s ^ 1);
        if (m0p >= 0) sink_tile(m0p);
        const float* sgp = ring + s * p.stage;
        // LDS -> registers first, LDS writes after: the compiler cannot move a ds_read across a ds_write of the same array
        if (zact) {
            const GsrcCoef4 gk = gcoef();
            for (int r0 = zy; r0 < R; r0 += 2 * PYz) {
                float4 g4[2], y4[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = min(r0 + u * PYz, R - 1);
                    g4[u] = *reinterpret_cast<const float4*>(&sgp[r * O + 4 * zx]);
                    y4[u] = two ? *reinterpret_cast<const float4*>(&sgp[off_y + r * O + 4 * zx]) : g4[u];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = r0 + u * PYz;
                    if (r < R) {
                        float4 v = gsrc_apply4(g4[u], y4[u], gk);
                        if (m0 + r >= M) v = f4zero();
                        *reinterpret_cast<float4*>(&dzs[r * LDZ + 4 * zx]) = v;
                    }
                }
            }
        }
        if (iact) {
            const SrcCoef4 xk = xcoef();
            float4 xr[NSK];
#pragma unroll
            for (int u = 0; u < NSK; ++u) {
                const int r = iy + u * PYi;
                if (r < R) {
                    xr[u] = *reinterpret_cast<const float4*>(&sgp[off_x + r * I + 4 * ix]);
                    yin[u] = xr[u];
                    if (sepy) yin[u] = *reinterpret_cast<const float4*>(&sgp[off_sy + r * I + 4 * ix]);
                    if (acc) go[u] = *reinterpret_cast<const float4*>(&sgp[off_go + r * I + 4 * ix]);
                }
            }
#pragma unroll
            for (int u = 0; u < NSK; ++u) {
                const int r = iy + u * PYi;
                if (r < R) {
                    float4 v = src_apply4(xr[u], xk);
                    if (m0 + r >= M) v = f4zero();
                    *reinterpret_cast<float4*>(&xs[r * LDX + 4 * ix]) = v;
                }
            }
        }
        __syncthreads();                          // dy', x' are complete; stage s and the previous dx tile are consumed

        // input gradient: 16 x 16 tiles d = rt * I16 + ct dealt to the waves; contraction over O in groups of 16: lane
        // (l16, lq) fetches k = 16g + 4lq .. +3 of both operands with one ds_read_b128 each and feeds component u to the u-th
        // MFMA of the group (the k order inside a group is free as long as A and B agree); the next group's operands are
        // fetched before this group's MFMAs issue
        for (int d = w; d < ND; d += 8) {
            const int rt = d / I16, ct = d - rt * I16;
            const float* arow = dzs + (16 * rt + l16) * LDZ;
            const float* brow = Wt + (16 * ct + l16) * LDZ;
            f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
            const int G = O >> 4;
            const float* ap = arow + 4 * lq;
            const float* bp = brow + 4 * lq;
            float4 a0 = f4zero(), b0 = f4zero(), a1 = f4zero(), b1 = f4zero();      // ping-pong operand sets (no copies)
            if (G > 0) { a0 = *reinterpret_cast<const float4*>(ap); b0 = *reinterpret_cast<const float4*>(bp); }
#pragma unroll 1
            for (int g = 0; g < G; g += 2) {
                if (g + 1 < G) { a1 = *reinterpret_cast<const float4*>(ap + 16 * (g + 1)); b1 = *reinterpret_cast<const float4*>(bp + 16 * (g + 1)); }
                __builtin_amdgcn_sched_barrier(0);
                c0 = mfma16(a0.x, b0.x, c0);
                c1 = mfma16(a0.y, b0.y, c1);
                c0 = mfma16(a0.z, b0.z, c0);
                c1 = mfma16(a0.w, b0.w, c1);
                if (g + 1 < G) {
                    if (g + 2 < G) { a0 = *reinterpret_cast<const float4*>(ap + 16 * (g + 2)); b0 = *reinterpret_cast<const float4*>(bp + 16 * (g + 2)); }
                    __builtin_amdgcn_sched_barrier(0);
                    c0 = mfma16(a1.x, b1.x, c0);
                    c1 = mfma16(a1.y, b1.y, c1);
                    c0 = mfma16(a1.z, b1.z, c0);
                    c1 = mfma16(a1.w, b1.w, c1);
                }
            }
            for (int k0 = 16 * G; k0 < O; k0 += 4) c0 = mfma16(arow[k0 + lq], brow[k0 + lq], c0);
            c0 += c1;
            const int col = 16 * ct + l16;
            if (col < I) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dxs[(16 * rt + 4 * lq + j) * LDX + col] = c0[j];
            }
        }
        // weight gradient: dW[o][i] += sum_r dy'[r][o] * x'[r][i]; NT independent accumulator chains per wave, no branches (a
        // wave with fewer tiles repeats its last one into an accumulator that is never stored); operands of k-step k+1 are
        // fetched before the MFMAs of k-step k issue
        auto wgrad = [&](auto ash_) {
            constexpr bool ASH = decltype(ash_)::value;
            // k order: step s of a 16-row block contracts rows {s, s+4, s+8, s+12} (lane group lq -> row 4*lq + s): with the odd
            // quad pitch the four groups then fall 16 banks apart (2-way = the wave64 minimum) instead of 4 banks (4-way)
            const float* np = (ASH ? dzs + 4 * lq * LDZ : xs + 4 * lq * LDX) + no;
            const float* wp = ASH ? xs + 4 * lq * LDX : dzs + 4 * lq * LDZ;
            const int nstep = ASH ? LDZ : LDX, wstep = ASH ? LDX : LDZ;
            float n0 = np[0], n1, w0[NT], w1[NT];             // ping-pong operand sets (no copies); R % 8 == 0
#pragma unroll
            for (int j = 0; j < NT; ++j) w0[j] = wp[wo[j]];
#pragma unroll 1
            for (int k0 = 0; k0 < R; k0 += 8) {
                n1 = np[nstep];
#pragma unroll
                for (int j = 0; j < NT; ++j) w1[j] = wp[wstep + wo[j]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j) aw[j] = ASH ? mfma16(n0, w0[j], aw[j]) : mfma16(w0[j], n0, aw[j]);
                np += ((k0 & 8) ? 14 : 2) * nstep; wp += ((k0 & 8) ? 14 : 2) * wstep;      // steps 0,1 | 2,3 of a block, then the next block
                if (k0 + 8 < R) {
                    n0 = np[0];
#pragma unroll
                    for (int j = 0; j < NT; ++j) w0[j] = wp[wo[j]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j) aw[j] = ASH ? mfma16(n1, w1[j], aw[j]) : mfma16(w1[j], n1, aw[j]);
            }
        };
        if (ashare) wgrad(std::true_type{}); else wgrad(std::false_type{});
        m0p = m0;
    }
    __syncthreads();
    if (m0p >= 0) sink_tile(m0p);

    // weight-gradient slab of this workgroup
    float* dst = p.slab + (size_t)blockIdx.x * O * I;
#pragma unroll
    for (int j = 0; j < NT; ++j)
        if (j < nt) {
            const int e = w + 8 * j;
            const int to = ashare ? e % Nn : e / Nn, ti = ashare ? e / Nn : e % Nn;
            const int i = 16 * ti + l16;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int o = 16 * to + 4 * lq + jj;
                if (o < O && i < I) dst[(size_t)o * I + i] = aw[j][jj];
            }
        }
    if (p.dx.finalize && sstats) {                // BatchNorm-backward sums of the sink: one partial row per workgroup
        __syncthreads();
        float* red = lds;
        float* slot = red + t * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < I4) {
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int yy = 0; yy < PYi; ++yy) {
                const float* o = red + (yy * I4 + t) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] += o[j];
            }
            float* pr = p.dx.partial + (size_t)blockIdx.x * 2 * p.dx.ldp;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pr[4 * t + j] = a8[j]; pr[p.dx.ldp + 4 * t + j] = a8[4 + j]; }
        }
    }
}

// out[j] = sum_s slab[s*n + j]; block = 32 columns x 32 slab lanes, 8 loads in flight per thread
__global__ __launch_bounds__(1024) void pw_slab_reduce_kernel(const float* slab, int n, int slabs, float* out) {
    __shared__ float red[32][33];
    const int j = blockIdx.x * 32 + threadIdx.x;
    float s = 0.f;
    if (j < n) {
        int k = threadIdx.y;
        for (; k + 7 * 32 < slabs; k += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + 32 * u) * n + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < slabs; k += 32) s += slab[(size_t)k * n + j];
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

// pitch of an LDS tile row of x floats: >= x + 4, a multiple of 4 with an odd quad count (16 rows x 4 k-lanes of an MFMA
// operand read then fall on distinct banks)
static int pw_pitch(int x) {
    int v = x + 4;
    if (((v >> 2) & 1) == 0) v += 4;
    return v;
}

// LDS floats of the kernel at R rows per tile with n_g gradient images (g [, y]) and n_x input-shaped images (x [, running
// gradient] [, separate sink y]) per stage; 0 when R breaks a per-thread bound of the kernel
struct PwGeom { int sg, sx, stage; size_t floats; };
static PwGeom pw_geom(int R, int I, int O, int n_g, int n_x) {
    PwGeom g{};
    const int O4 = O / 4, I4 = I / 4;
    if (R * O4 > PWD_T * PWD_NP || R * I4 > PWD_T * PWD_NP || R > 4 * (PWD_T / I4)) return g;
    const int ldz = pw_pitch(O), ldx = pw_pitch(I);
    g.sg = (R * O + 255) / 256 * 256;            // whole 1 KiB DMA pieces
    g.sx = (R * I + 255) / 256 * 256;
    g.stage = n_g * g.sg + n_x * g.sx;
    g.floats = (size_t)16 * ((I + 15) / 16) * ldz + 16 + (size_t)R * ldz + 16 + 2 * ((size_t)R * ldx + 16) + 4 * (size_t)O + 6 * (size_t)I + 2 * (size_t)g.stage;
    if (g.floats < 8 * PWD_T + 16) g.floats = 8 * PWD_T + 16;      // the statistics epilogue's scratch
    return g;
}
constexpr size_t PW_LDS_MAX = 160 * 1024;
// rows per tile: the largest of 128 / 64 / 32 / 16 whose two stages fit LDS (0: none)
static int pw_rows(int I, int O, int n_g, int n_x) {
    for (int R = 128; R >= 16; R >>= 1) {
        const PwGeom g = pw_geom(R, I, O, n_g, n_x);
        if (g.floats && g.floats * 4 <= PW_LDS_MAX) return R;
    }
    return 0;
}

struct PwPlan { bool ok; int grid; };
static PwPlan pw_plan(int M, int I, int O) {
    PwPlan pl{};
    const int lo = I < O ? I : O, hi = I < O ? O : I;
    // wide side <= 12 tiles of 16, narrow side <= 2 (24 weight-gradient tiles over 8 waves)
    pl.ok = M > 0 && (I & 3) == 0 && (O & 3) == 0 && lo >= 4 && lo <= 32 && hi <= 192 && pw_rows(I, O, 2, 3) > 0;
    if (!pl.ok) return pl;
    // persistent workgroups: one per CU, two where the common case (g, y | x, running gradient) leaves room for both
    const int R = pw_rows(I, O, 2, 2);
    const int occ = (int)(PW_LDS_MAX / (pw_geom(R, I, O, 2, 2).floats * 4)) >= 2 ? 2 : 1;
    static const char* genv = getenv("PQ_PW_GRID");
    int grid = genv ? atoi(genv) : 256 * occ;
    const int tiles = (M + 127) / 128;            // every workgroup gets a tile at any R
    if (grid > tiles) grid = tiles;
    if (grid < 1) grid = 1;
    pl.grid = grid;
    return pl;
}

// Policy: on the whole train step the fused kernel pays from 2^17 rows (the input-gradient chain is the critical path and
// the weight-gradient stream has slack, so merging the two only wins where dy is read from HBM twice).
// PQ_PW_BWD=0 disables it, PQ_PW_MIN_M=<rows> moves the threshold (A/B timing, tests).
extern "C" int32_t pq_conv1x1_bwd_pays(int32_t M, int32_t I, int32_t O) {
    const char* off = getenv("PQ_PW_BWD");
    const char* minm = getenv("PQ_PW_MIN_M");
    const int min_rows = minm ? atoi(minm) : 131072;
    if (off && off[0] == '0') return 0;
    return pw_plan(M, I, O).ok && M >= min_rows;
}
extern "C" int32_t pq_conv1x1_bwd_blocks(int32_t M, int32_t I, int32_t O) {
    const PwPlan pl = pw_plan(M, I, O);
    return pl.ok ? pl.grid : 0;
}
extern "C" int64_t pq_conv1x1_bwd_scratch_bytes(int32_t M, int32_t I, int32_t O) {
    const PwPlan pl = pw_plan(M, I, O);
    return pl.ok ? (int64_t)pl.grid * O * I * 4 : 0;
}

template <int R, int NT>
static int pw_launch(const PwP& p, int grid, size_t lds_bytes, hipStream_t st) {
    static size_t raised = 0;
    if (lds_bytes > raised) {
        if (hipFuncSetAttribute((const void*)pw_bwd_dma_kernel<R, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_MAX) != hipSuccess) {
            pq_set_error("pq_conv1x1_bwd: cannot raise the dynamic LDS limit");
            return PQ_ELAUNCH;
        }
        raised = PW_LDS_MAX;
    }
    hipLaunchKernelGGL((pw_bwd_dma_kernel<R, NT>), dim3(grid), dim3(PWD_T), lds_bytes, st, p);
    return PQ_OK;
}

extern "C" int pq_conv1x1_bwd(const pq_conv1x1_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->w && a->in.x && a->dx.g && a->dw && a->scratch, "null pointer");
    PQ_CHECK_ARG(a->M > 0 && a->I > 0 && a->O > 0, "empty shape");
    PQ_CHECK_ARG((a->dy.ldg & 3) == 0 && (a->dy.y == nullptr || (a->dy.ldy & 3) == 0) && (a->in.ld & 3) == 0 &&
                 (a->dx.ldg & 3) == 0 && (a->dx.ldy & 3) == 0, "pitches must be multiples of 4");
    PQ_CHECK_ARG(a->dy.ldg >= a->O && a->in.ld >= a->I && a->dx.ldg >= a->I, "pitch smaller than channel count");
    PQ_CHECK_ARG(!(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    PQ_CHECK_ARG(!(a->dx.finalize && a->dx.mean && a->dx.partial) || a->dx.ldp >= a->I, "stats pitch too small");
    if (act_is_special(a->dx.act) || act_is_special(a->in.act)) {
        pq_set_error("pq_conv1x1_bwd: logistic/tanh operands are not supported");
        return PQ_EUNSUPPORTED;
    }
    const PwPlan pl = pw_plan(a->M, a->I, a->O);
    if (!pl.ok) {
        pq_set_error("pq_conv1x1_bwd: shape M=%d I=%d O=%d is outside the fused kernel's range (pq_conv1x1_bwd_blocks() == 0): "
                     "use pq_conv1x1_dgrad + pq_conv1x1_wgrad", a->M, a->I, a->O);
        return PQ_EUNSUPPORTED;
    }
    const uintptr_t al = (uintptr_t)a->dy.g | (uintptr_t)a->dy.y | (uintptr_t)a->in.x | (uintptr_t)a->dx.g | (uintptr_t)a->dx.y;
    PQ_CHECK_ARG((al & 15) == 0, "operands must be 16-byte aligned");
    const bool need_y = a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || (a->dx.mean && a->dx.partial) || a->dx.qstate);
    const bool sep_y = need_y && (a->dx.y != a->in.x || a->dx.ldy != a->in.ld);
    const int n_g = a->dy.y ? 2 : 1, n_x = 1 + (a->dx.accumulate ? 1 : 0) + (sep_y ? 1 : 0);
    const int R = pw_rows(a->I, a->O, n_g, n_x);
    const PwGeom g = pw_geom(R, a->I, a->O, n_g, n_x);
    PwP p{};
    p.dy = a->dy; p.x = a->in; p.dx = a->dx; p.w = a->w; p.slab = a->scratch;
    p.M = a->M; p.I = a->I; p.O = a->O; p.LDZ = pw_pitch(a->O); p.LDX = pw_pitch(a->I); p.tiles = (a->M + R - 1) / R;
    p.sg = g.sg; p.sx = g.sx; p.stage = g.stage; p.sep_y = sep_y ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // weight-gradient tiles per wave: ceil(16x16 tiles / 8 waves)
    const int nt = (((a->O + 15) / 16) * ((a->I + 15) / 16) + 7) / 8;
    int rc;
#define PW_CASE(RR)                                                                                  \
    case RR:                                                                                         \
        rc = nt <= 1 ? pw_launch<RR, 1>(p, pl.grid, g.floats * 4, st)                                \
           : nt <= 2 ? pw_launch<RR, 2>(p, pl.grid, g.floats * 4, st)                                \
                     : pw_launch<RR, 3>(p, pl.grid, g.floats * 4, st);                               \
        break;
    switch (R) {
        PW_CASE(128)
        PW_CASE(64)
        PW_CASE(32)
        default:
        PW_CASE(16)
    }
#undef PW_CASE
    if (rc != PQ_OK) return rc;
    PQ_CHECK_LAUNCH();
    const int n = a->O * a->I;
    hipLaunchKernelGGL(pw_slab_reduce_kernel, dim3((n + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, n, pl.grid, a->dw);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
