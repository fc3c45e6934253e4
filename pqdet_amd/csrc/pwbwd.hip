// Fused backward of a narrow 1x1 convolution: input gradient AND weight gradient from ONE pass over dy.
//
// The pointwise layers at the high-resolution end of a MobileNet / RegNet body have a handful of channels on one side
// (16..32) and up to 192 on the other, over 10^5..10^6 rows: both backward GEMMs are HBM-bound, and both read the same
// gradient operand dy' = p*g + q*(y-mu) + r (two tensors).  Run as two kernels (pq_conv1x1_dgrad + pq_conv1x1_wgrad) the
// widest tensors of the network are read twice per step.  Here a workgroup walks 64-row tiles: it stages dy' and the
// layer input once in LDS, contracts over the output channels for the input gradient (-> pq_gsink epilogue with the
// BatchNorm-backward sums) and over the 64 rows for the weight gradient, which stays in registers across all tiles of
// the (persistent) workgroup; one slab per workgroup, summed by a small second kernel (deterministic).
//   mirrors: nn.Conv2d(k=1).backward inside the conv blocks of model/parser.py:389-410 (autograd in the reference).
#include "common.h"
#include <type_traits>

namespace pq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct PwP {
    pq_gsrc dy; pq_src x; pq_gsink dx;
    const float* w;      // [O][I]
    float* slab;         // [gridDim.x][O*I]
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;      // the sink's BatchNorm-backward sums into pq_bn_bstat accumulators
    pq_bn_bstat dys;                                            // dy's BatchNorm-backward sums (hand-over: no finalize launch in front)
    int M, I, O, LDZ, LDX, tiles;
};

constexpr int PW_BM = 64;     // rows per tile

// Diagnostic build only (-DPQ_STAMPS, scripts/pw_stamps.py): cycles a workgroup spends in the phases of a tile (first barrier, stage
// the operands, the two contractions, the sink), summed over its tiles, + the real time of its first and last instruction
#ifdef PQ_STAMPS
__device__ unsigned long long* pw_stamp_buf = nullptr;
#define PW_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define PW_T(v)
#endif

// TD: 16-wide column tiles of the input gradient per wave, ceil(I/16) <= TD (tiles past I are computed on whatever LDS holds there and
// never stored: no run-time tile count inside the MFMA loops, whose branches made the compiler drain lgkmcnt in front of every MFMA pair)
// NZ / NX: rows of the gradient pair / of the layer input ONE thread stages per tile (a thread owns one channel quad and every
// (256 / quads)-th row).  All of a tile's loads are issued as ONE batch: in the in-kernel stamps (scripts/pw_stamps.py) the staging
// phase of round 3's kernel -- batches of 2..6 rows, each a full memory round trip, then the input's -- was 50-60 % of a tile's time.
// OCC: workgroups per CU the register budget allows
template <int TD, int NZ, int NX, int OCC>
__global__ __launch_bounds__(256, OCC) void pw_bwd_kernel(PwP p) {
    constexpr bool OWIDE = TD <= 2;               // which side of the weight gradient is dealt to the waves in 16-wide tiles
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l16 = lane & 15, lq = lane >> 4;
    const int I = p.I, O = p.O, LDZ = p.LDZ, LDX = p.LDX;
    float* Ws = lds;                              // [O][LDX]     w[o][i]
    float* dzs = Ws + O * LDX + 16;               // [64][LDZ]    dy' tile
    float* xs = dzs + PW_BM * LDZ + 16;           // [64][LDX]    layer input tile; later the input-gradient tile
    const int I16 = (I + 15) >> 4, O16 = (O + 15) >> 4;

    // staging layouts: a thread owns one channel quad and walks rows
    const int O4 = O >> 2, I4 = I >> 2;
    const int zx = t % O4, zy = t / O4, PYz = 256 / O4;
    const int ix = t % I4, iy = t / I4, PYi = 256 / I4;
    const bool zact = zy < PYz, iact = iy < PYi;
    // per-channel coefficients live in LDS (re-read per row batch: keeps the register budget for 4 workgroups per CU)
    float* cg = xs + PW_BM * LDX + 16;            // [4][O]  p, q, mu, r of dy'
    float* cx = cg + 4 * O;                       // [2][I]  scale, shift of the layer input
    float* cs = cx + 2 * I;                       // [4][I]  scale, shift, mean, invstd of the sink
    const bool two = p.dy.y != nullptr, xaff = p.x.scale != nullptr, saff = p.dx.scale != nullptr;
    const bool sstats = p.dx.mean != nullptr && p.dx.partial != nullptr;
    if (p.dys.acc != nullptr) {                  // uniform: dy's coefficients from the sink kernel's accumulators (no finalize launch)
        for (int c = t; c < O; c += 256) {
            const GsrcCoef1 k1 = gsrc_coef1_bstat(p.dy, p.dys, c, blockIdx.x == 0);
            cg[c] = k1.p; cg[O + c] = k1.q; cg[2 * O + c] = k1.mu; cg[3 * O + c] = k1.r;
        }
    } else
    for (int c = t; c < O; c += 256) {
        cg[c] = two ? p.dy.p[c] : 1.f; cg[O + c] = two ? p.dy.q[c] : 0.f;
        cg[2 * O + c] = two ? p.dy.mu[c] : 0.f; cg[3 * O + c] = two ? p.dy.r[c] : 0.f;
    }
    for (int c = t; c < I; c += 256) {
        cx[c] = xaff ? p.x.scale[c] : 1.f; cx[I + c] = xaff ? p.x.shift[c] : 0.f;
        cs[c] = saff ? p.dx.scale[c] : 1.f; cs[I + c] = saff ? p.dx.shift[c] : 0.f;
        cs[2 * I + c] = sstats ? p.dx.mean[c] : 0.f; cs[3 * I + c] = sstats ? p.dx.invstd[c] : 0.f;
    }
    auto gcoef = [&]() {
        GsrcCoef4 k;
        k.two = two;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.p[j] = cg[4 * zx + j]; k.q[j] = cg[O + 4 * zx + j]; k.mu[j] = cg[2 * O + 4 * zx + j]; k.r[j] = cg[3 * O + 4 * zx + j]; }
        return k;
    };
    auto xcoef = [&]() {
        SrcCoef4 k;
        k.act = act_param(p.x.act);
        k.affine = xaff;
        k.q = p.x.qstate != nullptr;
        k.qs = k.q ? p.x.qstate[2] : 1.f;
        k.qinv = k.q ? 1.f / k.qs : 1.f;
        k.qzp = k.q ? p.x.qstate[3] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.a[j] = cx[4 * ix + j]; k.b[j] = cx[I + 4 * ix + j]; }
        return k;
    };
    auto scoef = [&]() {
        SinkCoef4 k;
        k.affine = saff;
        k.stats = sstats;
        k.act = act_param(p.dx.act);
        k.q = p.dx.qstate != nullptr;
        k.qinv = k.q ? 1.f / p.dx.qstate[2] : 0.f;
        k.qzp = k.q ? p.dx.qstate[3] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { k.a[j] = cs[4 * ix + j]; k.b[j] = cs[I + 4 * ix + j]; k.mean[j] = cs[2 * I + 4 * ix + j]; k.inv[j] = cs[3 * I + 4 * ix + j]; }
        return k;
    };
    const bool need_y = p.dx.finalize && (p.dx.act != PQ_ACT_LINEAR || sstats || p.dx.qstate != nullptr);

    for (int idx = t; idx < O * I; idx += 256) {
        const int o = idx / I, i = idx - o * I;
        Ws[o * LDX + i] = p.w[idx];
    }

    // The sink's y is normally the layer input itself (dx.y == in.x in every engine plan): the RAW input values a thread stages
    // stay in registers across the MFMA phase and serve the sink, instead of a second (L2-hit, but exposed) load per tile.
    constexpr bool KEEP = TD <= 6;                             // the 12-tile flavour (I up to 192) has no registers to spare
    float4 xraw[KEEP ? NX : 1];
    const bool same_y = KEEP && need_y && p.dx.y == p.x.x && p.dx.ldy == p.x.ld;

    f32x4 aw[3][2];      // [tile of the wide side: w, w+4, w+8][tile of the narrow side: 0, 1]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) aw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 s1 = f4zero(), s2 = f4zero();

#ifdef PQ_STAMPS
    unsigned long long tL = 0, tC = 0, tS = 0, tW = 0, nT = 0;
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime(), c_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int m0 = tile * PW_BM;
        PW_T(s0);
        __syncthreads();                          // previous tile's epilogue has read xs; Ws is written (first pass)
        PW_T(s1_);
        {
            // every load of the tile first (rows past the tile are not requested, rows past the matrix re-read its last row) ...
            // (a one-tensor operand reads g twice: `two ? load(y) : g` made the compiler WAIT for every g before it branched
            // around the y load -- 2 x NZ serialised round trips per tile, in round 3's kernel as well)
            float4 g[NZ], y[NZ], xv[NX];
            const float* yp = two ? p.dy.y : p.dy.g;
            const size_t yld = two ? (size_t)p.dy.ldy : (size_t)p.dy.ldg;
#pragma unroll
            for (int u = 0; u < NZ; ++u) {
                g[u] = f4zero(); y[u] = f4zero();
                if (zact && zy + u * PYz < PW_BM) {
                    const size_t m = (size_t)min(m0 + zy + u * PYz, p.M - 1);
                    g[u] = ldg4(p.dy.g + m * p.dy.ldg + 4 * zx);
                    y[u] = ldg4(yp + m * yld + 4 * zx);
                }
            }
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                xv[u] = f4zero();
                if (iact && iy + u * PYi < PW_BM) xv[u] = ldg4(p.x.x + (size_t)min(m0 + iy + u * PYi, p.M - 1) * p.x.ld + 4 * ix);
                if (KEEP) xraw[u] = xv[u];
            }
            // ... then transform and store
            if (zact) {
                const GsrcCoef4 gk = gcoef();
#pragma unroll
                for (int u = 0; u < NZ; ++u) {
                    const int r = zy + u * PYz;
                    if (r < PW_BM) {
                        float4 v = gsrc_apply4(g[u], y[u], gk);
                        if (m0 + r >= p.M) v = f4zero();
                        *reinterpret_cast<float4*>(&dzs[r * LDZ + 4 * zx]) = v;
                    }
                }
            }
            if (iact) {
                const SrcCoef4 xk = xcoef();
#pragma unroll
                for (int u = 0; u < NX; ++u) {
                    const int r = iy + u * PYi;
                    if (r < PW_BM) {
                        float4 v = src_apply4(xv[u], xk);
                        if (m0 + r >= p.M) v = f4zero();
                        *reinterpret_cast<float4*>(&xs[r * LDX + 4 * ix]) = v;
                    }
                }
            }
        }
        __syncthreads();
        PW_T(s2_);

        // input gradient: wave w owns rows 16w..16w+15 and all column tiles; contraction over the O output channels,
        // two k-steps per iteration (operands fetched from LDS before the MFMAs; KS interleaved accumulator sets keep
        // the narrow flavour's MFMA chain from depending on itself)
        constexpr int KS = TD <= 2 ? 2 : 1;
        f32x4 ad[KS][TD];
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
            for (int c = 0; c < TD; ++c) ad[s_][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* arow = dzs + (16 * w + l16) * LDZ + lq;
            const float* brow = Ws + lq * LDX + l16;
            int k0 = 0;
            const int kend = O & ~7;
            constexpr bool PRE = TD <= 2;          // operands of the next step leave LDS before this step's MFMAs (registers: narrow flavours only)
            float a0 = 0.f, a1 = 0.f, b0[TD], b1[TD];
            auto fetch = [&](int k, float& fa0, float& fa1, float (&fb0)[TD], float (&fb1)[TD]) {
                fa0 = arow[k]; fa1 = arow[k + 4];
#pragma unroll
                for (int c = 0; c < TD; ++c) { fb0[c] = brow[k * LDX + 16 * c]; fb1[c] = brow[(k + 4) * LDX + 16 * c]; }
            };
            if (PRE && kend > 0) fetch(0, a0, a1, b0, b1);
            for (; k0 < kend; k0 += 8) {
                if (PRE) {
                    float na0 = 0.f, na1 = 0.f, nb0[TD], nb1[TD];
                    const int kn = k0 + 8 < kend ? k0 + 8 : k0;        // (last step: re-reads its own operands, unused)
                    fetch(kn, na0, na1, nb0, nb1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < TD; ++c) { ad[0][c] = mfma16(a0, b0[c], ad[0][c]); ad[KS - 1][c] = mfma16(a1, b1[c], ad[KS - 1][c]); }
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = na0; a1 = na1;
#pragma unroll
                    for (int c = 0; c < TD; ++c) { b0[c] = nb0[c]; b1[c] = nb1[c]; }
                } else {
                    fetch(k0, a0, a1, b0, b1);
#pragma unroll
                    for (int c = 0; c < TD; ++c) { ad[0][c] = mfma16(a0, b0[c], ad[0][c]); ad[KS - 1][c] = mfma16(a1, b1[c], ad[KS - 1][c]); }
                }
            }
            for (; k0 < O; k0 += 4) {
                const float ta = arow[k0];
#pragma unroll
                for (int c = 0; c < TD; ++c) ad[0][c] = mfma16(ta, brow[k0 * LDX + 16 * c], ad[0][c]);
            }
            if (KS == 2) {
#pragma unroll
                for (int c = 0; c < TD; ++c) ad[0][c] += ad[KS - 1][c];
            }
        }
        // weight gradient: wave w owns the 16-wide tiles w, w+4, w+8 of the wide side x both tiles of the narrow side;
        // contraction over the 64 rows; all (up to 6) accumulator chains advance together
        {
            const float* ap = dzs + lq * LDZ + l16 + (OWIDE ? 16 * w : 0);
            const float* bp = xs + lq * LDX + l16 + (OWIDE ? 0 : 16 * w);
            const int wide16 = OWIDE ? O16 : I16, narrow16 = OWIDE ? I16 : O16;
            const int na = wide16 > w ? (wide16 - w + 3) >> 2 : 0;
            auto run = [&](auto NA_, auto NN_) {
                constexpr int NA = decltype(NA_)::value, NN = decltype(NN_)::value;
#pragma unroll 4
                for (int k0 = 0; k0 < PW_BM; k0 += 4) {
                    float wv[NA], nv[NN];
#pragma unroll
                    for (int a = 0; a < NA; ++a) wv[a] = OWIDE ? ap[k0 * LDZ + 64 * a] : bp[k0 * LDX + 64 * a];
#pragma unroll
                    for (int b2 = 0; b2 < NN; ++b2) nv[b2] = OWIDE ? bp[k0 * LDX + 16 * b2] : ap[k0 * LDZ + 16 * b2];
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int b2 = 0; b2 < NN; ++b2)
                            aw[a][b2] = OWIDE ? mfma16(wv[a], nv[b2], aw[a][b2]) : mfma16(nv[b2], wv[a], aw[a][b2]);
                }
            };
            using i1 = std::integral_constant<int, 1>; using i2 = std::integral_constant<int, 2>; using i3 = std::integral_constant<int, 3>;
            if (narrow16 > 1) { if (na == 3) run(i3{}, i2{}); else if (na == 2) run(i2{}, i2{}); else if (na == 1) run(i1{}, i2{}); }
            else { if (na == 3) run(i3{}, i1{}); else if (na == 2) run(i2{}, i1{}); else if (na == 1) run(i1{}, i1{}); }
        }
        __syncthreads();                          // all waves are done with xs (and dzs)
        PW_T(s3_);
#pragma unroll
        for (int c = 0; c < TD; ++c) {
            const int col = 16 * c + l16;
            if (col < I) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xs[(16 * w + 4 * lq + j) * LDX + col] = ad[0][c][j];
            }
        }
        __syncthreads();
        if (iact) {
            const SinkCoef4 sk = scoef();
            constexpr int SC = TD >= 12 ? 7 : NX;     // the sink's own loads (running gradient; y where it is not kept): one batch as well (widest flavour: two)
#pragma unroll
            for (int u0 = 0; u0 < NX; u0 += SC) {
                // three straight-line passes: every load (rows past the tile / the matrix: a clamped, valid row), every value, then the
                // stores.  With the loads and stores inside per-row `if`s the compiler could not count what was outstanding and put
                // s_waitcnt vmcnt(0) in front of every row: each row then waited for the previous row's STORE (stamps: 14-20 k cycles).
                float4 go[SC], yv[SC];
#pragma unroll
                for (int v = 0; v < SC; ++v) {
                    const int u = min(u0 + v, NX - 1), r = min(iy + u * PYi, PW_BM - 1);
                    const size_t m = (size_t)min(m0 + r, p.M - 1);
                    go[v] = f4zero(); yv[v] = f4zero();
                    if (p.dx.accumulate) go[v] = ldg4(sink_old(p.dx) + m * p.dx.ldg + 4 * ix);
                    if (same_y) yv[v] = xraw[KEEP ? u : 0];
                    else if (need_y) yv[v] = ldg4(p.dx.y + m * p.dx.ldy + 4 * ix);
                }
#pragma unroll
                for (int v = 0; v < SC; ++v) {
                    const int u = u0 + v, r = iy + u * PYi;
                    const bool ok = u < NX && r < PW_BM && m0 + r < p.M;
                    const float4 dz = f4add(*reinterpret_cast<const float4*>(&xs[min(r, PW_BM - 1) * LDX + 4 * ix]), go[v]);
                    float4 a1 = s1, a2 = s2;
                    go[v] = sink_apply4(p.dx, sk, dz, yv[v], a1, a2);
                    s1.x = ok ? a1.x : s1.x; s1.y = ok ? a1.y : s1.y; s1.z = ok ? a1.z : s1.z; s1.w = ok ? a1.w : s1.w;
                    s2.x = ok ? a2.x : s2.x; s2.y = ok ? a2.y : s2.y; s2.z = ok ? a2.z : s2.z; s2.w = ok ? a2.w : s2.w;
                }
#pragma unroll
                for (int v = 0; v < SC; ++v) {
                    const int u = u0 + v, r = iy + u * PYi, m = m0 + r;
                    if (u < NX && r < PW_BM && m < p.M) stg4(p.dx.g + (size_t)m * p.dx.ldg + 4 * ix, go[v]);
                }
            }
        }
#ifdef PQ_STAMPS
        { PW_T(s4_); tW += s1_ - s0; tL += s2_ - s1_; tC += s3_ - s2_; tS += s4_ - s3_; ++nT; }
#endif
    }
#ifdef PQ_STAMPS
    if (t == 0 && pw_stamp_buf) {
        unsigned long long* o = pw_stamp_buf + (size_t)blockIdx.x * 8;
        o[0] = tW; o[1] = tL; o[2] = tC; o[3] = tS; o[4] = r_begin; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = nT;
        o[7] = __builtin_amdgcn_s_memtime() - c_begin;
    }
#endif

    // weight-gradient slab of this workgroup
    float* dst = p.slab + (size_t)blockIdx.x * O * I;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int to = OWIDE ? w + 4 * a : b, ti = OWIDE ? b : w + 4 * a;
            const int i = 16 * ti + l16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 16 * to + 4 * lq + j;
                if (o < O && i < I) dst[(size_t)o * I + i] = aw[a][b][j];
            }
        }
    // BatchNorm-backward sums of the sink: one partial row per workgroup
    if (p.dx.finalize && sstats) {
        __syncthreads();
        float* red = lds;
        float* slot = red + t * 8;
        slot[0] = s1.x; slot[1] = s1.y; slot[2] = s1.z; slot[3] = s1.w;
        slot[4] = s2.x; slot[5] = s2.y; slot[6] = s2.z; slot[7] = s2.w;
        __syncthreads();
        if (t < I4) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int yy = 0; yy < PYi; ++yy) {
                const float* o = red + (yy * I4 + t) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += o[j];
            }
            if (p.stat_acc != nullptr) {          // no finalize launch: the consumer of the tensor's gradient derives the coefficients
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, p.dx.ldp, blockIdx.x, 0, 4 * t + j, acc[j]);
                    bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, p.dx.ldp, blockIdx.x, 1, 4 * t + j, acc[4 + j]);
                }
            } else {
                float* pr = p.dx.partial + (size_t)blockIdx.x * 2 * p.dx.ldp;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pr[4 * t + j] = acc[j]; pr[p.dx.ldp + 4 * t + j] = acc[4 + j]; }
            }
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

struct PwPlan { bool ok; int nz, nx, occ; int ldz, ldx, tiles, grid; size_t lds_bytes; };
static PwPlan pw_plan(int M, int I, int O) {
    PwPlan pl{};
    const int lo = I < O ? I : O, hi = I < O ? O : I;
    // wide side <= 12 tiles of 16 (3 per wave), narrow side <= 2
    pl.ok = M > 0 && (I & 3) == 0 && (O & 3) == 0 && lo >= 4 && lo <= 32 && hi <= 192;
    if (!pl.ok) return pl;
    pl.ldz = pw_pitch(O);
    pl.ldx = pw_pitch(I);
    size_t fl = (size_t)O * pl.ldx + 16 + (size_t)PW_BM * pl.ldz + 16 + (size_t)PW_BM * pl.ldx + 16 + 4 * (size_t)O + 6 * (size_t)I;
    if (fl < 2048 + 16) fl = 2048 + 16;
    pl.lds_bytes = fl * 4;
    pl.tiles = (M + PW_BM - 1) / PW_BM;
    int occ = (int)((160 * 1024) / pl.lds_bytes);
    // rows one thread stages per tile: the 256 threads cover 256 / quads rows per pass
    pl.nz = (PW_BM + 256 / (O / 4) - 1) / (256 / (O / 4));
    pl.nx = (PW_BM + 256 / (I / 4) - 1) / (256 / (I / 4));
    // register budget: three workgroups per CU while a thread holds at most 2 rows of the gradient pair (O <= 32), else two
    const int reg_occ = ((I + 15) / 16 <= 2 && pl.nz <= 2) ? 3 : 2;
    if (occ > reg_occ) occ = reg_occ;
    if (occ < 1) occ = 1;
    pl.occ = occ;
    int grid = 256 * occ;
    const int half = pl.tiles / 2 > 256 ? pl.tiles / 2 : 256;     // at least two tiles per workgroup once the chip is covered
    if (grid > half) grid = half;
    if (grid > pl.tiles) grid = pl.tiles;
    if (grid < 1) grid = 1;
    pl.grid = grid;
    return pl;
}

// Policy: on the whole train step the fused kernel pays from 2^18 rows (the input-gradient chain is the critical path and
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

#ifdef PQ_STAMPS
extern "C" int pq_debug_set_stamps_pw(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pq::pw_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
template <int TD, int NZ, int NX, int OCC>
static int pw_launch(const PwP& p, const PwPlan& pl, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)pw_bwd_kernel<TD, NZ, NX, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            pq_set_error("pq_conv1x1_bwd: cannot raise the dynamic LDS limit");
            return PQ_ELAUNCH;
        }
        attr_set = true;
    }
    PQ_LAUNCH((pw_bwd_kernel<TD, NZ, NX, OCC>), dim3(pl.grid), dim3(256), pl.lds_bytes, st, p);
    return PQ_OK;
}

extern "C" int pq_conv1x1_bwd(const pq_conv1x1_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->dy.g && a->w && a->in.x && a->dx.g && a->scratch, "null pointer");      // dw == NULL: deferred reduction
    PQ_SINK_ADDEND_CHECK(a->dx);
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
    PwP p{};
    p.dy = a->dy; p.x = a->in; p.dx = a->dx; p.w = a->w; p.slab = a->scratch;
    p.M = a->M; p.I = a->I; p.O = a->O; p.LDZ = pl.ldz; p.LDX = pl.ldx; p.tiles = pl.tiles;
    p.dys = a->dy_stat;
    if (p.dys.slots < 1) p.dys.slots = 1;
    if (a->dy_stat.acc) {
        PQ_CHECK_ARG(a->dy.y && a->dy.mu && a->dy_stat.bad && a->dy_stat.gamma && a->dy_stat.invstd && a->dy_stat.dgamma && a->dy_stat.dbeta &&
                     a->dy_stat.p && a->dy_stat.q && a->dy_stat.r && a->dy_stat.count > 0.f && a->dy_stat.ldp >= a->O && a->dy_stat.slots <= 4,
                     "dy_stat: incomplete");
    }
    if (a->dx.stat_acc) {
        PQ_CHECK_ARG(a->dx.stat_bad && a->dx.finalize && a->dx.mean && a->dx.partial && a->dx.ldp >= a->I && a->dx.stat_slots <= 4,
                     "dx.stat_acc needs stat_bad and a finalising sink with statistics");
        p.stat_acc = a->dx.stat_acc; p.stat_bad = a->dx.stat_bad; p.stat_slots = a->dx.stat_slots > 0 ? a->dx.stat_slots : 1;
        if (!stat_adds_ok(pl.grid, p.stat_slots)) {
            pq_set_error("pq_conv1x1_bwd: more than %d workgroups per accumulator slot", PQ_STAT_MAX_ADDS);
            return PQ_EUNSUPPORTED;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int i16 = (a->I + 15) / 16;
    int rc;
    // flavours: <input-gradient column tiles, rows of the gradient pair per thread, rows of the input per thread, workgroups per CU>
    if (i16 <= 1) {              // I <= 16: one row of the input per thread; O up to 192
        if (pl.nz <= 2) rc = pw_launch<1, 2, 1, 3>(p, pl, st);
        else if (pl.nz <= 7) rc = pw_launch<1, 7, 1, 2>(p, pl, st);
        else if (pl.nz <= 10) rc = pw_launch<1, 10, 1, 2>(p, pl, st);
        else rc = pw_launch<1, 13, 1, 2>(p, pl, st);
    } else if (i16 <= 2) {       // I <= 32
        if (pl.nz <= 2) rc = pw_launch<2, 2, 2, 3>(p, pl, st);
        else if (pl.nz <= 7) rc = pw_launch<2, 7, 2, 2>(p, pl, st);
        else if (pl.nz <= 10) rc = pw_launch<2, 10, 2, 2>(p, pl, st);
        else rc = pw_launch<2, 13, 2, 2>(p, pl, st);
    } else if (i16 <= 6) rc = pw_launch<6, 2, 7, 2>(p, pl, st);        // I <= 96, O <= 32
    else if (i16 <= 9) rc = pw_launch<9, 2, 10, 2>(p, pl, st);         // I <= 144
    else rc = pw_launch<12, 2, 13, 2>(p, pl, st);                      // I <= 192
    if (rc != PQ_OK) return rc;
    PQ_CHECK_LAUNCH();
    if (!a->dw) return PQ_OK;          // the workgroups' slabs stay in `scratch` (pq_conv1x1_bwd_reduce_item / pq_reduce_batch)
    const int n = a->O * a->I;
    PQ_LAUNCH(pw_slab_reduce_kernel, dim3((n + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, n, pl.grid, a->dw);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_conv1x1_bwd_reduce_item(const pq_conv1x1_bwd_args* a, pq_reduce_item* out) {
    PQ_CHECK_ARG(a && out && a->dw && a->scratch, "null pointer");
    const PwPlan pl = pw_plan(a->M, a->I, a->O);
    if (!pl.ok) { pq_set_error("pq_conv1x1_bwd_reduce_item: shape outside the fused kernel's range"); return PQ_EUNSUPPORTED; }
    const int n = a->O * a->I;
    *out = pq_reduce_item{a->scratch, a->dw, nullptr, (int64_t)n, 0, pl.grid, n, 0, n, 0, 0, 0, 0};
    return PQ_OK;
}

// pq_bn_bstat_ok for PQ_OP_CONV1X1_BWD (producer role)
int conv1x1_bwd_bstat_ok(const pq_conv1x1_bwd_args* a) {
    return a->dx.finalize && a->dx.mean != nullptr && a->dx.partial != nullptr && !a->dx.qstate && (a->I & 3) == 0;
}
