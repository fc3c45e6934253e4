// 1x1 convolution as exact-fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32) for gfx950.
//
//   gemm_nt : C[m][n] = sum_k A'[m][k] * Bop[n][k]      forward (A' = act(bn(x)), Bop = W[O][I])
//                                                        dgrad   (A' = dy',        Bop = W read as [k=o][n=i])
//   gemm_tn : C[o][i] = sum_m dY'[m][o] * X'[m][i]       wgrad, split over m into slabs
//
// Tiles are staged global -> registers (operand transform applied here, once per element) -> LDS, with the next
// k-tile's global loads in flight during the MFMA phase.  LDS rows are padded to 36 floats so the ds_read_b128
// fragment reads of a 32x32x2 operand (lane = row, 4 consecutive k per lane half) are bank-conflict free.
// The k index inside an 8-wide group is permuted identically for A and B (lane half h owns k = 4h..4h+3), which
// leaves the contraction unchanged.
#include "common.h"
#include "gemm_epi.h"

namespace pq {

constexpr int BK = 32;
constexpr int LDT = BK + 4;   // padded LDS row (floats)

struct GemmP {
    // A operand
    const float* a0; const float* a1;          // x | g , y
    const float* c0; const float* c1; const float* c2; const float* c3;   // scale,shift | p,q,mu,r
    const float* qstate;                       // AMODE 1, QAT: fake-quantise the transformed operand
    int lda0, lda1, act;
    int sub, OH, OW, IH, IW;                    // strided 1x1 conv: forward gathers A rows, the sink epilogue scatters rows
    // B operand
    const float* b; int ldb;
    int M, N, K;
    // epilogue 0/1
    float* out; const float* bias; float* partial; int ldo, ldp;
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;     // epilogue 1: statistics into pq_bn_stat accumulators
    // epilogue 2
    pq_gsink sink;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// AMODE: 0 plain, 1 forward affine+act, 2 gradient pair.  BKN: B stored [k][n].  EMODE: 0 store(+bias), 1 store+stats, 2 sink,
// 3 store + stats + per-channel min / max of the outputs (4-plane statistic rows: QAT observers)
// DB: double-buffered LDS + loads two k-tiles ahead (compute-bound shapes); !DB: one buffer (full-N streaming tiles).
template <int BM, int BN, int WM, int WN, int AMODE, bool BKN, int EMODE, bool DB, int PF = 1>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmP p) {
    constexpr int NWN = BN / WN;
    constexpr int TI = WM / 32, TJ = WN / 32;
    constexpr int AI = BM / 32;            // float4 A loads per thread per k-tile
    constexpr int BI = BN / 32;            // float4 B loads per thread per k-tile
    constexpr int NB = DB ? 2 : 1;
    constexpr int LDBK = BN + 4;           // row pitch of the [k][n] B image (BKN)
    static_assert((BM / WM) * NWN == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float As2[NB * BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs2[NB * BN * LDT];
    __shared__ float red[(EMODE == 3 ? 4 : 2) * 4 * 192];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w / NWN, wn = w % NWN;
    int bnx, bmy;              // XCD-aware numbering: one L2 serves all column tiles of a row panel
    xcd_block(bnx, bmy);
    const int m0 = bmy * BM, n0 = bnx * BN;
    const int M = p.M, N = p.N, K = p.K;
    const ActParam aact = act_param(p.act);
    const bool aq = AMODE == 1 && p.qstate != nullptr;          // uniform
    const float aqs = aq ? p.qstate[2] : 1.f, aqinv = aq ? 1.f / aqs : 1.f, aqzp = aq ? p.qstate[3] : 0.f;

    // ---- A load plan
    const int ac4 = t & 7, ar0 = t >> 3;
    const float* arow0[AI];
    const float* arow1[AI];
    bool avalid[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int m = m0 + ar0 + 32 * i;
        avalid[i] = m < M;
        size_t src = (size_t)(avalid[i] ? m : 0);
        if (EMODE != 2 && p.sub > 1) {
            int ox = (int)(src % p.OW); size_t tt = src / p.OW; int oy = (int)(tt % p.OH); size_t bb = tt / p.OH;
            src = (bb * p.IH + (size_t)oy * p.sub) * p.IW + (size_t)ox * p.sub;
        }
        arow0[i] = p.a0 + src * (size_t)p.lda0;
        arow1[i] = (AMODE == 2) ? p.a1 + src * (size_t)p.lda1 : nullptr;
    }
    // ---- B load plan
    constexpr int BNQ = BN / 4;                 // float4 per k-row (BKN): element e = t + 256*i -> (row e / BNQ, quad e % BNQ)
    const int bc4 = t & 7;                      // NK layout: quad within the 32-wide k-tile
    const int br0 = t >> 3;                     // NK layout: row (n) within a 32-row pass
    const bool bvec = BKN ? ((N & 3) == 0 && (p.ldb & 3) == 0) : ((p.ldb & 3) == 0);

    // Register staging sets (PF of them): the global loads of k-tile t+PF are in flight while tile t is multiplied.
    struct RegSet { float4 a[AI], a1[AI], b[BI], c0, c1, c2, c3; };
    const int Kpad = (K + 3) & ~3;

    // All loads are unconditional at clamped (always valid) addresses -- masking happens when the tile is written to
    // LDS -- so the compiler can issue a whole tile's loads back to back.
    auto load_tiles = [&](int k0, RegSet& r) {
        const int ka = k0 + 4 * ac4;
        if (ka < Kpad) {       // one branch around the whole batch (columns past K contribute nothing and are skipped)
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                r.a[i] = ldg4(arow0[i] + ka);
                if (AMODE == 2) r.a1[i] = ldg4(arow1[i] + ka);
            }
            if (AMODE == 1) {
                if (p.c0) { r.c0 = ldg4(p.c0 + ka); r.c1 = ldg4(p.c1 + ka); }
                else { r.c0 = make_float4(1.f, 1.f, 1.f, 1.f); r.c1 = f4zero(); }
            } else if (AMODE == 2) {
                r.c0 = ldg4(p.c0 + ka); r.c1 = ldg4(p.c1 + ka); r.c2 = ldg4(p.c2 + ka); r.c3 = ldg4(p.c3 + ka);
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (!BKN) {
                const int n = min(n0 + br0 + 32 * i, N - 1), kb = k0 + 4 * bc4;
                if (bvec) {     // rows >= N / k >= K read real (finite) weights: they only meet zeroed A or unstored C
                    r.b[i] = ldg4(p.b + (size_t)n * p.ldb + min(kb, Kpad - 4));
                } else {
                    const float* src = p.b + (size_t)n * p.ldb;
                    r.b[i] = make_float4(kb < K ? src[kb] : 0.f, kb + 1 < K ? src[kb + 1] : 0.f, kb + 2 < K ? src[kb + 2] : 0.f,
                                         kb + 3 < K ? src[kb + 3] : 0.f);
                }
            } else {
                const int e = t + 256 * i;
                const int k = min(k0 + e / BNQ, K - 1), n = n0 + 4 * (e % BNQ);
                if (bvec) {
                    r.b[i] = ldg4(p.b + (size_t)k * p.ldb + min(n, N - 4));
                } else {
                    const float* src = p.b + (size_t)k * p.ldb;
                    r.b[i] = make_float4(n < N ? src[n] : 0.f, n + 1 < N ? src[n + 1] : 0.f, n + 2 < N ? src[n + 2] : 0.f,
                                         n + 3 < N ? src[n + 3] : 0.f);
                }
            }
        }
    };

    auto store_tiles = [&](int k0, const RegSet& r, int buf) {
        float* As = As2 + buf * (BM * LDT);
        float* Bs = Bs2 + buf * (BN * LDT);
        const int ka = k0 + 4 * ac4;
        const float c0[4] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w}, c1[4] = {r.c1.x, r.c1.y, r.c1.z, r.c1.w};
        const float c2[4] = {r.c2.x, r.c2.y, r.c2.z, r.c2.w}, c3[4] = {r.c3.x, r.c3.y, r.c3.z, r.c3.w};
        bool kok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kok[j] = ka + j < K;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            float v[4] = {r.a[i].x, r.a[i].y, r.a[i].z, r.a[i].w};
            if (AMODE == 1) {
                // branch-free: (c0, c1) are (1, 0) when the operand has no affine; piecewise-linear activation family
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u = fmaf(c0[j], v[j], c1[j]);
                    v[j] = fminf(fmaf(aact.slope, fminf(u, 0.f), fmaxf(u, 0.f)), aact.hi);
                }
                if (aq) {        // QAT only: one uniform branch per quad, nothing on the float path
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fq_act(v[j], aqs, aqinv, aqzp);
                }
            } else if (AMODE == 2) {
                const float y[4] = {r.a1[i].x, r.a1[i].y, r.a1[i].z, r.a1[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf(c0[j], v[j], fmaf(c1[j], y[j] - c2[j], c3[j]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (avalid[i] && kok[j]) ? v[j] : 0.f;
            *reinterpret_cast<float4*>(&As[(ar0 + 32 * i) * LDT + 4 * ac4]) = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (!BKN) {
                // a clamped k (tile beyond K) holds duplicates of real weights; the matching A columns are zero
                *reinterpret_cast<float4*>(&Bs[(br0 + 32 * i) * LDT + 4 * bc4]) = r.b[i];
            } else {
                // B arrives as [k][n] rows: keep that layout in LDS (float4 row writes are conflict free) and let the
                // MFMA phase read its fragment with four ds_read_b32 (lanes walk n, so those are conflict free too)
                const int e = t + 256 * i;
                const int kr = e / BNQ, bq = e % BNQ;
                const bool kok = k0 + kr < K;
                const int n = n0 + 4 * bq;
                const float ev[4] = {r.b[i].x, r.b[i].y, r.b[i].z, r.b[i].w};
                float4 o;
                o.x = (kok && n + 0 < N) ? ev[0] : 0.f;
                o.y = (kok && n + 1 < N) ? ev[1] : 0.f;
                o.z = (kok && n + 2 < N) ? ev[2] : 0.f;
                o.w = (kok && n + 3 < N) ? ev[3] : 0.f;
                *reinterpret_cast<float4*>(&Bs[kr * LDBK + 4 * bq]) = o;
            }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & 31, lh = lane >> 5;
    // half = -1: the whole k-tile; 0 / 1: its first / second half (the LDS writes of the next tile are issued in between,
    // so that they execute in the shadow of the first half's MFMAs instead of after the last one)
    auto compute = [&](int buf, int half = -1) {
        const float* As = As2 + buf * (BM * LDT);
        const float* Bs = Bs2 + buf * (BN * LDT);
        const int ks0 = half == 1 ? BK / 16 : 0, ks1 = half == 0 ? BK / 16 : BK / 8;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            if (ks < ks0 || ks >= ks1) continue;
            float4 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                af[i] = *reinterpret_cast<const float4*>(&As[(wm * WM + 32 * i + lrow) * LDT + ks * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                if (!BKN) {
                    bf[j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + 32 * j + lrow) * LDT + ks * 8 + lh * 4]);
                } else {
                    const float* bp = &Bs[(ks * 8 + lh * 4) * LDBK + wn * WN + 32 * j + lrow];
                    bf[j] = make_float4(bp[0], bp[LDBK], bp[2 * LDBK], bp[3 * LDBK]);
                }
            }
            // k-major order: consecutive MFMAs write different accumulators, so none waits for the previous one's result
            {
                float ak[TI][4], bk[TJ][4];
#pragma unroll
                for (int i = 0; i < TI; ++i) { ak[i][0] = af[i].x; ak[i][1] = af[i].y; ak[i][2] = af[i].z; ak[i][3] = af[i].w; }
#pragma unroll
                for (int j = 0; j < TJ; ++j) { bk[j][0] = bf[j].x; bk[j][1] = bf[j].y; bk[j][2] = bf[j].z; bk[j][3] = bf[j].w; }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(ak[i][q], bk[j][q], acc[i][j]);
            }
        }
    };

    // AMODE 2 with y == NULL never reaches this kernel (host dispatches AMODE 0 instead).
    const int nk = (K + BK - 1) / BK;
    RegSet r0, r1;
    r0.c0 = r0.c1 = r0.c2 = r0.c3 = f4zero();
    r1.c0 = r1.c1 = r1.c2 = r1.c3 = f4zero();
    if (DB && PF == 2) {
        // One barrier per k-tile: tile t+1 goes into the other LDS buffer right after this wave's MFMAs of tile t (other
        // waves may still be multiplying); its global loads were issued two MFMA phases earlier (register sets alternate).
        load_tiles(0, r0);
        if (nk > 1) load_tiles(BK, r1);
        store_tiles(0, r0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 < nk) load_tiles((kt + 2) * BK, r0);
            compute(0, 0);
            if (kt + 1 < nk) store_tiles((kt + 1) * BK, r1, 1);
            compute(0, 1);
            __syncthreads();
            if (kt + 1 < nk) {
                if (kt + 3 < nk) load_tiles((kt + 3) * BK, r1);
                compute(1, 0);
                if (kt + 2 < nk) store_tiles((kt + 2) * BK, r0, 0);
                compute(1, 1);
                __syncthreads();
            }
        }
    } else if (DB) {
        // same, loads one k-tile ahead (big tiles: the MFMA phase alone covers the memory latency, registers are scarce)
        load_tiles(0, r0);
        store_tiles(0, r0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) load_tiles((kt + 1) * BK, r0);
            // pinned phase order (the scheduler otherwise sinks the loads behind half of the MFMAs and the LDS writes behind all
            // of them, which drains the MFMA pipe before every barrier): loads -> MFMA half -> LDS writes -> MFMA half -> barrier
            __builtin_amdgcn_sched_barrier(0);
            compute(kt & 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) store_tiles((kt + 1) * BK, r0, (kt + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(kt & 1, 1);
            __syncthreads();
        }
    } else {
        load_tiles(0, r0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
            store_tiles(kt * BK, r0, 0);
            __syncthreads();
            if (kt + 1 < nk) load_tiles((kt + 1) * BK, r0);
            compute(0);
        }
    }

    // ---------------------------------------------------------------- epilogue (gemm_epi.h: the accumulators leave in their native
    // layout -- lane = column, two 128-byte row segments per access -- without the LDS patches and per-element flag tests of round 3)
    float s1[TJ], s2[TJ];
    float mn[TJ], mx[TJ];                        // EMODE 3 only
#pragma unroll
    for (int j = 0; j < TJ; ++j) { s1[j] = s2[j] = 0.f; mn[j] = __builtin_huge_valf(); mx[j] = -__builtin_huge_valf(); }
    bool want_stats = false;
    EpiGeom eg;
    eg.m0w = m0 + __builtin_amdgcn_readfirstlane(wm) * WM; eg.n0w = n0 + __builtin_amdgcn_readfirstlane(wn) * WN;
    eg.lrow = lrow; eg.lh = lh; eg.M = M; eg.N = N;
    eg.full = m0 + BM <= M;

    if (EMODE == 2) {
        const pq_gsink& sk = p.sink;
        const bool affine = sk.scale != nullptr;
        const bool stats = sk.mean != nullptr && sk.partial != nullptr;
        const ActParam sact = act_param(sk.act);
        const bool qmask = sk.finalize && sk.qstate != nullptr;        // QAT: straight-through mask of the activation quantiser
        const float qinv = qmask ? 1.f / sk.qstate[2] : 0.f, qzp = qmask ? sk.qstate[3] : 0.f;
        const bool need_y = sk.finalize && (sk.act != PQ_ACT_LINEAR || stats || qmask);
        want_stats = sk.finalize && stats;
        if (p.sub <= 1) {
            const SinkFlags sf = sink_flags(sk);
            SinkCoef<TJ> sc;
            sink_coefs<TJ>(sk, sf, eg, sc);
            sink_epilogue<TI, TJ, 0>(acc, sk, sf, sc, eg, nullptr, nullptr, s1, s2);
        } else          // strided 1x1 conv (RegNet projection shortcuts): output rows scatter to every sub-th pixel
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = n0 + wn * WN + 32 * j + lrow;
            const bool nok = n < N;
            const int nc = min(n, N - 1);
            const float ca = affine ? sk.scale[nc] : 1.f, cb = affine ? sk.shift[nc] : 0.f;
            const float cm = stats ? sk.mean[nc] : 0.f, ci = stats ? sk.invstd[nc] : 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                // batch the 16 (y, old g) loads of this 32x32 tile at clamped addresses, then compute and store
                float yv[16], gv[16];
                size_t srow[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mc = min(m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
                    srow[r] = (size_t)mc;
                    if (p.sub > 1) {      // strided 1x1 conv: output row (b,oy,ox) scatters to input pixel (b,oy*sub,ox*sub)
                        const int ox = mc % p.OW, tt = mc / p.OW, oy = tt % p.OH, bb = tt / p.OH;
                        srow[r] = ((size_t)bb * p.IH + (size_t)oy * p.sub) * p.IW + (size_t)ox * p.sub;
                    }
                    yv[r] = need_y ? sk.y[srow[r] * sk.ldy + nc] : 0.f;
                    gv[r] = sk.accumulate ? sk.addend[srow[r] * sk.ldg + nc] : 0.f;      // (never NULL here: normalised by the host)
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float dz = acc[i][j][r] + gv[r];
                    if (need_y) {
                        const float u = fmaf(ca, yv[r], cb);
                        dz *= act_bwd(u, sact);
                        if (qmask) { const float qv = rintf(act_fwd(u, sact) * qinv) + qzp; dz = (qv >= 0.f && qv <= 255.f) ? dz : 0.f; }
                    }
                    if (m < M && nok) {
                        sk.g[srow[r] * sk.ldg + n] = dz;
                        if (stats) { s1[j] += dz; s2[j] = fmaf(dz, (yv[r] - cm) * ci, s2[j]); }
                    }
                }
            }
        }
    } else {
        want_stats = (EMODE == 1 || EMODE == 3) && (p.partial != nullptr || p.stat_acc != nullptr);
        if (eg.full) epi_store<TI, TJ, EMODE, true>(acc, p.out, p.ldo, p.bias, eg, s1, s2, mn, mx);
        else epi_store<TI, TJ, EMODE, false>(acc, p.out, p.ldo, p.bias, eg, s1, s2, mn, mx);
    }

    if (want_stats) {   // uniform across the block
        float* part = (EMODE == 2) ? p.sink.partial : p.partial;
        const int ldp = (EMODE == 2) ? p.sink.ldp : p.ldp;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (lh == 0) {
                red[(0 * 4 + wm) * 192 + wn * WN + 32 * j + lrow] = s1[j];
                red[(1 * 4 + wm) * 192 + wn * WN + 32 * j + lrow] = s2[j];
            }
        }
        if (EMODE == 3) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                mn[j] = fminf(mn[j], __shfl_xor(mn[j], 32));
                mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32));
                if (lh == 0) {
                    red[(2 * 4 + wm) * 192 + wn * WN + 32 * j + lrow] = mn[j];
                    red[(3 * 4 + wm) * 192 + wn * WN + 32 * j + lrow] = mx[j];
                }
            }
        }
        __syncthreads();
        if (t < BN) {
            constexpr int PLANES = EMODE == 3 ? 4 : 2;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < BM / WM; ++q) { a += red[(0 * 4 + q) * 192 + t]; b += red[(1 * 4 + q) * 192 + t]; }
            const int n = n0 + t;
            if (n < N && EMODE == 1 && p.stat_acc != nullptr) {   // no finalize launch: the first consumer derives the coefficients
                bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, ldp, bmy, 0, n, a);
                bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, ldp, bmy, 1, n, b);
            } else if (n < N) {
                float* dst = part + (size_t)bmy * PLANES * ldp;
                dst[n] = a;
                dst[ldp + n] = b;
                if (EMODE == 3) {
                    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
#pragma unroll
                    for (int q = 0; q < BM / WM; ++q) { lo = fminf(lo, red[(2 * 4 + q) * 192 + t]); hi = fmaxf(hi, red[(3 * 4 + q) * 192 + t]); }
                    dst[2 * ldp + n] = lo;
                    dst[3 * ldp + n] = hi;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ wgrad (TN)
struct GemmTP {
    pq_gsrc dy; pq_src x;
    float* slab;       // [splits][O*I (+O bias)] or final when splits==1
    int M, O, I, rows_per_split, want_bias;
    size_t slab_stride;
    int sub, OH, OW, IH, IW;     // strided 1x1 conv: x rows are gathered at (b, oy*sub, ox*sub)
    pq_bn_bstat dys;             // acc != NULL: dy's p / q / r derived per thread from the sink kernel's accumulators (never the writer)
};

template <int BO, int BIc>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTP p) {
    constexpr int WO = BO / 2, WI = BIc / 2, TO = WO / 32, TI = WI / 32;
    constexpr int OQ = BO / 4, IQ = BIc / 4;               // float4 per tile row
    constexpr int ORP = 256 / OQ, IRP = 256 / IQ;          // rows per pass
    constexpr int OL = 32 / ORP, IL = 32 / IRP;            // passes
    __shared__ __attribute__((aligned(16))) float Ys[32 * BO];
    __shared__ __attribute__((aligned(16))) float Xs[32 * BIc];
    __shared__ float bred[256 * 4];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wo = w >> 1, wi = w & 1;
    const int tiles_i = (p.I + BIc - 1) / BIc;
    int btile, split;
    xcd_block(btile, split);
    const int to = btile / tiles_i, ti = btile % tiles_i;
    const int o0 = to * BO, i0 = ti * BIc;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);

    const int yc4 = t % OQ, yr0 = t / OQ, xc4 = t % IQ, xr0 = t / IQ;
    const int oc = o0 + 4 * yc4, ic = i0 + 4 * xc4;
    const GsrcCoef4 gk = p.dys.acc != nullptr ? gsrc_coef4_bstat(p.dy, p.dys, oc, p.O, false) : gsrc_coef4(p.dy, oc, p.O);
    const SrcCoef4 xk = src_coef4(p.x, ic, p.I);
    const bool do_bias = p.want_bias && ti == 0;
    // unconditional loads at clamped coordinates (pitches are multiples of 4, so a whole quad is always readable);
    // rows / columns outside the problem are zeroed when the tile is written to LDS
    const int Opad = (p.O + 3) & ~3, Ipad = (p.I + 3) & ~3;
    const int ocl = min(oc, Opad - 4), icl = min(ic, Ipad - 4);

    float4 ry[OL], ry1[OL], rx[IL];
    float4 bsum = f4zero();

    auto load_tiles = [&](int mb) {
        if (oc < Opad) {       // one branch per batch: lanes whose quad lies past O skip their loads
#pragma unroll
            for (int l = 0; l < OL; ++l) {
                const int m = min(mb + yr0 + ORP * l, mend - 1);
                ry[l] = ldg4(p.dy.g + (size_t)m * p.dy.ldg + ocl);
                if (gk.two) ry1[l] = ldg4(p.dy.y + (size_t)m * p.dy.ldy + ocl);
            }
        }
        if (ic < Ipad) {
#pragma unroll
            for (int l = 0; l < IL; ++l) {
                const int m = min(mb + xr0 + IRP * l, mend - 1);
                size_t xr = (size_t)m;
                if (p.sub > 1) {
                    const int ox = m % p.OW, tt = m / p.OW, oy = tt % p.OH, bb = tt / p.OH;
                    xr = ((size_t)bb * p.IH + (size_t)oy * p.sub) * p.IW + (size_t)ox * p.sub;
                }
                rx[l] = ldg4(p.x.x + xr * p.x.ld + icl);
            }
        }
    };
    auto store_tiles = [&](int mb) {
#pragma unroll
        for (int l = 0; l < OL; ++l) {
            int r = yr0 + ORP * l;
            bool ok = (mb + r) < mend;
            float4 v = gsrc_apply4(ry[l], ry1[l], gk);
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (ok && oc + j < p.O) ? e[j] : 0.f;
            v = make_float4(e[0], e[1], e[2], e[3]);
            bsum = f4add(bsum, v);
            *reinterpret_cast<float4*>(&Ys[r * BO + 4 * yc4]) = v;
        }
#pragma unroll
        for (int l = 0; l < IL; ++l) {
            int r = xr0 + IRP * l;
            bool ok = (mb + r) < mend;
            float4 v = src_apply4(rx[l], xk);
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (ok && ic + j < p.I) ? e[j] : 0.f;
            *reinterpret_cast<float4*>(&Xs[r * BIc + 4 * xc4]) = make_float4(e[0], e[1], e[2], e[3]);
        }
    };

    f32x16 acc[TO][TI];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lrow = lane & 31, lh = lane >> 5;
    if (mbeg < mend) load_tiles(mbeg);
    for (int mb = mbeg; mb < mend; mb += 32) {
        __syncthreads();
        store_tiles(mb);
        __syncthreads();
        if (mb + 32 < mend) load_tiles(mb + 32);
        // operands of step kk+1 are fetched from LDS before the MFMAs of step kk are issued (one accumulator per wave on the
        // 64x64 tile: without the prefetch every pair of MFMAs waited for its own LDS round trip)
        float af[TO], bf[TI];
#pragma unroll
        for (int a = 0; a < TO; ++a) af[a] = Ys[lh * BO + wo * WO + 32 * a + lrow];
#pragma unroll
        for (int b = 0; b < TI; ++b) bf[b] = Xs[lh * BIc + wi * WI + 32 * b + lrow];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float an[TO], bn[TI];
            if (kk + 1 < 16) {
#pragma unroll
                for (int a = 0; a < TO; ++a) an[a] = Ys[(2 * (kk + 1) + lh) * BO + wo * WO + 32 * a + lrow];
#pragma unroll
                for (int b = 0; b < TI; ++b) bn[b] = Xs[(2 * (kk + 1) + lh) * BIc + wi * WI + 32 * b + lrow];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch above the MFMAs (the scheduler sinks it back otherwise)
#pragma unroll
            for (int a = 0; a < TO; ++a)
#pragma unroll
                for (int b = 0; b < TI; ++b) acc[a][b] = mfma32(af[a], bf[b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < 16) {
#pragma unroll
                for (int a = 0; a < TO; ++a) af[a] = an[a];
#pragma unroll
                for (int b = 0; b < TI; ++b) bf[b] = bn[b];
            }
        }
    }

    float* dst = p.slab + (size_t)split * p.slab_stride;
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) {
            const int i = i0 + wi * WI + 32 * b + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * WO + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < p.O && i < p.I) dst[(size_t)o * p.I + i] = acc[a][b][r];
            }
        }
    if (do_bias) {   // uniform per block
        __syncthreads();
        bred[t * 4 + 0] = bsum.x; bred[t * 4 + 1] = bsum.y; bred[t * 4 + 2] = bsum.z; bred[t * 4 + 3] = bsum.w;
        __syncthreads();
        if (t < OQ) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = t; q < 256; q += OQ)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += bred[q * 4 + j];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (o0 + 4 * t + j < p.O) dst[(size_t)p.O * p.I + o0 + 4 * t + j] = s[j];
        }
    }
}

// out[j] = sum_s slab[s*stride + j]; block = 32 columns x 32 split lanes, 8 loads in flight per thread
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* slab, size_t stride, int splits, float* out0, int n0, float* out1, int n1) {
    __shared__ float red[32][33];
    const int n = n0 + n1;
    const int j = blockIdx.x * 32 + threadIdx.x;
    float s = 0.f;
    if (j < n) {
        int k = threadIdx.y;
        for (; k + 7 * 32 < splits; k += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + 32 * u) * stride + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < splits; k += 32) s += slab[(size_t)k * stride + j];
    }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && j < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
        if (j < n0) out0[j] = t; else out1[j - n0] = t;
    }
}

// few slabs, many columns (the usual wgrad case: <= 64 row splits of a large O x I matrix): one thread per column, 8 loads in
// flight, every wave reads 256 contiguous bytes per slab
__global__ __launch_bounds__(256) void slab_reduce_cols_kernel(const float* slab, size_t stride, int splits, float* out0, int n0, float* out1, int n1) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n0 + n1) return;
    const float* src = slab + j;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += src[(size_t)k * stride];
    if (j < n0) out0[j] = s; else out1[j - n0] = s;
}

}  // namespace pq

using namespace pq;

// ------------------------------------------------------------------------------------------------ host dispatch
// Block-tile choice.
//  * streaming tiles 128 x (32*TJ), TJ = ceil(N/32) <= 6, one block per 128 rows covering ALL N columns: the A tile is
//    read once and the block writes one contiguous slab of the output -- used whenever that still fills the chip;
//  * otherwise the best of 128x128 / 128x64 / 64x64 / 128x32 by padding waste and block count (double-buffered LDS).
struct TileCand { int bm, bn; int occ; float eff; };
// occ = workgroups of that tile shape that fit on a CU together (registers / LDS), eff = relative efficiency of its inner loop
static const TileCand kTiles[4] = {{128, 128, 2, 1.00f}, {128, 64, 2, 0.98f}, {64, 64, 3, 0.95f}, {128, 32, 4, 0.85f}};
struct TilePick { int kind; int idx; int bm, bn; };     // kind 0: kTiles[idx]; kind 1: streaming with TJ = idx
static TilePick pick_tile(int M, int N) {
    const int tj = (N + 31) / 32;
    if (tj <= 6 && (long)(M + 127) / 128 >= 384) return {1, tj, 128, 32 * tj};
    // cost model: the chip holds 256*occ workgroups at once; co-resident workgroups share their CU, so one "wave" of
    // workgroups costs occ * tile area, and a partially filled last wave costs as much as a full one.  Reproduces the
    // measured ranking of the three big tiles on every mid / low resolution layer shape (PQ_GEMM_TILE=0..3 forces one).
    int best = 0;
    float best_cost = 3.4e38f;
    for (int c = 0; c < 4; ++c) {
        const long gm = (M + kTiles[c].bm - 1) / kTiles[c].bm, gn = (N + kTiles[c].bn - 1) / kTiles[c].bn;
        const long slots = 256L * kTiles[c].occ;
        const long rounds = (gm * gn + slots - 1) / slots;
        const float cost = (float)rounds * kTiles[c].occ * kTiles[c].bm * kTiles[c].bn / kTiles[c].eff;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return {0, best, kTiles[best].bm, kTiles[best].bn};
}
// gemm2.hip (LDS-DMA ring kernels)
int g2_conv1x1_fwd(const pq_conv1x1_fwd_args* a, int bm, hipStream_t st, pq_gemm_plan* plan);
int g2_conv1x1_dgrad(const pq_conv1x1_dgrad_args* a, int bm, hipStream_t st, pq_gemm_plan* plan);
int g2_conv1x1_wgrad(const pq_conv1x1_wgrad_args* a, int* splits_out, size_t* stride_out, hipStream_t st, pq_gemm_plan* plan);
int g2_wgrad_splits(int M, int I, int O);

// Row-block height of the GEMM (M rows, N output columns, contraction K; kind 0 forward, 1 data gradient).  The BatchNorm
// statistics rows are sized from it at plan time, so it is a pure function of the shape.  K % 16 == 0 (K >= 64) shapes are
// tiled for the LDS-DMA kernels: the largest tile that still gives every CU two workgroups; the data gradient stages two
// tensors per row, so its row block is 64 wherever that keeps two workgroups per CU in LDS.
static int row_bm(int M, int N, int K, int kind) {
    const bool g2 = K >= (kind == 0 ? 128 : 64) && (K & 15) == 0 && N >= 64;
    if (!g2) return pick_tile(M, N).bm;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
    if (kind == 0) return (t128 >= 400 || t12864 >= 400) ? 128 : 64;
    // data gradient: 128-row tiles stage two tensors per row (126-150 KB of LDS: one workgroup per CU) -- only where 64-row tiles would
    // mean more than ~16 rounds (threshold 512 / 1024 / 2048 / never: 10.35 / 10.20 / 10.12 / 10.13 ms per mobilenetv2-fpn step)
    return t128 >= 2048 ? 128 : 64;
}
extern "C" int32_t pq_conv1x1_blocks_k(int32_t M, int32_t N, int32_t K, int32_t kind) {
    const int bm = row_bm(M, N, K, kind);
    return (M + bm - 1) / bm;
}
extern "C" int32_t pq_conv1x1_blocks(int32_t M, int32_t N) { return pq_conv1x1_blocks_k(M, N, 0, 0); }

// best column width for a fixed row-block height (same cost model as pick_tile)
static TilePick pick_tile_bm(int M, int N, int bm) {
    TilePick t = pick_tile(M, N);
    if (t.bm == bm) return t;
    int best = -1;
    float best_cost = 3.4e38f;
    for (int c = 0; c < 4; ++c) {
        if (kTiles[c].bm != bm) continue;
        const long gm = (M + kTiles[c].bm - 1) / kTiles[c].bm, gn = (N + kTiles[c].bn - 1) / kTiles[c].bn;
        const long slots = 256L * kTiles[c].occ;
        const long rounds = (gm * gn + slots - 1) / slots;
        const float cost = (float)rounds * kTiles[c].occ * kTiles[c].bm * kTiles[c].bn / kTiles[c].eff;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return {0, best, kTiles[best].bm, kTiles[best].bn};
}
template <int AMODE, bool BKN, int EMODE>
static void launch_nt(const GemmP& p, hipStream_t st, int bm, pq_gemm_plan* plan) {
    TilePick t = pick_tile_bm(p.M, p.N, bm);
    // the 128 x 192 streaming tile of a gradient sink keeps 96 accumulators + the sink's operands per lane; its epilogue indexed the
    // accumulator array by a loop variable the compiler did not unroll, so the whole array lived in a 448-byte private segment
    // (scratch): two 96-wide column tiles instead (the contraction of these layers is
    // 32-64 long, so reading the A rows twice is cheap; the sink's columns are not duplicated): 10.07-10.10 -> 10.01-10.07 ms
    if (EMODE == 2 && t.kind == 1 && t.bn == 192) { t.idx = 3; t.bn = 96; }
    if (plan) { *plan = pq_gemm_plan{1, t.bm, t.bn, t.kind == 0 ? 2 : 1, AMODE, EMODE, 0, BKN ? 1 : 0}; return; }   // query only
    dim3 blk(256);
    dim3 grid((p.N + t.bn - 1) / t.bn, (p.M + t.bm - 1) / t.bm);
    if (t.kind == 0) {
        switch (t.idx) {
            case 0: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 128, 64, 64, AMODE, BKN, EMODE, true, 1>), grid, blk, 0, st, p); break;
            case 1: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 64, 64, 32, AMODE, BKN, EMODE, true, 2>), grid, blk, 0, st, p); break;
            case 2: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<64, 64, 32, 32, AMODE, BKN, EMODE, true, 2>), grid, blk, 0, st, p); break;
            default: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 32, 32, 32, AMODE, BKN, EMODE, true, 2>), grid, blk, 0, st, p); break;
        }
    } else {
        switch (t.idx) {
            case 1: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 32, 32, 32, AMODE, BKN, EMODE, false>), grid, blk, 0, st, p); break;
            case 2: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 64, 32, 64, AMODE, BKN, EMODE, false>), grid, blk, 0, st, p); break;
            case 3: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 96, 32, 96, AMODE, BKN, EMODE, false>), grid, blk, 0, st, p); break;
            case 4: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 128, 32, 128, AMODE, BKN, EMODE, false>), grid, blk, 0, st, p); break;
            case 5: PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 160, 32, 160, AMODE, BKN, EMODE, false>), grid, blk, 0, st, p); break;
            default:
                // (never reached with a gradient sink, see above: not instantiated either -- tests/test_codeobj.py asserts that no kernel
                // of the library has a private segment)
                if constexpr (EMODE != 2) PQ_LAUNCH_BINDABLE((gemm_nt_kernel<128, 192, 32, 192, AMODE, BKN, EMODE, false>), grid, blk, 0, st, p);
                break;
        }
    }
}

static int conv1x1_fwd_impl(const pq_conv1x1_fwd_args* a, void* stream, pq_gemm_plan* plan) {
    PQ_CHECK_ARG(a && a->in.x && a->w, "null pointer");
    // out == NULL: statistics only (the first pass of a fused expand + depthwise block: pq_xdw_fwd recomputes the values)
    PQ_CHECK_ARG(a->out || ((a->partial || a->stat_acc) && !a->minmax && !a->bias), "out == NULL needs statistics (no min / max, no bias)");
    PQ_CHECK_ARG(a->M > 0 && a->I > 0 && a->O > 0, "empty shape");
    PQ_CHECK_ARG((a->in.ld & 3) == 0 && (((uintptr_t)a->in.x) & 15) == 0, "input pitch/pointer must be 16B aligned");
    PQ_CHECK_ARG((!a->out || a->ldo >= a->O) && a->in.ld >= a->I, "pitch smaller than channel count");
    PQ_CHECK_ARG(a->partial == nullptr || a->ldp >= a->O, "stats pitch too small");
    PQ_CHECK_ARG(a->sub >= 1, "sub must be >= 1");
    if (act_is_special(a->in.act)) { pq_set_error("pq_conv1x1_fwd: logistic/tanh operands must be materialised first"); return PQ_EUNSUPPORTED; }
    GemmP p{};
    p.a0 = a->in.x; p.lda0 = a->in.ld; p.c0 = a->in.scale; p.c1 = a->in.shift; p.act = a->in.act; p.qstate = a->in.qstate;
    p.sub = a->sub; p.OH = a->OH; p.OW = a->OW; p.IH = a->IH; p.IW = a->IW;
    p.b = a->w; p.ldb = a->I; p.M = a->M; p.N = a->O; p.K = a->I;
    p.out = a->out; p.bias = a->bias; p.partial = a->partial; p.ldo = a->ldo; p.ldp = a->ldp;
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    PQ_CHECK_ARG(a->stat_acc == nullptr || (a->stat_bad != nullptr && a->minmax == 0 && a->ldp >= a->O), "stat_acc needs stat_bad, no minmax");
    if (a->stat_acc && !stat_adds_ok(pq_conv1x1_blocks_k(a->M, a->O, a->I, 0), a->stat_slots)) {
        pq_set_error("pq_conv1x1_fwd: more than %d row blocks per accumulator slot (use `partial` rows + pq_bn_fwd_finalize)", PQ_STAT_MAX_ADDS);
        return PQ_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool xform = a->in.scale != nullptr || a->in.act != PQ_ACT_LINEAR || a->in.qstate != nullptr;
    const int bm = row_bm(a->M, a->O, a->I, 0);
    const bool minmax = a->partial != nullptr && a->minmax != 0;      // QAT only: the round-1 kernels carry the 4-plane epilogue
    const bool stats = a->partial != nullptr || a->stat_acc != nullptr;
    if (!minmax) {
        const int rc = g2_conv1x1_fwd(a, bm, st, plan);
        if (rc < 0) return rc;
        if (rc == 1) { if (plan) return PQ_OK; PQ_CHECK_LAUNCH(); return PQ_OK; }
    }
    if (a->in.stat.acc != nullptr) {      // only the LDS-DMA ring kernels derive the operand's BatchNorm from accumulators
        pq_set_error("pq_conv1x1_fwd: in.stat is not supported for this shape (ask pq_bn_stat_ok first)");
        return PQ_EUNSUPPORTED;
    }
    if (xform) {
        if (minmax) launch_nt<1, false, 3>(p, st, bm, plan);
        else if (stats) launch_nt<1, false, 1>(p, st, bm, plan); else launch_nt<1, false, 0>(p, st, bm, plan);
    } else {
        if (minmax) launch_nt<0, false, 3>(p, st, bm, plan);
        else if (stats) launch_nt<0, false, 1>(p, st, bm, plan); else launch_nt<0, false, 0>(p, st, bm, plan);
    }
    if (plan) return PQ_OK;
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
extern "C" int pq_conv1x1_fwd(const pq_conv1x1_fwd_args* a, void* stream) { return conv1x1_fwd_impl(a, stream, nullptr); }
extern "C" int pq_conv1x1_fwd_plan(const pq_conv1x1_fwd_args* a, pq_gemm_plan* out) {
    PQ_CHECK_ARG(out, "null pointer");
    return conv1x1_fwd_impl(a, nullptr, out);
}
// pq_bn_stat_ok for PQ_OP_CONV1X1_FWD: both kernel families can ADD their statistics (role 0, no 4-plane rows); only the ring
// kernels derive the operand's coefficients from accumulators (role 1)
int conv1x1_fwd_stat_ok(const pq_conv1x1_fwd_args* a, int role) {
    if (!a || a->minmax) return 0;
    if (role == 0) return 1;
    pq_conv1x1_fwd_args b = *a;
    b.in.stat.acc = reinterpret_cast<int64_t*>(uintptr_t(256));        // ask the dispatch as if the hand-over were on
    if (!b.in.scale) b.in.scale = b.in.shift = reinterpret_cast<const float*>(uintptr_t(256));
    pq_gemm_plan pl{};
    if (conv1x1_fwd_impl(&b, nullptr, &pl) != PQ_OK) return 0;
    return pl.family == 2 && pl.amode == 3;
}

static int conv1x1_dgrad_impl(const pq_conv1x1_dgrad_args* a, void* stream, pq_gemm_plan* plan) {
    PQ_CHECK_ARG(a && a->dy.g && a->w && a->dx.g, "null pointer");
    PQ_SINK_ADDEND_CHECK(a->dx);
    PQ_CHECK_ARG(a->M > 0 && a->I > 0 && a->O > 0, "empty shape");
    PQ_CHECK_ARG((a->dy.ldg & 3) == 0 && (a->dy.y == nullptr || (a->dy.ldy & 3) == 0), "gradient pitch must be a multiple of 4");
    PQ_CHECK_ARG(!(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    if (act_is_special(a->dx.act)) { pq_set_error("pq_conv1x1_dgrad: logistic/tanh sinks are not supported"); return PQ_EUNSUPPORTED; }
    if (a->sub > 1) {
        PQ_CHECK_ARG(a->OH > 0 && a->OW > 0 && a->M % (a->OH * a->OW) == 0, "strided dgrad: M must be B*OH*OW");
        PQ_CHECK_ARG((a->OH - 1) * a->sub < a->IH && (a->OW - 1) * a->sub < a->IW, "strided dgrad: input plane too small");
        PQ_CHECK_ARG(!a->dx.finalize, "strided dgrad: the sink must be accumulate-only (finalize it with pq_route_bwd)");
    }
    GemmP p{};
    p.a0 = a->dy.g; p.a1 = a->dy.y; p.lda0 = a->dy.ldg; p.lda1 = a->dy.ldy;
    p.c0 = a->dy.p; p.c1 = a->dy.q; p.c2 = a->dy.mu; p.c3 = a->dy.r;
    p.sub = a->sub > 1 ? a->sub : 1; p.OH = a->OH; p.OW = a->OW; p.IH = a->IH; p.IW = a->IW;
    p.b = a->w; p.ldb = a->I; p.M = a->M; p.N = a->I; p.K = a->O;
    p.sink = a->dx;
    if (!p.sink.addend) p.sink.addend = p.sink.g;      // the kernels of THIS file read the running gradient through `addend` (no select: SGPRs)
    if (a->dy_stat.acc && !plan) {
        PQ_CHECK_ARG(a->dy.y && a->dy.mu && a->dy_stat.bad && a->dy_stat.gamma && a->dy_stat.invstd && a->dy_stat.dgamma && a->dy_stat.dbeta &&
                     a->dy_stat.p && a->dy_stat.q && a->dy_stat.r && a->dy_stat.count > 0.f && a->dy_stat.ldp >= a->O && a->dy_stat.slots <= 4,
                     "dy_stat: incomplete");
    }
    if (a->dx.stat_acc) {     // the sink's BatchNorm-backward sums go to accumulators (pq_bn_bstat), not to rows
        PQ_CHECK_ARG(a->dx.stat_bad && a->sub <= 1 && a->dx.finalize && a->dx.mean && a->dx.partial && a->dx.ldp >= a->I && a->dx.stat_slots <= 4,
                     "dx.stat_acc needs stat_bad, sub == 1 and a finalising sink with statistics");
        if (!stat_adds_ok(pq_conv1x1_blocks_k(a->M, a->I, a->O, 1), a->dx.stat_slots)) {
            pq_set_error("pq_conv1x1_dgrad: more than %d row blocks per accumulator slot (use `partial` rows + pq_bn_bwd_finalize)", PQ_STAT_MAX_ADDS);
            return PQ_EUNSUPPORTED;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int bm = row_bm(a->M, a->I, a->O, 1);
    {
        const int rc = g2_conv1x1_dgrad(a, bm, st, plan);
        if (rc < 0) return rc;
        if (rc == 1) { if (plan) return PQ_OK; PQ_CHECK_LAUNCH(); return PQ_OK; }
    }
    // only the ring kernels (gemm2.hip) have the atomic sink epilogue: in the register-staged family the extra path made the
    // 128 x 192 sink instantiations spill to scratch (the only scratch users of the library)
    if ((a->dx.stat_acc || a->dy_stat.acc) && !plan) {
        pq_set_error("pq_conv1x1_dgrad: dx.stat_acc / dy_stat are not supported for this shape (pq_bn_bstat_ok)");
        return PQ_EUNSUPPORTED;
    }
    if (a->wt) {        // transposed weights [I][ceil4(O)] (pad columns zero): B rows are K-contiguous like the forward operand
        p.b = a->wt; p.ldb = (a->O + 3) & ~3;
        if (a->dy.y) launch_nt<2, false, 2>(p, st, bm, plan); else launch_nt<0, false, 2>(p, st, bm, plan);
    } else {
        if (a->dy.y) launch_nt<2, true, 2>(p, st, bm, plan); else launch_nt<0, true, 2>(p, st, bm, plan);
    }
    if (plan) return PQ_OK;
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
extern "C" int pq_conv1x1_dgrad(const pq_conv1x1_dgrad_args* a, void* stream) { return conv1x1_dgrad_impl(a, stream, nullptr); }
extern "C" int pq_conv1x1_dgrad_plan(const pq_conv1x1_dgrad_args* a, pq_gemm_plan* out) {
    PQ_CHECK_ARG(out, "null pointer");
    return conv1x1_dgrad_impl(a, nullptr, out);
}

// pq_bn_bstat_ok for PQ_OP_CONV1X1_DGRAD: producer (role 0: the sink adds) and consumer (role 1: dy's coefficients derived) roles both
// need the shape to dispatch to the ring kernels (family 2)
int conv1x1_dgrad_bstat_ok(const pq_conv1x1_dgrad_args* a, int role) {
    if (a->sub > 1) return 0;
    pq_conv1x1_dgrad_args b = *a;
    if (role == 0) {
        if (!(a->dx.finalize && a->dx.mean != nullptr && a->dx.partial != nullptr && !a->dx.qstate)) return 0;
    } else {
        if (!a->dy.y || !a->dy.mu) return 0;
        b.dy_stat.acc = reinterpret_cast<int64_t*>(uintptr_t(256));     // ask the dispatch as if the hand-over were on
    }
    pq_gemm_plan pl{};
    if (conv1x1_dgrad_impl(&b, nullptr, &pl) != PQ_OK || pl.family != 2) return 0;
    return role == 0 || pl.amode == 4;
}

static void wgrad_plan(int M, int I, int O, int& tiles, int& splits, int& rows) {
    // 64x64 output tiles everywhere: four times more tiles than 128x128 means four times fewer row splits for the same
    // number of workgroups, i.e. longer contraction loops per workgroup and a quarter of the slab traffic; measured equal
    // or faster on every layer shape of the nets
    const int b = 64;
    tiles = ((O + b - 1) / b) * ((I + b - 1) / b);
    static const char* tg_env = getenv("PQ_TN_TARGET");             // A/B only
    const int target = tg_env ? atoi(tg_env) : 512;
    int want = (target + tiles - 1) / tiles;
    int maxs = (M + 255) / 256;
    splits = want < 1 ? 1 : (want > maxs ? maxs : want);
    if (splits < 1) splits = 1;
    rows = (((M + splits - 1) / splits) + 31) / 32 * 32;
    splits = (M + rows - 1) / rows;
}

extern "C" int64_t pq_conv1x1_wgrad_scratch_bytes(int32_t M, int32_t I, int32_t O) {
    int tiles, splits, rows;
    wgrad_plan(M, I, O, tiles, splits, rows);
    const int s2 = g2_wgrad_splits(M, I, O);
    if (s2 > splits) splits = s2;
    return (int64_t)splits * ((int64_t)O * I + O) * 4;
}

// item != NULL: no launch, describe the slabs a deferred call (dw == NULL) leaves in `scratch` (pq_reduce_batch)
static int conv1x1_wgrad_impl(const pq_conv1x1_wgrad_args* a, void* stream, pq_gemm_plan* plan, pq_reduce_item* item = nullptr) {
    PQ_CHECK_ARG(a && a->dy.g && a->in.x && (a->scratch || plan), "null pointer");
    PQ_CHECK_ARG(!item || a->dw, "the item query needs the gradient's address");
    const bool deferred = a->dw == nullptr;
    PQ_CHECK_ARG(a->M > 0 && a->I > 0 && a->O > 0, "empty shape");
    PQ_CHECK_ARG((a->dy.ldg & 3) == 0 && (a->in.ld & 3) == 0 && (a->dy.y == nullptr || (a->dy.ldy & 3) == 0), "pitches must be multiples of 4");
    if (act_is_special(a->in.act)) { pq_set_error("pq_conv1x1_wgrad: logistic/tanh operands must be materialised first"); return PQ_EUNSUPPORTED; }
    if (a->dy_stat.acc && !plan) {
        PQ_CHECK_ARG(a->dy.y && a->dy.mu && a->dy_stat.bad && a->dy_stat.gamma && a->dy_stat.invstd && a->dy_stat.count > 0.f &&
                     a->dy_stat.ldp >= a->O && a->dy_stat.slots <= 4, "dy_stat: incomplete");
    }
    int tiles, splits, rows;
    hipStream_t st = (hipStream_t)stream;
    {
        int s2 = 0; size_t stride2 = 0;
        const int rc = g2_conv1x1_wgrad(a, &s2, &stride2, st, plan);
        if (rc < 0) return rc;
        if (rc == 1) {
            const int n0 = a->O * a->I;
            if (item) *item = pq_reduce_item{a->scratch, a->dw, nullptr, (int64_t)stride2, 0, s2, n0, 0, n0, 0, 0, 0, 0};
            if (plan) return PQ_OK;
            PQ_CHECK_LAUNCH();
            if (deferred) return PQ_OK;
            if (s2 <= 64) PQ_LAUNCH(slab_reduce_cols_kernel, dim3((n0 + 255) / 256), dim3(256), 0, st, (const float*)a->scratch,
                                             stride2, s2, a->dw, n0, (float*)nullptr, 0);
            else PQ_LAUNCH(slab_reduce_kernel, dim3((n0 + 31) / 32), dim3(32, 32), 0, st, (const float*)a->scratch, stride2, s2,
                                    a->dw, n0, (float*)nullptr, 0);
            PQ_CHECK_LAUNCH();
            return PQ_OK;
        }
    }
    wgrad_plan(a->M, a->I, a->O, tiles, splits, rows);
    if (item) {
        const int n0 = a->O * a->I, n1 = a->dbias ? a->O : 0;
        *item = pq_reduce_item{a->scratch, a->dw, a->dbias, (int64_t)a->O * a->I + a->O, 0, splits, n0, n1, n0 + n1, 0, 0, 0, 0};
    }
    if (plan) {
        *plan = pq_gemm_plan{1, 64, 64, 1, (a->dy.y ? 1 : 0) | ((a->in.scale || a->in.act != PQ_ACT_LINEAR) ? 2 : 0), 0, splits, 0};
        return PQ_OK;
    }
    GemmTP p{};
    p.dy = a->dy; p.x = a->in; p.slab = a->scratch; p.M = a->M; p.O = a->O; p.I = a->I;
    p.dys = a->dy_stat;
    if (p.dys.slots < 1) p.dys.slots = 1;
    p.rows_per_split = rows; p.want_bias = a->dbias != nullptr;
    p.sub = a->sub > 1 ? a->sub : 1; p.OH = a->OH; p.OW = a->OW; p.IH = a->IH; p.IW = a->IW;
    if (a->sub > 1) {
        PQ_CHECK_ARG(a->OH > 0 && a->OW > 0 && a->M % (a->OH * a->OW) == 0, "strided wgrad: M must be B*OH*OW");
        PQ_CHECK_ARG((a->OH - 1) * a->sub < a->IH && (a->OW - 1) * a->sub < a->IW, "strided wgrad: input plane too small");
    }
    p.slab_stride = (size_t)a->O * a->I + a->O;
    dim3 grid(tiles, splits), blk(256);
    PQ_LAUNCH((gemm_tn_kernel<64, 64>), grid, blk, 0, st, p);
    PQ_CHECK_LAUNCH();
    if (deferred) return PQ_OK;
    const int n0 = a->O * a->I, n1 = a->dbias ? a->O : 0;
    if (splits <= 64) {
        PQ_LAUNCH(slab_reduce_cols_kernel, dim3((n0 + n1 + 255) / 256), dim3(256), 0, st, (const float*)a->scratch,
                           p.slab_stride, splits, a->dw, n0, a->dbias, n1);
        PQ_CHECK_LAUNCH();
        return PQ_OK;
    }
    const int rb = (n0 + n1 + 31) / 32;
    PQ_LAUNCH(slab_reduce_kernel, dim3(rb), dim3(32, 32), 0, st, (const float*)a->scratch, p.slab_stride, splits,
                       a->dw, n0, a->dbias, n1);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
extern "C" int pq_conv1x1_wgrad(const pq_conv1x1_wgrad_args* a, void* stream) { return conv1x1_wgrad_impl(a, stream, nullptr); }
extern "C" int pq_conv1x1_wgrad_plan(const pq_conv1x1_wgrad_args* a, pq_gemm_plan* out) {
    PQ_CHECK_ARG(out, "null pointer");
    return conv1x1_wgrad_impl(a, nullptr, out);
}
extern "C" int pq_conv1x1_wgrad_reduce_item(const pq_conv1x1_wgrad_args* a, pq_reduce_item* out) {
    PQ_CHECK_ARG(out, "null pointer");
    pq_gemm_plan pl{};
    return conv1x1_wgrad_impl(a, nullptr, &pl, out);
}
