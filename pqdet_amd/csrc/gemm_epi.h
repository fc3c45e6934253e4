// Epilogues shared by the two NT GEMM kernel families (gemm.hip: register-staged, gemm2.hip: LDS-DMA ring).
//
// The accumulators of v_mfma_f32_32x32x2_f32 leave in their NATIVE layout: lane = output column (lrow = lane & 31), 16 rows per
// lane and 32 x 32 block, row(r) = (r & 3) + 8 * (r >> 2) + 4 * lh (lh = lane >> 5).  Every global access of a wave instruction
// is then two 128-byte row segments (full-rate dword accesses), there is no LDS transpose, per-column coefficients and
// statistics are ONE value per lane, and a tile that lies inside the matrix takes a path without masks or branches.  Addresses:
// a wave-uniform pointer to (first row of the block + rowc(r), first column) -- an SGPR pair, SALU arithmetic -- plus ONE 32-bit
// per-lane element offset (4 * lh rows down, lrow columns right) shared by all 16 accesses of a block; tiles that stick out of
// the matrix clamp / mask per lane instead (edge tiles only).
// Round 3's epilogue went through per-wave LDS patches and tested row / column / alignment flags per element: 6-15 k of a tile's
// 35-95 k cycles in the in-kernel stamps (scripts/gemm_stamps.py).
#pragma once
#include <type_traits>
#include "common.h"

namespace pq {

struct EpiGeom {
    int m0w, n0w;        // first row / column of the WAVE's tile (wave-uniform: built from readfirstlane'd wave coordinates)
    int lrow, lh;        // lane & 31, lane >> 5
    int M, N;
    bool full;           // every ROW of the workgroup tile lies inside the matrix (uniform); columns past N are masked per lane
};
__device__ __forceinline__ int epi_rowc(int r) { return (r & 3) + 8 * (r >> 2); }

// ---------------------------------------------------------------------------------------------- store (+ bias, + statistics)
// EMODE 0: store; 1: + per-column sum / sum of squares into t1 / t2; 3: + per-column min / max (QAT observers)
// out == NULL (uniform; EMODE 1 only, pq_conv1x1_fwd_args.out): statistics ONLY, nothing is stored -- the first pass of a fused
// expand + depthwise block (irblock.hip), whose expanded tensor is never materialised
template <int TI, int TJ, int EMODE, bool F>
__device__ __forceinline__ void epi_store(const f32x16 (&acc)[TI][TJ], float* out, int ldo, const float* bias, const EpiGeom& g,
                                          float (&t1)[TJ], float (&t2)[TJ], float (&mn)[TJ], float (&mx)[TJ]) {
    // every column block's bias first: loaded inside the block loop, the (conditional) load's wait sat in front of each block's first
    // store as s_waitcnt vmcnt(0) -- with or without a bias it drained the PREVIOUS block's 16 stores, TJ - 1 store round trips per tile
    float bjv[TJ];
    const bool st = out != nullptr;                                     // uniform
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = g.n0w + 32 * j + g.lrow;
        bjv[j] = (bias != nullptr && n < g.N) ? bias[n] : 0.f;
    }
    // (an opaque use in straight-line code: the compiler waits for the conditional loads HERE, once; without it every later
    // lane-masked store block re-waited for them with vmcnt(0), which also drains the stores issued since)
#pragma unroll
    for (int j = 0; j < TJ; ++j) asm volatile("" : "+v"(bjv[j]));
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = g.n0w + 32 * j + g.lrow;
        const bool nok = n < g.N;                                       // (widths that are not multiples of the tile: the last column tile)
        const unsigned oo = 4u * g.lh * (unsigned)ldo + (nok ? g.lrow : 0);
        const float bj = bjv[j];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int mb = g.m0w + 32 * i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = epi_rowc(r);
                const float v = acc[i][j][r] + bj;
                const bool ok = nok && (F || mb + 4 * g.lh + row < g.M);
                if (F) { if (nok && st) (out + (size_t)(mb + row) * (size_t)ldo + (g.n0w + 32 * j))[oo] = v; }
                else if (ok && st) out[(size_t)(mb + 4 * g.lh + row) * ldo + n] = v;
                if (EMODE != 0 && ok) {
                    t1[j] += v; t2[j] = fmaf(v, v, t2[j]);
                    if (EMODE == 3) { mn[j] = fminf(mn[j], v); mx[j] = fmaxf(mx[j], v); }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- gradient sink
// (pq_gsink.addend is never NULL here: both GEMM hosts normalise it to g -- a select in the kernel costs SGPRs these tiles do not have)
struct SinkFlags { bool affine, stats, qmask, need_y, acc; };
__device__ __forceinline__ SinkFlags sink_flags(const pq_gsink& sk) {
    SinkFlags f;
    f.affine = sk.scale != nullptr;
    f.stats = sk.mean != nullptr && sk.partial != nullptr;
    // QAT: the tensor is consumed as fake_quant(act(...)): straight-through mask of its activation observer (arithmetic of
    // qat.hip / sink_apply4: the multiply sits inside rintf, nothing here can contract)
    f.qmask = sk.finalize && sk.qstate != nullptr;
    f.need_y = sk.finalize && (sk.act != PQ_ACT_LINEAR || f.stats || f.qmask);
    f.acc = sk.accumulate != 0;
    return f;
}
template <int TJ> struct SinkCoef { float a[TJ], b[TJ], m[TJ], i[TJ]; };
template <int TJ>
__device__ __forceinline__ void sink_coefs(const pq_gsink& sk, const SinkFlags& f, const EpiGeom& g, SinkCoef<TJ>& c) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = g.n0w + 32 * j + g.lrow;
        const int nc = n < g.N ? n : 0;
        c.a[j] = f.affine ? sk.scale[nc] : 1.f; c.b[j] = f.affine ? sk.shift[nc] : 0.f;
        c.m[j] = f.stats ? sk.mean[nc] : 0.f; c.i[j] = f.stats ? sk.invstd[nc] : 0.f;
    }
}
// the 16 (y, running gradient) values of block (i, j) of the wave's tile
template <bool F>
__device__ __forceinline__ void sink_block_loads(const pq_gsink& sk, const SinkFlags& f, const EpiGeom& g, int i, int j, float* yd, float* gd) {
    if (F) {
        // columns past N (widths that are not multiples of the tile): the block's base column is clamped to the last block that still
        // has a valid column and the lane to the last valid column -- never an address past the row's end (the last rows of the
        // tensor would otherwise read past the allocation); the values are discarded by the column mask of the store
        const int nb = min(g.n0w + 32 * j, (g.N - 1) & ~31);                           // uniform
        const unsigned lc = (unsigned)min(g.lrow, g.N - 1 - nb);
        const unsigned oy = 4u * g.lh * (unsigned)sk.ldy + lc, og = 4u * g.lh * (unsigned)sk.ldg + lc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = g.m0w + 32 * i + epi_rowc(r), n = nb;                       // uniform
            yd[r] = f.need_y ? (sk.y + (size_t)m * (size_t)sk.ldy + n)[oy] : 0.f;
            gd[r] = f.acc ? (sk.addend + (size_t)m * (size_t)sk.ldg + n)[og] : 0.f;
        }
    } else {
        const int n = g.n0w + 32 * j + g.lrow;
        const int nc = n < g.N ? n : 0;
        const int mb = g.m0w + 32 * i + 4 * g.lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mc = min(mb + epi_rowc(r), g.M - 1);
            yd[r] = f.need_y ? sk.y[(size_t)mc * sk.ldy + nc] : 0.f;
            gd[r] = f.acc ? sk.addend[(size_t)mc * sk.ldg + nc] : 0.f;
        }
    }
}
// NPRE: blocks whose operands were requested before the epilogue (pre_y / pre_g, indexed j * TI + i); 0: the blocks load here, one
// block ahead of the one being finished (two blocks' operands live: the streaming tiles of gemm.hip have up to six per wave)
template <int TI, int TJ, bool F, bool Q, int NPRE>
__device__ __forceinline__ void sink_emit(const f32x16 (&acc)[TI][TJ], const pq_gsink& sk, const SinkFlags& f, const SinkCoef<TJ>& c,
                                          const EpiGeom& g, const float (*pre_y)[16], const float (*pre_g)[16], float qinv, float qzp,
                                          float (&t1)[TJ], float (&t2)[TJ]) {
    const ActParam sact = act_param(sk.act);
    const unsigned og = 4u * g.lh * (unsigned)sk.ldg + g.lrow;
    constexpr bool PRE = NPRE > 0;
    constexpr int NBLK = TI * TJ;
    float yl[2][16], gl[2][16];
    if (!PRE) sink_block_loads<F>(sk, f, g, 0, 0, yl[0], gl[0]);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int j = b / TI, i = b % TI;
        if (!PRE && b + 1 < NBLK) sink_block_loads<F>(sk, f, g, (b + 1) % TI, (b + 1) / TI, yl[(b + 1) & 1], gl[(b + 1) & 1]);
        const int n = g.n0w + 32 * j + g.lrow;
        const bool nok = n < g.N;
        const int mb = g.m0w + 32 * i;
        const float* yy = PRE ? pre_y[PRE ? b : 0] : yl[b & 1];
        const float* gg = PRE ? pre_g[PRE ? b : 0] : gl[b & 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = epi_rowc(r);
            float dz = acc[i][j][r] + gg[r];
            const float uu = fmaf(c.a[j], yy[r], c.b[j]);
            dz *= f.need_y ? act_bwd(uu, sact) : 1.f;
            if (Q) { const float qv = rintf(act_fwd(uu, sact) * qinv) + qzp; dz = (qv >= 0.f && qv <= 255.f) ? dz : 0.f; }
            const bool ok = nok && (F || mb + 4 * g.lh + row < g.M);
            if (F) { if (nok) (sk.g + (size_t)(mb + row) * (size_t)sk.ldg + (g.n0w + 32 * j))[og] = dz; }
            else if (ok) sk.g[(size_t)(mb + 4 * g.lh + row) * sk.ldg + n] = dz;
            if (ok) {
                t1[j] += dz;
                t2[j] = fmaf(dz, (yy[r] - c.m[j]) * c.i[j], t2[j]);
            }
        }
        if (!PRE) __builtin_amdgcn_sched_barrier(0);        // at most two blocks' operands live at a time
    }
}
// dispatch on the two uniform flags (tile inside the matrix, QAT mask)
template <int TI, int TJ, int NPRE>
__device__ __forceinline__ void sink_epilogue(const f32x16 (&acc)[TI][TJ], const pq_gsink& sk, const SinkFlags& f, const SinkCoef<TJ>& c,
                                              const EpiGeom& g, const float (*pre_y)[16], const float (*pre_g)[16],
                                              float (&t1)[TJ], float (&t2)[TJ]) {
    const float qinv = f.qmask ? 1.f / sk.qstate[2] : 0.f, qzp = f.qmask ? sk.qstate[3] : 0.f;
    if (f.qmask) {
        if (g.full) sink_emit<TI, TJ, true, true, NPRE>(acc, sk, f, c, g, pre_y, pre_g, qinv, qzp, t1, t2);
        else sink_emit<TI, TJ, false, true, 0>(acc, sk, f, c, g, pre_y, pre_g, qinv, qzp, t1, t2);
    } else if (g.full) {
        sink_emit<TI, TJ, true, false, NPRE>(acc, sk, f, c, g, pre_y, pre_g, qinv, qzp, t1, t2);
    } else {
        sink_emit<TI, TJ, false, false, 0>(acc, sk, f, c, g, pre_y, pre_g, qinv, qzp, t1, t2);
    }
}

}  // namespace pq
