// Depthwise 3x3 (pad 1, stride 1|2) as LDS-staged NHWC tiles, gfx950.
//
// A workgroup (256 threads) owns an OTH x OTW output tile of one image for a chunk of 32 channels (8 float4 quads).
// The input tile with its halo is staged ONCE into LDS with coalesced 128-byte-per-pixel loads; the producer's
// BatchNorm affine + activation (or, in the backward kernel, the closed-form BN backward  dy = p*g + q*(y-mu) + r) is
// applied at staging time, so every element is loaded from HBM and transformed exactly once.  Threads then slide a
// 3 x (TW*S+2) register window over the LDS tile (lane = 8 quads x 32 pixel lanes).  Pixel pitch in LDS is 9 quads
// (144 B) to spread the 16-lane ds_read_b128 groups over the banks.
//
// Forward only since round 4: the LDS-tiled backward kernels (round 2) were superseded by the walking kernels of dwwalk.hip
// (pq_dw3x3_bwd parts 5 / 6) on every shape and have been removed; planes the walking backward does not take (stride 2 with an odd
// input size) use pq_dw3x3_dgrad + pq_dw3x3_wgrad (dwconv.hip).
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

int pq_dw_fwd_stream(const pq_dw3x3_fwd_args* a, void* stream);     // dwconv.hip
int dw_walk_blocks(int B, int IH, int IW, int C, int stride);       // dwwalk.hip
int dw_walk_launch(const pq_dw3x3_bwd_args* a, void* stream);
int dw_fwd_walk_blocks(int B, int H, int W, int C);
int dw_fwd_walk_launch(const pq_dw3x3_fwd_args* a, void* stream);

// Forward dispatch: stride 1 = the walking kernel (dwwalk.hip) at every plane size; stride 2 = the LDS-tiled kernel below while the
// planes are small (<= 64x64 outputs at batch 16: few, fat workgroups per image), the streaming register-window kernel of
// dwconv.hip on the big early planes (measured with scripts/dw_bench.py).
static bool dw_fwd_use_walk(int stride) { return stride == 1; }
static bool dw_fwd_use_tiled(int OH, int OW, int stride) { (void)stride; return (long)OH * OW <= 64 * 64; }

extern "C" int32_t pq_dw3x3_fwd_blocks(int32_t B, int32_t OH, int32_t OW, int32_t C, int32_t stride) {
    if (dw_fwd_use_walk(stride)) return dw_fwd_walk_blocks(B, OH, OW, C);
    if (!dw_fwd_use_tiled(OH, OW, stride)) return ew_geom((long)B * OH * OW, C).gx;
    int th, tw; fwd_tiles(OH, OW, stride, th, tw);
    return B * th * tw;
}
extern "C" int pq_dw3x3_fwd(const pq_dw3x3_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->in.x && a->w && a->out, "null pointer");
    PQ_CHECK_ARG(dwt_check(a->B, a->IH, a->IW, a->OH, a->OW, a->C, a->stride), "bad shape (need stride 1|2, pad 1)");
    PQ_CHECK_ARG((a->in.ld & 3) == 0 && (a->ldo & 3) == 0, "pitches must be multiples of 4");
    if (act_is_special(a->in.act)) { pq_set_error("pq_dw3x3_fwd: logistic/tanh operands must be materialised first"); return PQ_EUNSUPPORTED; }
    if (dw_fwd_use_walk(a->stride)) {
        PQ_CHECK_ARG(a->stat_acc == nullptr || (a->stat_bad != nullptr && a->minmax == 0 && a->ldp >= a->C), "stat_acc needs stat_bad, no minmax");
        PQ_CHECK_ARG(a->stat_acc == nullptr || stat_adds_ok(pq_dw3x3_fwd_blocks(a->B, a->OH, a->OW, a->C, a->stride), a->stat_slots), "too many row blocks per accumulator slot");
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
    PQ_CHECK_ARG(a->stat_acc == nullptr || stat_adds_ok(pq_dw3x3_fwd_blocks(a->B, a->OH, a->OW, a->C, a->stride), a->stat_slots), "too many row blocks per accumulator slot");
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    fwd_tiles(a->OH, a->OW, a->stride, p.tilesH, p.tilesW);
    dim3 grid(a->B * p.tilesH * p.tilesW, (p.C4 + CBQ - 1) / CBQ), blk(256);
    hipStream_t st = (hipStream_t)stream;
    PQ_LAUNCH((dwt_fwd_kernel<2>), grid, blk, 0, st, p);          // (stride 1 never gets here: walking kernel)
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

// pq_bn_stat_ok for PQ_OP_DW_FWD, both roles: every stride-1 plane (the walking kernel takes them all) and the stride-2 planes
// the LDS-tiled kernel takes (outputs up to 64 x 64); the streaming stride-2 forward of the big planes has neither role
int dw_fwd_stat_ok(const pq_dw3x3_fwd_args* a, int role) {
    (void)role;
    return a && !a->minmax && !a->in.qstate && (dw_fwd_use_walk(a->stride) || dw_fwd_use_tiled(a->OH, a->OW, a->stride));
}

extern "C" int32_t pq_dw3x3_bwd_walk_blocks(int32_t B, int32_t IH, int32_t IW, int32_t C, int32_t stride) {
    if (B <= 0 || IH <= 0 || IW <= 0 || C <= 0 || (stride != 1 && stride != 2)) return 0;
    return dw_walk_blocks(B, IH, IW, C, stride);
}

extern "C" int pq_dw3x3_bwd(const pq_dw3x3_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->part >= 0 && a->part <= 6, "null pointer / bad part");
    if (a->part == 5) PQ_SINK_ADDEND_CHECK(a->dx);
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
            const int nwalk = pq_dw3x3_bwd_walk_blocks(a->B, a->IH, a->IW, a->C, a->stride);
            PQ_SINK_ACC_CHECK(a->dx, a->C, nwalk);
            PQ_CHECK_ARG(nwalk <= 65535, "more than 65535 walkers (grid.y): split the batch");
        }
        const int rc = dw_walk_launch(a, stream);
        if (rc != PQ_OK) return rc;
        PQ_CHECK_LAUNCH();
        return PQ_OK;
    }
    pq_set_error("pq_dw3x3_bwd: parts 0-4 (the LDS-tiled backward of round 2) were removed; use part 5 + 6, or pq_dw3x3_dgrad + pq_dw3x3_wgrad");
    return PQ_EUNSUPPORTED;
}
