// 1x1 convolution GEMMs, second generation: operand tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into
// a ring of S stages, with the loads of D = S-1 k-tiles in flight behind a COUNTED s_waitcnt vmcnt + one raw s_barrier per
// k-tile.  No staging registers, no ds_write, no address masking in the loop; the operand transform (BatchNorm affine +
// activation clamp of the forward operand, the closed-form BatchNorm backward of the gradient pair) is applied to the
// fragments after the ds_read, in the shadow of the 64-cycle v_mfma_f32_32x32x2_f32.  The per-k coefficients of a k-tile
// ride in the same ring (global_load_lds_dword into a 256-byte slot per wave).
//
//   gemm2_nt : C[m][n] = sum_k A'[m][k] * B[n][k]     forward (A' = clamp(scale*x+shift), B = W[O][I])
//                                                      dgrad   (A' = p*g + q*(y-mu) + r,  B = Wt[I][O])
//   gemm2_tn : C[o][i] = sum_m dY'[m][o] * X'[m][i]    wgrad, rows split into slabs (same slab reduce as gemm.hip)
//
// LDS images are lane-linear per LDS-DMA piece (64 lanes x 16 B = 1 KiB): a k-tile row is 32 floats = 8 chunks of 16 B and
// the chunk index is XOR-swizzled with (row >> 1) & 7 on the SOURCE address and again on the fragment read, which makes the
// ds_read_b128 fragment reads (lane = row, lane half = k group) bank-conflict free without padding.
//
// The LDS-DMA instructions are issued from inline asm: hipcc then neither drains vmcnt(0) in front of every ds_read (it does
// for its own builtin, because the DMA "may alias" the LDS array) nor counts them; ordering is by hand:
//   RAW  issuing wave: s_waitcnt vmcnt(N) -> s_barrier -> any wave may ds_read the tile
//   WAR  a stage is refilled only after the barrier that follows its last ds_read (whose data the MFMAs have consumed).
//
// mirrors: nn.Conv2d(k=1) forward / backward inside the conv blocks of the reference, model/parser.py:389-410.
#include "common.h"
#include "gemm_epi.h"
#include "ldsdma.h"

namespace pq {

// Diagnostic build only (python -m pqdet_amd.build -DPQ_STAMPS --tag=stamps; scripts/gemm_stamps.py): s_memtime at four points of
// every workgroup (entry, first k-tile landed, main loop done, epilogue done) + s_memrealtime at both ends, into a buffer the
// kernel never reads.  The shipped library has none of it.
#ifdef PQ_STAMPS
__device__ unsigned long long* pq_stamp_buf = nullptr;
#define PQ_STAMP(i)                                                                                                        \
    do {                                                                                                                   \
        if (threadIdx.x == 0 && pq_stamp_buf)                                                                              \
            pq_stamp_buf[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] =                                        \
                ((i) == 4 || (i) == 5 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime());                              \
    } while (0)
#else
#define PQ_STAMP(i)
#endif

// wait until at most `tiles` k-tiles (NL loads each) of this wave are still in flight; tiles in 0..2
template <int NL>
__device__ __forceinline__ void wait_tiles(int tiles) {
    if (tiles >= 2) wait_vmcnt<2 * NL>();
    else if (tiles == 1) wait_vmcnt<NL>();
    else wait_vmcnt<0>();
}
// B operand of the half k-tile at the end of a contraction that is 16 modulo 32 (see gemm2_nt_kernel)
__device__ __attribute__((aligned(16))) float g2_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int G2_STAT_KMAX = 1280;       // AMODE 3: longest contraction whose coefficient table fits behind the ring
__device__ __forceinline__ f32x16 mfma32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

struct G2P {
    const float* a0; const float* a1;                                    // x | g, y    [M][lda]
    const float* c0; const float* c1; const float* c2; const float* c3;  // scale, shift | p, q, mu, r
    int lda0, lda1;
    float lo, hi;                                                         // clamp of the forward operand (relu6: 0, 6)
    const float* b; int ldb;                                              // [N][ldb], contraction index contiguous
    int M, N, K;
    float* out; const float* bias; float* partial; int ldo, ldp;          // EMODE 0 / 1
    int64_t* stat_acc; uint32_t* stat_bad; int stat_slots;                // EMODE 1: statistics into pq_bn_stat accumulators
    pq_gsink sink;                                                        // EMODE 2
    pq_bn_stat stat;                                                      // AMODE 3: the operand's BatchNorm, derived in the prologue
    pq_bn_bstat bstat;                                                    // AMODE 4: the gradient operand's BatchNorm backward, derived in the prologue
};

// AMODE 0: A as stored; 1: clamp(c0*a + c1, lo, hi); 2: c0*g + c1*(y - c2) + c3; 3: as 1, but (c0, c1) = BatchNorm scale / shift
// DERIVED in the prologue from the producer's statistic accumulators (pq_bn_stat: no finalize launch in between) into an LDS table
// behind the ring -- nothing rides in the ring for them.   EMODE 0: store(+bias); 1: + statistics;
// AMODE 4: as 2, but (c0, c1, c3) = p, q, r of the BatchNorm backward DERIVED in the prologue from the sink kernel's accumulators
// (pq_bn_bstat: no finalize launch in between) into a [4][Kp] LDS table behind the ring together with c2 = mu.
// 2: gradient sink (native-layout dword accesses, gemm_epi.h; the host still requires 16-byte aligned rows for the DMA operands).   K % 32 == 0, or K % 32 == 16 (RegNetX widths 48 / 80 /
// 176 / 240 / 528): the last k-tile is then half valid -- its upper four chunks of B come from g2_zero, those of A and of the
// coefficient vectors repeat the lower four (finite values times zero), so the loop itself is unchanged.
template <int BM, int BN, int AMODE, int EMODE, int S>
__global__ __launch_bounds__(256, 2) void gemm2_nt_kernel(G2P p) {     // two waves per SIMD: at most 256 registers
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32;
    constexpr bool PAIR = AMODE == 2 || AMODE == 4;                      // two-tensor gradient operand (g, y)
    // AMODE 2: the staged pair is turned into dy' = p g + q (y - mu) + r IN PLACE, once per k-tile, by all 256 threads (8 floats
    // each), instead of at every fragment read: a tile row is read by both column waves, and every wave read the four coefficient
    // quads (p, q, mu, r) of a k-step as 1 KiB broadcasts -- per k-tile of a 64-row tile 32 KB of (g, y) + 64 KB of coefficients
    // through the LDS return path against 32 KB of B: the read port, not the matrix pipe, set the k-tile time (round-4 stamps).
    // Now: 24.5 KB for the pass + 16 KB of dy' fragments.  One more barrier per k-tile.
    constexpr bool PREX = AMODE == 2;
    constexpr int NA = PAIR ? 2 : 1;
    constexpr int A_FL = BM * 32, B_FL = BN * 32;
    constexpr int CW = AMODE == 2 ? 128 : (AMODE == 1 ? 64 : 0);        // coefficient floats per wave per stage
    constexpr int NC = AMODE == 2 ? 2 : (AMODE == 1 ? 1 : 0);
    constexpr int STAGE = NA * A_FL + B_FL + 4 * CW;                     // floats
    constexpr int PA = BM / 32, PB = BN / 32;                            // 1 KiB pieces per wave per k-tile
    constexpr int NL = NA * PA + PB + NC;                                // LDS-DMA instructions per wave per k-tile
    constexpr int D = S - 1;
    static_assert(D >= 1 && D <= 3, "ring depth");
    static_assert(2 * NL <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    PQ_STAMP(0); PQ_STAMP(4);
    int bnx, bmy;
    xcd_block(bnx, bmy);
    const int m0 = bmy * BM, n0 = bnx * BN;
    const int M = p.M, N = p.N, K = p.K;
    const int nk = (K + 31) / 32;
    const bool tail = (K & 31) != 0;

    // ---- source pointers of this lane's share of a k-tile (advanced by 32 floats per issued tile)
    const float* pa[NA][PA];
    const float* pb[PB];
    const float* pc[NC > 0 ? NC : 1];
    unsigned hiA = 0, hiB = 0;                    // bit q: this lane's chunk of piece q is one of the upper four of a k-tile
    {
        const int sl = lane & 7;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int r = 8 * (w * PA + q) + (lane >> 3);
            const int c = sl ^ ((r >> 1) & 7);
            if (c >= 4) hiA |= 1u << q;
            const size_t row = (size_t)min(m0 + r, M - 1);
            pa[0][q] = p.a0 + row * (size_t)p.lda0 + 4 * c;
            if (PAIR) pa[NA - 1][q] = p.a1 + row * (size_t)p.lda1 + 4 * c;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int r = 8 * (w * PB + q) + (lane >> 3);
            const int c = sl ^ ((r >> 1) & 7);
            if (c >= 4) hiB |= 1u << q;
            pb[q] = p.b + (size_t)min(n0 + r, N - 1) * (size_t)p.ldb + 4 * c;
        }
        if (AMODE == 1) pc[0] = (lane < 32 ? p.c0 : p.c1) + (lane & 31);
        if (AMODE == 2) { pc[0] = (lane < 32 ? p.c0 : p.c1) + (lane & 31); pc[NC - 1] = (lane < 32 ? p.c2 : p.c3) + (lane & 31); }
    }
    const unsigned lbase = lds_addr(lds);
    auto issue = [&](int s, int t) {              // k-tile t into stage s
        const unsigned sb = lbase + (unsigned)(s * STAGE) * 4u;
        const bool half = tail && t == nk - 1;    // uniform
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                const float* src = pa[a][q] - ((half && ((hiA >> q) & 1u)) ? 16 : 0);
                glds16(src, __builtin_amdgcn_readfirstlane(sb + (unsigned)(a * A_FL + (w * PA + q) * 256) * 4u));
                pa[a][q] += 32;
            }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const float* src = (half && ((hiB >> q) & 1u)) ? g2_zero : pb[q];
            glds16(src, __builtin_amdgcn_readfirstlane(sb + (unsigned)(NA * A_FL + (w * PB + q) * 256) * 4u));
            pb[q] += 32;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* src = pc[c] - ((half && (lane & 16)) ? 16 : 0);
            glds4(src, __builtin_amdgcn_readfirstlane(sb + (unsigned)(NA * A_FL + B_FL + w * CW + 64 * c) * 4u));
            pc[c] += 32;
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = (lrow >> 1) & 7;
    const float lo = p.lo, hi = p.hi;
    const int Kp = nk * 32;
    float* ctab = lds + S * STAGE;                 // AMODE 3: [2][Kp] scale, shift; AMODE 4: [4][Kp] p, q, mu, r
    auto compute = [&](int s, int kt) {
        const float* As = lds + s * STAGE;
        const float* Bs = As + NA * A_FL;
        const float* Cs = Bs + B_FL + w * CW;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((2 * ks + lh) ^ swz) << 2;
            float4 af[TI], yf[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                af[i] = *reinterpret_cast<const float4*>(&As[(wm * WM + 32 * i + lrow) * 32 + ch]);
                if (PAIR && !PREX) yf[i] = *reinterpret_cast<const float4*>(&As[A_FL + (wm * WM + 32 * i + lrow) * 32 + ch]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + 32 * j + lrow) * 32 + ch]);
            float ak[TI][4], bk[TJ][4];
#pragma unroll
            for (int j = 0; j < TJ; ++j) { bk[j][0] = bf[j].x; bk[j][1] = bf[j].y; bk[j][2] = bf[j].z; bk[j][3] = bf[j].w; }
#pragma unroll
            for (int i = 0; i < TI; ++i) { ak[i][0] = af[i].x; ak[i][1] = af[i].y; ak[i][2] = af[i].z; ak[i][3] = af[i].w; }
            if (AMODE == 1 || AMODE == 3) {
                const float4 c0 = *reinterpret_cast<const float4*>(AMODE == 1 ? &Cs[ks * 8 + 4 * lh] : &ctab[kt * 32 + ks * 8 + 4 * lh]);
                const float4 c1 = *reinterpret_cast<const float4*>(AMODE == 1 ? &Cs[32 + ks * 8 + 4 * lh] : &ctab[Kp + kt * 32 + ks * 8 + 4 * lh]);
                const float s4[4] = {c0.x, c0.y, c0.z, c0.w}, h4[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ak[i][q] = __builtin_amdgcn_fmed3f(fmaf(s4[q], ak[i][q], h4[q]), lo, hi);
            } else if (PAIR && !PREX) {
                const float* T4 = &ctab[kt * 32 + ks * 8 + 4 * lh];
                const float4 c0 = *reinterpret_cast<const float4*>(AMODE == 2 ? &Cs[ks * 8 + 4 * lh] : T4);
                const float4 c1 = *reinterpret_cast<const float4*>(AMODE == 2 ? &Cs[32 + ks * 8 + 4 * lh] : T4 + Kp);
                const float4 c2 = *reinterpret_cast<const float4*>(AMODE == 2 ? &Cs[64 + ks * 8 + 4 * lh] : T4 + 2 * Kp);
                const float4 c3 = *reinterpret_cast<const float4*>(AMODE == 2 ? &Cs[96 + ks * 8 + 4 * lh] : T4 + 3 * Kp);
                const float p4[4] = {c0.x, c0.y, c0.z, c0.w}, q4[4] = {c1.x, c1.y, c1.z, c1.w};
                const float m4[4] = {c2.x, c2.y, c2.z, c2.w}, r4[4] = {c3.x, c3.y, c3.z, c3.w};
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const float yy[4] = {yf[i].x, yf[i].y, yf[i].z, yf[i].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) ak[i][q] = fmaf(p4[q], ak[i][q], fmaf(q4[q], yy[q] - m4[q], r4[q]));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32x2(ak[i][q], bk[j][q], acc[i][j]);
        }
    };

    // the in-place pass of AMODE 2: thread -> (chunk slot t & 7, rows t / 8 + 32 i); the slot of a row holds k-quad slot ^ swizzle(row),
    // the same quad for all rows of a thread (32 rows apart), so its four coefficient quads are read once per k-tile
    auto xform = [&](int s) {
        float* As = lds + s * STAGE;
        const float* Cw = As + NA * A_FL + B_FL + w * CW;
        const int sl = t & 7, r0 = t >> 3;
        const int c4 = (sl ^ ((r0 >> 1) & 7)) << 2;
        const float4 cp = *reinterpret_cast<const float4*>(&Cw[c4]), cq = *reinterpret_cast<const float4*>(&Cw[32 + c4]);
        const float4 cm = *reinterpret_cast<const float4*>(&Cw[64 + c4]), cr = *reinterpret_cast<const float4*>(&Cw[96 + c4]);
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            float* gp = &As[(r0 + 32 * i) * 32 + 4 * sl];
            const float4 g4 = *reinterpret_cast<const float4*>(gp), y4 = *reinterpret_cast<const float4*>(gp + A_FL);
            *reinterpret_cast<float4*>(gp) = make_float4(fmaf(cp.x, g4.x, fmaf(cq.x, y4.x - cm.x, cr.x)), fmaf(cp.y, g4.y, fmaf(cq.y, y4.y - cm.y, cr.y)),
                                                         fmaf(cp.z, g4.z, fmaf(cq.z, y4.z - cm.z, cr.z)), fmaf(cp.w, g4.w, fmaf(cq.w, y4.w - cm.w, cr.w)));
        }
    };
    // ---- main loop: ring of S stages, D tiles in flight
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nk) issue(d, d);
    if (AMODE == 3) {
        // BatchNorm coefficients of the operand from the producer's accumulators, while the first tiles are in flight; logical
        // workgroup (0, 0) also stores the vectors / running statistics for every later consumer (pq_bn_stat)
        const bool writer = bmy == 0 && bnx == 0;
        for (int k = K + t; k < Kp; k += 256) { ctab[k] = 0.f; ctab[Kp + k] = 0.f; }
        auto put = [&](int c, const BnCoef& bc) { ctab[c] = bc.scale; ctab[Kp + c] = bc.shift; };
        // all of a thread's accumulator loads in flight at once: one memory round trip for K <= 1280 (one slot) / 512 (four)
        if (p.stat.slots == 1) bn_stat_coefs<5>(p.stat, t, 256, K, writer, put);
        else for (int k0 = t; k0 < K; k0 += 512) bn_stat_coefs<2>(p.stat, k0, 256, K, writer, put);
        __syncthreads();
    }
    if (AMODE == 4) {
        // the same for the BatchNorm BACKWARD of the gradient operand (pq_bn_bstat): the sums the sink kernel added -> p, q, r per
        // contraction channel; logical workgroup (0, 0) also stores them with dgamma / dbeta
        const bool writer = bmy == 0 && bnx == 0;
        for (int k = K + t; k < Kp; k += 256) { ctab[k] = 0.f; ctab[Kp + k] = 0.f; ctab[2 * Kp + k] = 0.f; ctab[3 * Kp + k] = 0.f; }
        auto put = [&](int c, const GsrcCoef1& k1) { ctab[c] = k1.p; ctab[Kp + c] = k1.q; ctab[2 * Kp + c] = k1.mu; ctab[3 * Kp + c] = k1.r; };
        if (p.bstat.slots == 1) bn_bstat_coefs<5>(p.bstat, p.c2, t, 256, K, writer, put);
        else for (int k0 = t; k0 < K; k0 += 512) bn_bstat_coefs<2>(p.bstat, p.c2, k0, 256, K, writer, put);
        __syncthreads();
    }
    // ---- main loop; the last k-tile is peeled so that the loads of the epilogue can be requested under its MFMAs
    for (int kt = 0; kt < nk - 1; ++kt) {
        const int issued = min(nk, kt + D);
        wait_tiles<NL>(issued - kt - 1);
        __builtin_amdgcn_s_barrier();
#ifdef PQ_STAMPS
        if (kt == 0) PQ_STAMP(1);
#endif
        if (PREX) { xform(kt % S); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // (raw barrier: the LDS writes are drained by hand)
        if (kt + D < nk) issue((kt + D) % S, kt + D);
        compute(kt % S, kt);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (PREX) { xform((nk - 1) % S); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#ifdef PQ_STAMPS
    if (nk == 1) PQ_STAMP(1);
#endif

    // ---------------------------------------------------------------- epilogue (gemm_epi.h: native accumulator layout, no LDS
    // transpose, branch-free path for tiles inside the matrix).  EMODE 2, one or two blocks per wave: the gradient sink's operands
    // (the layer input y for the activation derivative and the BatchNorm-backward sums, the running gradient of an accumulating
    // sink) and its coefficients are requested HERE, in front of the last k-tile's MFMAs, so that their HBM round trip does not
    // open the epilogue -- where every co-resident workgroup of a single-round grid would wait for the same burst at once.
    constexpr int NB = TI * TJ;
#ifdef PQ_G2_NOHOIST
    constexpr bool HOIST = false;
#else
    constexpr bool HOIST = EMODE == 2 && NB <= 2;
#endif
    EpiGeom eg;
    eg.m0w = m0 + __builtin_amdgcn_readfirstlane(wm) * WM; eg.n0w = n0 + __builtin_amdgcn_readfirstlane(wn) * WN;
    eg.lrow = lrow; eg.lh = lh; eg.M = M; eg.N = N;
    eg.full = m0 + BM <= M;
    const pq_gsink& sk = p.sink;
    SinkFlags sf{};
    SinkCoef<TJ> sc;
    float yv[HOIST ? NB : 1][16], gv[HOIST ? NB : 1][16];
    if (EMODE == 2) {
        sf = sink_flags(sk);
        sink_coefs<TJ>(sk, sf, eg, sc);
        if (HOIST && eg.full) {                    // (edge tiles load in the epilogue: their clamped per-lane addresses are not worth keeping)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int i = 0; i < TI; ++i) sink_block_loads<true>(sk, sf, eg, i, j, yv[HOIST ? j * TI + i : 0], gv[HOIST ? j * TI + i : 0]);
        }
    }
    compute((nk - 1) % S, nk - 1);
    __builtin_amdgcn_s_barrier();                 // every wave is done reading the ring: it becomes epilogue scratch
    PQ_STAMP(2);

    float* red = lds;                              // [2][2][BN]
    bool want_stats = false;
    float t1[TJ], t2[TJ], mn_[TJ], mx_[TJ];        // per-column sums of this lane
#pragma unroll
    for (int j = 0; j < TJ; ++j) t1[j] = t2[j] = mn_[j] = mx_[j] = 0.f;
    if (EMODE == 2) {
        want_stats = sk.finalize && sf.stats;
        sink_epilogue<TI, TJ, HOIST ? NB : 0>(acc, sk, sf, sc, eg, yv, gv, t1, t2);
    } else {
        want_stats = (EMODE == 1) && (p.partial != nullptr || p.stat_acc != nullptr);
        if (eg.full) epi_store<TI, TJ, EMODE, true>(acc, p.out, p.ldo, p.bias, eg, t1, t2, mn_, mx_);
        else epi_store<TI, TJ, EMODE, false>(acc, p.out, p.ldo, p.bias, eg, t1, t2, mn_, mx_);
    }
    if (want_stats) {   // uniform across the block
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            t1[j] += __shfl_xor(t1[j], 32);
            t2[j] += __shfl_xor(t2[j], 32);
            if (lh == 0) {
                red[(0 * 2 + wm) * BN + wn * WN + 32 * j + lrow] = t1[j];
                red[(1 * 2 + wm) * BN + wn * WN + 32 * j + lrow] = t2[j];
            }
        }
        PQ_STAMP(6);
        float* part = (EMODE == 2) ? p.sink.partial : p.partial;
        const int ldp = (EMODE == 2) ? p.sink.ldp : p.ldp;
        __syncthreads();
        if (t < BN) {
            const float a = red[(0 * 2 + 0) * BN + t] + red[(0 * 2 + 1) * BN + t];
            const float b = red[(1 * 2 + 0) * BN + t] + red[(1 * 2 + 1) * BN + t];
            const int n = n0 + t;
            if (n < N) {
                if ((EMODE == 1 || EMODE == 2) && p.stat_acc != nullptr) {       // no finalize launch: the consumer derives the coefficients
                    bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, ldp, bmy, 0, n, a);
                    bn_stat_add(p.stat_acc, p.stat_bad, p.stat_slots, ldp, bmy, 1, n, b);
                } else {
                    float* dst = part + (size_t)bmy * 2 * ldp;
                    dst[n] = a;
                    dst[ldp + n] = b;
                }
            }
        }
    }
#ifdef PQ_STAMPS
    __syncthreads();
    PQ_STAMP(3); PQ_STAMP(5);
#endif
}

// ------------------------------------------------------------------------------------------------ wgrad (TN)
// C[o][i] = sum_m dY'[m][o] * X'[m][i] on a 128 x 128 output tile, 8 waves.  Both operands are row-major [m][channels], i.e.
// the contraction index is the SLOW one: an MFMA operand register wants 32 consecutive channels of one row.  Read like that
// (ds_read_b32) every MFMA needs two fresh LDS instructions and the LDS issue port, not the matrix pipe, sets the pace (78 TF
// asymptote measured on gemm_tn and on a b32 ring kernel alike).  Here a lane OWNS adjacent channels instead -- 2 of dY'
// (one ds_read_b64) and 4 of X' (one ds_read_b128) -- so one row pair feeds 2 x 4 = 8 MFMAs from three LDS reads; the MFMA
// "row" index is just a label, so the C fragment of MFMA (a, b) holds o = 2*row + a, i = 4*col + b.  The 8 waves split the tile
// as 2 halves of o x 4 row groups of the 32-row k-tile (each element of a staged tile is read from LDS by exactly one
// wave); the four row groups add their 64 x 128 accumulators through LDS in fixed order at the end (deterministic).
struct G2TP {
    const float* g; const float* y; const float* p; const float* q; const float* mu; const float* r; int ldg, ldy;
    const float* x; const float* scale; const float* shift; int ldx; float lo, hi;
    float* slab; size_t slab_stride;
    int M, O, I, rows_per_split;
    pq_bn_bstat bstat;                                    // acc != NULL: p / q / r derived per lane from the sink kernel's accumulators
};

// TWO: dy' = p*g + q*(y-mu) + r (else dy' = g).  XA: x' = clamp(scale*x + shift, lo, hi) (else x' = x).
// rows_per_split % 32 == 0 and M % 32 == 0.
template <bool TWO, bool XA, int S, int NW>
__global__ __launch_bounds__(NW * 64) void gemm2_tn_kernel(G2TP p) {
    constexpr int RG = NW / 2;                            // row groups of a 32-row k-tile (the other factor 2: halves of o)
    constexpr int KK = 16 / RG;                           // row pairs per row group and k-tile
    constexpr int PPW = 16 / NW;                          // 1 KiB pieces per tensor, wave and k-tile
    constexpr int BO = 128, BI = 128;
    constexpr int NY = TWO ? 2 : 1;
    constexpr int Y_FL = 32 * BO, X_FL = 32 * BI;
    constexpr int STAGE = NY * Y_FL + X_FL;
    constexpr int NL = PPW * (NY + 1);                    // LDS-DMA instructions per wave per k-tile
    constexpr int D = S - 1;
    static_assert(2 * NL <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, og = w & 1, rg = w >> 1;
    const int lrow = lane & 31, lh = lane >> 5;
    const int tiles_i = (p.I + BI - 1) / BI;
    int btile, split;
    xcd_block(btile, split);
    const int to = btile / tiles_i, ti = btile % tiles_i;
    const int o0 = to * BO, i0 = ti * BI;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nk = (mend - mbeg) / 32;
    const int Opad = (p.O + 3) & ~3, Ipad = (p.I + 3) & ~3;

    // ---- source pointers: piece q of a [32][128] tile image covers two rows (256 floats)
    const float* py[NY][PPW];
    const float* px[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int f = (w * PPW + q) * 256 + lane * 4;
        const int r = f / BO, co = min(o0 + f % BO, Opad - 4), ci = min(i0 + f % BI, Ipad - 4);
        py[0][q] = p.g + (size_t)(mbeg + r) * (size_t)p.ldg + co;
        if (TWO) py[NY - 1][q] = p.y + (size_t)(mbeg + r) * (size_t)p.ldy + co;
        px[q] = p.x + (size_t)(mbeg + r) * (size_t)p.ldx + ci;
    }
    const size_t sg = (size_t)32 * p.ldg, sy = (size_t)32 * p.ldy, sx = (size_t)32 * p.ldx;
    const unsigned lbase = lds_addr(lds);
    auto issue = [&](int s) {
        const unsigned sb = lbase + (unsigned)(s * STAGE) * 4u;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            glds16(py[0][q], __builtin_amdgcn_readfirstlane(sb + (unsigned)((w * PPW + q) * 256) * 4u));
            py[0][q] += sg;
            if (TWO) {
                glds16(py[NY - 1][q], __builtin_amdgcn_readfirstlane(sb + (unsigned)(Y_FL + (w * PPW + q) * 256) * 4u));
                py[NY - 1][q] += sy;
            }
            glds16(px[q], __builtin_amdgcn_readfirstlane(sb + (unsigned)(NY * Y_FL + (w * PPW + q) * 256) * 4u));
            px[q] += sx;
        }
    };

    // prologue tiles first, then this lane's per-column coefficients (they never change: registers).  These are the only
    // compiler-visible global loads of the kernel; the empty asm below makes the compiler wait for them HERE, once (it
    // drains the prologue tiles with them, which the first k-tile needs anyway) -- left alone it puts its vmcnt(0) at their
    // first use inside the k-loop, where it would drain the LDS-DMA ring on every iteration.
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nk) issue(d);
    float cp[2], cq[2], cm[2], cr[2], xs[4], xh[4];
    if (TWO && p.bstat.acc != nullptr) {                  // uniform: no finalize launch in front -- both channels of the lane in one round trip
        const int oa = min(o0 + og * 64 + 2 * lrow, p.O - 1);
        // (never the writer of the vectors: the data-gradient kernel of the layer is -- the engine hands over only with one)
        bn_bstat_coefs<2>(p.bstat, p.mu, oa, 1, p.O, false,
                          [&](int c, const GsrcCoef1& k1) { const int a = c - oa; cp[a] = k1.p; cq[a] = k1.q; cm[a] = k1.mu; cr[a] = k1.r; });
        if (oa + 1 >= p.O) { cp[1] = cp[0]; cq[1] = cq[0]; cm[1] = cm[0]; cr[1] = cr[0]; }          // (clamped duplicate: its rows are never stored)
    } else
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int o = min(o0 + og * 64 + 2 * lrow + a, p.O - 1);
        cp[a] = TWO ? p.p[o] : 1.f; cq[a] = TWO ? p.q[o] : 0.f; cm[a] = TWO ? p.mu[o] : 0.f; cr[a] = TWO ? p.r[o] : 0.f;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int i = min(i0 + 4 * lrow + b, p.I - 1);
        xs[b] = (XA && p.scale) ? p.scale[i] : 1.f; xh[b] = (XA && p.scale) ? p.shift[i] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) asm volatile("" : "+v"(cp[a]), "+v"(cq[a]), "+v"(cm[a]), "+v"(cr[a]));
#pragma unroll
    for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(xs[b]), "+v"(xh[b]));
    const float lo = p.lo, hi = p.hi;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int s) {
        const float* Gs = lds + s * STAGE + (2 * KK * rg + lh) * BO + og * 64 + 2 * lrow;
        const float* Xs = lds + s * STAGE + NY * Y_FL + (2 * KK * rg + lh) * BI + 4 * lrow;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float2 g2 = *reinterpret_cast<const float2*>(&Gs[2 * kk * BO]);
            float av[2] = {g2.x, g2.y};
            if (TWO) {
                const float2 y2 = *reinterpret_cast<const float2*>(&Gs[Y_FL + 2 * kk * BO]);
                av[0] = fmaf(cp[0], av[0], fmaf(cq[0], y2.x - cm[0], cr[0]));
                av[1] = fmaf(cp[1], av[1], fmaf(cq[1], y2.y - cm[1], cr[1]));
            }
            const float4 x4 = *reinterpret_cast<const float4*>(&Xs[2 * kk * BI]);
            float bv[4] = {x4.x, x4.y, x4.z, x4.w};
            if (XA) {
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[b] = __builtin_amdgcn_fmed3f(fmaf(xs[b], bv[b], xh[b]), lo, hi);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mfma32x2(av[a], bv[b], acc[a][b]);
        }
    };

    for (int kt = 0; kt < nk; ++kt) {
        const int issued = min(nk, kt + D);
        wait_tiles<NL>(issued - kt - 1);
        __builtin_amdgcn_s_barrier();
        if (kt + D < nk) issue((kt + D) % S);
        compute(kt % S);
    }

    // ---- the four row groups add up through one 128 x 128 LDS tile, in fixed order
    float* T = lds;                                        // 64 KiB of the ring
#pragma unroll 1
    for (int round = 0; round < RG; ++round) {
        __syncthreads();                                   // round 0: every wave is done reading the ring
        if (rg == round) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = og * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + a;
                    float4* dst = reinterpret_cast<float4*>(&T[o * BI + 4 * lrow]);
                    float4 v = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                    if (round > 0) v = f4add(v, *dst);
                    *dst = v;
                }
        }
    }
    __syncthreads();
    float* dst = p.slab + (size_t)split * p.slab_stride;
    const bool vec = (p.I & 3) == 0 && (((uintptr_t)dst) & 15) == 0;
#pragma unroll
    for (int u = 0; u < 4096 / (NW * 64); ++u) {
        const int idx = t + NW * 64 * u;
        const int ro = idx >> 5, c4 = (idx & 31) * 4;
        const int o = o0 + ro, i = i0 + c4;
        if (o < p.O && i < p.I) {
            const float4 v = *reinterpret_cast<const float4*>(&T[ro * BI + c4]);
            float* d = dst + (size_t)o * p.I + i;
            if (vec && i + 3 < p.I) stg4(d, v);
            else { const float e[4] = {v.x, v.y, v.z, v.w}; for (int b = 0; b < 4 && i + b < p.I; ++b) d[b] = e[b]; }
        }
    }
}

}  // namespace pq

using namespace pq;

// ------------------------------------------------------------------------------------------------ host side
#ifdef PQ_STAMPS
extern "C" int pq_debug_set_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pq::pq_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
static bool g2_enabled() { return true; }
static bool g2_raise_lds(const void* fn, size_t bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

template <int BM, int BN, int AMODE, int EMODE, int S>
static int g2_launch_nt(const G2P& p, hipStream_t st, pq_gemm_plan* plan) {
    if (plan) { *plan = pq_gemm_plan{2, BM, BN, S, AMODE, EMODE, 0, 0}; return PQ_OK; }      // query only (pq_conv1x1_*_plan)
    constexpr int NA = (AMODE == 2 || AMODE == 4) ? 2 : 1;
    constexpr int CW = AMODE == 2 ? 128 : (AMODE == 1 ? 64 : 0);
    constexpr int STAGE = NA * BM * 32 + BN * 32 + 4 * CW;
    constexpr size_t ring0 = (size_t)S * STAGE * 4, epi = (size_t)(4 * BN) * 4;
    constexpr size_t ring_max = ring0 + (AMODE == 3 ? (size_t)2 * G2_STAT_KMAX * 4 : (AMODE == 4 ? (size_t)4 * G2_STAT_KMAX * 4 : 0));
    constexpr size_t bytes_max = ring_max > epi ? ring_max : epi;
    static bool ok = g2_raise_lds((const void*)gemm2_nt_kernel<BM, BN, AMODE, EMODE, S>, bytes_max);
    if (!ok) { pq_set_error("gemm2: cannot raise the dynamic LDS limit"); return PQ_ELAUNCH; }
    // AMODE 3: the coefficient table behind the ring is sized by THIS contraction (a fixed 10 KB pushed the 128 x 64 x 3-stage ring
    // over 80 KB, i.e. down to one workgroup per CU: +13 us on the K = 512 layers)
    const size_t ring = ring0 + (AMODE == 3 ? (size_t)2 * ((p.K + 31) / 32 * 32) * 4 : (AMODE == 4 ? (size_t)4 * ((p.K + 31) / 32 * 32) * 4 : 0));
    const size_t bytes = ring > epi ? ring : epi;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    // (BINDABLE: the fork event of a side-stream op behind this call is bound to THIS dispatch -- valid only while the op launches
    // nothing after it: pq_conv1x1_fwd / _dgrad are single-kernel ops; a trailing kernel added to them must take the event instead)
    PQ_LAUNCH_BINDABLE((gemm2_nt_kernel<BM, BN, AMODE, EMODE, S>), grid, dim3(256), bytes, st, p);
    return PQ_OK;
}

// Column-tile width for a row-block height the caller has fixed (pq_conv1x1_blocks decides BM from (M, N) alone because the
// statistics rows are sized from it): 128-wide tiles while they still give every CU two workgroups.
static int g2_pick_bn(int M, int N, int bm) {
    if (N <= 64) return 64;
    const long tiles128 = (long)((M + bm - 1) / bm) * ((N + 127) / 128);
    return tiles128 >= 512 ? 128 : 64;
}

template <int AMODE, int EMODE>
static int g2_dispatch_nt(const G2P& p, int bm, hipStream_t st, pq_gemm_plan* plan) {
    const int bn = g2_pick_bn(p.M, p.N, bm);
    // ring depth by LDS budget: two workgroups per CU (<= 80 KB each) -- except AMODE 4 (coefficients derived from accumulators, behind
    // PQ_BN_BWD_ATOMIC=all): its 128 x 64 x 3 ring takes 120 KB + up to 20 KB of table, i.e. ONE workgroup per CU, part of why that
    // scope measured slower (profiles/r4_bn_bwd_handover_scope.txt)
    // Ring depth TWO everywhere (round 5).  Three stages are 1-3 % faster per launch ALONE, but a 64 x 64 pair-staging workgroup then
    // holds 80 KB: two of them own a CU's whole LDS and no weight-gradient workgroup of the side stream can become resident beside
    // them (nor can a second main workgroup start on a CU where a weight-gradient workgroup sits).  With <= 54 KB per workgroup the
    // two streams interleave on every CU: 8.65 -> 8.55 ms per mobilenetv2-fpn step (-DPQ_G2_RING3 restores the old depth for A/B).
#ifdef PQ_G2_RING3
    constexpr int R3 = 3;
#else
    constexpr int R3 = 2;
#endif
    if constexpr (EMODE == 2) {
        // gradient sink: tiles of at most two 32 x 32 blocks per wave (the sink's operands of a whole tile are prefetched into
        // registers; a 128 x 128 sink tile would need 288 of them: one wave per SIMD)
        if (bm == 128) return g2_launch_nt<128, 64, AMODE, EMODE, R3>(p, st, plan);
        if (AMODE == 2 || AMODE == 4) return bn == 128 ? g2_launch_nt<64, 128, AMODE, EMODE, 2>(p, st, plan) : g2_launch_nt<64, 64, AMODE, EMODE, R3>(p, st, plan);
        return bn == 128 ? g2_launch_nt<64, 128, AMODE, EMODE, R3>(p, st, plan) : g2_launch_nt<64, 64, AMODE, EMODE, R3>(p, st, plan);
    } else {
        if (bm == 128) return bn == 128 ? g2_launch_nt<128, 128, AMODE, EMODE, 2>(p, st, plan) : g2_launch_nt<128, 64, AMODE, EMODE, R3>(p, st, plan);
        return bn == 128 ? g2_launch_nt<64, 128, AMODE, EMODE, R3>(p, st, plan) : g2_launch_nt<64, 64, AMODE, EMODE, R3>(p, st, plan);
    }
}

static bool act_clamp(int act, float& lo, float& hi) {
    const float inf = __builtin_huge_valf();
    if (act == PQ_ACT_LINEAR) { lo = -inf; hi = inf; return true; }
    if (act == PQ_ACT_RELU) { lo = 0.f; hi = inf; return true; }
    if (act == PQ_ACT_RELU6) { lo = 0.f; hi = 6.f; return true; }
    return false;
}

// returns 1 when the launch was taken, 0 when the shape / operands are outside this kernel family (caller falls back),
// < 0 on error
int g2_conv1x1_fwd(const pq_conv1x1_fwd_args* a, int bm, hipStream_t st, pq_gemm_plan* plan) {
    if (!g2_enabled() || a->sub > 1 || a->in.qstate || (a->I & 15) || a->I < 128 || a->O < 64) return 0;
    if ((a->in.ld & 3) || (((uintptr_t)a->in.x) & 15) || (((uintptr_t)a->w) & 15)) return 0;
    G2P p{};
    if (!act_clamp(a->in.act, p.lo, p.hi)) return 0;
    const bool from_stat = a->in.stat.acc != nullptr;
    if (from_stat && a->I > G2_STAT_KMAX) return 0;
    const bool xform = a->in.scale != nullptr || from_stat;
    if (!xform && a->in.act != PQ_ACT_LINEAR) return 0;
    p.a0 = a->in.x; p.lda0 = a->in.ld; p.c0 = a->in.scale; p.c1 = a->in.shift;
    p.b = a->w; p.ldb = a->I; p.M = a->M; p.N = a->O; p.K = a->I;
    p.out = a->out; p.bias = a->bias; p.partial = a->partial; p.ldo = a->ldo; p.ldp = a->ldp;
    p.stat_acc = a->stat_acc; p.stat_bad = a->stat_bad; p.stat_slots = a->stat_slots > 0 ? a->stat_slots : 1;
    p.stat = a->in.stat;
    const bool stats = a->partial != nullptr || a->stat_acc != nullptr;
    int rc;
    if (from_stat) rc = stats ? g2_dispatch_nt<3, 1>(p, bm, st, plan) : g2_dispatch_nt<3, 0>(p, bm, st, plan);
    else if (xform) rc = stats ? g2_dispatch_nt<1, 1>(p, bm, st, plan) : g2_dispatch_nt<1, 0>(p, bm, st, plan);
    else rc = stats ? g2_dispatch_nt<0, 1>(p, bm, st, plan) : g2_dispatch_nt<0, 0>(p, bm, st, plan);
    return rc == PQ_OK ? 1 : rc;
}

int g2_conv1x1_dgrad(const pq_conv1x1_dgrad_args* a, int bm, hipStream_t st, pq_gemm_plan* plan) {
    if (!g2_enabled() || a->sub > 1 || !a->wt || (a->O & 15) || a->O < 64 || a->I < 64) return 0;
    const pq_gsink& sk = a->dx;
    const bool stats = sk.mean != nullptr && sk.partial != nullptr;
    const bool need_y = sk.finalize && (sk.act != PQ_ACT_LINEAR || stats || sk.qstate != nullptr);
    if ((a->I & 3) || (sk.ldg & 3) || (((uintptr_t)sk.g) & 15) || (need_y && ((sk.ldy & 3) || (((uintptr_t)sk.y) & 15)))) return 0;
    if ((((uintptr_t)a->dy.g) & 15) || (a->dy.y && (((uintptr_t)a->dy.y) & 15)) || (((uintptr_t)a->wt) & 15)) return 0;
    G2P p{};
    p.a0 = a->dy.g; p.a1 = a->dy.y; p.lda0 = a->dy.ldg; p.lda1 = a->dy.ldy;
    p.c0 = a->dy.p; p.c1 = a->dy.q; p.c2 = a->dy.mu; p.c3 = a->dy.r;
    p.b = a->wt; p.ldb = a->O; p.M = a->M; p.N = a->I; p.K = a->O;      // O % 16 == 0: the transposed rows have no pad
    p.sink = a->dx;
    if (!p.sink.addend) p.sink.addend = p.sink.g;      // gemm_epi.h reads the running gradient through `addend`
    p.stat_acc = a->dx.stat_acc; p.stat_bad = a->dx.stat_bad; p.stat_slots = a->dx.stat_slots > 0 ? a->dx.stat_slots : 1;   // (checked by the caller)
    const bool from_bstat = a->dy_stat.acc != nullptr;       // (the caller checked the record)
    if (from_bstat && (!a->dy.y || a->O > G2_STAT_KMAX)) return 0;
    p.bstat = a->dy_stat;
    if (p.bstat.slots < 1) p.bstat.slots = 1;
    const int rc = from_bstat ? g2_dispatch_nt<4, 2>(p, bm, st, plan)
                              : (a->dy.y ? g2_dispatch_nt<2, 2>(p, bm, st, plan) : g2_dispatch_nt<0, 2>(p, bm, st, plan));
    return rc == PQ_OK ? 1 : rc;
}

// ---- wgrad
struct G2WPlan { bool ok; int tiles, splits, rows; };
static G2WPlan g2_wgrad_plan(int M, int I, int O) {
    G2WPlan pl{};
    // 128 x 128 tiles: shapes that would leave more than a third of a tile empty stay on the 64 x 64 gemm_tn kernel
    static const char* tn_env = getenv("PQ_G2_TN");                 // "0": every weight gradient on the 64 x 64 register-staged kernel (A/B)
    pl.ok = g2_enabled() && !(tn_env && tn_env[0] == '0') && (M % 32) == 0 && (I & 3) == 0 && (O & 3) == 0 &&
            ((O - 1) % 128) >= 80 && ((I - 1) % 128) >= 80;
    if (!pl.ok) return pl;
    pl.tiles = ((O + 127) / 128) * ((I + 127) / 128);
    // one workgroup per CU (96 KiB of LDS, 272 registers): tiles x splits <= 256 in ONE round, and every workgroup must get a contraction
    // of at least 256 rows (8 k-tiles) or the fixed costs (ring fill, row-group reduction, slab) dominate: shorter ones stay
    // on gemm_tn
    // (re-measured in round 3 with the main chain as the limiter: 128 workgroups, half the CUs, 10.08-10.11 ms against 10.09-10.11;
    // 96 / 64: 10.2-10.26 / 10.41-10.46)
    static const char* tg_env = getenv("PQ_G2_TN_TARGET");          // A/B only
    const int target = tg_env ? atoi(tg_env) : 256;
    int splits = target / pl.tiles;
    const int maxs = M / 256;
    if (splits > maxs) splits = maxs;
    if (splits < 1 || (long)pl.tiles * splits < 200) { pl.ok = false; return pl; }
    pl.rows = (((M + splits - 1) / splits) + 31) / 32 * 32;
    pl.splits = (M + pl.rows - 1) / pl.rows;
    return pl;
}
int g2_wgrad_splits(int M, int I, int O) { const G2WPlan pl = g2_wgrad_plan(M, I, O); return pl.ok ? pl.splits : 0; }

template <bool TWO, bool XA, int NW>
static int g2_launch_tn_nw(const G2TP& p, const G2WPlan& pl, hipStream_t st, pq_gemm_plan* plan) {
    // two stages, 96 KiB (round 5): with three (144 KiB) a weight-gradient workgroup owned its CU -- no main-stream workgroup could
    // start there for its whole ~80 us life; with two, one ring-GEMM / depthwise workgroup fits beside it: 8.55 -> 8.45 ms per step
#ifdef PQ_G2_TN3
    constexpr int S = 3;
#else
    constexpr int S = 2;
#endif
    if (plan) { *plan = pq_gemm_plan{2, 128, 128, S, (TWO ? 1 : 0) | (XA ? 2 : 0), 0, pl.splits, NW}; return PQ_OK; }
    constexpr size_t ring_b = (size_t)S * ((TWO ? 2 : 1) * 32 * 128 + 32 * 128) * 4;        // 96 / 64 KiB
    constexpr size_t bytes = ring_b > 65536 ? ring_b : 65536;                                // >= the 64 KiB epilogue tile
    static bool ok = g2_raise_lds((const void*)gemm2_tn_kernel<TWO, XA, S, NW>, bytes);
    if (!ok) { pq_set_error("gemm2: cannot raise the dynamic LDS limit"); return PQ_ELAUNCH; }
    PQ_LAUNCH((gemm2_tn_kernel<TWO, XA, S, NW>), dim3(pl.tiles, pl.splits), dim3(NW * 64), bytes, st, p);
    return PQ_OK;
}
template <bool TWO, bool XA>
static int g2_launch_tn(const G2TP& p, const G2WPlan& pl, hipStream_t st, pq_gemm_plan* plan) {
    return g2_launch_tn_nw<TWO, XA, 4>(p, pl, st, plan);      // (8 waves: 1 % faster alone, slower in the step -- DESIGN 5.1)
}

// the GEMM half of pq_conv1x1_wgrad (the caller runs the slab reduction); 1 = taken, 0 = not eligible
int g2_conv1x1_wgrad(const pq_conv1x1_wgrad_args* a, int* splits_out, size_t* stride_out, hipStream_t st, pq_gemm_plan* plan) {
    if (a->sub > 1 || a->dbias || a->in.qstate) return 0;
    const G2WPlan pl = g2_wgrad_plan(a->M, a->I, a->O);
    if (!pl.ok) return 0;
    if ((a->dy.ldg & 3) || (a->in.ld & 3) || (a->dy.y && (a->dy.ldy & 3))) return 0;
    if ((((uintptr_t)a->dy.g) & 15) || (a->dy.y && (((uintptr_t)a->dy.y) & 15)) || (((uintptr_t)a->in.x) & 15)) return 0;
    G2TP p{};
    if (!act_clamp(a->in.act, p.lo, p.hi)) return 0;
    const bool xa = a->in.scale != nullptr || a->in.act != PQ_ACT_LINEAR;
    p.g = a->dy.g; p.y = a->dy.y; p.p = a->dy.p; p.q = a->dy.q; p.mu = a->dy.mu; p.r = a->dy.r; p.ldg = a->dy.ldg; p.ldy = a->dy.ldy;
    p.x = a->in.x; p.scale = a->in.scale; p.shift = a->in.shift; p.ldx = a->in.ld;
    p.slab = a->scratch; p.slab_stride = (size_t)a->O * a->I + a->O;
    p.M = a->M; p.O = a->O; p.I = a->I; p.rows_per_split = pl.rows;
    p.bstat = a->dy_stat;
    if (p.bstat.slots < 1) p.bstat.slots = 1;
    int rc;
    if (a->dy.y) rc = xa ? g2_launch_tn<true, true>(p, pl, st, plan) : g2_launch_tn<true, false>(p, pl, st, plan);
    else rc = xa ? g2_launch_tn<false, true>(p, pl, st, plan) : g2_launch_tn<false, false>(p, pl, st, plan);
    if (rc != PQ_OK) return rc;
    *splits_out = pl.splits;
    *stride_out = p.slab_stride;
    return 1;
}
