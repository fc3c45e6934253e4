/*
 * pqdet_hip.h -- C-ABI of libpqdet_hip.so (gfx950 / MI355X), the drop-in compute boundary of pqdet_amd.
 *
 * The reference (eleflea/PQDet) is pure Python on PyTorch and has no FFI of its own: every entry point below
 * replaces arithmetic that the reference delegates to ATen/cuDNN/torchvision from the Python call sites cited
 * next to it (paths are relative to the reference repo).  The host side (pqdet_amd/, Python, mirroring the
 * reference's Python API) binds these with ctypes on raw device pointers; see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - all tensors fp32, activations NHWC with an explicit channel pitch `ld` (floats) so a tensor may live
 *     inside a wider (concat / padded) buffer; weights keep PyTorch's OIHW layout (checkpoint compatible);
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*), never synchronises, never
 *     allocates: scratch is passed in (`pq_*_scratch_bytes` gives sizes);
 *   - return 0 on success, a negative PQ_E* code on bad arguments / launch failure (pq_last_error() has text);
 *   - "virtual tensors": a conv+BN(+act) output is stored RAW (pre-BN) and consumers apply the per-channel affine
 *     and activation while loading (pq_src).  Gradients w.r.t. a raw tensor are likewise applied on load from
 *     (g, y) pairs (pq_gsrc):  dy = p*g + q*(y-mu) + r, the closed form of BatchNorm's backward.
 */
#ifndef PQDET_HIP_H
#define PQDET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PQ_OK 0
#define PQ_EINVAL (-1)
#define PQ_ELAUNCH (-2)
#define PQ_EUNSUPPORTED (-3)

/* activation codes -- reference model/parser.py:106-112 (ACTIVATION_MAP) + 'linear' */
enum { PQ_ACT_LINEAR = 0, PQ_ACT_RELU = 1, PQ_ACT_RELU6 = 2, PQ_ACT_LEAKY = 3, PQ_ACT_LOGISTIC = 4, PQ_ACT_TANH = 5 };
/* box loss kinds -- reference model/loss.py:62-81 */
enum { PQ_BOX_L1 = 0, PQ_BOX_GIOU = 1, PQ_BOX_DIOU = 2, PQ_BOX_CIOU = 3, PQ_BOX_IOU = 4 };

/* Train-mode BatchNorm statistics handed from the kernel that produces a tensor to the FIRST kernel that consumes it, without
 * a pq_bn_fwd_finalize launch in between (the dependent launch + its dispatch gap cost ~10 us per layer inside a training step;
 * nn.BatchNorm2d semantics unchanged: model/parser.py:404-405).
 *   producer: adds its per-row-block partial sums (sum x, sum x^2 per channel; fp32 values) into `acc` with 64-bit INTEGER
 *             atomics on a fixed-point image of each partial (unit 2^-50, two limbs per value: the sum is exact, hence
 *             independent of the arrival order: run-to-run deterministic).  `slots` replicas (row block b adds to slot
 *             b % slots) keep same-address contention low.  A non-finite / out-of-range partial sets *bad instead.
 *   consumer: every workgroup derives mean / invstd / scale / shift of the channels it touches from `acc` in its prologue
 *             (mean and biased variance E[x^2] - mean^2 in fp64 like pq_bn_fwd_finalize, invstd = 1 / sqrt(var + eps) in fp32:
 *             within an ulp of the finalize kernel's value); the first workgroup of the grid also stores the four vectors (for every later consumer and for the
 *             backward) and updates the running statistics and num_batches_tracked.
 * acc and *bad must be zero when the producer starts (the engine clears the arena of all layers with one pq_fill per step). */
typedef struct pq_bn_stat {
    int64_t* acc;            /* [slots][2][ldp][2]: slot, statistic (sum, sum of squares), channel, limb (low, high); NULL: unused */
    uint32_t* bad;
    const float* gamma;
    const float* beta;
    float* running_mean;     /* updated in place by the consumer's first workgroup; may be NULL */
    float* running_var;
    int64_t* num_batches_tracked;
    float* mean;             /* out [C] */
    float* invstd;
    float* scale;
    float* shift;
    int32_t slots, ldp;
    float count, eps, momentum;
    float inv_count;         /* 1 / count */
} pq_bn_stat;

/* The same hand-over for the BACKWARD sums of a train-mode BatchNorm (S1 = sum dz, S2 = sum dz * xhat per channel): the kernel
 * whose gradient sink finalises the tensor ADDS its per-row-block partials into `acc` (same exact fixed-point image, same slot
 * rule as pq_bn_stat) instead of writing rows for pq_bn_bwd_finalize, and the ONE kernel that consumes the tensor's gradient
 * operand derives  p = gamma*invstd,  q = -p*invstd*S2/count,  r = -p*S1/count  in its prologue (fp64 like the finalize
 * kernel; the accumulated sums are exact, i.e. run-to-run deterministic); its first workgroup per channel chunk also stores
 * dgamma = S2, dbeta = S1 (the parameter gradients) and the p / q / r vectors.  Producers: pq_conv1x1_dgrad (the shapes
 * that run on the ring kernels: pq_bn_bstat_ok), pq_conv1x1_bwd (dx_stat_*); consumer: pq_dw3x3_bwd part 5 (dy_stat) -- the
 * depthwise layers, whose gradient operand has exactly one reader.
 * acc and *bad must be zero when the producer starts.  train == 0 (frozen statistics): q = r = 0. */
typedef struct pq_bn_bstat {
    int64_t* acc;            /* [slots][2][ldp][2]; NULL: unused (the gsrc's p / q / r vectors are read instead) */
    uint32_t* bad;
    const float* gamma;
    const float* invstd;
    float* dgamma;           /* out [C] */
    float* dbeta;
    float* p;                /* out [C] */
    float* q;
    float* r;
    int32_t slots, ldp;
    float count;             /* rows of the tensor */
    int32_t train;
} pq_bn_bstat;

/* forward-value operand:  v[row][c] = act(scale[c] * x[row*ld + c] + shift[c]);  scale==NULL => identity affine;
 * with qstate the value is then fake-quantised (quint8 [0,255], per tensor): (clamp(rint(v/s)+zp) - zp) * s */
typedef struct pq_src {
    const float* x;
    const float* scale;
    const float* shift;
    int32_t ld;
    int32_t act;
    const float* qstate; /* QAT: the consumer sees fake_quant(v) with this activation observer {min,max,scale,zp}; NULL otherwise */
    pq_bn_stat stat;     /* stat.acc != NULL: scale / shift are DERIVED from these statistics by this call (see pq_bn_stat; `scale` and
                            `shift` above are then ignored).  Only entry points for which pq_bn_stat_ok(code, args, 1) says so */
} pq_src;

/* gradient operand: v[row][c] = p[c]*g[row*ldg+c] + q[c]*(y[row*ldy+c] - mu[c]) + r[c];  y==NULL => v = g */
typedef struct pq_gsrc {
    const float* g;
    const float* y;
    const float* p;
    const float* q;
    const float* mu;
    const float* r;
    int32_t ldg;
    int32_t ldy;
} pq_gsrc;

/* gradient destination for a (possibly virtual) tensor T = act(scale*y + shift):
 *   dz arrives from a consumer; if accumulate, dz += g[...]; if finalize, g = dz * act'(scale*y+shift) and, when
 *   mean!=NULL, per-channel partial sums  S1 = sum g,  S2 = sum g*(y-mean)*invstd  are written to
 *   partial[block][0|1][ldp]  (one row pair per row-block; reduced by pq_bn_bwd_finalize). */
typedef struct pq_gsink {
    float* g;
    const float* y;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
    float* partial;
    int32_t ldg;
    int32_t ldy;
    int32_t act;
    int32_t accumulate;
    int32_t finalize;    /* 1: last contribution -- apply act'(y), emit the BatchNorm-backward sums; 3 (pq_route_bwd and the other
                            element-wise sinks only): the same WITHOUT the store -- g already holds exactly this gradient (it is the call's
                            gout: a shortcut whose conv branch uses the sum's gradient buffer as its own), only the sums are wanted */
    int32_t ldp;
    const float* qstate; /* QAT: T is consumed as fake_quant(act(...)) with this observer state {min,max,scale,zp}: finalize also
                            applies the straight-through mask (0 where the value falls outside [0,255] steps).  NULL otherwise */
    int64_t* stat_acc;   /* != NULL (only with finalize && mean && partial): the two BatchNorm-backward sums per channel are ADDED to
                            this accumulator (pq_bn_bstat.acc, pitch ldp) instead of written as rows of `partial` (which stays
                            non-NULL: it is what marks a sink with statistics) -- no pq_bn_bwd_finalize launch follows, the kernels
                            that read the tensor's gradient operand derive p / q / r themselves (pq_bn_bstat in their gsrc's *_stat).
                            Honoured by the calls pq_bn_bstat_ok(code, args, 0) accepts */
    uint32_t* stat_bad;
    int32_t stat_slots;
    int32_t pad_;
    const float* addend; /* with accumulate: the running gradient is READ from here (same pitch ldg) instead of from g -- the first
                            contribution may live in another tensor's gradient (a linear shortcut passes its output's gradient on
                            unchanged, model/parser.py:149-156: no copy, no launch).  NULL: read g.  Honoured by every entry with a
                            pq_gsink except pq_qact_bwd */
} pq_gsink;

/* ------------------------------------------------------------------ 1x1 convolution = fp32 MFMA GEMM
 * replaces nn.Conv2d(k=1) forward/backward inside the conv blocks built at model/parser.py:389-410 and run by
 * AnyModel.forward (model/interpreter.py:45-46).
 *   fwd  : out[m][o] = sum_i in'[m][i] * w[o][i] (+ bias[o]);  rows m = B*H*W, optional spatial subsampling
 *          (sub>1: row (b,oy,ox) reads input pixel (b,oy*sub,ox*sub) -- strided 1x1 conv);
 *          partial!=NULL: also emit per-channel partial sums of out and out^2 (train-mode BN statistics).
 *   dgrad: dz[m][i] = sum_o dy'[m][o] * w[o][i]  -> sink (sub>1: scattered to the strided input pixels)
 *   wgrad: dw[o][i] = sum_m dy'[m][o] * in'[m][i]   (split over m into `splits` slabs, then reduced) */
typedef struct pq_conv1x1_fwd_args {
    pq_src in;
    const float* w;      /* [O][I] */
    const float* bias;   /* [O] or NULL */
    float* out;          /* [M][ldo] raw; NULL: statistics ONLY (partial / stat_acc required, no bias, no minmax): nothing is stored --
                          * the pass in front of pq_xdw_fwd, which recomputes the values */
    float* partial;      /* [pq_conv1x1_blocks_k(M, O, I, 0)][2][ldp] or NULL ([..][4][ldp] with minmax) */
    int32_t M, I, O, ldo, ldp;
    int32_t sub, OH, OW, IH, IW;   /* sub==1: plain; else strided gather of input rows */
    int32_t minmax;      /* != 0: `partial` rows have 4 planes: sum, sum of squares, per-channel MIN and MAX of the outputs of the
                          * row block (QAT: pq_bn_fwd_finalize turns them into the activation observer's update, no pass over
                          * the tensor) */
    int64_t* stat_acc;   /* != NULL: the statistics are ADDED to this accumulator (pq_bn_stat.acc, pitch ldp) instead of written
                          * as rows of `partial` (which may then be NULL) */
    uint32_t* stat_bad;
    int32_t stat_slots;
} pq_conv1x1_fwd_args;
int pq_conv1x1_fwd(const pq_conv1x1_fwd_args* a, void* stream);
/* row-blocks (partial-sum rows) a GEMM with M rows and N output columns uses: fwd N=O, dgrad sink N=I */
int32_t pq_conv1x1_blocks(int32_t M, int32_t N);
/* the same count for a known contraction length K (kind 0: forward, N = O, K = I; kind 1: data gradient, N = I, K = O).
 * pq_conv1x1_fwd / _dgrad tile their rows by this function of (M, N, K, kind): `partial` (forward) and the sink's `partial`
 * (data gradient) must be sized with it.  pq_conv1x1_blocks(M, N) == pq_conv1x1_blocks_k(M, N, 0, 0) only describes
 * contractions that are not multiples of 32. */
int32_t pq_conv1x1_blocks_k(int32_t M, int32_t N, int32_t K, int32_t kind);

typedef struct pq_conv1x1_dgrad_args {
    pq_gsrc dy;
    const float* w;      /* [O][I] */
    pq_gsink dx;
    int32_t M, I, O;
    int32_t sub, OH, OW, IH, IW;   /* sub>1 (strided 1x1 conv): row (b,oy,ox) of dz lands on sink pixel (b,oy*sub,ox*sub); the
                                      sink must then be accumulate-only (finalize==0) -- untouched pixels keep their value */
    const float* wt;     /* optional transposed copy of w, [I][ceil4(O)] with zero pad columns (pq_transpose_batch writes that
                            layout): K-contiguous B operand, faster LDS path */
    pq_bn_bstat dy_stat; /* dy_stat.acc != NULL: the coefficients of dy are DERIVED from these sums in the kernel's prologue (no
                            pq_bn_bwd_finalize launch in front of this call; dy.p / dy.q / dy.r are ignored, dy.mu is still read).
                            Ring kernels only: pq_bn_bstat_ok(PQ_OP_CONV1X1_DGRAD, args, 1) */
} pq_conv1x1_dgrad_args;
int pq_conv1x1_dgrad(const pq_conv1x1_dgrad_args* a, void* stream);

typedef struct pq_conv1x1_wgrad_args {
    pq_gsrc dy;
    pq_src in;
    float* dw;           /* [O][I]; NULL: deferred reduction (pq_reduce_batch), the slabs stay in `scratch` */
    float* dbias;        /* [O] or NULL: column sums of dy' (deferred form: non-NULL only says that the column is wanted) */
    float* scratch;      /* >= pq_conv1x1_wgrad_scratch_bytes(M,I,O) */
    int32_t M, I, O;
    int32_t sub, OH, OW, IH, IW;   /* sub>1: in' rows are gathered at (b,oy*sub,ox*sub) as in the forward */
    pq_bn_bstat dy_stat; /* as in the data-gradient arguments above; every kernel of this entry derives per lane */
} pq_conv1x1_wgrad_args;
int pq_conv1x1_wgrad(const pq_conv1x1_wgrad_args* a, void* stream);

/* Fused backward of a NARROW 1x1 convolution (min(I,O) <= 32, max <= 192, many rows): dgrad + wgrad of the two entries
 * above from ONE pass over dy' (both GEMMs are HBM-bound there and dy' is the widest operand).  Same operand meaning as
 * pq_conv1x1_dgrad (dy, w, dx) and pq_conv1x1_wgrad (in, dw); stride 1 only, no bias gradient.
 * pq_conv1x1_bwd_blocks(M,I,O): number of workgroups = partial-sum rows the sink emits; 0 => shape outside the fused
 * kernel's range (use the two separate entries; pq_conv1x1_bwd then returns PQ_EUNSUPPORTED).
 * pq_conv1x1_bwd_pays(M,I,O): 1 when the fused kernel is supported AND faster on a whole step than the two entries. */
typedef struct pq_conv1x1_bwd_args {
    pq_gsrc dy;
    const float* w;      /* [O][I] */
    pq_src in;
    pq_gsink dx;
    float* dw;           /* [O][I]; NULL: deferred reduction (pq_reduce_batch) */
    float* scratch;      /* >= pq_conv1x1_bwd_scratch_bytes(M,I,O) */
    int32_t M, I, O;
    pq_bn_bstat dy_stat; /* as in the data-gradient arguments above */
} pq_conv1x1_bwd_args;
int pq_conv1x1_bwd(const pq_conv1x1_bwd_args* a, void* stream);
int32_t pq_conv1x1_bwd_blocks(int32_t M, int32_t I, int32_t O);
int32_t pq_conv1x1_bwd_pays(int32_t M, int32_t I, int32_t O);
int64_t pq_conv1x1_bwd_scratch_bytes(int32_t M, int32_t I, int32_t O);
int64_t pq_conv1x1_wgrad_scratch_bytes(int32_t M, int32_t I, int32_t O);

/* Which kernel instantiation a pq_conv1x1_{fwd,dgrad,wgrad} call with exactly these arguments launches (pure host
 * function of the argument struct: shapes, pitches, pointer alignment, operand kinds; nothing is launched, no device
 * memory is touched).  The parity tests assert it for every case, so an eligibility rule cannot silently send a shape
 * to a kernel that no test checks, and list the instantiations the BASELINE nets select at their benchmark shapes. */
typedef struct pq_gemm_plan {
    int32_t family;   /* 1: register-staged kernels (gemm.hip); 2: LDS-DMA ring kernels (gemm2.hip) */
    int32_t bm, bn;   /* output tile: rows x columns (wgrad: O x I) */
    int32_t stages;   /* family 2: ring depth; family 1: 2 = double-buffered LDS, 1 = single buffer (streaming full-N tiles) */
    int32_t amode;    /* operand transform: 0 as stored, 1 BatchNorm affine + activation, 2 gradient pair (closed-form BN backward);
                         wgrad: bit 0 = gradient pair, bit 1 = transformed input operand */
    int32_t emode;    /* epilogue: 0 store (+bias), 1 + BatchNorm statistics, 2 gradient sink, 3 + per-channel min / max (QAT) */
    int32_t splits;   /* wgrad: row splits (slabs to reduce); 0 otherwise */
    int32_t variant;  /* family 1: B operand stored [k][n] (1) or [n][k] (0); family 2 wgrad: waves per workgroup */
} pq_gemm_plan;
int pq_conv1x1_fwd_plan(const pq_conv1x1_fwd_args* a, pq_gemm_plan* out);
int pq_conv1x1_dgrad_plan(const pq_conv1x1_dgrad_args* a, pq_gemm_plan* out);
int pq_conv1x1_wgrad_plan(const pq_conv1x1_wgrad_args* a, pq_gemm_plan* out);

/* ------------------------------------------------------------------ depthwise 3x3 (groups == channels), pad 1
 * replaces nn.Conv2d(groups=C, k=3) fwd/bwd (mobilenetv2-fpn: 26 layers).  w is OIHW [C][1][3][3]. */
typedef struct pq_dw3x3_fwd_args {
    pq_src in;
    const float* w;
    float* out;          /* raw, pitch ldo */
    float* partial;      /* [nblk][2][ldp] or NULL; nblk = pq_dw3x3_fwd_blocks(B, OH, OW, C, stride); [nblk][4][ldp] with minmax */
    int32_t B, IH, IW, OH, OW, C, stride, ldo, ldp;
    int32_t minmax;      /* as in pq_conv1x1_fwd_args */
    int64_t* stat_acc;   /* as in pq_conv1x1_fwd_args */
    uint32_t* stat_bad;
    int32_t stat_slots;
} pq_dw3x3_fwd_args;
int pq_dw3x3_fwd(const pq_dw3x3_fwd_args* a, void* stream);
int32_t pq_dw3x3_fwd_blocks(int32_t B, int32_t OH, int32_t OW, int32_t C, int32_t stride);   /* statistics rows written by pq_dw3x3_fwd */
int32_t pq_dw3x3_blocks(int32_t B, int32_t H, int32_t W, int32_t C);   /* row-blocks of the streaming pq_dw3x3_dgrad sink */

typedef struct pq_dw3x3_dgrad_args {
    pq_gsrc dy;          /* at output resolution */
    const float* w;
    pq_gsink dx;         /* at input resolution */
    int32_t B, IH, IW, OH, OW, C, stride;
} pq_dw3x3_dgrad_args;
int pq_dw3x3_dgrad(const pq_dw3x3_dgrad_args* a, void* stream);

typedef struct pq_dw3x3_wgrad_args {
    pq_gsrc dy;
    pq_src in;
    float* dw;           /* [C][9] */
    float* scratch;      /* >= pq_dw3x3_wgrad_scratch_bytes */
    int32_t B, IH, IW, OH, OW, C, stride;
} pq_dw3x3_wgrad_args;
int pq_dw3x3_wgrad(const pq_dw3x3_wgrad_args* a, void* stream);
int64_t pq_dw3x3_wgrad_scratch_bytes(int32_t B, int32_t OH, int32_t OW, int32_t C);

/* depthwise backward, both gradients from ONE pass over (g_out, y_out, y_in): the WALKING kernel (any plane size, stride 1|2,
 * even input sizes at stride 2; needs in.x == dx.y, i.e. the sink is the conv's input tensor).  A workgroup walks down a
 * 16-column strip with (dy', input) row rings in LDS; the sink's statistics rows and the weight-gradient rows in `scratch` are
 * one per workgroup (pq_dw3x3_bwd_walk_blocks).  4 tensor passes instead of the 7 of pq_dw3x3_dgrad + pq_dw3x3_wgrad, which stay
 * as the fallback for the planes this kernel does not take.  (Parts 0-4, the LDS-tiled kernels of round 2, were removed.) */
typedef struct pq_dw3x3_bwd_args {
    pq_gsrc dy;
    pq_src in;
    const float* w;
    pq_gsink dx;
    float* dw;
    float* scratch;      /* >= pq_dw3x3_bwd_walk_blocks * C * 9 floats */
    int32_t B, IH, IW, OH, OW, C, stride;
    int32_t part;        /* 5: both gradients, per-workgroup weight-gradient rows left in `scratch` (dw unused);
                            6: reduce the rows a part-5 launch left in `scratch` into dw (the only launch that may run on
                               another stream; dy, in, w, dx unused) */
    pq_bn_bstat dy_stat; /* part 5, dy_stat.acc != NULL: the coefficients of dy are DERIVED from these sums (no pq_bn_bwd_finalize
                            launch in front of this call; dy.p / dy.q / dy.r are ignored, dy.mu is still read) */
} pq_dw3x3_bwd_args;
int pq_dw3x3_bwd(const pq_dw3x3_bwd_args* a, void* stream);
/* workgroups of the walking kernel = rows of the sink's statistics = rows [C][9] of `scratch` */
int32_t pq_dw3x3_bwd_walk_blocks(int32_t B, int32_t IH, int32_t IW, int32_t C, int32_t stride);

/* ------------------------------------------------------------------ fused inverted-residual block (MobileNetV2)
 * replaces the conv(1x1, I -> E) -> BatchNorm -> act -> conv(3x3, groups = E, stride 1|2, pad 1) chain of an inverted-residual block
 * (model/parser.py:389-410 as mobilenetv2-fpn.cfg lays the triples out) WITHOUT materialising the expanded E-channel tensor:
 *   out[b,oy,ox,c] = sum_taps wd[c][tap] * mid'(sum_i we[c][i] * in'[b, oy*s+ky-1, ox*s+kx-1, i])      (zero padding of mid')
 * mid'(v) = act(mid.scale[c] * v + mid.shift[c]): the BatchNorm of the expanded tensor; its train-mode statistics come from
 * pq_xstat_fwd in front of this call (vectors mid.scale / mid.shift), or from a statistics-only pq_conv1x1_fwd (out == NULL), either
 * finalised or handed over (mid.stat, pq_bn_stat: this call is the first consumer, its first workgroup per chunk stores the vectors).
 * mid.x is ignored.  in.act must be PQ_ACT_LINEAR (the expand layer of an inverted-residual block reads a linear bottleneck or a
 * materialised sum): the operand's affine is folded into the weights.  The call emits the depthwise output's own statistics like
 * pq_dw3x3_fwd (partial rows or stat_acc).
 * pq_xdw_fwd_blocks: partial-sum rows = walkers of the launch; 0 => shape outside the kernel's range (I in {16, 24}; stride 2
 * needs even plane sizes) -- use the unfused entries.  yexp != NULL: the raw expanded tensor is ALSO written (pitch ldy; for
 * a backward that reads it and for tests). */
typedef struct pq_xdw_fwd_args {
    pq_src in;           /* block input, (B, IH, IW, I) */
    const float* we;     /* [E][I] expand weights */
    pq_src mid;          /* BatchNorm + activation of the expanded tensor */
    float* yexp;         /* optional [B*IH*IW][ldy] raw expanded tensor, or NULL */
    const float* wd;     /* [E][1][3][3] depthwise weights */
    float* out;          /* [B*OH*OW][ldo] raw depthwise output */
    float* partial;      /* [pq_xdw_fwd_blocks][2][ldp] or NULL */
    int32_t B, IH, IW, OH, OW, I, E, stride, ldy, ldo, ldp;
    int64_t* stat_acc;   /* as in pq_conv1x1_fwd_args */
    uint32_t* stat_bad;
    int32_t stat_slots;
} pq_xdw_fwd_args;
int pq_xdw_fwd(const pq_xdw_fwd_args* a, void* stream);
int32_t pq_xdw_fwd_blocks(int32_t B, int32_t IH, int32_t IW, int32_t I, int32_t E, int32_t stride);

/* Train-mode BatchNorm statistics of the expanded tensor y = we x' of such a block WITHOUT computing it (nn.BatchNorm2d semantics,
 * model/parser.py:404-405): per expanded channel c, sum y_c = we_c . s and sum y_c^2 = we_c^T G we_c with s = sum x', G = sum x' x'^T
 * over the M pixels -- one pass over the I-channel input (workgroup partials added into exact fixed-point accumulators in `scratch`:
 * order independent, run-to-run deterministic), then a finalize that writes mean / invstd / scale / shift and the running statistics
 * exactly like pq_bn_fwd_finalize(train).  I in {16, 24}.  in.stat is honoured (pq_bn_stat, consumer role): the block input's own
 * BatchNorm may be handed over by its producer. */
typedef struct pq_xstat_fwd_args {
    pq_src in;           /* block input, M rows of I channels */
    const float* we;     /* [E][I] */
    const float* gamma;
    const float* beta;
    float* running_mean; /* updated in place; may be NULL */
    float* running_var;
    int64_t* num_batches_tracked;
    float* mean;         /* out [E] */
    float* invstd;
    float* scale;
    float* shift;
    float* scratch;      /* >= pq_xstat_fwd_scratch_bytes(M, I), 16-byte aligned: exact accumulators; ZERO before the first call, the
                          * call leaves them zero again (self re-arming) */
    int32_t M, I, E;
    float eps, momentum;
    double* moments;     /* out, optional: [I * I + I] fp64 -- G (row-major) and s, what pq_xdw_bwd needs of the forward */
} pq_xstat_fwd_args;
int pq_xstat_fwd(const pq_xstat_fwd_args* a, void* stream);
int64_t pq_xstat_fwd_scratch_bytes(int32_t M, int32_t I);

/* Backward of the chain pq_xdw_fwd computes, in one pass over (g_d, y_d, x) with the expanded tensor recomputed (csrc/irbwd.hip):
 * gradient w.r.t. the block input (through the sink dx: activation derivative, BatchNorm-backward sums, accumulation), the two weight
 * gradients and the BatchNorm parameter gradients of the expanded tensor.  Train-mode statistics only (the closed-form BatchNorm
 * backward of the expanded tensor is folded analytically into the expand layer's two products, which needs the moments of x' that
 * pq_xstat_fwd kept in `moments`).  Three launches: the walking kernel, a fold over its partial rows, a fix-up pass over the
 * I-channel gradient.  dx.partial (if the sink has statistics) has pq_xdw_bwd_fix_blocks(M, I) rows.
 * pq_xdw_bwd_blocks == 0: shape outside the kernel's range (today: 16 -> 96 stride 2, 24 -> 144 stride 1 | 2). */
typedef struct pq_xdw_bwd_args {
    pq_src in;           /* block input x (linear operand), (B, IH, IW, I) */
    const float* we;     /* [E][I] */
    const float* mid_scale; const float* mid_shift; const float* mid_mean; const float* mid_invstd; const float* mid_gamma;   /* [E] */
    int32_t mid_act;     /* PQ_ACT_RELU6 */
    int32_t pad_;
    pq_gsrc dy;          /* gradient pair of the depthwise output, (B, OH, OW, E) */
    const float* wd;     /* [E][1][3][3] */
    pq_gsink dx;         /* gradient sink of the block input */
    float* dwe;          /* out [E][I] */
    float* dwd;          /* out [E][9] */
    float* dgamma;       /* out [E]: BatchNorm of the expanded tensor */
    float* dbeta;
    const double* moments;   /* [I * I + I] from pq_xstat_fwd */
    float* scratch;      /* >= pq_xdw_bwd_scratch_bytes, 16-byte aligned */
    float* gexp;         /* optional (tests): gradient w.r.t. the BatchNorm output of the expanded tensor, [B*IH*IW][ldge] */
    int32_t ldge;
    int32_t B, IH, IW, OH, OW, I, E, stride;
} pq_xdw_bwd_args;
int pq_xdw_bwd(const pq_xdw_bwd_args* a, void* stream);
int32_t pq_xdw_bwd_blocks(int32_t B, int32_t IH, int32_t IW, int32_t I, int32_t E, int32_t stride);
int32_t pq_xdw_bwd_fix_blocks(int32_t M, int32_t I);
int64_t pq_xdw_bwd_scratch_bytes(int32_t B, int32_t IH, int32_t IW, int32_t I, int32_t E, int32_t stride);

/* ------------------------------------------------------------------ forward-only layers of the classifier cfgs
 * replaces FC (model/parser.py:122-137), nn.MaxPool2d / nn.AdaptiveAvgPool2d (:441-449), dense k x k nn.Conv2d beyond the grouped
 * 3 x 3 kernels (:389-410, as im2col + pq_conv1x1_fwd) and logistic / tanh as an ordinary conv activation (:106-112).  The
 * reference has no training path for these nets, so there is no backward. */
#define PQ_LAYER_MAXPOOL 0   /* nn.MaxPool2d(k, stride, pad), -inf padding */
#define PQ_LAYER_AVGPOOL 1   /* nn.AdaptiveAvgPool2d((OH, OW)) */
#define PQ_LAYER_IM2COL 2    /* out[B*OH*OW][ldo], column = ci*k*k + ky*k + kx (OIHW weight-row order), zero padding, pad columns 0 */
#define PQ_LAYER_FC 3        /* out[o] = act(bias[o] + sum_n w[o][n] * in'(n)), n over the NCHW-flattened (1, C, IH, IW) input */
#define PQ_LAYER_ACT 4       /* out = act(scale * x + shift) materialised at pitch ldo (logistic / tanh allowed) */
#define PQ_LAYER_PADROWS 5   /* out[r][0..ldo) = w[r][0..k), then zeros; r < O */
typedef struct pq_layer_fwd_args {
    pq_src in;           /* NHWC virtual source (B, IH, IW, C); IM2COL with nchw != 0: the raw (B, C, IH, IW) network input */
    const float* w;      /* FC: [O][C*IH*IW]; PADROWS: [O][k] */
    const float* bias;   /* FC: [O] or NULL */
    float* out;
    int32_t kind, B, IH, IW, OH, OW, C, k, stride, pad, ldo, O, act, nchw;
} pq_layer_fwd_args;
int pq_layer_fwd(const pq_layer_fwd_args* a, void* stream);

/* ------------------------------------------------------------------ grouped 3x3 convolution, pad 1, stride 1|2
 * replaces nn.Conv2d(k=3, groups=C/gw) fwd/bwd in the conv blocks of model/parser.py:389-410 (regnetx-600m-fpn/pan: group
 * width 24 in the backbone, 16 in the neck; also a dense 3x3 with C == gw <= 32).  gw % 4 == 0, gw <= 32, C % gw == 0.
 * w is OIHW [C][gw][3][3].  Implicit GEMM on the fp32 MFMA, no im2col buffer.
 *   fwd  : out[b,oy,ox,o] = sum_{tap,i} in'[b, oy*s+ky-1, ox*s+kx-1, grp(o)*gw+i] * w[o][i][tap]   (+ BN partial sums)
 *   dgrad: transposed convolution of dy' -> sink (rows of the input plane)
 *   wgrad: dw[o][i][tap] = sum_pixels dy'[p][o] * in'[p*s+tap-1][i] */
typedef struct pq_gconv3x3_fwd_args {
    pq_src in;
    const float* w;
    float* out;          /* raw, pitch ldo */
    float* partial;      /* [pq_gconv3x3_blocks(B, OH, OW)][2][ldp] or NULL */
    int32_t B, IH, IW, OH, OW, C, gw, stride, ldo, ldp;
    int64_t* stat_acc;   /* statistics hand-over, as in pq_conv1x1_fwd_args; in.stat is honoured too */
    uint32_t* stat_bad;
    int32_t stat_slots;
} pq_gconv3x3_fwd_args;
int pq_gconv3x3_fwd(const pq_gconv3x3_fwd_args* a, void* stream);
/* partial-sum rows written by a launch over a (B, H, W) pixel grid: forward H,W = output plane; dgrad sink H,W = input plane */
int32_t pq_gconv3x3_blocks(int32_t B, int32_t H, int32_t W);

typedef struct pq_gconv3x3_dgrad_args {
    pq_gsrc dy;          /* on the (B, OH, OW) plane */
    const float* w;
    pq_gsink dx;         /* on the (B, IH, IW) plane */
    int32_t B, IH, IW, OH, OW, C, gw, stride;
} pq_gconv3x3_dgrad_args;
int pq_gconv3x3_dgrad(const pq_gconv3x3_dgrad_args* a, void* stream);

typedef struct pq_gconv3x3_wgrad_args {
    pq_gsrc dy;
    pq_src in;
    float* dw;           /* [C][gw][3][3]; NULL: deferred reduction (pq_reduce_batch) */
    float* scratch;      /* >= pq_gconv3x3_wgrad_scratch_bytes(B, OH, OW, C, gw) */
    int32_t B, IH, IW, OH, OW, C, gw, stride;
} pq_gconv3x3_wgrad_args;
int pq_gconv3x3_wgrad(const pq_gconv3x3_wgrad_args* a, void* stream);
int64_t pq_gconv3x3_wgrad_scratch_bytes(int32_t B, int32_t OH, int32_t OW, int32_t C, int32_t gw);

/* ------------------------------------------------------------------ network stem: dense 3x3 conv on the NCHW image
 * (3 input channels, pad 1, stride 1|2), the first section of every shipped detector cfg. */
typedef struct pq_stem3x3_fwd_args {
    const float* image;  /* NCHW [B][CI][IH][IW] (dataset/augment.py:396 ToTensor layout) */
    const float* w;      /* [O][CI][3][3] */
    float* out;          /* NHWC raw, pitch ldo */
    float* partial;
    int32_t B, CI, IH, IW, OH, OW, O, stride, ldo, ldp;
    int32_t minmax;      /* as in pq_conv1x1_fwd_args */
} pq_stem3x3_fwd_args;
int pq_stem3x3_fwd(const pq_stem3x3_fwd_args* a, void* stream);

typedef struct pq_stem3x3_wgrad_args {
    pq_gsrc dy;
    const float* image;
    float* dw;
    float* scratch;
    int32_t B, CI, IH, IW, OH, OW, O, stride;
} pq_stem3x3_wgrad_args;
int pq_stem3x3_wgrad(const pq_stem3x3_wgrad_args* a, void* stream);
int64_t pq_stem3x3_wgrad_scratch_bytes(int32_t B, int32_t OH, int32_t OW, int32_t O, int32_t CI);

/* ------------------------------------------------------------------ BatchNorm2d statistics (train / eval)
 * replaces nn.BatchNorm2d (model/parser.py:404-405): eps 1e-5, momentum 0.1, biased variance for normalisation,
 * unbiased for running_var, num_batches_tracked += 1. */
typedef struct pq_bn_fwd_finalize_args {
    const float* partial;   /* [nblk][2][ldp]: sum x, sum x^2 */
    const float* gamma;
    const float* beta;
    float* running_mean;    /* updated in place (train) / read (eval) */
    float* running_var;
    int64_t* num_batches_tracked;   /* may be NULL */
    float* mean;            /* out [C] */
    float* invstd;          /* out [C] */
    float* scale;           /* out [C]: gamma*invstd */
    float* shift;           /* out [C]: beta - mean*scale */
    int32_t nblk, ldp, C, train;
    float count, eps, momentum;
    double* scratch;        /* >= pq_bn_finalize_scratch_bytes(C), or NULL: one block per 64 channels walks all rows */
    uint32_t* tickets;      /* pq_bn_finalize_tickets(C) counters, zero before the first use (self re-arming), or NULL */
    const float* sf;        /* QAT (fused conv-BN): the conv ran with weights scaled by sf[c]; BatchNorm acts on out/sf.  NULL otherwise */
    /* QAT, activation observer folded into the finalize (train only): `partial` rows have 4 planes (the producer ran with
     * minmax != 0); the per-channel extremes of act(scale*raw + shift) follow from the raw ones (monotone per channel), the
     * tensor's min / max go into the moving-average observer `qstate` = {min, max, scale, zero_point} exactly as
     * pq_qact_fwd(out == NULL) would after a pass over the tensor.  planes == 0 means 2. */
    int32_t planes;         /* 2 | 4 */
    int32_t act;            /* activation of the tensor (PQ_ACT_*) */
    float* qstate;          /* observer state to update, or NULL */
    uint32_t* mm;           /* 3 words {0xffffffff, 0, 0} before the first use (self re-arming) */
    float avg_const;        /* moving-average constant (0.01) */
} pq_bn_fwd_finalize_args;
int pq_bn_fwd_finalize(const pq_bn_fwd_finalize_args* a, void* stream);

/* Inference / frozen statistics: the per-channel vectors of EVERY BatchNorm of a net from its running statistics in ONE launch
 * (one workgroup per table entry; the table lives in device memory).  Same arithmetic as pq_bn_fwd_finalize with train == 0:
 * mean = rm*sf, invstd = 1/sqrt(rv+eps)/sf, scale = gamma*invstd, shift = beta - rm*gamma/sqrt(rv+eps)   (sf = 1 without QAT) */
typedef struct pq_bn_eval_item {
    const float* gamma; const float* beta; const float* running_mean; const float* running_var;
    float* mean; float* invstd; float* scale; float* shift;
    const float* sf;        /* QAT weight scale per channel or NULL */
    float eps;
    int32_t C;
} pq_bn_eval_item;
typedef struct pq_bn_eval_args {
    const pq_bn_eval_item* items;   /* device pointer */
    int32_t n_items;
} pq_bn_eval_args;
int pq_bn_eval_batch(const pq_bn_eval_args* a, void* stream);

typedef struct pq_bn_bwd_finalize_args {
    const float* partial;   /* [nblk][2][ldp]: sum g, sum g*xhat */
    const float* gamma;
    const float* invstd;
    float* dgamma;
    float* dbeta;
    float* p;               /* out [C]: coefficient vectors of pq_gsrc */
    float* q;
    float* r;
    int32_t nblk, ldp, C, train;
    float count;
    double* scratch;        /* as in pq_bn_fwd_finalize_args; the two may share buffers when issued on one stream */
    uint32_t* tickets;
} pq_bn_bwd_finalize_args;
int pq_bn_bwd_finalize(const pq_bn_bwd_finalize_args* a, void* stream);
/* long partial lists are reduced by several blocks per channel group through this scratch (deterministic order) */
int64_t pq_bn_finalize_scratch_bytes(int32_t C);
/* Can the op `code` (PQ_OP_*) called with exactly `args` (its argument struct) take part in the statistics hand-over of
 * pq_bn_stat?  role 0: as the PRODUCER (statistics into stat_acc instead of `partial` rows; the kernel this shape dispatches to
 * has the atomic epilogue), role 1: as the first CONSUMER of its input operand (in.stat / a.stat derived in the prologue).
 * Pure host function. */
int32_t pq_bn_stat_ok(int32_t code, const void* args, int32_t role);
/* the same question for the backward sums (pq_bn_bstat): role 0 = this call's sink honours pq_gsink.stat_acc (PQ_OP_CONV1X1_DGRAD on the
 * ring kernels, PQ_OP_CONV1X1_BWD, PQ_OP_DW_BWD part 5, PQ_OP_ROUTE_BWD), role 1 = this call can derive its gradient operand's
 * coefficients from its dy_stat (PQ_OP_CONV1X1_DGRAD on the ring kernels, PQ_OP_CONV1X1_WGRAD, PQ_OP_CONV1X1_BWD, PQ_OP_DW_BWD part 5).
 * Exactly the kernels of role 1 that have a gradient sink of their own store the derived vectors / dgamma / dbeta (the weight-gradient
 * entry never does): hand over only when one of the readers is such a call. */
int32_t pq_bn_bstat_ok(int32_t code, const void* args, int32_t role);
/* bytes of a pq_bn_stat.acc for C channels with pitch ldp and `slots` replicas, and the replica count to use for nblk row blocks */
int64_t pq_bn_stat_bytes(int32_t ldp, int32_t slots);
int32_t pq_bn_stat_slots(int32_t nblk);
/* At most 4096 partials may be added into ONE slot of an accumulator (the low limb of the fixed-point image is read back as a
 * signed 64-bit integer): every entry point that takes stat_acc / dx_stat_acc checks its own row-block count / slots against that
 * and returns PQ_EUNSUPPORTED beyond it.  Test hook: values[i] is added as the partial of row block i (plane 0, channel 0) with
 * exactly the producers' arithmetic; acc ([slots][2][ldp][2] int64) and *bad must be zero before. */
int pq_bn_stat_selftest(const float* values, int32_t n, int32_t slots, int32_t ldp, int64_t* acc, uint32_t* bad, void* stream);
int32_t pq_bn_finalize_tickets(int32_t C);

/* ------------------------------------------------------------------ shortcut / route / upsample
 * forward  (model/parser.py:149-156 ShortCut, :178-183 Route, :450-452 UpsamplingNearest2d):
 *   out[b,y,x,coff+c] = a'[b,y/up,x/up,c] (+ b'[b,y,x,c])
 * backward: dz[b,ys,xs,c] = sum_{dy,dx<up} gout[b,ys*up+dy,xs*up+dx,coff+c]  -> sink */
typedef struct pq_copy_args {
    pq_src a;
    pq_src b;            /* b.x == NULL: no second addend */
    float* out;
    int32_t B, H, W, C, ldo, coff, up;   /* H,W = OUTPUT size */
    const float* gate;   /* scale_channels (model/parser.py:158-168): out *= sigmoid(gate[b][c]); raw gate [B][ldgate] or NULL */
    int32_t ldgate;
} pq_copy_args;
int pq_copy_fwd(const pq_copy_args* a, void* stream);

typedef struct pq_route_bwd_args {
    const float* gout;
    pq_gsink dx;
    int32_t B, H, W, C, ldgo, coff, up;  /* H,W = SOURCE (dx) size */
    const float* gate;   /* scale_channels backward: dz = gout * sigmoid(gate[b][c]) (up == 1) */
    int32_t ldgate;
    int32_t bcast;       /* global average pool backward: gout is [B][ldgo], dz[b,p,c] = gout[b][c] * mul */
    float mul;
    pq_gsink dx2;        /* dx2.g != NULL: a SECOND destination of the same dz (both inputs of a shortcut add receive the output's
                            gradient, model/parser.py:149-156): gout is read once for the two sinks.  up == 1, no gate / bcast */
} pq_route_bwd_args;
int pq_route_bwd(const pq_route_bwd_args* a, void* stream);
int32_t pq_ew_blocks(int32_t rows, int32_t C);   /* row-blocks used by elementwise sinks (partial rows) */

/* reductions over the pixels of each image (squeeze-and-excitation blocks of regnety-400m-fpn.cfg):
 *   gout == NULL: out[b][c] = mul * sum_p a'[b,p,c]                (avgpool with height = width = 1)
 *   gout != NULL: out[b][c] = s(1-s) * mul * sum_p gout[b,p,c] * a'[b,p,c], s = sigmoid(gate[b][c])   (scale_channels gate gradient) */
typedef struct pq_pool_reduce_args {
    pq_src a;
    const float* gout; int32_t ldgo;
    const float* gate; int32_t ldgate;
    float* out; int32_t ldo;
    int32_t B, HW, C;
    float mul;
} pq_pool_reduce_args;
int pq_pool_reduce(const pq_pool_reduce_args* a, void* stream);

/* ------------------------------------------------------------------ YOLO head: decode + loss
 * decode: model/parser.py:206-235; loss: model/loss.py:22-115 (+ tools.py:357-477 IoU family).
 * Element (b,h,w,a,k) of the head lives at  base[b*sb + h*sh + w*sw + (a*(5+C)+k)*sc]  (floats), so both the
 * engine's NHWC head buffer and an NCHW tensor (standalone Decode/YOLOLayer API) are addressed in place. */
typedef struct pq_head_view {
    const float* base;
    int64_t sb, sh, sw, sc;
} pq_head_view;

typedef struct pq_decode_args {
    pq_head_view raw;
    float* out;          /* [B][rows_total][5+C], this head starts at row `row0`, cells in (h,w,a) order */
    int32_t B, H, W, A, C, stride, rows_total, row0;
} pq_decode_args;
int pq_decode(const pq_decode_args* a, void* stream);
/* gradient of decode for the standalone API: draw = dpred * d(decode)/d(raw); `pred` is the decoded output */
typedef struct pq_decode_bwd_args {
    const float* pred;
    const float* dpred;  /* [B][H][W][A][5+C] */
    float* draw;         /* addressed through `view` (base ignored) */
    pq_head_view view;
    int32_t B, H, W, A, C, stride;
} pq_decode_bwd_args;
int pq_decode_bwd(const pq_decode_bwd_args* a, void* stream);

typedef struct pq_yolo_loss_args {
    pq_head_view in;     /* raw head (from_raw=1) or decoded pred (from_raw=0; then sb.. describe pred) */
    const float* label;  /* [B][H][W][A][6+C] */
    const float* boxes;  /* [B][M][4] zero padded */
    float* partial;      /* scratch: [B][nblk][3] floats, nblk = pq_yolo_loss_blocks(H,W,A) */
    float* losses;       /* out [4]: loss, box, conf, cls  (each already .mean over batch) */
    int32_t* nan_flag;   /* out: set to 1 when loss is NaN (model/loss.py:110-114), never cleared by the kernel; may be NULL.  May
                          * point into pinned host memory (pq_host_device_pointer): the host then polls it without a copy */
    /* backward only */
    const float* gscale; /* [3] upstream weights for (box, conf, cls) */
    float* grad;         /* gradient w.r.t. `in` (same addressing as `in`) */
    int64_t gsb, gsh, gsw, gsc;
    int32_t B, H, W, A, C, M, stride, box_loss, from_raw;
    float ignore_thresh, l1_gain;
    /* whole-step glue (all optional; the reference sums the per-head 4-tuples in DetectionModel.forward, model/interpreter.py:77-85):
     * forward: `losses` is row `head_slot` of an [n_heads][4] table; with totals != NULL this call -- which must be the LAST of
     * the table's heads in stream order -- also writes totals[0..3] = the column sums of the table and totals[4 + h] = head h's
     * total, i.e. the dict DetectionModel.forward returns, in one buffer;
     * backward (used when gscale == NULL): upstream gradient SCALARS w.r.t. totals[0..3] and totals[4 + head_slot]; any may be
     * NULL (= 0).  The weights of (box, conf, cls) are gup[0] + gup[1 | 2 | 3] + gup[4]. */
    float* totals;
    int32_t head_slot, n_heads;
    const float* gup[5];
    int32_t phase;       /* forward: 0 both launches; 1 the per-cell kernel only (block partials); 2 the finalize only (partials ->
                            losses row, totals) -- lets a caller put a stream join between the two (the finalize with totals reads
                            the other heads' rows) */
    const struct pq_yolo_loss_args* also[2];   /* backward only: up to two more heads (HOST pointers to their argument structs, read
                            at call time) whose gradients the SAME launch computes -- the per-head kernels are latency bound
                            (~26 us each whatever the head size), three of them at the start of a backward cost one; same B.
                            NULL: this head only */
} pq_yolo_loss_args;
int pq_yolo_loss_fwd(const pq_yolo_loss_args* a, void* stream);
int pq_yolo_loss_bwd(const pq_yolo_loss_args* a, void* stream);
int32_t pq_yolo_loss_blocks(int32_t H, int32_t W, int32_t A);

/* broadcast IoU family on (n,4) x (n,4) pairs (tools.py:357-477), kind = PQ_BOX_IOU/GIOU/DIOU/CIOU */
int pq_iou_pairs(const float* b1, const float* b2, float* out, int32_t n, int32_t kind, void* stream);

/* ------------------------------------------------------------------ Adam (trainer.py:160-163, torch.optim.Adam)
 * one launch over the flat parameter / gradient / moment buffers. */
typedef struct pq_adam_args {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, grad_scale;
} pq_adam_args;
int pq_adam_step(const pq_adam_args* a, void* stream);
/* trainer.py:236-239 sparse-L1 on BN gammas: grad += ratio*sign(weight) over [n] */
int pq_sign_l1(float* grad, const float* weight, int64_t n, float ratio, void* stream);
int pq_fill(float* dst, float value, int64_t n, void* stream);

/* ------------------------------------------------------------------ post-process
 * recover: dataset/base_sample.py:98-139 (+ voc/coco/visdrone affine); in place like the reference:
 *   coords -= [dx,dy,dx,dy]; /= ratio; clamp to [0, orig-1]; prob *= conf.  out (B,N,4+C). */
typedef struct pq_recover_args {
    float* pred;         /* [B][N][5+C]; when mutate!=0 it is also updated in place like the reference does */
    const float* delta;  /* [B][2] (dy,dx) as returned by the affine fn, or [1][2] with per_image=0 */
    const float* ratio;  /* [B] or [1] */
    const float* orig;   /* [B][2] (h,w) or [1][2] */
    float* out;          /* [B][N][4+C] */
    int32_t B, N, C, per_image, mutate;
} pq_recover_args;
int pq_recover(const pq_recover_args* a, void* stream);

/* Label assignment for a batch: TrainDataset.create_label + collate_batch padding (dataset/train_dataset.py:16-43,109-150).
 *   gt [B][G][6] = x1,y1,x2,y2,class,mix-up weight (rows >= counts[b] ignored; counts NULL: all G rows are boxes)
 *   label[s] [B][H[s]][W[s]][3][6+C] (fully written: zeros, mix-up weight 1, then the assigned rows in box order)
 *   boxes[s] [B][cap][4] zero padded, insertion order, duplicates kept; box_counts [B][3] rows used (may be NULL).
 *   The reference pads the lists to the batch maximum; any cap >= that maximum is equivalent for the loss (all-zero rows
 *   have IoU 0 with every prediction), cap = 3*G always suffices. */
typedef struct pq_create_label_args {
    const float* gt;
    const int32_t* counts;
    const float* anchors;    /* [9][2] (w,h), scale-major: 3 per scale */
    float* label[3];
    float* boxes[3];
    int32_t* box_counts;
    int32_t H[3], W[3], stride[3];
    int32_t B, G, C, gt_per_grid, cap;
    float iou_threshold;
} pq_create_label_args;
int pq_create_label(const pq_create_label_args* a, void* stream);

/* NMS: tools.py:540-566 torch_nms (+ torchvision batched_nms, restated).  One image per call.
 *   in  : bboxes [N][4+C] (recovered), score_thr, iou_thr
 *   out : dets [cap][6] rows (x1,y1,x2,y2,score,class) in score-descending order, *count = K (device int32)
 *   scratch sized by pq_nms_scratch_bytes(N, C).  Entirely on device, no host sync. */
typedef struct pq_nms_args {
    const float* bboxes;
    float* dets;
    int32_t* count;
    void* scratch;
    int64_t scratch_bytes;
    int32_t N, C, cap;
    float score_thr, iou_thr;
    int32_t cand_cap;    /* > 0: compact the (box, class) pairs above score_thr first and sort only cand_cap of them
                            (ordered compaction, same tie order); *count = -1 if there were more -- call again with a
                            larger value or 0.  0: sort all N*C pairs (any candidate count) */
    int32_t B;           /* > 1: a whole batch in one call: bboxes is [B][N][4+C], every image is suppressed on its own (classes and
                            the coordinate offset of batched_nms are per image, as in one torch_nms call per image); the kept
                            rows of all images come out in descending score order with their image in img[] (a stable sort by
                            image gives the per-image lists); cap / cand_cap / count then cover the whole batch */
    int32_t* img;        /* [cap] image index per kept row (B > 1), else NULL */
} pq_nms_args;
int pq_nms(const pq_nms_args* a, void* stream);
/* which sort the calling thread's last pq_nms call took: 0 = one workgroup for the whole call (capacity <= 8192), 1 = one
 * workgroup per image (batched call, capacity <= 8192 per image: the kept rows then come GROUPED BY IMAGE, descending score inside
 * an image), 2 = the library radix sorts (rows of a batch in descending score order across images) */
int32_t pq_nms_last_path(void);
int64_t pq_nms_scratch_bytes(int32_t N, int32_t C);
int64_t pq_nms_batch_scratch_bytes(int32_t B, int32_t N, int32_t C);

/* ------------------------------------------------------------------ quantisation-aware training (BASELINE config 5)
 * replaces what the reference gets from torch.ao's QAT modules (tools.py:256-290, model/parser.py:143-183,
 * model/interpreter.py:27-53): FusedMovingAvgObsFakeQuantize observers + fake-quantise (activations quint8 affine,
 * weights qint8 symmetric, per tensor) and the fused conv-BN weight scaling.  state = {min, max, scale, zero_point}. */
typedef struct pq_qat_weight_item {
    const float* w;             /* [O][K] float weights */
    const float* gamma;         /* BatchNorm weight or NULL (heads) */
    const float* running_var;
    float* wq;                  /* out: fake-quantised (w * sf), what the conv kernels read */
    float* sf;                  /* out [O]: gamma / sqrt(running_var + eps) (1 without BatchNorm) */
    float* state;               /* observer state of this weight */
    const float* dwq;           /* backward: gradient w.r.t. wq (from the wgrad kernels) */
    float* dw;                  /* backward out: gradient w.r.t. w */
    float* dgamma;              /* backward: += contribution through sf (after the BatchNorm's own dgamma) */
    float eps;
    int32_t O, K;
} pq_qat_weight_item;
typedef struct pq_qat_weight_args {
    const pq_qat_weight_item* items;   /* device table, one entry per conv layer */
    int32_t n_items, observe, quantise;
    float avg_const;
} pq_qat_weight_args;
int pq_qat_weight_prep(const pq_qat_weight_args* a, void* stream);   /* forward: all layers, one launch */
int pq_qat_weight_grad(const pq_qat_weight_args* a, void* stream);   /* backward: dw = dwq*mask*sf, dgamma += ..., one launch */

/* out[r][c] = fq(act(scale[c]*x[r][c] + shift[c])) with the observer of `state` updated first (observe != 0) */
typedef struct pq_qact_args {
    pq_src in;
    float* out;              /* [rows][ldo] (flat: [rows]); NULL: only update the observer (consumers quantise on load) */
    float* state;
    float* partial;          /* >= 2 * pq_qact_blocks(elements) floats */
    int64_t rows;
    int32_t C, ldo, flat, observe, quantise;
    float avg_const;
} pq_qact_args;
int pq_qact_fwd(const pq_qact_args* a, void* stream);
int32_t pq_qact_blocks(int64_t elems);
/* g = gq * [value inside the quantisation range] * act'(u), written through dx (which also yields the BN-backward sums) */
typedef struct pq_qact_bwd_args {
    const float* gq;         /* gradient w.r.t. the quantised tensor, pitch ldgq */
    pq_gsink dx;             /* g, y, scale, shift, mean, invstd, partial ([pq_ew_blocks(rows, C)][2][ldp]); act */
    const float* state;
    int32_t rows, C, ldgq, quantise;
} pq_qact_bwd_args;
int pq_qact_bwd(const pq_qact_bwd_args* a, void* stream);

/* ------------------------------------------------------------------ batched 2-D transposes (weights for the dgrad GEMMs)
 * dst[c][r] = src[r][c] for every item, ONE launch for the whole table (the table itself lives in device memory);
 * dst rows have a pitch of ceil4(rows) floats (16-byte aligned rows; the pad is never written).
 * tile0 = number of 32x32 tiles of all previous items, tiles_c = ceil(cols/32). */
typedef struct pq_transpose_item {
    const float* src;    /* [rows][cols] */
    float* dst;          /* [cols][ceil4(rows)] */
    int32_t rows, cols, tile0, tiles_c;
} pq_transpose_item;
typedef struct pq_transpose_args {
    const pq_transpose_item* items;   /* device pointer */
    int32_t n_items, total_tiles;
} pq_transpose_args;
int pq_transpose_batch(const pq_transpose_args* a, void* stream);

/* ------------------------------------------------------------------ batched partial-sum reduction (weight gradients)
 * Every weight-gradient entry splits its contraction over workgroups and leaves one partial slab per split in its `scratch`; the
 * sum over the splits used to be a launch of its own per layer (83 launches per mobilenetv2-fpn step).  DEFERRED form: call the
 * entry with dw == NULL (pq_conv1x1_wgrad, pq_conv1x1_bwd, pq_gconv3x3_wgrad; for the walking depthwise backward simply do not
 * issue part 6) -- the slabs stay in `scratch` -- and sum the slabs of MANY layers with ONE pq_reduce_batch launch later (the
 * engine: once per group of layers, in front of the gradient-ready mark of a DDP bucket).  pq_*_reduce_item(args with the real dw)
 * describes the slabs the entry leaves behind (pure host function, no launch); pq_reduce_batch_plan assigns the workgroups.
 * out[e] = sum_{s < splits} slab[(e / ng) * gstride + s * stride + e % ng]   in a FIXED order (bit-identical replays).
 * Work mirrored: the weight gradients torch.autograd accumulates behind trainer.py:231-233. */
typedef struct pq_reduce_item {
    const float* slab;
    float* out0;         /* columns [0, n0) */
    float* out1;         /* columns [n0, n0 + n1) or NULL (bias gradient column of pq_conv1x1_wgrad) */
    int64_t stride;      /* floats between the partial rows of one column group */
    int64_t gstride;     /* floats between column groups (grouped 3x3: one group of gw output channels) */
    int32_t splits, n0, n1;
    int32_t ng;          /* columns per group (n0 + n1 when there is one group) */
    int32_t perm_gw;     /* > 0: grouped-3x3 layout, column k = tap * gw + i of row o of its group -> out0[((g*gw + o) * gw + i) * 9 + tap] */
    int32_t lanes;       /* split lanes per column, 4 or 16 (pq_reduce_batch_plan) */
    int32_t block0;      /* first workgroup of the item (pq_reduce_batch_plan) */
    int32_t pad_;        /* floats per column unit, 1 or 4 (pq_reduce_batch_plan) */
} pq_reduce_item;
typedef struct pq_reduce_batch_args {
    const pq_reduce_item* items;      /* DEVICE copy of the planned table */
    int32_t n_items, n_blocks;
} pq_reduce_batch_args;
int32_t pq_reduce_batch_plan(pq_reduce_item* host_items, int32_t n_items);     /* fills lanes / block0; returns n_blocks */
int pq_reduce_batch(const pq_reduce_batch_args* a, void* stream);
int pq_conv1x1_wgrad_reduce_item(const pq_conv1x1_wgrad_args* a, pq_reduce_item* out);
int pq_conv1x1_bwd_reduce_item(const pq_conv1x1_bwd_args* a, pq_reduce_item* out);
int pq_dw3x3_bwd_reduce_item(const pq_dw3x3_bwd_args* a, pq_reduce_item* out);         /* a: the part-6 arguments */
int pq_gconv3x3_wgrad_reduce_item(const pq_gconv3x3_wgrad_args* a, pq_reduce_item* out);

/* ------------------------------------------------------------------ batched executor
 * The engine compiles one training / inference step into an array of pq_op and replays it with ONE host call. */
enum {
    PQ_OP_CONV1X1_FWD = 1, PQ_OP_CONV1X1_DGRAD, PQ_OP_CONV1X1_WGRAD, PQ_OP_DW_FWD, PQ_OP_DW_DGRAD, PQ_OP_DW_WGRAD,
    PQ_OP_STEM_FWD, PQ_OP_STEM_WGRAD, PQ_OP_BN_FWD_FIN, PQ_OP_BN_BWD_FIN, PQ_OP_COPY_FWD, PQ_OP_ROUTE_BWD,
    PQ_OP_DECODE, PQ_OP_LOSS_FWD, PQ_OP_LOSS_BWD, PQ_OP_ADAM, PQ_OP_FILL, PQ_OP_DW_BWD,
    PQ_OP_GCONV_FWD, PQ_OP_GCONV_DGRAD, PQ_OP_GCONV_WGRAD, PQ_OP_TRANSPOSE,
    PQ_OP_QAT_WPREP, PQ_OP_QAT_WGRAD, PQ_OP_QACT_FWD, PQ_OP_QACT_BWD, PQ_OP_POOL_REDUCE, PQ_OP_CONV1X1_BWD,
    PQ_OP_BN_EVAL,
    PQ_OP_LAYER_FWD,
    PQ_OP_XDW_FWD, PQ_OP_XSTAT_FWD, PQ_OP_XDW_BWD,
    PQ_OP_REDUCE_BATCH
};
typedef struct pq_fill_args { float* dst; float value; int64_t n; } pq_fill_args;
#define PQ_OPF_SIDE 1   /* op may run on the side stream of pq_run_ops2 (it only feeds the optimiser, e.g. weight gradients) */
#define PQ_OPF_JOIN 2   /* main-stream op that reads what side-stream ops wrote: `stream` waits for the side stream before it */
#define PQ_OPF_LAZY 4   /* with PQ_OPF_SIDE: not urgent -- the op waits for the fork of the next side op (or the next join / the end of
                         * the call) instead of forcing a fork of its own (every fork costs the main queue ~5 us: small reductions) */
#define PQ_OPF_NOFORK 8 /* with PQ_OPF_SIDE: everything the op reads was written by EARLIER SIDE ops of this call (or before the call):
                         * it is queued behind them without a fork event, i.e. it does not wait for the main stream's position
                         * (loss kernels of the side-stream detection heads: they start when their head conv is done, not when
                         * the main stream reaches the end of the forward) */
typedef struct pq_op {
    int32_t code;
    int32_t flags;
    union {
        pq_conv1x1_fwd_args c1f; pq_conv1x1_dgrad_args c1d; pq_conv1x1_wgrad_args c1w;
        pq_dw3x3_fwd_args dwf; pq_dw3x3_dgrad_args dwd; pq_dw3x3_wgrad_args dww; pq_dw3x3_bwd_args dwb;
        pq_stem3x3_fwd_args stf; pq_stem3x3_wgrad_args stw;
        pq_gconv3x3_fwd_args gcf; pq_gconv3x3_dgrad_args gcd; pq_gconv3x3_wgrad_args gcw;
        pq_bn_fwd_finalize_args bnf; pq_bn_bwd_finalize_args bnb;
        pq_copy_args cp; pq_route_bwd_args rb;
        pq_decode_args dec; pq_yolo_loss_args loss; pq_adam_args adam; pq_fill_args fill; pq_transpose_args tr;
        pq_qat_weight_args qw; pq_qact_args qa; pq_qact_bwd_args qb; pq_pool_reduce_args pr; pq_conv1x1_bwd_args c1b; pq_bn_eval_args bne;
        pq_layer_fwd_args lay;
        pq_xdw_fwd_args xdf; pq_xstat_fwd_args xsf; pq_xdw_bwd_args xdb;
        pq_reduce_batch_args red;
    } u;
} pq_op;
int pq_run_ops(const pq_op* ops, int32_t n, void* stream);
/* device address of a PINNED host allocation (hipHostGetDevicePointer): lets a kernel write a flag the host polls without a
 * device-to-host copy or an event on the stream (the NaN trap of the loss).  PQ_EUNSUPPORTED when the memory is not mapped. */
int pq_host_device_pointer(void* host_ptr, void** device_ptr);
/* two-stream replay: ops flagged PQ_OPF_SIDE are launched on `side_stream` after an event recorded on `stream` at their
 * position (they see everything launched before them); `stream` waits for the side stream at the end of the call.
 * Lets the weight-gradient GEMMs overlap the latency-bound data-gradient chain of the backward pass. */
int pq_run_ops2(const pq_op* ops, int32_t n, void* stream, void* side_stream);
/* the same without the final join: `stream` does NOT wait for `side_stream` when the call returns (PQ_OPF_JOIN ops inside still
 * do).  For callers that replay one op array in pieces and hand each piece's results to a third stream (data-parallel gradient
 * buckets: the communication stream waits for both streams, the compute streams never stall); the caller joins once at the end. */
int pq_run_ops2_nojoin(const pq_op* ops, int32_t n, void* stream, void* side_stream);
/* profiling variant: brackets every op with HIP events on `stream`, synchronises at the end and writes the elapsed
 * milliseconds of op k to ms[k] (used by bench.py for the live roofline figure; never on the timed path) */
int pq_run_ops_timed(const pq_op* ops, int32_t n, void* stream, float* ms);
/* same for the two-stream replay: every op is bracketed on the stream it is launched on, so the figures are the in-situ
 * durations (overlapping kernels share the chip), i.e. what a rocprofv3 kernel trace of the training step reports */
int pq_run_ops2_timed(const pq_op* ops, int32_t n, void* stream, void* side_stream, float* ms);
/* Test hook: while enabled, every kernel launch of the library notes "kernel expression | enclosing function" (the pretty name of a
 * templated dispatcher carries its template arguments) in a set; pq_launch_log(enable) clears the set and switches the log,
 * pq_launch_log_dump writes the entries, newline separated, and returns the bytes needed.  The suite uses it to prove that every
 * kernel instantiation the plans of the BASELINE nets launch is also launched by a value test. */
void pq_launch_log(int32_t enable);
int64_t pq_launch_log_dump(char* buf, int64_t cap);
/* index of the first failing op of the last pq_run_ops (or -1) and its message */
int pq_last_failed_op(void);
const char* pq_last_error(void);
const char* pq_version(void);
/* sizeof() of the argument struct of op `code` (0: pq_op, 100: pq_recover_args, 101: pq_nms_args,
 * 102: pq_decode_bwd_args, 103: pq_transpose_item, 104: pq_create_label_args,
 * 105: pq_qat_weight_item, 106: pq_bn_eval_item, 107: pq_bn_stat, 108: pq_src) -- lets a foreign-language binding verify its
 * struct layout at load time */
int32_t pq_abi_sizeof(int32_t code);

#ifdef __cplusplus
}
#endif
#endif
