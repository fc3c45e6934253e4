// Elementwise / reduction kernels: BatchNorm statistics finalisation (fwd + bwd), shortcut / route / upsample
// forward copy and backward routing, Adam, fill.  All HBM-bound; float4 per lane, rows contiguous across lanes.
#include "common.h"
#include "qatmath.h"

namespace pq {

// ------------------------------------------------------------------------------------------------ BN finalize
// partial is [nblk][2][ldp] (one row pair per producer row-block).  Block = 16 channels x 16 row lanes (a wave reads 64
// contiguous bytes of four partial rows); grid = (ceil(C/16), R).  Each block sums its slice of the rows with 8 independent
// loads in flight per thread: the kernel is pure load latency and sits on the layer-to-layer critical path, so the rows
// are spread over many lanes (a 128-row list is ONE batch of loads per thread; 64 x 4 lanes needed four dependent rounds).  R > 1 (long partial lists, up to 8192 rows for the
// 1M-pixel layers): slices go to a double scratch, an agent-scope ticket elects the last block of a channel group, which
// adds the R slices in fixed order (deterministic) and finalises; it also re-arms the ticket for the next launch.
constexpr int BNF_CH = 16, BNF_RL = 16, BNF_UNROLL = 8, BNF_MAXR = 64;

__device__ __forceinline__ bool bn_reduce_partials(const float* partial, int nblk, int ldp, int C, double* scratch,
                                                    unsigned* tickets, double& t1, double& t2, int planes = 2) {
    __shared__ double sh[2][BNF_RL][BNF_CH];
    __shared__ unsigned s_ticket;
    const int cl = threadIdx.x & (BNF_CH - 1), rl = threadIdx.x / BNF_CH;
    const int cg = blockIdx.x, c = cg * BNF_CH + cl;
    const int R = gridDim.y, r = blockIdx.y;
    const int per = (nblk + R - 1) / R;
    const int k0 = r * per, k1 = min(nblk, k0 + per);
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int k = k0 + rl;
        for (; k + (BNF_UNROLL - 1) * BNF_RL < k1; k += BNF_UNROLL * BNF_RL) {
            float a[BNF_UNROLL], b[BNF_UNROLL];
#pragma unroll
            for (int u = 0; u < BNF_UNROLL; ++u) {
                const float* row = partial + (size_t)(k + u * BNF_RL) * planes * ldp;
                a[u] = row[c];
                b[u] = row[ldp + c];
            }
#pragma unroll
            for (int u = 0; u < BNF_UNROLL; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
        }
        for (; k < k1; k += BNF_RL) {
            const float* row = partial + (size_t)k * planes * ldp;
            s1 += (double)row[c];
            s2 += (double)row[ldp + c];
        }
    }
    sh[0][rl][cl] = s1;
    sh[1][rl][cl] = s2;
    __syncthreads();
    t1 = t2 = 0.0;
    if (rl == 0) {
#pragma unroll
        for (int q = 0; q < BNF_RL; ++q) { t1 += sh[0][q][cl]; t2 += sh[1][q][cl]; }
    }
    if (R == 1) return rl == 0 && c < C;
    double* mine = scratch + ((size_t)(cg * R + r) * 2) * BNF_CH;
    if (rl == 0) { mine[cl] = t1; mine[BNF_CH + cl] = t2; }
    __threadfence();                                   // release the slice (agent scope: L2s are per XCD)
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(&tickets[cg], 1u);
    __syncthreads();
    if (s_ticket != (unsigned)(R - 1)) return false;   // uniform per block
    __threadfence();                                   // acquire the other blocks' slices
    // every row lane adds a quarter of the slices (independent loads in flight), then the four sums are combined in
    // fixed order: the result does not depend on which block came last
    double u1 = 0.0, u2 = 0.0;
    {
        int q = rl;
        for (; q + 3 * BNF_RL < R; q += 4 * BNF_RL) {
            double x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double* sl = scratch + ((size_t)(cg * R + q + u * BNF_RL) * 2) * BNF_CH;
                x[u] = sl[cl];
                y[u] = sl[BNF_CH + cl];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { u1 += x[u]; u2 += y[u]; }
        }
        for (; q < R; q += BNF_RL) {
            const double* sl = scratch + ((size_t)(cg * R + q) * 2) * BNF_CH;
            u1 += sl[cl];
            u2 += sl[BNF_CH + cl];
        }
    }
    __syncthreads();
    sh[0][rl][cl] = u1;
    sh[1][rl][cl] = u2;
    __syncthreads();
    if (rl == 0) {
        t1 = t2 = 0.0;
#pragma unroll
        for (int q = 0; q < BNF_RL; ++q) { t1 += sh[0][q][cl]; t2 += sh[1][q][cl]; }
    }
    if (threadIdx.x == 0) tickets[cg] = 0u;            // re-arm (the next launch is stream-ordered after this one)
    return rl == 0 && c < C;
}

// QAT: per-channel extremes of the raw conv outputs from planes 2 / 3 of 4-plane partial rows (one block walks all rows; a
// separate walk after the sums: one combined walk with four loads per row measured 1.39 against 1.07 ms per step).
__device__ __forceinline__ void bn_reduce_minmax(const float* partial, int nblk, int ldp, int C, float& lo, float& hi) {
    __shared__ float shm2[2][BNF_RL][BNF_CH];
    const int cl = threadIdx.x & (BNF_CH - 1), rl = threadIdx.x / BNF_CH;
    const int c = blockIdx.x * BNF_CH + cl;
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    if (c < C) {
        int k = rl;
        for (; k + (BNF_UNROLL - 1) * BNF_RL < nblk; k += BNF_UNROLL * BNF_RL) {
            float a[BNF_UNROLL], b[BNF_UNROLL];
#pragma unroll
            for (int u = 0; u < BNF_UNROLL; ++u) {
                const float* row = partial + (size_t)(k + u * BNF_RL) * 4 * ldp;
                a[u] = row[2 * ldp + c];
                b[u] = row[3 * ldp + c];
            }
#pragma unroll
            for (int u = 0; u < BNF_UNROLL; ++u) { mn = fminf(mn, a[u]); mx = fmaxf(mx, b[u]); }
        }
        for (; k < nblk; k += BNF_RL) {
            const float* row = partial + (size_t)k * 4 * ldp;
            mn = fminf(mn, row[2 * ldp + c]);
            mx = fmaxf(mx, row[3 * ldp + c]);
        }
    }
    shm2[0][rl][cl] = mn;
    shm2[1][rl][cl] = mx;
    __syncthreads();
    lo = __builtin_huge_valf(); hi = -__builtin_huge_valf();
    if (rl == 0) {
#pragma unroll
        for (int q = 0; q < BNF_RL; ++q) { lo = fminf(lo, shm2[0][q][cl]); hi = fmaxf(hi, shm2[1][q][cl]); }
    }
}
// float <-> unsigned key with the same order (atomicMin / atomicMax on the keys = min / max of the floats, any arrival order)
__device__ __forceinline__ unsigned f2key(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(BNF_CH * BNF_RL) void bn_fwd_finalize_kernel(pq_bn_fwd_finalize_args a) {
    const int c = blockIdx.x * BNF_CH + (threadIdx.x & (BNF_CH - 1));
    if (a.train) {
        double t1, t2;
        // QAT with the observer folded in (4-plane rows, gridDim.y == 1): raw per-channel extremes first, by every lane
        const bool mmode = a.planes == 4 && a.qstate != nullptr;
        float rlo = 0.f, rhi = 0.f;
        if (mmode) bn_reduce_minmax(a.partial, a.nblk, a.ldp, a.C, rlo, rhi);
        const bool owner = bn_reduce_partials(a.partial, a.nblk, a.ldp, a.C, a.scratch, a.tickets, t1, t2, a.planes == 4 ? 4 : 2);
        if (!owner && !mmode) return;
        float vlo = __builtin_huge_valf(), vhi = -__builtin_huge_valf();
        if (owner) {
            const double n = (double)a.count;
            const double mean = t1 / n;
            double var = t2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
            // QAT: the statistics are those of out = conv_orig * sf.  BatchNorm on conv_orig = out / sf equals BatchNorm on `out`
            // with eps scaled by sf^2 (and the sign of sf), so everything downstream keeps working on the raw tensor.
            const double sf = a.sf ? (double)a.sf[c] : 1.0;
            const float inv = (float)((sf < 0.0 ? -1.0 : 1.0) / sqrt(var + (double)a.eps * sf * sf));
            const float g = a.gamma[c], b = a.beta[c];
            a.mean[c] = (float)mean;
            a.invstd[c] = inv;
            const float sc = g * inv;
            const float sh = b - (float)mean * sc;
            a.scale[c] = sc;
            a.shift[c] = sh;
            if (a.running_mean) {
                const double unb = (n > 1.0 ? var * n / (n - 1.0) : var) / (sf * sf);
                a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)(mean / sf);
                a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
            }
            if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
            if (mmode) {
                // the observed tensor is act(sc * raw + sh): monotone per channel, so its extremes are the images of the raw ones
                // (same fmaf + activation as every consumer and as qact_minmax4_kernel: bit-identical to a pass over the tensor)
                const ActParam act = act_param(a.act);
                const float e0 = act_fwd(fmaf(sc, rlo, sh), act), e1 = act_fwd(fmaf(sc, rhi, sh), act);
                vlo = fminf(e0, e1); vhi = fmaxf(e0, e1);
            }
        }
        if (mmode) {
            // block extremes -> two order-preserving atomics on the layer's scratch; the last block of the launch (ticket) feeds the
            // moving-average observer and re-arms the scratch.  mm = {key of the minimum, key of the maximum, ticket}
            __shared__ float sv[2][BNF_CH];
            __shared__ unsigned s_last;
            __syncthreads();
            if (threadIdx.x < BNF_CH) { sv[0][threadIdx.x] = vlo; sv[1][threadIdx.x] = vhi; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float lo = sv[0][0], hi = sv[1][0];
#pragma unroll
                for (int q = 1; q < BNF_CH; ++q) { lo = fminf(lo, sv[0][q]); hi = fmaxf(hi, sv[1][q]); }
                atomicMin(&a.mm[0], f2key(lo));
                atomicMax(&a.mm[1], f2key(hi));
                __threadfence();
                s_last = atomicAdd(&a.mm[2], 1u);
                if (s_last == gridDim.x - 1) {
                    __threadfence();
                    const float tlo = key2f(atomicMax(&a.mm[0], 0xffffffffu)), thi = key2f(atomicMin(&a.mm[1], 0u));   // read + re-arm
                    a.mm[2] = 0u;
                    qat_observe(a.qstate, tlo, thi, 0, 255, false, a.avg_const);
                }
            }
        }
    } else if (threadIdx.x < BNF_CH && blockIdx.y == 0 && c < a.C) {
        const float inv = 1.f / sqrtf(a.running_var[c] + a.eps);
        const float sf = a.sf ? a.sf[c] : 1.f;              // QAT: the raw tensor is conv_orig * sf
        a.mean[c] = a.running_mean[c] * sf;
        a.invstd[c] = inv / sf;
        a.scale[c] = a.gamma[c] * inv / sf;
        a.shift[c] = a.beta[c] - a.running_mean[c] * (a.gamma[c] * inv);
    }
}

// every BatchNorm of a net from its running statistics, one workgroup per layer (same expressions as the eval branch above)
__global__ __launch_bounds__(256) void bn_eval_batch_kernel(const pq_bn_eval_item* items) {
    const pq_bn_eval_item it = items[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256) {
        const float inv = 1.f / sqrtf(it.running_var[c] + it.eps);
        const float sf = it.sf ? it.sf[c] : 1.f;
        it.mean[c] = it.running_mean[c] * sf;
        it.invstd[c] = inv / sf;
        it.scale[c] = it.gamma[c] * inv / sf;
        it.shift[c] = it.beta[c] - it.running_mean[c] * (it.gamma[c] * inv);
    }
}

__global__ __launch_bounds__(BNF_CH * BNF_RL) void bn_bwd_finalize_kernel(pq_bn_bwd_finalize_args a) {
    const int c = blockIdx.x * BNF_CH + (threadIdx.x & (BNF_CH - 1));
    double t1, t2;
    if (!bn_reduce_partials(a.partial, a.nblk, a.ldp, a.C, a.scratch, a.tickets, t1, t2)) return;
    const float inv = a.invstd[c];
    const float pc = a.gamma[c] * inv;
    a.dbeta[c] = (float)t1;
    a.dgamma[c] = (float)t2;
    a.p[c] = pc;
    if (a.train) {
        a.q[c] = (float)(-(double)pc * (double)inv * t2 / (double)a.count);
        a.r[c] = (float)(-(double)pc * t1 / (double)a.count);
    } else {   // statistics frozen: dy = gamma*invstd*g
        a.q[c] = 0.f;
        a.r[c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ copy / route
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// 4 consecutive channels at an address that may not be 16-byte aligned / may straddle the end of the channel range
// (concatenations of tensors whose channel counts are not multiples of 4): lanes with c0 + j >= C are not touched
__device__ __forceinline__ float4 ldg4_masked(const float* p, int c0, int C) {
    return make_float4(c0 < C ? p[0] : 0.f, c0 + 1 < C ? p[1] : 0.f, c0 + 2 < C ? p[2] : 0.f, c0 + 3 < C ? p[3] : 0.f);
}
__device__ __forceinline__ void stg4_masked(float* p, float4 v, int c0, int C) {
    if (c0 < C) p[0] = v.x;
    if (c0 + 1 < C) p[1] = v.y;
    if (c0 + 2 < C) p[2] = v.z;
    if (c0 + 3 < C) p[3] = v.w;
}

struct CopyP { pq_copy_args a; int C4; int unal; };

__global__ __launch_bounds__(256) void copy_fwd_kernel(CopyP p) {
    const pq_copy_args& a = p.a;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + threadIdx.x, c0 = c4 * 4;
    // (with a.a.stat / a.b.stat: row lane 0 derives its quad's coefficients from the producer's accumulators and shares them
    // through LDS; in the first row block it also stores the vectors for every later consumer)
    __shared__ float cshare[2][64][8];
    const bool cok = c4 < p.C4;
    const bool lead = cok && threadIdx.y == 0;
    const SrcCoef4 ka = src_coef4_shared(a.a, cok ? c0 : 0, a.C, lead, bx == 0, cshare[0][threadIdx.x]);
    const SrcCoef4 kb = src_coef4_shared(a.b, cok ? c0 : 0, a.C, lead, bx == 0, cshare[1][threadIdx.x]);
    if (!cok) return;
    const bool two = a.b.x != nullptr;
    const int rows = a.B * a.H * a.W;
    const int SH = a.H / a.up, SW = a.W / a.up;
    // four rows per step, all their loads before the first store (the compiler may not move a load above a store that could
    // alias it, so a one-row loop exposes a full memory round trip per row)
    constexpr int U = 4;
    const int stride = gridDim.x * blockDim.y;
    for (int r0 = bx * blockDim.y + threadIdx.y; r0 < rows; r0 += U * stride) {
        float4 va[U], vb[U], vz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = min(r0 + u * stride, rows - 1);
            size_t sr = (size_t)r;
            if (a.up > 1) {
                const int x = (int)(r % a.W);
                const int t2 = r / a.W;
                const int y = (int)(t2 % a.H);
                const int b = t2 / a.H;
                sr = ((size_t)b * SH + y / a.up) * SW + x / a.up;
            }
            va[u] = ldg4(a.a.x + sr * a.a.ld + c0);
            vb[u] = two ? ldg4(a.b.x + (size_t)r * a.b.ld + c0) : f4zero();
            vz[u] = a.gate ? ldg4(a.gate + (size_t)(r / (a.H * a.W)) * a.ldgate + c0) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * stride;
            if (r >= rows) break;
            float4 v = src_apply4(va[u], ka);
            if (two) v = f4add(v, src_apply4(vb[u], kb));
            if (a.gate) {                                   // scale_channels: multiply by the per-image channel gate sigmoid(z)
                const float4 z = vz[u];
                v = make_float4(v.x * sigmoidf_(z.x), v.y * sigmoidf_(z.y), v.z * sigmoidf_(z.z), v.w * sigmoidf_(z.w));
            }
            float* dst = a.out + (size_t)r * a.ldo + a.coff + c0;
            if (p.unal) stg4_masked(dst, v, c0, a.C); else stg4(dst, v);
        }
    }
}

struct RouteP { pq_route_bwd_args a; int C4; int unal; };

__global__ __launch_bounds__(256) void route_bwd_kernel(RouteP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pq_route_bwd_args& a = p.a;
    int bx, by;
    xcd_block(bx, by);
    const int c4 = by * blockDim.x + threadIdx.x, c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const SinkCoef4 sk = sink_coef4(a.dx, c0, a.C);
    const int rows = a.B * a.H * a.W;
    const int OH = a.H * a.up, OW = a.W * a.up;
    float4 s1 = f4zero(), s2 = f4zero();
    const int stride = gridDim.x * blockDim.y;
    if (a.dx2.g != nullptr) {
        // shortcut add: both inputs receive the output's gradient -- one read of gout, two sinks (each with its own accumulate /
        // activation derivative / statistics), four rows per step with every operand in flight before the first store
        const SinkCoef4 sk2 = sink_coef4(a.dx2, c0, a.C);
        float4 t1 = f4zero(), t2 = f4zero();
        constexpr int U = 4;
        for (int r0 = bx * blockDim.y + threadIdx.y; r0 < rows && cok; r0 += U * stride) {
            float4 dz[U];
            SinkOps so[U], so2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = (size_t)min(r0 + u * stride, rows - 1);
                dz[u] = ldg4(a.gout + r * a.ldgo + a.coff + c0);
                so[u] = sink_load4(a.dx, sk, r, c0);
                so2[u] = sink_load4(a.dx2, sk2, r, c0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * stride;
                if (r >= rows) break;
                sink_store4(a.dx, sk, (size_t)r, c0, dz[u], so[u], s1, s2);
                sink_store4(a.dx2, sk2, (size_t)r, c0, dz[u], so2[u], t1, t2);
            }
        }
        float* part1 = (a.dx.finalize && a.dx.mean) ? a.dx.partial : nullptr;
        float* part2 = (a.dx2.finalize && a.dx2.mean) ? a.dx2.partial : nullptr;
        if (part1) block_reduce_store_stats<256>(s1, s2, part1, a.dx.ldp, c0, a.C, lds, bx, 2, sink_acc(a.dx));
        if (part1 && part2) __syncthreads();
        if (part2) block_reduce_store_stats<256>(t1, t2, part2, a.dx2.ldp, c0, a.C, lds, bx, 2, sink_acc(a.dx2));
        return;
    }
    if ((a.dx.finalize & 2) && !a.bcast && a.up == 1 && !a.gate && !p.unal) {
        // statistics only (finalize = 3): g already holds this gradient (it IS gout: the conv branch of a shortcut with an activation
        // uses the sum's gradient buffer), only the BatchNorm-backward sums over (gout, y) are wanted -- two reads per element, no store
        constexpr int U = 4;
        for (int r0 = bx * blockDim.y + threadIdx.y; r0 < rows && cok; r0 += U * stride) {
            float4 dz[U], yv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = (size_t)min(r0 + u * stride, rows - 1);
                dz[u] = ldg4(a.gout + r * a.ldgo + a.coff + c0);
                yv[u] = ldg4(a.dx.y + r * (size_t)a.dx.ldy + c0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r0 + u * stride < rows) (void)sink_apply4(a.dx, sk, dz[u], yv[u], s1, s2);
        }
    } else if (!a.bcast && a.up == 1 && !a.gate && !p.unal) {
        // the common case (shortcut / route / strided-projection finalize): four rows per step, the upstream gradient AND the
        // sink's operands of all four in flight before the first store
        constexpr int U = 4;
        for (int r0 = bx * blockDim.y + threadIdx.y; r0 < rows && cok; r0 += U * stride) {
            float4 dz[U];
            SinkOps so[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = (size_t)min(r0 + u * stride, rows - 1);
                dz[u] = ldg4(a.gout + r * a.ldgo + a.coff + c0);
                so[u] = sink_load4(a.dx, sk, r, c0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * stride;
                if (r >= rows) break;
                sink_store4(a.dx, sk, (size_t)r, c0, dz[u], so[u], s1, s2);
            }
        }
    } else
    for (int r = bx * blockDim.y + threadIdx.y; r < rows && cok; r += stride) {
        float4 dz = f4zero();
        if (a.bcast) {                                  // global average pool: every pixel of image b receives gout[b] * mul
            const float* gp = a.gout + (size_t)(r / (a.H * a.W)) * a.ldgo + a.coff + c0;
            const float4 gb = p.unal ? ldg4_masked(gp, c0, a.C) : ldg4(gp);
            dz = make_float4(gb.x * a.mul, gb.y * a.mul, gb.z * a.mul, gb.w * a.mul);
        } else if (a.up == 1) {
            const float* gp = a.gout + (size_t)r * a.ldgo + a.coff + c0;
            dz = p.unal ? ldg4_masked(gp, c0, a.C) : ldg4(gp);
            if (a.gate) {                               // scale_channels: d(feature) = gout * sigmoid(z)
                const float4 z = ldg4(a.gate + (size_t)(r / (a.H * a.W)) * a.ldgate + c0);
                dz = make_float4(dz.x * sigmoidf_(z.x), dz.y * sigmoidf_(z.y), dz.z * sigmoidf_(z.z), dz.w * sigmoidf_(z.w));
            }
        } else {
            const int x = (int)(r % a.W);
            const int t2 = r / a.W;
            const int y = (int)(t2 % a.H);
            const int b = t2 / a.H;
            for (int dy = 0; dy < a.up; ++dy)
                for (int dx = 0; dx < a.up; ++dx) {
                    const size_t orow = ((size_t)b * OH + (size_t)y * a.up + dy) * OW + (size_t)x * a.up + dx;
                    const float* gp = a.gout + orow * a.ldgo + a.coff + c0;
                    dz = f4add(dz, p.unal ? ldg4_masked(gp, c0, a.C) : ldg4(gp));
                }
        }
        sink_emit4(a.dx, sk, (size_t)r, c0, dz, s1, s2);
    }
    float* part = (a.dx.finalize && a.dx.mean) ? a.dx.partial : nullptr;
    if (part) block_reduce_store_stats<256>(s1, s2, part, a.dx.ldp, c0, a.C, lds, bx, 2, sink_acc(a.dx));
}

// ------------------------------------------------------------------------------------------------ pooled reductions
// out[b][c] = mul * sum_p a'[b,p,c]                                   (global average pool, model/parser.py avgpool)
// out[b][c] = s(1-s) * sum_p gout[b,p,c] * a'[b,p,c],  s = sigmoid(gate[b][c])   (gradient of a scale_channels gate)
struct PoolP { pq_pool_reduce_args a; int C4; };

__global__ __launch_bounds__(256) void pool_reduce_kernel(PoolP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pq_pool_reduce_args& a = p.a;
    const int b = blockIdx.x;
    const int c4 = blockIdx.y * blockDim.x + threadIdx.x, c0 = c4 * 4;
    const bool cok = c4 < p.C4;
    const SrcCoef4 ka = src_coef4(a.a, c0, a.C);
    float4 acc = f4zero();
    if (cok)
        for (int q = threadIdx.y; q < a.HW; q += blockDim.y) {
            const size_t r = (size_t)b * a.HW + q;
            float4 v = src_apply4(ldg4(a.a.x + r * a.a.ld + c0), ka);
            if (a.gout) {
                const float4 g = ldg4(a.gout + r * a.ldgo + c0);
                v = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
            }
            acc = f4add(acc, v);
        }
    float* slot = lds + (size_t)(threadIdx.y * blockDim.x + threadIdx.x) * 4;
    slot[0] = acc.x; slot[1] = acc.y; slot[2] = acc.z; slot[3] = acc.w;
    __syncthreads();
    if (threadIdx.y == 0 && cok) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < (int)blockDim.y; ++y) {
            const float* o = lds + (size_t)(y * blockDim.x + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] += o[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c0 + j >= a.C) break;
            float v = t[j] * a.mul;
            if (a.gate) { const float s = sigmoidf_(a.gate[(size_t)b * a.ldgate + c0 + j]); v *= s * (1.f - s); }
            a.out[(size_t)b * a.ldo + c0 + j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ batched transpose
__global__ __launch_bounds__(256) void transpose_batch_kernel(const pq_transpose_item* items, int n) {
    __shared__ float tile[32][33];
    const int t = blockIdx.x;
    int lo = 0, hi = n - 1;                      // last item whose tile0 <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    const pq_transpose_item it = items[lo];
    const int local = t - it.tile0;
    const int r0 = (local / it.tiles_c) * 32, c0 = (local % it.tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < it.rows && c < it.cols) ? it.src[(size_t)r * it.cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < it.cols && r < it.rows) it.dst[(size_t)c * ((it.rows + 3) & ~3) + r] = tile[tx][ty + 8 * k];   // rows padded to 16 B
    }
}

// ------------------------------------------------------------------------------------------------ Adam etc.
__global__ __launch_bounds__(256) void adam_kernel(pq_adam_args a) {
    const int64_t n4 = a.n / 4;
    const float step = a.lr / a.bias_correction1;
    const float rs2 = 1.f / sqrtf(a.bias_correction2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p4 = ldg4(a.param + 4 * i), g4 = ldg4(a.grad + 4 * i), m4 = ldg4(a.exp_avg + 4 * i), v4 = ldg4(a.exp_avg_sq + 4 * i);
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g = gg[j] * a.grad_scale;
            if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, pp[j], g);
            mm[j] = a.beta1 * mm[j] + (1.f - a.beta1) * g;
            vv[j] = a.beta2 * vv[j] + (1.f - a.beta2) * g * g;
            const float denom = sqrtf(vv[j]) * rs2 + a.eps;
            pp[j] -= step * (mm[j] / denom);
        }
        stg4(a.param + 4 * i, make_float4(pp[0], pp[1], pp[2], pp[3]));
        stg4(a.exp_avg + 4 * i, make_float4(mm[0], mm[1], mm[2], mm[3]));
        stg4(a.exp_avg_sq + 4 * i, make_float4(vv[0], vv[1], vv[2], vv[3]));
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float g = a.grad[i] * a.grad_scale;
        if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, a.param[i], g);
        const float m = a.beta1 * a.exp_avg[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.exp_avg_sq[i] + (1.f - a.beta2) * g * g;
        a.exp_avg[i] = m;
        a.exp_avg_sq[i] = v;
        a.param[i] -= step * (m / (sqrtf(v) * rs2 + a.eps));
    }
}

__global__ void fill_kernel(float* dst, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void sign_l1_kernel(float* grad, const float* w, int64_t n, float ratio) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = w[i];
        grad[i] += ratio * (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
    }
}

}  // namespace pq

using namespace pq;

static int bn_row_groups(int nblk, bool have_scratch) {
    // One block per 16 channels walks the whole partial list up to 4096 rows: measured on the headline step with a limit of
    // 256 -> 11.54 ms, 1024 -> 11.39, 2048/4096/8192 -> 11.35 (the ticket + two device fences of the multi-block path cost more
    // than 16 lanes reading 4096 rows from L2).  Longer lists split the rows.
    const int r1max = 4096;
    if (!have_scratch || nblk <= r1max) return 1;
    const int r = (nblk + 127) / 128;
    return r > BNF_MAXR ? BNF_MAXR : r;
}
extern "C" int64_t pq_bn_finalize_scratch_bytes(int32_t C) { return (int64_t)((C + BNF_CH - 1) / BNF_CH) * BNF_MAXR * 2 * BNF_CH * 8; }
extern "C" int32_t pq_bn_finalize_tickets(int32_t C) { return (C + BNF_CH - 1) / BNF_CH; }

extern "C" int pq_bn_fwd_finalize(const pq_bn_fwd_finalize_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->gamma && a->beta && a->mean && a->invstd && a->scale && a->shift, "null pointer");
    PQ_CHECK_ARG(a->C > 0, "empty");
    PQ_CHECK_ARG(!a->train || (a->partial && a->nblk > 0 && a->count > 0.f), "train mode needs partial sums");
    PQ_CHECK_ARG(a->train || (a->running_mean && a->running_var), "eval mode needs running statistics");
    const bool mmode = a->train && a->planes == 4 && a->qstate != nullptr;
    PQ_CHECK_ARG(a->planes == 0 || a->planes == 2 || a->planes == 4, "planes must be 2 or 4");
    PQ_CHECK_ARG(!mmode || a->mm, "the folded observer needs its 3-word scratch");
    PQ_LAUNCH(bn_fwd_finalize_kernel, dim3((a->C + BNF_CH - 1) / BNF_CH, (a->train && !mmode) ? bn_row_groups(a->nblk, a->scratch && a->tickets) : 1),
                       dim3(BNF_CH * BNF_RL), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_bn_eval_batch(const pq_bn_eval_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->items && a->n_items > 0, "bad args");
    PQ_LAUNCH(bn_eval_batch_kernel, dim3(a->n_items), dim3(256), 0, (hipStream_t)stream, a->items);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

// test hook: thread i adds values[i] as the partial of row block i (plane 0, channel 0) -- the producers' bn_stat_add, nothing else
__global__ void bn_stat_selftest_kernel(const float* v, int n, int64_t* acc, uint32_t* bad, int slots, int ldp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bn_stat_add(acc, bad, slots, ldp, i, 0, 0, v[i]);
}
extern "C" int pq_bn_stat_selftest(const float* values, int32_t n, int32_t slots, int32_t ldp, int64_t* acc, uint32_t* bad, void* stream) {
    PQ_CHECK_ARG(values && acc && bad && n > 0 && slots >= 1 && slots <= 4 && ldp >= 1, "bad arguments");
    if (!stat_adds_ok(n, slots)) { pq_set_error("pq_bn_stat_selftest: more than %d partials per accumulator slot", PQ_STAT_MAX_ADDS); return PQ_EUNSUPPORTED; }
    PQ_LAUNCH(bn_stat_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, values, n, acc, bad, slots, ldp);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_bn_bwd_finalize(const pq_bn_bwd_finalize_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->partial && a->gamma && a->invstd && a->dgamma && a->dbeta && a->p && a->q && a->r, "null pointer");
    PQ_CHECK_ARG(a->C > 0 && a->nblk > 0 && a->count > 0.f, "empty");
    PQ_LAUNCH_BINDABLE(bn_bwd_finalize_kernel, dim3((a->C + BNF_CH - 1) / BNF_CH, bn_row_groups(a->nblk, a->scratch && a->tickets)),
                       dim3(BNF_CH * BNF_RL), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_copy_fwd(const pq_copy_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->a.x && a->out, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->up >= 1, "bad shape");
    PQ_CHECK_ARG(a->ldo % 4 == 0 && a->a.ld % 4 == 0 && (!a->b.x || a->b.ld % 4 == 0), "pitches must be multiples of 4");
    PQ_CHECK_ARG(!a->gate || (a->C % 4 == 0 && a->coff % 4 == 0), "gated copy needs channel count / offset multiples of 4");
    PQ_CHECK_ARG(a->H % a->up == 0 && a->W % a->up == 0, "output size must be a multiple of the upsample factor");
    PQ_CHECK_ARG(!(a->b.x && a->up != 1), "second addend needs up==1");
    if (act_is_special(a->a.act) || (a->b.x && act_is_special(a->b.act))) { pq_set_error("pq_copy_fwd: logistic/tanh operands are not supported yet"); return PQ_EUNSUPPORTED; }
    CopyP p{*a, (a->C + 3) / 4, ((a->C & 3) || (a->coff & 3)) ? 1 : 0};   // odd widths / offsets: masked scalar stores
    EwGeom g = ew_geom((long)a->B * a->H * a->W, a->C);
    PQ_LAUNCH(copy_fwd_kernel, dim3(g.gx, g.gy), dim3(g.cx, g.py), 0, (hipStream_t)stream, p);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_route_bwd(const pq_route_bwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->gout && a->dx.g, "null pointer");
    PQ_SINK_ADDEND_CHECK(a->dx);
    if (a->dx2.g) PQ_SINK_ADDEND_CHECK(a->dx2);
    PQ_CHECK_ARG(a->B > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->up >= 1, "bad shape");
    PQ_CHECK_ARG(a->ldgo % 4 == 0 && a->dx.ldg % 4 == 0, "pitches must be multiples of 4");
    PQ_CHECK_ARG(!(a->gate || a->bcast) || (a->C % 4 == 0 && a->coff % 4 == 0), "gated / broadcast routing needs multiples of 4");
    PQ_CHECK_ARG(!(a->dx.finalize && (a->dx.act != PQ_ACT_LINEAR || a->dx.mean)) || a->dx.y, "sink needs y to finalize");
    if (a->dx.finalize & 2) {       // statistics only: dx.g already holds gout (the same buffer), nothing is stored
        PQ_CHECK_ARG(a->dx.g == a->gout + a->coff && a->dx.ldg == a->ldgo && !a->dx.accumulate && a->dx.act == PQ_ACT_LINEAR && a->dx.mean &&
                     a->dx.partial && a->dx.y && !a->dx2.g && a->up == 1 && !a->gate && !a->bcast && (a->C & 3) == 0 && (a->coff & 3) == 0 &&
                     (a->dx.ldy & 3) == 0 && !a->dx.qstate,
                     "finalize = 3 (statistics only): in-place linear sink with statistics, plain routing of a multiple of 4 channels");
    }
    if (a->dx2.g) {
        PQ_CHECK_ARG(a->up == 1 && !a->gate && !a->bcast && (a->C & 3) == 0 && (a->coff & 3) == 0 && a->dx2.ldg % 4 == 0,
                     "two sinks: plain routing of a multiple of 4 channels only");
        PQ_CHECK_ARG(!(a->dx2.finalize && (a->dx2.act != PQ_ACT_LINEAR || a->dx2.mean)) || a->dx2.y, "second sink needs y to finalize");
        PQ_CHECK_ARG(a->dx2.g != a->dx.g, "the two sinks must be different tensors");
        if (act_is_special(a->dx2.act)) { pq_set_error("pq_route_bwd: logistic/tanh sinks are not supported yet"); return PQ_EUNSUPPORTED; }
    }
    if (act_is_special(a->dx.act)) { pq_set_error("pq_route_bwd: logistic/tanh sinks are not supported yet"); return PQ_EUNSUPPORTED; }
    RouteP p{*a, (a->C + 3) / 4, ((a->C & 3) || (a->coff & 3)) ? 1 : 0};
    EwGeom g = ew_geom((long)a->B * a->H * a->W, a->C);
    PQ_SINK_ACC_CHECK(a->dx, a->C, g.gx);
    if (a->dx2.g) PQ_SINK_ACC_CHECK(a->dx2, a->C, g.gx);
    PQ_LAUNCH(route_bwd_kernel, dim3(g.gx, g.gy), dim3(g.cx, g.py), (size_t)g.cx * g.py * 8 * sizeof(float),
                       (hipStream_t)stream, p);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_pool_reduce(const pq_pool_reduce_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->a.x && a->out, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->HW > 0 && a->C > 0 && a->C % 4 == 0 && a->a.ld % 4 == 0 && (!a->gout || a->ldgo % 4 == 0), "bad shape");
    if (act_is_special(a->a.act)) { pq_set_error("pq_pool_reduce: logistic/tanh operands are not supported"); return PQ_EUNSUPPORTED; }
    PoolP p{*a, a->C / 4};
    const int cx = pick_cx(p.C4), py = 256 / cx, gy = (p.C4 + cx - 1) / cx;
    PQ_LAUNCH(pool_reduce_kernel, dim3(a->B, gy), dim3(cx, py), (size_t)cx * py * 4 * sizeof(float), (hipStream_t)stream, p);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_transpose_batch(const pq_transpose_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->items && a->n_items > 0 && a->total_tiles > 0, "bad args");
    PQ_LAUNCH(transpose_batch_kernel, dim3(a->total_tiles), dim3(256), 0, (hipStream_t)stream, a->items, a->n_items);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_adam_step(const pq_adam_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->param && a->grad && a->exp_avg && a->exp_avg_sq, "null pointer");
    PQ_CHECK_ARG(a->n > 0, "empty");
    PQ_CHECK_ARG(((uintptr_t)a->param & 15) == 0 && ((uintptr_t)a->grad & 15) == 0 && ((uintptr_t)a->exp_avg & 15) == 0 &&
                 ((uintptr_t)a->exp_avg_sq & 15) == 0, "buffers must be 16B aligned");
    int64_t nb = (a->n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    PQ_LAUNCH(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_fill(float* dst, float value, int64_t n, void* stream) {
    PQ_CHECK_ARG(dst && n >= 0, "bad args");
    if (n == 0) return PQ_OK;
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    PQ_LAUNCH(fill_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dst, value, n);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_sign_l1(float* grad, const float* weight, int64_t n, float ratio, void* stream) {
    PQ_CHECK_ARG(grad && weight && n > 0, "bad args");
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    PQ_LAUNCH(sign_l1_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, grad, weight, n, ratio);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
