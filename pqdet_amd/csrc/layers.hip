// Forward-only layers of the classifier cfgs (reference model/parser.py:122-137 FC, :389-410 dense k x k convolutions, :441-449 maxpool /
// avgpool, :106-112 logistic / tanh as a conv activation): small, HBM-bound gather kernels.  The reference has no training path
// for these nets (AnyModel has no caller with a loss), so there is no backward here; the engine refuses them in a train graph.
//
//   MAXPOOL  nn.MaxPool2d(k, stride, pad): padding is -inf, i.e. the maximum over the taps that lie inside the plane
//   AVGPOOL  nn.AdaptiveAvgPool2d((OH, OW)): cell (i, j) averages rows floor(i*IH/OH) .. ceil((i+1)*IH/OH)-1 (columns alike)
//   IM2COL   patch matrix of a dense k x k convolution, one row per output pixel, column = ci*k*k + ky*k + kx (the order of an
//            OIHW weight row), zero padding, pad columns zero: the convolution is then pq_conv1x1_fwd on it (fp32 MFMA GEMM)
//   FC       nn.Linear on the NCHW-flattened tensor (reference: x.view(input)): one wave per output
//   ACT      act(scale * x + shift) materialised, logistic and tanh included (consumers only apply the piecewise-linear family)
//   PADROWS  weight rows [O][k] copied to pitch ldo with zero fill (7 x 7 x 3 = 147 floats per row is not a multiple of 4)
#include "common.h"

namespace pq {

__device__ __forceinline__ float src_value(const pq_src& s, float v, int c, const ActParam& ap) {
    const float u = s.scale ? fmaf(s.scale[c], v, s.shift[c]) : v;
    return ap.special ? act_fwd_slow(u, ap.code) : act_fwd(u, ap);
}

__global__ __launch_bounds__(256) void pool2d_kernel(pq_layer_fwd_args a) {
    const ActParam ap = act_param(a.in.act);
    const long total = (long)a.B * a.OH * a.OW * a.C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % a.C);
        long t = idx / a.C;
        const int ox = (int)(t % a.OW); t /= a.OW;
        const int oy = (int)(t % a.OH);
        const int b = (int)(t / a.OH);
        int y0, y1, x0, x1;
        if (a.kind == PQ_LAYER_MAXPOOL) {
            y0 = oy * a.stride - a.pad; y1 = y0 + a.k; x0 = ox * a.stride - a.pad; x1 = x0 + a.k;
            y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, a.IH); x1 = min(x1, a.IW);
        } else {
            y0 = (oy * a.IH) / a.OH; y1 = ((oy + 1) * a.IH + a.OH - 1) / a.OH;
            x0 = (ox * a.IW) / a.OW; x1 = ((ox + 1) * a.IW + a.OW - 1) / a.OW;
        }
        float acc = a.kind == PQ_LAYER_MAXPOOL ? -__builtin_huge_valf() : 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const float v = src_value(a.in, a.in.x[(((size_t)b * a.IH + y) * a.IW + x) * a.in.ld + c], c, ap);
                acc = a.kind == PQ_LAYER_MAXPOOL ? fmaxf(acc, v) : acc + v;
            }
        if (a.kind == PQ_LAYER_AVGPOOL) acc = acc / (float)((y1 - y0) * (x1 - x0));
        a.out[(((size_t)b * a.OH + oy) * a.OW + ox) * a.ldo + c] = acc;
    }
}

__global__ __launch_bounds__(256) void im2col_kernel(pq_layer_fwd_args a) {
    const ActParam ap = act_param(a.in.act);
    const int kk = a.k * a.k, K = a.C * kk;
    const long total = (long)a.B * a.OH * a.OW * a.ldo;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = (int)(idx % a.ldo);
        long t = idx / a.ldo;
        const int ox = (int)(t % a.OW); t /= a.OW;
        const int oy = (int)(t % a.OH);
        const int b = (int)(t / a.OH);
        float v = 0.f;
        if (col < K) {
            const int ci = col / kk, tap = col - ci * kk, ky = tap / a.k, kx = tap - ky * a.k;
            const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
            if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) {
                if (a.nchw) v = a.in.x[(((size_t)b * a.C + ci) * a.IH + iy) * a.IW + ix];       // the network input: as stored
                else v = src_value(a.in, a.in.x[(((size_t)b * a.IH + iy) * a.IW + ix) * a.in.ld + ci], ci, ap);
            }
        }
        a.out[idx] = v;
    }
}

// one wave per output: out[o] = act(bias[o] + sum_n w[o][n] * in'(n)), n in NCHW order (c = n / (IH*IW), then row, column)
__global__ __launch_bounds__(256) void fc_kernel(pq_layer_fwd_args a) {
    const ActParam ap = act_param(a.in.act), oact = act_param(a.act);
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= a.O) return;
    const int hw = a.IH * a.IW, N = a.C * hw;
    float acc = 0.f;
    for (int n = lane; n < N; n += 64) {
        const int c = n / hw, p = n - c * hw;
        acc = fmaf(a.w[(size_t)o * N + n], src_value(a.in, a.in.x[(size_t)p * a.in.ld + c], c, ap), acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        const float u = acc + (a.bias ? a.bias[o] : 0.f);
        a.out[o] = oact.special ? act_fwd_slow(u, oact.code) : act_fwd(u, oact);
    }
}

__global__ __launch_bounds__(256) void act_kernel(pq_layer_fwd_args a) {
    const ActParam ap = act_param(a.in.act);
    const long rows = (long)a.B * a.IH * a.IW, total = rows * a.ldo;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % a.ldo);
        const long r = idx / a.ldo;
        a.out[idx] = c < a.C ? src_value(a.in, a.in.x[(size_t)r * a.in.ld + c], c, ap) : 0.f;
    }
}

__global__ __launch_bounds__(256) void padrows_kernel(pq_layer_fwd_args a) {
    const long total = (long)a.O * a.ldo;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % a.ldo);
        const long r = idx / a.ldo;
        a.out[idx] = c < a.k ? a.w[(size_t)r * a.k + c] : 0.f;
    }
}

}  // namespace pq

using namespace pq;

static unsigned blocks_for(long total) {
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    return (unsigned)(nb < 1 ? 1 : nb);
}

extern "C" int pq_layer_fwd(const pq_layer_fwd_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->out, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    switch (a->kind) {
        case PQ_LAYER_MAXPOOL: case PQ_LAYER_AVGPOOL:
            PQ_CHECK_ARG(a->in.x && a->B > 0 && a->IH > 0 && a->IW > 0 && a->OH > 0 && a->OW > 0 && a->C > 0 && a->ldo >= a->C, "bad shape");
            if (a->kind == PQ_LAYER_MAXPOOL) {
                PQ_CHECK_ARG(a->k >= 1 && a->stride >= 1 && a->pad >= 0 && 2 * a->pad <= a->k, "bad window");
                PQ_CHECK_ARG(a->OH == (a->IH + 2 * a->pad - a->k) / a->stride + 1 && a->OW == (a->IW + 2 * a->pad - a->k) / a->stride + 1, "output size");
            }
            PQ_LAUNCH(pool2d_kernel, dim3(blocks_for((long)a->B * a->OH * a->OW * a->C)), dim3(256), 0, st, *a);
            break;
        case PQ_LAYER_IM2COL:
            PQ_CHECK_ARG(a->in.x && a->B > 0 && a->C > 0 && a->k >= 1 && a->stride >= 1 && a->pad >= 0, "bad shape");
            PQ_CHECK_ARG(a->ldo >= a->C * a->k * a->k && (a->ldo & 3) == 0, "patch pitch must cover C*k*k and be a multiple of 4");
            PQ_CHECK_ARG(a->OH == (a->IH + 2 * a->pad - a->k) / a->stride + 1 && a->OW == (a->IW + 2 * a->pad - a->k) / a->stride + 1, "output size");
            PQ_CHECK_ARG(!a->nchw || (a->in.scale == nullptr && a->in.act == PQ_ACT_LINEAR), "the NCHW network input carries no transform");
            PQ_LAUNCH(im2col_kernel, dim3(blocks_for((long)a->B * a->OH * a->OW * a->ldo)), dim3(256), 0, st, *a);
            break;
        case PQ_LAYER_FC:
            PQ_CHECK_ARG(a->in.x && a->w && a->B == 1 && a->C > 0 && a->IH > 0 && a->IW > 0 && a->O > 0, "bad shape (nn.Linear on x.view(input): one image)");
            PQ_LAUNCH(fc_kernel, dim3((a->O + 3) / 4), dim3(256), 0, st, *a);
            break;
        case PQ_LAYER_ACT:
            PQ_CHECK_ARG(a->in.x && a->B > 0 && a->IH > 0 && a->IW > 0 && a->C > 0 && a->ldo >= a->C, "bad shape");
            PQ_LAUNCH(act_kernel, dim3(blocks_for((long)a->B * a->IH * a->IW * a->ldo)), dim3(256), 0, st, *a);
            break;
        case PQ_LAYER_PADROWS:
            PQ_CHECK_ARG(a->w && a->O > 0 && a->k > 0 && a->ldo >= a->k, "bad shape");
            PQ_LAUNCH(padrows_kernel, dim3(blocks_for((long)a->O * a->ldo)), dim3(256), 0, st, *a);
            break;
        default:
            pq_set_error("pq_layer_fwd: unknown kind %d", a->kind);
            return PQ_EINVAL;
    }
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
