// Observer arithmetic of quantisation-aware training, shared by qat.hip (activation / weight observers) and ew.hip (the
// activation observer folded into pq_bn_fwd_finalize).  Rounding boundaries are part of the contract (oracle/qat.py restates
// torch.fused_moving_avg_obs_fake_quant bit for bit): floating-point contraction is switched off inside the function, so it
// means the same in a translation unit that is not compiled with -ffp-contract=off.
#pragma once
#include "common.h"

namespace pq {

constexpr double QAT_SMALL_SCALE = 6.1e-5;

// state = {min, max, scale, zero_point}
__device__ inline void qat_observe(float* state, float lo, float hi, int qmin, int qmax, bool symmetric, float avg_const) {
#pragma clang fp contract(off)
    float mn = state[0], mx = state[1];
    if (isinf(mn)) { mn = lo; mx = hi; }                  // first batch initialises
    else { mn = mn + avg_const * (lo - mn); mx = mx + avg_const * (hi - mx); }
    state[0] = mn; state[1] = mx;
    double l = mn, h = mx;
    if (l < 0 && h > 0 && symmetric) {
        const int smin = -((qmax - qmin) / 2 + 1), smax = (qmax - qmin) / 2;
        const double m = fmax(fabs(l / smin), fabs(h / smax));
        l = m * smin; h = m * smax;
    }
    l = fmin(l, 0.0); h = fmax(h, 0.0);
    l = (double)(float)l; h = (double)(float)h;
    double scale = (h - l) / (double)(qmax - qmin);
    const float fs = (float)scale;
    if (fs == 0.f || isinf(1.f / fs)) scale = 0.1;
    if (scale < QAT_SMALL_SCALE) {
        const double org = scale;
        scale = QAT_SMALL_SCALE;
        if (l == 0.0) h = QAT_SMALL_SCALE * (qmax - qmin);
        else if (h == 0.0) l = -QAT_SMALL_SCALE * (qmax - qmin);
        else { const double amp = QAT_SMALL_SCALE / org; l *= amp; h *= amp; }
    }
    const double zmin = qmin - l / scale, zmax = qmax - h / scale;
    const double emin = abs(qmin) - fabs(l / scale), emax = abs(qmax) - fabs(h / scale);
    double z0 = emin < emax ? zmin : zmax;
    if (l < 0 && h > 0 && symmetric) z0 = (double)(qmin + qmax) / 2;
    int zp;
    if (z0 < qmin) zp = qmin; else if (z0 > qmax) zp = qmax; else zp = (int)rint(z0);
    state[2] = (float)scale;
    state[3] = (float)zp;
}

}  // namespace pq
