// Node-level kernels of the step engine's GINConv layers (CausalGIN, model.py:166-264):
//     GINConv(Sequential(Linear(H,H), BatchNorm1d(H), ReLU(), Linear(H,H), ReLU()))      (model.py:188-194; eps = 0, not trained)
//     h_i = relu(W2 relu(BN(W1 (h_{i-1} + sum_{j -> v} h_{i-1}[j]) + b1)) + b2)
// The unweighted aggregation runs on k_espmm with unit coefficients, the two Linear layers on the MFMA GEMMs (bias /
// ReLU / BatchNorm-statistics epilogues); what is left are the row-wise passes between them, all in the G-lanes-per-row,
// 16 B-per-lane shape of k_bn_bwd (UR rows per pass with every load issued first, per-workgroup column pre-reduction):
//   k_gin_bn_relu   y = relu(BN(t1))                                   (forward; also the running-statistics update)
//   k_gin_dots      s1 = sum_rows m dy, s2 = sum_rows m dy x_hat       (m = [y > 0], x_hat = (t1 - mean) rstd: the two
//                                                                       BatchNorm-backward sums BEHIND the ReLU mask)
//   k_gin_bn_bwd    dt1 = gamma rstd (m dy - s1 / n - x_hat s2 / n)    (+ its column sums = d b1, zero up to rounding)
//   k_gin_mask      dz = dh [h > 0]                                    (+ column sums = d b2 of the layer below)
#pragma once
#include "engine_kernels.hpp"

namespace cal {

struct GinRowArgs {
    const float* a;        // first input  [N,W]: t1 (fwd) / dy (dots, bn_bwd) / dh (mask)
    const float* a2;       // mask only: a second partial of dh added to the first (per-graph GINConv backward), or null
    const float* y;        // relu(BN(t1)) [N,W] (dots, bn_bwd) or h (mask)
    const float* t1;       // pre-BatchNorm activations (dots, bn_bwd)
    float* out;            // y (fwd) / dt1 (bn_bwd) / dz (mask)
    BNRef bn;
    const double* dot_sum; const double* dot_prod;      // finalised sums (bn_bwd)
    Acc acc0, acc1;        // dots: s1, s2; bn_bwd / mask: column sums of the output (acc0)
};

// MODE 0 fwd, 1 dots, 2 bn_bwd, 3 mask
template <int VEC, int G, int MODE>
__global__ void __launch_bounds__(256) k_gin_rows(const GinRowArgs p, int N, int W, int rows_per_block) {
    __shared__ double lds[256 * (VEC == 4 ? 4 : 1)];
    constexpr int RPB = 256 / G, UR = 4;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<VEC>;
    for (int c = l * VEC; c - l * VEC < W; c += G * VEC) {
        const bool cok = c < W;
        const int cc = min(c, W - VEC);
        float mean[VEC], rstd[VEC], sc[VEC], sh[VEC], gs[VEC], m1[VEC], m2[VEC];
        double cs0[VEC], cs1[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { cs0[j] = cs1[j] = 0.0; mean[j] = 0.f; rstd[j] = 1.f; sc[j] = 1.f; sh[j] = 0.f; gs[j] = 0.f; m1[j] = m2[j] = 0.f; }
        if (MODE != 3) {
            BNRaw raw[VEC];
            double ds[VEC], dp[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                raw[j] = bn_raw_load(p.bn, cc + j);
                ds[j] = MODE == 2 ? p.dot_sum[cc + j] : 0.0; dp[j] = MODE == 2 ? p.dot_prod[cc + j] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                bn_raw_pin(raw[j]);
                bn_raw_mean_rstd(p.bn, raw[j], mean[j], rstd[j]);
                sc[j] = raw[j].g * rstd[j]; sh[j] = raw[j].b - mean[j] * sc[j];
                gs[j] = sc[j];
                m1[j] = (float)(ds[j] * (double)p.bn.inv_n); m2[j] = (float)(dp[j] * (double)p.bn.inv_n);
                if (MODE == 0 && p.bn.update && blockIdx.x == 0 && grp == 0 && cok && c + j < W) bn_raw_update_running(p.bn, raw[j], c + j);
            }
        }
        for (int r0 = rbeg + grp; r0 < rend; r0 += RPB * UR) {
            V va[UR], vy[UR], vt[UR], v2[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const size_t r = (size_t)min(r0 + u * RPB, rend - 1);
                va[u] = V::ld(p.a + r * W + cc);
                v2[u] = (MODE == 3 && p.a2) ? V::ld(p.a2 + r * W + cc) : V::zero();
                vy[u] = MODE != 0 ? V::ld(p.y + r * W + cc) : V::zero();
                vt[u] = (MODE == 1 || MODE == 2) ? V::ld(p.t1 + r * W + cc) : V::zero();
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) { va[u].pin(); if (MODE != 0) vy[u].pin(); if (MODE == 1 || MODE == 2) vt[u].pin(); if (MODE == 3) { v2[u].pin(); va[u].add(v2[u]); } }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * RPB;
                const bool rok = r < rend && cok;
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float av = va[u].get(j);
                    if (MODE == 0) {
                        o[j] = fmaxf(fmaf(av, sc[j], sh[j]), 0.f);
                    } else if (MODE == 3) {
                        o[j] = vy[u].get(j) > 0.f ? av : 0.f;
                        if (rok) cs0[j] += (double)o[j];
                    } else {
                        const float g = vy[u].get(j) > 0.f ? av : 0.f;           // gradient behind the ReLU
                        const float xn = (vt[u].get(j) - mean[j]) * rstd[j];
                        if (MODE == 1) {
                            if (rok) { cs0[j] += (double)g; cs1[j] += (double)g * (double)xn; }
                        } else {
                            o[j] = gs[j] * (g - m1[j] - xn * m2[j]);
                            if (rok) cs0[j] += (double)o[j];
                        }
                    }
                }
                if (MODE != 1 && rok) {
                    V ov;
                    if constexpr (VEC == 4) ov.v = make_float4(o[0], o[1], o[2], o[3]); else ov.v = o[0];
                    ov.st(p.out + (size_t)r * W + c);
                }
            }
        }
        if (MODE != 0 && p.acc0.on()) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                block_col_atomic(cs0[j], l * VEC + j, grp, RPB, G * VEC, cok, p.acc0, c + j, lds);
                if (MODE == 1) block_col_atomic(cs1[j], l * VEC + j, grp, RPB, G * VEC, cok, p.acc1, c + j, lds);
            }
        }
    }
}

}  // namespace cal
