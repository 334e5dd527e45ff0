// Node-level feature layer for narrow inputs (F <= FEAT_FMAX; SURVEY.md 8d config 5: F = 10, N = 160 k rows, H = 256):
//   forward   h0 = relu(BN0(x0) W_feat)                                  (model.py:90-91, gcn_conv.py:75-77 with gfn=True)
//   backward  d W_feat = BN0(x0)^T dZ,  d gamma0 / d beta0               (autograd of the same lines)
// As [160k, 10] x [10, 256] and its two transposes these were three launches of the 64 x 64 / 128 x 128 tiled GEMMs: 113 us
// forward; backward 104 us (dZ W^T, only for BatchNorm-backward sums) + 94 us (the weight gradient), each streaming the
// 164 MB of dZ that k_bn_bwd (84 us) had just written.  With ten features the products are a handful of FMAs per output: a lane
// group per row keeps its four columns of W_feat in registers and the kernels run at the speed of their ONE matrix pass --
// forward: write h0; backward: the last BatchNorm-backward never stores dZ at all.
//
// Backward algebra (per feature f, column c; xn = (x0 - mean0) rstd0, y0 = gamma0 xn + beta0, dZ the gradient of the layer's
// pre-ReLU output):   M[f][c] = sum_r xn[r][f] dZ[r][c],   S[c] = sum_r dZ[r][c]
//   d W_feat[f][c] = gamma0_f M[f][c] + beta0_f S[c]
//   d gamma0_f     = sum_r xn[r][f] (dZ W^T)[r][f] = sum_c W[f][c] M[f][c]
//   d beta0_f      = sum_r (dZ W^T)[r][f]           = sum_c W[f][c] S[c]
// so one pass accumulates the (F + 1) x H sums (fp32 per lane over its <= ~40 rows, fp64 partial row per workgroup, summed in a
// fixed order by k_stats_final) and k_feat_bwd_final turns them into the three gradients.
#pragma once
#include "engine_kernels.hpp"

namespace cal {

constexpr int FEAT_FMAX = 16;

// y of feature f for the row held by this lane group: the value sits in lane f of the group
template <int G>
__device__ __forceinline__ float group_bcast(float v, int f) {
    if constexpr (G == 64) return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), f));
    else return __shfl(v, f, G);
}

struct FeatFwdArgs {
    const float* x0;      // [N, F]
    const float* W;       // [F, H]
    float* out;           // [N, H]
    BNRef bn0;
    Acc st_sum, st_sq;    // column statistics of the output (BatchNorm of the first backbone layer) or off
};

// grid (row blocks), 256 threads = 256 / G rows at a time; lane l of a group: columns 4 l .. 4 l + 3, and feature l of the row
template <int G, int FP>
__global__ void __launch_bounds__(256) k_feat_fwd_rows(const FeatFwdArgs a, int N, int H, int F, int rows_per_block) {
    __shared__ double lds[256 * 4];
    constexpr int RPB = 256 / G, UR = 4;
    warm_kernargs<sizeof(FeatFwdArgs) + 32>();
    const int grp = threadIdx.x / G, l = threadIdx.x % G, c = l * 4;
    const bool cok = c < H;
    const int cld = cok ? c : 0, fl = min(l, F - 1);
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<4>;
    V w[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f) w[f] = f < F ? V::ld(a.W + (size_t)f * H + cld) : V::zero();
    float sc, sh;
    bn_scale_shift(a.bn0, fl, sc, sh);
    if (a.bn0.update && blockIdx.x == 0 && threadIdx.x < F) bn_update_running(a.bn0, threadIdx.x);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = rbeg + grp; r0 < rend; r0 += RPB * UR) {
        float y[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) y[u] = a.x0[(size_t)min(r0 + u * RPB, rend - 1) * F + fl];
#pragma unroll
        for (int u = 0; u < UR; ++u) asm volatile("" : "+v"(y[u]));
#pragma unroll
        for (int u = 0; u < UR; ++u) y[u] = fmaf(y[u], sc, sh);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int r = r0 + u * RPB;
            V acc = V::zero();
#pragma unroll
            for (int f = 0; f < FP; ++f)
                if (f < F) acc.fma(group_bcast<G>(y[u], f), w[f]);
            acc.relu();
            if (r < rend && cok) {
                acc.st(a.out + (size_t)r * H + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double v = acc.get(j); s1[j] += v; s2[j] += v * v; }
            }
        }
    }
    if (a.st_sum.on()) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            block_col_atomic(s1[j], l * 4 + j, grp, RPB, G * 4, cok, a.st_sum, c + j, lds);
            block_col_atomic(s2[j], l * 4 + j, grp, RPB, G * 4, cok, a.st_sq, c + j, lds);
        }
    }
}

struct FeatBwdRowsArgs {
    // the last BatchNorm backward (k_bn_bwd's problem: dy = gamma rstd (dyh - m1 - xn m2) [x > 0])
    const float* dyh;     // [N, H] gradient of BN_1's output
    const float* x;       // [N, H] h0 (BN_1's input, post-ReLU)
    BNRef bn;
    const double* dot_sum;
    const double* dot_prod;
    // the feature layer
    const float* x0;      // [N, F]
    BNRef bn0;
    double* parts;        // [blocks][(F + 1) * H]: rows 0..F-1 = M, row F = S
};

template <int G, int FP>
__global__ void __launch_bounds__(256) k_bn_bwd_feat(const FeatBwdRowsArgs a, int N, int H, int F, int rows_per_block) {
    constexpr int RPB = 256 / G, UR = 2, NC = G * 4;
    __shared__ float red[RPB * 4 * NC];                  // 16 KB: four features at a time (or S as doubles: RPB * NC * 8 B)
    warm_kernargs<sizeof(FeatBwdRowsArgs) + 32>();
    const int grp = threadIdx.x / G, l = threadIdx.x % G, c = l * 4;
    const bool cok = c < H;
    const int cb = cok ? c : 0, fl = min(l, F - 1);
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<4>;
    float mean[4], rstd[4], gs[4], m1[4], m2[4];
    {
        double ds[4], dp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gs[j] = a.bn.gamma ? a.bn.gamma[cb + j] : 1.f;
            ds[j] = a.dot_sum[cb + j]; dp[j] = a.dot_prod[cb + j];
        }
        bn_mean_rstd_v<4>(a.bn, cb, mean, rstd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gs[j] *= rstd[j];
            m1[j] = (float)(ds[j] * (double)a.bn.inv_n);
            m2[j] = (float)(dp[j] * (double)a.bn.inv_n);
        }
    }
    float mean0, rstd0;
    bn_mean_rstd(a.bn0, fl, mean0, rstd0);
    V M[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f) M[f] = V::zero();
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = rbeg + grp; r0 < rend; r0 += RPB * UR) {
        V d[UR], xv[UR];
        float xn[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const size_t r = (size_t)min(r0 + u * RPB, rend - 1);
            d[u] = V::ld(a.dyh + r * H + cb); xv[u] = V::ld(a.x + r * H + cb);
            xn[u] = a.x0[r * F + fl];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) { d[u].pin(); xv[u].pin(); asm volatile("" : "+v"(xn[u])); }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const bool live = r0 + u * RPB < rend && cok;
            V dz;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xnj = (xv[u].get(j) - mean[j]) * rstd[j];
                float t = gs[j] * (d[u].get(j) - m1[j] - xnj * m2[j]);
                if (!(xv[u].get(j) > 0.f) || !live) t = 0.f;
                o[j] = t;
                cs[j] += (double)t;
            }
            dz.v = make_float4(o[0], o[1], o[2], o[3]);
            const float xnl = (xn[u] - mean0) * rstd0;
#pragma unroll
            for (int f = 0; f < FP; ++f)
                if (f < F) M[f].fma(group_bcast<G>(xnl, f), dz);
        }
    }
    // the workgroup's (F + 1) x H sums: lane groups folded through LDS four features at a time, group 0 writes the partial row
    double* prow = a.parts + (size_t)blockIdx.x * (F + 1) * H;
#pragma unroll
    for (int f0 = 0; f0 < FP; f0 += 4) {
        if (f0 >= F) break;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(red + ((grp * 4 + q) * NC + c)) = M[f0 + q].v;
        __syncthreads();
        if (grp == 0 && cok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (f0 + q < F) {
                    double t[4] = {0.0, 0.0, 0.0, 0.0};
                    for (int k = 0; k < RPB; ++k) {
                        const float4 v = *reinterpret_cast<const float4*>(red + ((k * 4 + q) * NC + c));
                        t[0] += (double)v.x; t[1] += (double)v.y; t[2] += (double)v.z; t[3] += (double)v.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) prow[(size_t)(f0 + q) * H + c + j] = t[j];
                }
            }
        }
        __syncthreads();
    }
    double* dred = reinterpret_cast<double*>(red);
#pragma unroll
    for (int j = 0; j < 4; ++j) dred[(grp * NC) + c + j] = cs[j];
    __syncthreads();
    if (grp == 0 && cok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double t = 0.0;
            for (int k = 0; k < RPB; ++k) t += dred[k * NC + c + j];
            prow[(size_t)F * H + c + j] = t;
        }
    }
}

// sums [(F + 1), H] (fp64, from k_stats_final) -> d W_feat [F, H] (fp32 slab, committed by k_finish) and the BatchNorm-0 row
// {d beta0 [F], d gamma0 [F]} (fp64, committed by k_finish).  grid (F), 256 threads
__global__ void __launch_bounds__(256) k_feat_bwd_final(const double* __restrict__ sums, const float* __restrict__ W, const BNRef bn0,
                                                        float* __restrict__ dW, double* __restrict__ bnrow, int H, int F) {
    __shared__ double red[2][256];
    const int f = blockIdx.x;
    const float g = bn0.gamma ? bn0.gamma[f] : 1.f, b = bn0.beta ? bn0.beta[f] : 0.f;
    double dg = 0.0, db = 0.0;
    for (int c = threadIdx.x; c < H; c += 256) {
        const double m = sums[(size_t)f * H + c], s = sums[(size_t)F * H + c];
        const double w = (double)W[(size_t)f * H + c];
        dW[(size_t)f * H + c] = (float)((double)g * m + (double)b * s);
        dg += w * m; db += w * s;
    }
    red[0][threadIdx.x] = db; red[1][threadIdx.x] = dg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { bnrow[f] = red[0][0]; bnrow[F + f] = red[1][0]; }
}

}  // namespace cal
