// GATv1 attention layer on the GraphPlan CSR (fp32): what the reference gets from PyG's
// GATConv(hidden, hidden/heads, heads=heads, dropout=p) at model.py:340,390.
//
//   z = x W  (dense, done by the GEMM)                       [N, K*D]
//   a_dst[i,k] = <z[i,k,:], att[k,:D]>,  a_src[j,k] = <z[j,k,:], att[k,D:]>
//   e = LeakyReLU_slope(a_dst[i] + a_src[j]);  alpha = softmax over the incoming edges of i
//   (self loops of the input dropped, one loop per node added), exp(e-max)/(sum+1e-16);
//   alpha~ = alpha * keep/(1-p) in training;  out[i,k,:] = sum_j alpha~ z[j,k,:] + bias.
//
// One G-lane group per destination row; every lane owns VEC consecutive columns of one head and
// walks the row's slots itself, so the per-head softmax needs no cross-lane traffic in the forward.
// Only (max, denominator) per (node, head) are kept for the backward; alpha is recomputed.
// Roofline: HBM-bound gather, same bytes as cal_spmm_fwd plus 3*E'*K*4 for the logits.
#include "common.hpp"
#include "gat_common.hpp"
#include <algorithm>

namespace cal {

// XCD-contiguous row blocks (see k_espmm): workgroup w takes block (w % 8) * (blocks / 8) + w / 8, so each XCD's L2 serves
// the gathers of one contiguous eighth of the (block-diagonal) batch
__device__ __forceinline__ int xcd_block() {
    const int per = gridDim.x >> 3;
    return (int)blockIdx.x < 8 * per ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// thread per (node, head)
__global__ void k_gat_scores(const float* __restrict__ z, const float* __restrict__ att,
                             float* __restrict__ adst, float* __restrict__ asrc, int N, int K, int D) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * K) return;
    int k = t % K;
    const float* zp = z + (size_t)t * D;          // [v, k, :] is contiguous at (v*K + k)*D
    const float* ad = att + (size_t)k * 2 * D;
    float sd = 0.f, ss = 0.f;
    for (int d = 0; d < D; ++d) {
        float zv = zp[d];
        sd = fmaf(zv, ad[d], sd);
        ss = fmaf(zv, ad[D + d], ss);
    }
    adst[t] = sd;
    asrc[t] = ss;
}

// Same scores with coalesced 16 B reads (D = 4 * 2^n): G lanes walk one row of z, the D/4 lanes of a head reduce
// their partial dots with shuffles.  The thread-per-(node, head) version reads 64 consecutive floats per thread --
// 0.9 TB/s at config 5 (179 us per layer for a 164 MB read).
template <int G>
__global__ void __launch_bounds__(256) k_gat_scores_v(const float* __restrict__ z, const float* __restrict__ att,
                                                      float* __restrict__ adst, float* __restrict__ asrc, int N, int K, int D) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int H = K * D, LH = D / 4;
    for (int i = blockIdx.x * RPB + g; i < N; i += gridDim.x * RPB) {
        for (int c = l * 4; c < H; c += G * 4) {
            const int k = c / D, d = c % D;
            const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)i * H + c);
            const float4 ad = *reinterpret_cast<const float4*>(att + (size_t)k * 2 * D + d);
            const float4 as = *reinterpret_cast<const float4*>(att + (size_t)k * 2 * D + D + d);
            float sd = zv.x * ad.x + zv.y * ad.y + zv.z * ad.z + zv.w * ad.w;
            float ss = zv.x * as.x + zv.y * as.y + zv.z * as.z + zv.w * as.w;
            for (int o = LH / 2; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, 64); ss += __shfl_xor(ss, o, 64); }
            if (d == 0) { adst[(size_t)i * K + k] = sd; asrc[(size_t)i * K + k] = ss; }
        }
    }
}

template <int VEC, int G>
__global__ void __launch_bounds__(256) k_gat_fwd(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                 const int* __restrict__ eid, const float* __restrict__ z,
                                                 const float* __restrict__ adst, const float* __restrict__ asrc,
                                                 const float* __restrict__ bias, int relu, float slope, float p,
                                                 uint64_t seed, int64_t E, float* __restrict__ out,
                                                 float* __restrict__ mx, float* __restrict__ den, int N, int K, int D,
                                                 const uint64_t* __restrict__ ctr) {
    constexpr int RPB = 256 / G;
    seed = step_seed(seed, ctr);
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = xcd_block() * RPB + g;
    if (i >= N) return;
    const int H = K * D;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        const int k = c / D;
        const float ad = adst[(size_t)i * K + k];
        const float eself = lrelu(ad + asrc[(size_t)i * K + k], slope);
        // neighbours four at a time: ids, then the four source scores (pass 1) / the four z rows (pass 2) are requested
        // together from clamped slots -- one dependent round trip per FOUR neighbours instead of per neighbour
        float m = eself;
        for (int s = s0; s < s1; s += 4) {
            int j[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) j[q] = nbr[min(s + q, s1 - 1)];
            float as[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) as[q] = asrc[(size_t)j[q] * K + k];
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmaxf(m, lrelu(ad + as[q], slope));      // a repeated last slot does not change a max
        }
        float lsum = 0.f;
        V acc = V::zero();
        for (int s = s0; s < s1; s += 4) {
            int j[4], id[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int sq = min(s + q, s1 - 1); j[q] = nbr[sq]; id[q] = eid[sq]; }
            float as[4];
            V zv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { as[q] = asrc[(size_t)j[q] * K + k]; zv[q] = V::ld(z + (size_t)j[q] * H + c); }
#pragma unroll
            for (int q = 0; q < 4; ++q) zv[q].pin();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pe = s + q < s1 ? expf(lrelu(ad + as[q], slope) - m) : 0.f;
                lsum += pe;
                acc.fma(pe * keep_scale(seed, id[q], k, K, p, inv_keep), zv[q]);
            }
        }
        {
            float pe = expf(eself - m);
            lsum += pe;
            acc.fma(pe * keep_scale(seed, E + i, k, K, p, inv_keep), V::ld(z + (size_t)i * H + c));
        }
        const float dn = lsum + 1e-16f;
        acc.scale(1.f / dn);
        if (bias) acc.add(V::ld(bias + c));
        if (relu) acc.relu();
        acc.st(out + (size_t)i * H + c);
        if (c % D == 0) { mx[(size_t)i * K + k] = m; den[(size_t)i * K + k] = dn; }
    }
}

// Backward pass 1+2 over the by-destination CSR.  Lanes of one head (LH = D/VEC of them, a power
// of two) reduce their partial dots with shuffles.  draw[id*K + k] (id = edge id, or E + i for the
// loop of node i) receives d(raw logit); dadst[i,k] the row sum.
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_gat_bwd_dst(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                     const int* __restrict__ eid, const float* __restrict__ z,
                                                     const float* __restrict__ adst, const float* __restrict__ asrc,
                                                     const float* __restrict__ mx, const float* __restrict__ den,
                                                     const float* __restrict__ gout, float slope, float p,
                                                     uint64_t seed, int64_t E, float* __restrict__ draw,
                                                     float* __restrict__ dadst, int N, int K, int D,
                                                     const uint64_t* __restrict__ ctr) {
    constexpr int RPB = 256 / G;
    seed = step_seed(seed, ctr);
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = xcd_block() * RPB + g;
    if (i >= N) return;
    const int H = K * D;
    const int LH = D / VEC;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        const int k = c / D;
        const bool head_lead = (c % D) == 0;
        const float ad = adst[(size_t)i * K + k];
        const float m = mx[(size_t)i * K + k], dn = den[(size_t)i * K + k];
        const V gi = V::ld(gout + (size_t)i * H + c);
        // slots s0..s1 (s1 = the node's own loop), four at a time with every load of a batch requested together
        float S = 0.f;
        for (int s = s0; s <= s1; s += 4) {
            int j[4];
            int64_t id[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sq = min(s + q, s1);
                j[q] = sq < s1 ? nbr[min(sq, max(s1 - 1, 0))] : i;
                id[q] = sq < s1 ? (int64_t)eid[min(sq, max(s1 - 1, 0))] : E + i;
            }
            V zv[4];
            float as[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { zv[q] = V::ld(z + (size_t)j[q] * H + c); as[q] = asrc[(size_t)j[q] * K + k]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) zv[q].pin();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dot = gi.dot(zv[q]);
                for (int o = LH / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
                const float alpha = expf(lrelu(ad + as[q], slope) - m) / dn;
                const float dalpha = dot * keep_scale(seed, id[q], k, K, p, inv_keep);
                if (s + q <= s1) {
                    S = fmaf(alpha, dalpha, S);
                    if (head_lead) draw[id[q] * K + k] = dalpha;
                }
            }
        }
        float rowsum = 0.f;
        for (int s = s0; s <= s1; s += 4) {
            int j[4];
            int64_t id[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sq = min(s + q, s1);
                j[q] = sq < s1 ? nbr[min(sq, max(s1 - 1, 0))] : i;
                id[q] = sq < s1 ? (int64_t)eid[min(sq, max(s1 - 1, 0))] : E + i;
            }
            float as[4], da[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { as[q] = asrc[(size_t)j[q] * K + k]; da[q] = draw[id[q] * K + k]; }   // written by this head's lead lane above
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float raw = ad + as[q];
                const float alpha = expf(lrelu(raw, slope) - m) / dn;
                const float dalpha = __shfl(da[q], (threadIdx.x & 63) & ~(LH - 1), 64);
                const float de = alpha * (dalpha - S);
                const float dr = de * (raw > 0.f ? 1.f : slope);
                if (s + q <= s1) {
                    rowsum += dr;
                    if (head_lead) draw[id[q] * K + k] = dr;
                }
            }
        }
        if (head_lead) dadst[(size_t)i * K + k] = rowsum;
    }
}

// dz[j,k,:] = sum_{s: src = j} alpha~_s g[dst_s,k,:] + alpha~_loop g[j,k,:]
//           + dadst[j,k] att[k,:D] + dasrc[j,k] att[k,D:]
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_gat_bwd_src(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                     const int* __restrict__ eid, const float* __restrict__ att,
                                                     const float* __restrict__ adst, const float* __restrict__ asrc,
                                                     const float* __restrict__ mx, const float* __restrict__ den,
                                                     const float* __restrict__ gout, const float* __restrict__ dadst,
                                                     const float* __restrict__ draw, float* __restrict__ dasrc, float slope, float p,
                                                     uint64_t seed, int64_t E, float* __restrict__ dz, int N, int K, int D,
                                                     const uint64_t* __restrict__ ctr) {
    constexpr int RPB = 256 / G;
    seed = step_seed(seed, ctr);
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int j = xcd_block() * RPB + g;
    if (j >= N) return;
    const int H = K * D;
    const int s0 = rowptr[j], s1 = rowptr[j + 1];
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        const int k = c / D, d = c % D;
        const float as = asrc[(size_t)j * K + k];
        V acc = V::zero();
        float das = 0.f;      // d a_src[j,k] = sum of d(raw logit) over the edges leaving j and its loop (was a kernel of its own)
        for (int s = s0; s <= s1; s += 4) {
            int i4[4];
            int64_t id[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sq = min(s + q, s1);
                i4[q] = sq < s1 ? nbr[min(sq, max(s1 - 1, 0))] : j;
                id[q] = sq < s1 ? (int64_t)eid[min(sq, max(s1 - 1, 0))] : E + j;
            }
            V gv[4];
            float ad4[4], mx4[4], dn4[4], dr4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gv[q] = V::ld(gout + (size_t)i4[q] * H + c);
                ad4[q] = adst[(size_t)i4[q] * K + k]; mx4[q] = mx[(size_t)i4[q] * K + k]; dn4[q] = den[(size_t)i4[q] * K + k];
                dr4[q] = draw[id[q] * K + k];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) gv[q].pin();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = s + q <= s1;
                const float alpha = expf(lrelu(ad4[q] + as, slope) - mx4[q]) / dn4[q];
                acc.fma(ok ? alpha * keep_scale(seed, id[q], k, K, p, inv_keep) : 0.f, gv[q]);
                das += ok ? dr4[q] : 0.f;
            }
        }
        if (d == 0) dasrc[(size_t)j * K + k] = das;
        acc.fma(dadst[(size_t)j * K + k], V::ld(att + (size_t)k * 2 * D + d));
        acc.fma(das, V::ld(att + (size_t)k * 2 * D + D + d));
        acc.st(dz + (size_t)j * H + c);
    }
}

// partial sums for d att: part[blk, k, 0:D] = sum_v dadst[v,k] z[v,k,:], part[blk, k, D:2D] = sum_v dasrc[v,k] z[v,k,:]
__global__ void __launch_bounds__(256) k_gat_datt_part(const float* __restrict__ z, const float* __restrict__ dadst,
                                                       const float* __restrict__ dasrc, float* __restrict__ part,
                                                       int N, int K, int D, int rows_per_block) {
    const int H = K * D;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const int k = c / D, d = c % D;
        float a = 0.f, b = 0.f;
        int r = r0;
        for (; r + 8 <= r1; r += 8) {           // eight independent row reads in flight per thread
            float zv[8], da[8], ds[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                zv[u] = z[(size_t)(r + u) * H + c];
                da[u] = dadst[(size_t)(r + u) * K + k];
                ds[u] = dasrc[(size_t)(r + u) * K + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a = fmaf(da[u], zv[u], a); b = fmaf(ds[u], zv[u], b); }
        }
        for (; r < r1; ++r) {
            float zv = z[(size_t)r * H + c];
            a = fmaf(dadst[(size_t)r * K + k], zv, a);
            b = fmaf(dasrc[(size_t)r * K + k], zv, b);
        }
        part[(size_t)blockIdx.x * 2 * H + (size_t)k * 2 * D + d] = a;
        part[(size_t)blockIdx.x * 2 * H + (size_t)k * 2 * D + D + d] = b;
    }
}

__global__ void __launch_bounds__(256) k_gat_datt_finish(const float* __restrict__ part, int nparts, int n,
                                                         float* __restrict__ datt) {
    __shared__ float red[256];
    int c = blockIdx.x * 16 + (threadIdx.x & 15);
    float s = finish_colsum(part, nparts, n, c, c < n, red);
    if ((threadIdx.x >> 4) == 0 && c < n) datt[c] = s;
}

// keep mask (1/0) as floats, [E + N, K]: row e < E for original edge e, row E + i for node i's loop
__global__ void k_gat_mask(uint64_t seed, int64_t rows, int K, float p, float* __restrict__ mask) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * K) return;
    mask[t] = keep_scale(seed, t / K, (int)(t % K), K, p, 1.f);
}

}  // namespace cal

using namespace cal;

static inline bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int gat_rows_per_block(int64_t N) {
    int64_t rpb = (N + 511) / 512;
    return (int)(rpb < 32 ? 32 : rpb);
}

// z [N,K*D] (= x W), att [K,2D] (first D: target half, last D: source half), bias [K*D] or null.
// Outputs: out [N,K*D]; saved for backward: adst, asrc, mx, den, each [N,K].
// p > 0 applies attention dropout with the counter-based mask of `seed` (cal_gat_dropout_mask
// materialises the same mask for tests).
namespace cal {
int gat_forward(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst, const float* z,
                const float* att, const float* bias, int relu, float slope, float p, uint64_t seed, const uint64_t* ctr,
                float* out, float* adst, float* asrc, float* mx, float* den, int64_t N, int64_t E,
                int64_t K, int64_t D, hipStream_t stream) {
    if (N == 0) return 0;
    CAL_REQUIRE(K > 0 && D > 0, "bad head shape");
    CAL_REQUIRE(p >= 0.f && p < 1.f, "dropout p must be in [0,1)");
    int64_t H = K * D;
    bool vec_ok = (D % 4 == 0) && aligned16(z) && aligned16(out) && (!bias || aligned16(bias));
    if (vec_ok && pow2(D / 4) && aligned16(att) && D / 4 <= 64) {
        const int G = std::max(group_for((int)H, 4), (int)(D / 4));      // a head's lanes must sit inside one row group
        const int blocks = (int)std::min<int64_t>(cdiv(N, 256 / G), 4096);
        if (G <= 8) hipLaunchKernelGGL((k_gat_scores_v<8>), dim3(blocks), dim3(256), 0, stream, z, att, adst, asrc, (int)N, (int)K, (int)D);
        else if (G == 16) hipLaunchKernelGGL((k_gat_scores_v<16>), dim3(blocks), dim3(256), 0, stream, z, att, adst, asrc, (int)N, (int)K, (int)D);
        else if (G == 32) hipLaunchKernelGGL((k_gat_scores_v<32>), dim3(blocks), dim3(256), 0, stream, z, att, adst, asrc, (int)N, (int)K, (int)D);
        else hipLaunchKernelGGL((k_gat_scores_v<64>), dim3(blocks), dim3(256), 0, stream, z, att, adst, asrc, (int)N, (int)K, (int)D);
    } else {
        hipLaunchKernelGGL(k_gat_scores, dim3(cdiv(N * K, 256)), dim3(256), 0, stream, z, att, adst, asrc, (int)N, (int)K, (int)D);
    }
    CAL_CHECK_LAUNCH("k_gat_scores");
    CAL_DISPATCH_VG((int)H, vec_ok, {
        hipLaunchKernelGGL((k_gat_fwd<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr_dst, nbr_dst, eid_dst,
                           z, adst, asrc, bias, relu, slope, p, seed, E, out, mx, den, (int)N, (int)K, (int)D, ctr);
    });
    CAL_CHECK_LAUNCH("k_gat_fwd");
    return 0;
}
}  // namespace cal

CAL_EXPORT int cal_gat_fwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst, const float* z,
                           const float* att, const float* bias, int relu, float slope, float p, uint64_t seed,
                           float* out, float* adst, float* asrc, float* mx, float* den, int64_t N, int64_t E,
                           int64_t K, int64_t D, void* stream_) {
    return gat_forward(rowptr_dst, nbr_dst, eid_dst, z, att, bias, relu, slope, p, seed, nullptr, out, adst, asrc, mx, den,
                       N, E, K, D, (hipStream_t)stream_);
}

CAL_EXPORT int64_t cal_gat_bwd_ws(int64_t N, int64_t E, int64_t K, int64_t D) {
    int64_t nb = N == 0 ? 1 : cdiv(N, gat_rows_per_block(N));
    return (E + N) * K + 2 * N * K + nb * 2 * K * D + 16;
}

// gout [N,K*D]: gradient at the layer output (already masked by the ReLU if one was fused).
// Outputs dz [N,K*D], datt [K,2D].  ws: cal_gat_bwd_ws floats.
namespace cal {
int gat_backward(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                 const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src, const float* z,
                 const float* att, const float* adst, const float* asrc, const float* mx, const float* den,
                 const float* gout, float slope, float p, uint64_t seed, const uint64_t* ctr, float* dz, float* datt,
                 float* ws, int64_t N, int64_t E, int64_t K, int64_t D, hipStream_t stream, float* part_out, int* nparts) {
    // part_out != null: the per-block partial sums of d att go there ([*nparts][2 K D] floats, cal_gat_datt_parts(N) rows)
    // and the caller reduces them (the step engine folds that into its final k_finish); datt is not written then
    int64_t H = K * D;
    int rpb = gat_rows_per_block(N);
    int nb = N == 0 ? 0 : cdiv(N, rpb);
    float* draw = ws;
    float* dadst = draw + ((E + N) * K + 3) / 4 * 4;
    float* dasrc = dadst + (N * K + 3) / 4 * 4;
    float* part = part_out ? part_out : dasrc + (N * K + 3) / 4 * 4;
    if (nparts) *nparts = nb;
    if (N > 0) {
        bool vec_ok = (D % 4 == 0) && pow2(D / 4) && aligned16(z) && aligned16(gout) && aligned16(dz) && aligned16(att);
        CAL_REQUIRE(vec_ok || pow2(D), "head dim must be a power of two (or 4 * a power of two)");
        CAL_DISPATCH_VG((int)H, vec_ok, {
            hipLaunchKernelGGL((k_gat_bwd_dst<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr_dst, nbr_dst,
                               eid_dst, z, adst, asrc, mx, den, gout, slope, p, seed, E, draw, dadst, (int)N, (int)K, (int)D, ctr);
        });
        CAL_CHECK_LAUNCH("k_gat_bwd_dst");
        CAL_DISPATCH_VG((int)H, vec_ok, {
            hipLaunchKernelGGL((k_gat_bwd_src<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr_src, nbr_src,
                               eid_src, att, adst, asrc, mx, den, gout, dadst, draw, dasrc, slope, p, seed, E, dz, (int)N, (int)K, (int)D, ctr);
        });
        CAL_CHECK_LAUNCH("k_gat_bwd_src");
        int threads = (int)(H > 256 ? 256 : ((H + 63) / 64) * 64);
        hipLaunchKernelGGL(k_gat_datt_part, dim3(nb), dim3(threads), 0, stream, z, dadst, dasrc, part, (int)N, (int)K, (int)D, rpb);
        CAL_CHECK_LAUNCH("k_gat_datt_part");
    }
    if (!part_out) {
        hipLaunchKernelGGL(k_gat_datt_finish, dim3(cdiv(2 * H, 16)), dim3(256), 0, stream, part, nb, (int)(2 * H), datt);
        CAL_CHECK_LAUNCH("k_gat_datt_finish");
    }
    return 0;
}
int gat_datt_parts(int64_t N) { return N == 0 ? 0 : (int)cdiv(N, gat_rows_per_block(N)); }
}  // namespace cal

CAL_EXPORT int cal_gat_bwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                           const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src, const float* z,
                           const float* att, const float* adst, const float* asrc, const float* mx, const float* den,
                           const float* gout, float slope, float p, uint64_t seed, float* dz, float* datt, float* ws,
                           int64_t N, int64_t E, int64_t K, int64_t D, void* stream_) {
    return gat_backward(rowptr_dst, nbr_dst, eid_dst, rowptr_src, nbr_src, eid_src, z, att, adst, asrc, mx, den, gout, slope, p,
                        seed, nullptr, dz, datt, ws, N, E, K, D, (hipStream_t)stream_, nullptr, nullptr);
}

CAL_EXPORT int cal_gat_dropout_mask(uint64_t seed, int64_t E, int64_t N, int64_t K, float p, float* mask, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int64_t n = (E + N) * K;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gat_mask, dim3(cdiv(n, 256)), dim3(256), 0, stream, seed, E + N, (int)K, p, mask);
    CAL_CHECK_LAUNCH("k_gat_mask");
    return 0;
}
