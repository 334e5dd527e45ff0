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
#include <cstdlib>

namespace cal {

// XCD-contiguous row blocks (see k_espmm): workgroup w takes block (w % 8) * (blocks / 8) + w / 8, so each XCD's L2 serves
// the gathers of one contiguous eighth of the (block-diagonal) batch
__device__ __forceinline__ int xcd_block() {
    const int per = gridDim.x >> 3;
    return (int)blockIdx.x < 8 * per ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }
constexpr float GAT_LOG2E = 1.4426950408889634f;      // softmax weights as base-2 exponentials (v_exp_f32) of log2(e)-scaled logits

// sum over the 16 lanes of a DPP row (every lane gets the total): four v_add_f32 with row_ror modifiers instead of four
// ds_bpermute round trips -- a head of 64 columns is exactly one row of 16 lanes x 4 columns
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}
template <int LH>
__device__ __forceinline__ float head_sum(float v) {
    if constexpr (LH == 16) return row16_sum(v);
    else {
#pragma unroll
        for (int o = LH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
}

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

// Single-pass forward with the attention scores fused (vectorised layout: a head's D/4 lanes sit side by side in the row
// group).  k_gat_scores + k_gat_fwd cost the latency of five dependent load rounds per destination row
// (rowptr -> nbr -> a_src (max pass) -> nbr again -> a_src + z rows) with every wave holding one row: 241 + 33 us per layer at
// config 5 = 17 % of the HBM roofline.  Here a row costs three rounds (rowptr | nbr, eid, own z row | neighbour z rows):
//   * a_src[j,k] is recomputed from the z row of j that the aggregation fetches anyway (a dot with att[k,D:] and log2(D/4)
//     shuffles), a_dst[i,k] / a_src[i,k] of the row's own node come from its own z row and are STORED for the backward,
//     so the separate scores kernel (a full extra read of z) and the a_src gather are gone;
//   * the edge softmax is computed online (running max m, accumulator and denominator rescaled by exp(m - m')): no max pass;
//     the final m and the denominator are what the two-pass version stored;
//   * neighbours go eight at a time (hub rows of the BA graphs: 150 slots were 38 dependent batches of four).
template <int VEC, int G, int LH>
__global__ void __launch_bounds__(256) k_gat_fwd_fused(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                       const int* __restrict__ eid, const float* __restrict__ z,
                                                       const float* __restrict__ att, const float* __restrict__ bias, int relu,
                                                       float slope, float p, uint64_t seed, int64_t E, float* __restrict__ out,
                                                       float* __restrict__ adst, float* __restrict__ asrc,
                                                       float* __restrict__ mx, float* __restrict__ den, int N, int K,
                                                       const uint64_t* __restrict__ ctr) {
    constexpr int RPB = 256 / G, NB = 4, D = LH * VEC;
    seed = step_seed(seed, ctr);
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = xcd_block() * RPB + g;
    if (i >= N) return;
    const int H = K * D;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    constexpr float LOG2E = 1.4426950408889634f;
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        const int k = c / D, d = c % D, hl = (threadIdx.x & 63) & ~(LH - 1);      // first lane of this head in the wave
        const V att_d = V::ld(att + (size_t)k * 2 * D + d), att_s = V::ld(att + (size_t)k * 2 * D + D + d);
        const V zi = V::ld(z + (size_t)i * H + c);
        const float ad = head_sum<LH>(zi.dot(att_d)), as_i = head_sum<LH>(zi.dot(att_s));
        if (d == 0) { adst[(size_t)i * K + k] = ad; asrc[(size_t)i * K + k] = as_i; }
        // the node's own loop starts the running softmax (base-2 exponentials of log2(e)-scaled logits)
        float m = lrelu(ad + as_i, slope) * LOG2E, lsum = 1.f;
        V acc = zi;
        acc.scale(keep_scale(seed, E + i, k, K, p, inv_keep));
        for (int s = s0; s < s1; s += NB) {
            int j[NB], id[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) { const int sq = min(s + q, s1 - 1); j[q] = nbr[sq]; id[q] = eid[sq]; }
            V zv[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) zv[q] = V::ld(z + (size_t)j[q] * H + c);
            // dropout keep factors: lane (d / VEC) % NB of a head hashes slot q = that index, the others read it by shuffle --
            // one hash per batch instead of one per slot (every lane of a wave executes every hash)
            float kq[NB];
            if (p > 0.f) {
                const int myq = (d / VEC) % NB;
                const int myid = myq == 0 ? id[0] : (myq == 1 ? id[1] : (myq == 2 ? id[2] : id[3]));
                const float mine = keep_scale(seed, myid, k, K, p, inv_keep);
#pragma unroll
                for (int q = 0; q < NB; ++q) kq[q] = LH >= NB ? __shfl(mine, hl + q, 64) : keep_scale(seed, id[q], k, K, p, inv_keep);
            } else {
#pragma unroll
                for (int q = 0; q < NB; ++q) kq[q] = 1.f;
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) zv[q].pin();
            float e[NB], mn = m;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                e[q] = lrelu(ad + head_sum<LH>(zv[q].dot(att_s)), slope) * LOG2E;
                if (s + q < s1) mn = fmaxf(mn, e[q]);
            }
            const float sc = exp2f(m - mn);                 // one rescale per batch
            lsum *= sc;
            acc.scale(sc);
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const float pe = s + q < s1 ? exp2f(e[q] - mn) : 0.f;
                lsum += pe;
                acc.fma(pe * kq[q], zv[q]);
            }
            m = mn;
        }
        const float dn = lsum + 1e-16f;
        acc.scale(1.f / dn);
        if (bias) acc.add(V::ld(bias + c));
        if (relu) acc.relu();
        acc.st(out + (size_t)i * H + c);
        if (d == 0) { mx[(size_t)i * K + k] = m * (1.f / LOG2E); den[(size_t)i * K + k] = dn; }
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
                if (LH == 16) dot = row16_sum(dot);
                else for (int o = LH / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
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

// The same with the first NC slots of a row kept in registers between the two passes.  The softmax backward needs
// S = sum_s alpha_s dalpha_s over the whole row before any d(raw logit) can be formed, so k_gat_bwd_dst walks the row twice
// and parks dalpha in `draw` in between: rowptr -> nbr/eid -> z rows + a_src -> (write dalpha) -> nbr/eid -> a_src + dalpha
// read-back -> write: six dependent rounds per row, 459 us per layer at config 5 (12 % of the HBM roofline with the
// source-side kernel).  BA(m = 2) rows have ~5 slots: with (raw logit, dalpha, slot id) of the first NC = 8 slots held in
// registers a typical row is rowptr -> nbr/eid -> z rows + a_src -> write, and only a hub's slots beyond the eighth take the
// read-back path.
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_gat_bwd_dst_c(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                       const int* __restrict__ eid, const float* __restrict__ z,
                                                       const float* __restrict__ adst, const float* __restrict__ asrc,
                                                       const float* __restrict__ mx, const float* __restrict__ den,
                                                       const float* __restrict__ gout, float slope, float p,
                                                       uint64_t seed, int64_t E, float* __restrict__ draw,
                                                       float* __restrict__ dadst, int N, int K, int D,
                                                       const uint64_t* __restrict__ ctr) {
    constexpr int RPB = 256 / G, NC = 8;
    seed = step_seed(seed, ctr);
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = xcd_block() * RPB + g;
    if (i >= N) return;
    const int H = K * D;
    const int LH = D / VEC;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];          // slots s0 .. s1 - 1 are edges, "slot" s1 is the node's own loop
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        const int k = c / D;
        const bool head_lead = (c % D) == 0;
        const float ad = adst[(size_t)i * K + k];
        const float m = mx[(size_t)i * K + k], dn = den[(size_t)i * K + k];
        const V gi = V::ld(gout + (size_t)i * H + c);
        const float rdn = 1.f / dn;
        // ---- cached slots: s0 .. s0 + NC - 1 (clamped to the loop slot s1), gathered in two halves of four (registers) -------
        int cid[NC];                                           // slot id (edge id, or E + i for the loop): < 2^31
        float craw[NC], cda[NC], cal[NC];
        float S = 0.f;
        {
            int j[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int sq = min(s0 + q, s1);
                const int se = max(min(min(sq, s1 - 1), (int)E - 1), 0);   // a valid slot to read, unconditionally (masked for the loop)
                const int jn = nbr[se], en = eid[se];
                j[q] = sq < s1 ? jn : i;
                cid[q] = sq < s1 ? en : (int)E + i;
            }
            float as[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) as[q] = asrc[(size_t)j[q] * K + k];
            // dropout keep factors: lane q of a head hashes cached slot q, the others read it by shuffle (one hash per row
            // instead of eight: every lane of a wave executes every hash)
            float ckeep[NC];
            if (p > 0.f && LH >= NC) {
                const int myq = (c % D) / VEC % NC;
                int myid = cid[0];
#pragma unroll
                for (int q = 1; q < NC; ++q) myid = myq == q ? cid[q] : myid;
                const float mine = keep_scale(seed, myid, k, K, p, inv_keep);
                const int hl = (threadIdx.x & 63) & ~(LH - 1);
#pragma unroll
                for (int q = 0; q < NC; ++q) ckeep[q] = __shfl(mine, hl + q, 64);
            } else {
#pragma unroll
                for (int q = 0; q < NC; ++q) ckeep[q] = keep_scale(seed, cid[q], k, K, p, inv_keep);
            }
#pragma unroll
            for (int h2 = 0; h2 < NC; h2 += 4) {
                if (h2 > 0 && s0 + h2 > s1) break;             // the second half holds no slot of this row
                V zv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) zv[q] = V::ld(z + (size_t)j[h2 + q] * H + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) zv[q].pin();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dot = gi.dot(zv[q]);
                    if (LH == 16) dot = row16_sum(dot);
                    else for (int o = LH / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
                    craw[h2 + q] = ad + as[h2 + q];
                    cda[h2 + q] = dot * ckeep[h2 + q];
                    cal[h2 + q] = exp2f((lrelu(craw[h2 + q], slope) - m) * GAT_LOG2E) * rdn;
                    if (s0 + h2 + q <= s1) S = fmaf(cal[h2 + q], cda[h2 + q], S);
                }
            }
        }
        // ---- slots beyond the cache (hub rows): dalpha parked in `draw`, as in k_gat_bwd_dst -------------------------------
        for (int s = s0 + NC; s <= s1; s += 4) {
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
                if (LH == 16) dot = row16_sum(dot);
                else for (int o = LH / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
                const float alpha = exp2f((lrelu(ad + as[q], slope) - m) * GAT_LOG2E) * rdn;
                const float dalpha = dot * keep_scale(seed, id[q], k, K, p, inv_keep);
                if (s + q <= s1) {
                    S = fmaf(alpha, dalpha, S);
                    if (head_lead) draw[id[q] * K + k] = dalpha;
                }
            }
        }
        // ---- d(raw logit) = alpha (dalpha - S) lrelu'(raw) ------------------------------------------------------------------
        float rowsum = 0.f;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            if (s0 + q <= s1) {
                const float dr = cal[q] * (cda[q] - S) * (craw[q] > 0.f ? 1.f : slope);
                rowsum += dr;
                if (head_lead) draw[(int64_t)cid[q] * K + k] = dr;
            }
        }
        for (int s = s0 + NC; s <= s1; s += 4) {
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
                const float alpha = exp2f((lrelu(raw, slope) - m) * GAT_LOG2E) * rdn;
                const float dalpha = __shfl(da[q], (threadIdx.x & 63) & ~(LH - 1), 64);
                const float dr = alpha * (dalpha - S) * (raw > 0.f ? 1.f : slope);
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
                const float alpha = exp2f((lrelu(ad4[q] + as, slope) - mx4[q]) * GAT_LOG2E) / dn4[q];
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

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the H = 256, K = 4 (head dim 64) layer -- PyG's default heads = 4 (model.py:319) at BASELINE config 5's width -- with
// one WAVE per row and the row's slots in LANES.  The kernels above issue one memory instruction per slot AND per quantity
// (nbr, eid, a_src, the z / g row, the d(raw logit) store: ~45 per row in the backward), every one of them 16+ cycles of the
// CU's address path whatever it moves: at config 5 they are bound by memory-instruction issue, not by bytes.  Here lane t of the
// wave owns slot t of the row (edges, then the node's own loop): its neighbour id, edge id, the FOUR heads' a_src / a_dst / max /
// denominator / d(raw logit) travel as one coalesced or one 16 B-per-lane instruction per 64 slots, the per-slot-per-head
// arithmetic (logit, exp, dropout hash, softmax backward) runs lane-parallel, and only the feature-row gathers remain per
// slot: their row address comes out of the slot lanes by v_readlane (an SGPR base), their coefficient by four v_readlane and a
// select on the lane's head, the per-head dot products go back into the slot lane by v_readlane + a select on the lane id.  Exactly deg + 1 row
// gathers, eight in flight.  (scripts/micro/gather_lds.hip: the same restructuring of the plain aggregation, 128 -> 101 us.)
// ------------------------------------------------------------------------------------------------------------------------
namespace gw {
constexpr int K = 4, D = 64, H = 256;
__device__ __forceinline__ float rdl(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ float sel4(float a0, float a1, float a2, float a3, int k) { return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : a3)); }
__device__ __forceinline__ float wave_sum(float v) {       // every lane gets the total of the 64
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int h) { return h == 0 ? v.x : (h == 1 ? v.y : (h == 2 ? v.z : v.w)); }

// sum over the 16 lanes of a DPP row, NB independent values at once, as FUSED v_add_f32_dpp (the update_dpp form compiles to
// v_mov_b32_dpp + v_add, and the SLP vectorizer packed the adds of two slots into v_pk_add_f32 behind ~14 register moves per slot:
// 44 VALU instructions per gathered row, of which 9 were the work).  A DPP read needs two wait states after the VALU write of
// its source: the leading s_nop covers the compiler's last write, inside the block the other values' instructions do (NB >= 3)
// or an s_nop per step.
template <int NB>
__device__ __forceinline__ void row16_sum_n(float (&v)[NB]) {
    static_assert(NB >= 1 && NB <= 4, "batch of 1..4");
#define CAL_DPP(r, ctl) "v_add_f32_dpp " r ", " r ", " r " " ctl " row_mask:0xf bank_mask:0xf\n"
    if constexpr (NB == 4) {
        asm volatile("s_nop 1\n"
                     CAL_DPP("%0", "row_ror:8") CAL_DPP("%1", "row_ror:8") CAL_DPP("%2", "row_ror:8") CAL_DPP("%3", "row_ror:8")
                     CAL_DPP("%0", "row_ror:4") CAL_DPP("%1", "row_ror:4") CAL_DPP("%2", "row_ror:4") CAL_DPP("%3", "row_ror:4")
                     CAL_DPP("%0", "row_ror:2") CAL_DPP("%1", "row_ror:2") CAL_DPP("%2", "row_ror:2") CAL_DPP("%3", "row_ror:2")
                     CAL_DPP("%0", "row_ror:1") CAL_DPP("%1", "row_ror:1") CAL_DPP("%2", "row_ror:1") CAL_DPP("%3", "row_ror:1")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    } else if constexpr (NB == 3) {
        asm volatile("s_nop 1\n"
                     CAL_DPP("%0", "row_ror:8") CAL_DPP("%1", "row_ror:8") CAL_DPP("%2", "row_ror:8")
                     CAL_DPP("%0", "row_ror:4") CAL_DPP("%1", "row_ror:4") CAL_DPP("%2", "row_ror:4")
                     CAL_DPP("%0", "row_ror:2") CAL_DPP("%1", "row_ror:2") CAL_DPP("%2", "row_ror:2")
                     CAL_DPP("%0", "row_ror:1") CAL_DPP("%1", "row_ror:1") CAL_DPP("%2", "row_ror:1")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    } else if constexpr (NB == 2) {
        asm volatile("s_nop 1\n"
                     CAL_DPP("%0", "row_ror:8") CAL_DPP("%1", "row_ror:8") "s_nop 0\n"
                     CAL_DPP("%0", "row_ror:4") CAL_DPP("%1", "row_ror:4") "s_nop 0\n"
                     CAL_DPP("%0", "row_ror:2") CAL_DPP("%1", "row_ror:2") "s_nop 0\n"
                     CAL_DPP("%0", "row_ror:1") CAL_DPP("%1", "row_ror:1")
                     : "+v"(v[0]), "+v"(v[1]));
    } else {
        asm volatile("s_nop 1\n" CAL_DPP("%0", "row_ror:8") "s_nop 1\n" CAL_DPP("%0", "row_ror:4") "s_nop 1\n" CAL_DPP("%0", "row_ror:2")
                     "s_nop 1\n" CAL_DPP("%0", "row_ror:1")
                     : "+v"(v[0]));
    }
#undef CAL_DPP
}

// 5..8 values: two blocks
template <int NB>
__device__ __forceinline__ void row16_sum_b(float (&v)[NB]) {
    if constexpr (NB <= 4) row16_sum_n<NB>(v);
    else {
        float a[4] = {v[0], v[1], v[2], v[3]}, b[NB - 4];
#pragma unroll
        for (int u = 4; u < NB; ++u) b[u - 4] = v[u];
        row16_sum_n<4>(a);
        row16_sum_n<NB - 4>(b);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a[u];
#pragma unroll
        for (int u = 4; u < NB; ++u) v[u] = b[u - 4];
    }
}
// four values summed over the 64 lanes (every lane gets the totals): fused row sums, then the two cross-row exchanges
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
    row16_sum_n<4>(v);
#pragma unroll
    for (int h = 0; h < 4; ++h) { v[h] += __shfl_xor(v[h], 16, 64); v[h] += __shfl_xor(v[h], 32, 64); }
}

// att_s / ad arrive scaled by log2(e) (lrelu is positively homogeneous: lrelu(x) log2e = lrelu(x log2e)).  hm = the keep decisions
// of this lane's HEAD for the chunk's 64 slots, one bit per slot: a dropped (slot, head) adds nothing to the row, the kept ones
// are scaled by 1 / (1 - p) once per row (after the loop).  The kernel is VALU-co-bound at eight waves per SIMD: the decisions
// come from ONE hash per 16 slots (lane (head, l) hashes slot 16 g + l for its head; a ballot and two shifts put the head's 16
// bits into every lane of the head) instead of four per slot lane -- a row of <= 16 slots paid 64 VALU instructions for them.
template <int NB, bool DROP>
__device__ __forceinline__ void fwd_batch(Vec<4>& acc, float& m, float& lsum, const float* __restrict__ z, const Vec<4>& att_s, float ad,
                                          float slope, int jl, uint64_t hm, int q, int c) {
    Vec<4> zv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) zv[u] = Vec<4>::ld(z + (size_t)__builtin_amdgcn_readlane(jl, q + u) * H + c);
#pragma unroll
    for (int u = 0; u < NB; ++u) zv[u].pin();
    float e[NB], mn = m;
#pragma unroll
    for (int u = 0; u < NB; ++u) e[u] = zv[u].dot(att_s);
    row16_sum_n<NB>(e);
#pragma unroll
    for (int u = 0; u < NB; ++u) { e[u] = lrelu(ad + e[u], slope); mn = fmaxf(mn, e[u]); }
    const float sc = __builtin_amdgcn_exp2f(m - mn);   // one rescale of the running softmax per batch (arguments <= 0: the raw v_exp_f32,
                                                     // without exp2f's denormal-range rescue -- a term below 2^-126 of the row maximum is 0)
    lsum *= sc;
    acc.scale(sc);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const float pe = __builtin_amdgcn_exp2f(e[u] - mn);
        lsum += pe;
        const float pk = DROP ? (((hm >> (q + u)) & 1ull) ? pe : 0.f) : pe;
        acc.fma(pk, zv[u]);
    }
    m = mn;
}
}  // namespace gw

constexpr int FWD_NB = 4;      // z rows in flight per wave in the forward (8: spills at seven / eight waves per SIMD)
#define GW_SWITCH(REM, CALL) switch (REM) { case 7: CALL(7); break; case 6: CALL(6); break; case 5: CALL(5); break; case 4: CALL(4); break; \
                                            case 3: CALL(3); break; case 2: CALL(2); break; case 1: CALL(1); break; default: break; }

// GATConv forward, one pass (a_src of a neighbour recomputed from the z row the aggregation fetches, online edge softmax).
// WPE: waves per SIMD the register allocator has to make room for (amdgpu_waves_per_eu): a wave is a chain of ~4 dependent round
// trips per row, so the kernels are as fast as the number of rows in flight until VALU issue or the fabric takes over
// (forward: 5 -> 8 waves 178 -> 160 us, then the VALU diet above 160 -> 135 us; by-destination backward 6 -> 7 waves 213 -> 206 us).
template <bool DROP, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_gat_fwd_w(const int* __restrict__ rowptr, const int* __restrict__ nbr, const int* __restrict__ eid,
                                                   const float* __restrict__ z, const float* __restrict__ att, const float* __restrict__ bias,
                                                   int relu, float slope, float p, uint64_t seed, int64_t E, float* __restrict__ out,
                                                   float* __restrict__ adst, float* __restrict__ asrc, float* __restrict__ mx,
                                                   float* __restrict__ den, int N, const uint64_t* __restrict__ ctr) {
    using namespace gw;
    using V = Vec<4>;
    seed = step_seed(seed, ctr);
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(xcd_block() * 4 + (int)(threadIdx.x >> 6));
    if (i >= N) return;
    const int c = lane * 4, k = lane >> 4, d = c & (D - 1);
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    const V zi = V::ld(z + (size_t)i * H + c);
    const V att_d = V::ld(att + k * 2 * D + d);
    V att_s = V::ld(att + k * 2 * D + D + d);
    const float inv_keep = DROP ? 1.f / (1.f - p) : 1.f;
    const float ad = row16_sum(zi.dot(att_d)), as_i = row16_sum(zi.dot(att_s));
    att_s.scale(GAT_LOG2E);
    const float ad2 = ad * GAT_LOG2E;
    float m = lrelu(ad + as_i, slope) * GAT_LOG2E, lsum = 1.f;          // the node's own loop starts the running softmax
    V acc = zi;
    if (DROP) { if (keep_scale(seed, E + i, k, K, p, 1.f) == 0.f) acc = V::zero(); }
    for (int base = s0; base < s1; base += 64) {
        const int sl = min(base + lane, s1 - 1);
        int jl = nbr[sl], eg = DROP ? eid[min(base + (lane & 15), s1 - 1)] : 0;      // eg: edge id of slot (lane & 15), the first group
        asm volatile("" : "+v"(jl), "+v"(eg));
        const int cnt = min(64, s1 - base);
        uint64_t hm = 0;
        if (DROP) {
            for (int g = 0; 16 * g < cnt; ++g) {
                if (g > 0) eg = eid[min(base + 16 * g + (lane & 15), s1 - 1)];
                const uint64_t b = __ballot(keep_scale(seed, eg, k, K, p, 1.f) != 0.f);       // bit 16 head + l: (slot 16 g + l, head)
                const uint32_t w = (k & 2) ? (uint32_t)(b >> 32) : (uint32_t)b;
                hm |= (uint64_t)((w >> ((k & 1) * 16)) & 0xffffu) << (16 * g);
            }
        }
        int q = 0;
        for (; q + FWD_NB <= cnt; q += FWD_NB) fwd_batch<FWD_NB, DROP>(acc, m, lsum, z, att_s, ad2, slope, jl, hm, q, c);
#define GW_CALL(NB) fwd_batch<(NB < FWD_NB ? NB : 1), DROP>(acc, m, lsum, z, att_s, ad2, slope, jl, hm, q, c)
        GW_SWITCH(cnt - q, GW_CALL)
#undef GW_CALL
    }
    const float dn = lsum + 1e-16f;
    acc.scale(inv_keep / dn);
    if (bias) acc.add(V::ld(bias + c));
    if (relu) acc.relu();
    acc.st(out + (size_t)i * H + c);
    // per-(node, head) scalars for the backward: the four heads' values as ONE 16 B store each
    const float mo = m * (1.f / GAT_LOG2E);
    const float4 a4 = make_float4(rdl(ad, 0), rdl(ad, 16), rdl(ad, 32), rdl(ad, 48)), s4 = make_float4(rdl(as_i, 0), rdl(as_i, 16), rdl(as_i, 32), rdl(as_i, 48));
    const float4 m4 = make_float4(rdl(mo, 0), rdl(mo, 16), rdl(mo, 32), rdl(mo, 48)), d4 = make_float4(rdl(dn, 0), rdl(dn, 16), rdl(dn, 32), rdl(dn, 48));
    if (lane < 4) {
        float* dst = lane == 0 ? adst : (lane == 1 ? asrc : (lane == 2 ? mx : den));
        const float4 v = lane == 0 ? a4 : (lane == 1 ? s4 : (lane == 2 ? m4 : d4));
        *reinterpret_cast<float4*>(dst + (size_t)i * K) = v;
    }
}

// Backward over the by-destination CSR: d(raw logit) of every slot -> draw[id, 0:4], their row sums -> dadst[i, 0:4].
// Slot t of the row = edge s0 + t (t < deg) or the node's own loop (t = deg).  Rows of <= 64 slots (all but a few hubs of a BA
// graph) are ONE sweep: dalpha of the slots stays in the slot lanes until S = sum alpha dalpha is known; longer rows sweep twice
// (the second sweep gathers the z rows again instead of parking dalpha in memory).
template <bool DROP, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_gat_bwd_dst_w(const int* __restrict__ rowptr, const int* __restrict__ nbr, const int* __restrict__ eid,
                                                       const float* __restrict__ z, const float* __restrict__ adst, const float* __restrict__ asrc,
                                                       const float* __restrict__ mx, const float* __restrict__ den, const float* __restrict__ gout,
                                                       float slope, float p, uint64_t seed, int64_t E, float* __restrict__ draw,
                                                       float* __restrict__ dadst, int N, const uint64_t* __restrict__ ctr) {
    using namespace gw;
    using V = Vec<4>;
    seed = step_seed(seed, ctr);
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(xcd_block() * 4 + (int)(threadIdx.x >> 6));
    if (i >= N) return;
    const int c = lane * 4;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    const int deg = s1 - s0, nsl = deg + 1;
    const V gi = V::ld(gout + (size_t)i * H + c);
    const float inv_keep = DROP ? 1.f / (1.f - p) : 1.f;
    if (nsl <= 16) {
        // Rows of at most 16 slots (98 % of a BA graph's rows): lane (head k, l) owns (slot l, head k) -- ONE logit, exp, hash and
        // d(raw logit) per lane instead of four per slot lane (the VALU cost of a wave instruction does not depend on how many
        // lanes are on: a 5-slot row paid 4 x 64-lane instructions for 5 x 4 values), a head's dot product moves into its slot
        // lane by one compare + select, and the two row reductions are 16-lane DPP sums inside the head.
        const int k = lane >> 4, l = lane & 15;
        const bool valid = l < nsl;
        int jl = i, el = 0;
        if (deg > 0) {
            const int se = s0 + min(l, deg - 1);
            jl = nbr[se]; el = eid[se];
            asm volatile("" : "+v"(jl), "+v"(el));
            if (l >= deg) jl = i;
        }
        const int idl = l < deg ? el : (int)E + i;
        const float as = asrc[(size_t)jl * K + k];
        const float adk = adst[(size_t)i * K + k], mk = mx[(size_t)i * K + k], dnk = den[(size_t)i * K + k];
        float dal = 0.f;
        auto batch = [&](auto nbc, int q) {
            constexpr int NB = decltype(nbc)::value;
            V zv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) zv[u] = V::ld(z + (size_t)__builtin_amdgcn_readlane(jl, q + u) * H + c);
#pragma unroll
            for (int u = 0; u < NB; ++u) zv[u].pin();
            float dot[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) dot[u] = gi.dot(zv[u]);
            row16_sum_b<NB>(dot);                                       // <g_i, z_j> of the lane's head, in its 16 lanes
#pragma unroll
            for (int u = 0; u < NB; ++u) dal = l == q + u ? dot[u] : dal;
        };
        int q = 0;
        for (; q + 8 <= nsl; q += 8) batch(std::integral_constant<int, 8>(), q);
#define GW_CALL(NB) batch(std::integral_constant<int, NB>(), q)
        GW_SWITCH(nsl - q, GW_CALL)
#undef GW_CALL
        const float raw = adk + as;
        const float al = valid ? __builtin_amdgcn_exp2f((lrelu(raw, slope) - mk) * GAT_LOG2E) / dnk : 0.f;
        const float da = dal * (DROP ? keep_scale(seed, idl, k, K, p, inv_keep) : 1.f);
        float s1[1] = {al * da};
        row16_sum_n<1>(s1);
        const float de = al * (da - s1[0]) * (raw > 0.f ? 1.f : slope);
        float r1[1] = {de};
        row16_sum_n<1>(r1);
        if (valid) draw[(size_t)idl * K + k] = de;
        if (l == 0) dadst[(size_t)i * K + k] = r1[0];
        return;
    }
    const float4 ad4 = ld4(adst + (size_t)i * K), m4 = ld4(mx + (size_t)i * K), dn4 = ld4(den + (size_t)i * K);     // uniform addresses
    const float4 rdn4 = make_float4(1.f / dn4.x, 1.f / dn4.y, 1.f / dn4.z, 1.f / dn4.w);      // once per row (uniform values)
    float al[4], da[4], raw[4];
    int idl = 0;
    bool valid = false;
    // one chunk of <= 64 slots: ids and a_src lane-parallel, dalpha through the z gathers, then alpha / dalpha of this lane's slot
    auto chunk = [&](int base) {
        const int t = base + lane;
        valid = t < nsl;
        int jl = i, el = 0;
        if (deg > 0) {
            const int se = s0 + min(t, deg - 1);
            jl = nbr[se]; el = eid[se];
            asm volatile("" : "+v"(jl), "+v"(el));
            if (t >= deg) jl = i;
        }
        idl = t < deg ? el : (int)E + i;
        const float4 as4 = ld4(asrc + (size_t)jl * K);
        float dal[4] = {0.f, 0.f, 0.f, 0.f};
        const int cnt = min(64, nsl - base);
        auto batch = [&](auto nbc, int q) {
            constexpr int NB = decltype(nbc)::value;
            V zv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) zv[u] = V::ld(z + (size_t)__builtin_amdgcn_readlane(jl, q + u) * H + c);
#pragma unroll
            for (int u = 0; u < NB; ++u) zv[u].pin();
            float dot[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) dot[u] = gi.dot(zv[u]);
            row16_sum_b<NB>(dot);                                       // <g_i, z_j> per head, in every lane of the head
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const bool mine = lane == q + u;                        // ... and into the slot's lane (one compare, four selects)
#pragma unroll
                for (int h = 0; h < 4; ++h) dal[h] = mine ? rdl(dot[u], 16 * h) : dal[h];
            }
        };
        int q = 0;
        for (; q + 8 <= cnt; q += 8) batch(std::integral_constant<int, 8>(), q);
#define GW_CALL(NB) batch(std::integral_constant<int, NB>(), q)
        GW_SWITCH(cnt - q, GW_CALL)
#undef GW_CALL
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            raw[h] = comp(ad4, h) + comp(as4, h);
            // (the exponent is <= 0 up to rounding: m is the row maximum -- the raw v_exp_f32 without exp2f's denormal-range rescue)
            al[h] = valid ? __builtin_amdgcn_exp2f((lrelu(raw[h], slope) - comp(m4, h)) * GAT_LOG2E) * comp(rdn4, h) : 0.f;
            da[h] = dal[h] * (DROP ? keep_scale(seed, idl, h, K, p, inv_keep) : 1.f);
            if (DROP) __builtin_amdgcn_sched_barrier(0);
        }
    };
    float S[4] = {0.f, 0.f, 0.f, 0.f}, rowsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&]() {
        float de[4], ds[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) { de[h] = al[h] * (da[h] - S[h]) * (raw[h] > 0.f ? 1.f : slope); ds[h] = de[h]; }
        wave_sum4(ds);
#pragma unroll
        for (int h = 0; h < 4; ++h) rowsum[h] += ds[h];
        if (valid) *reinterpret_cast<float4*>(draw + (size_t)idl * K) = make_float4(de[0], de[1], de[2], de[3]);
    };
    if (nsl <= 64) {
        chunk(0);
#pragma unroll
        for (int h = 0; h < 4; ++h) S[h] = al[h] * da[h];
        wave_sum4(S);
        finish();
    } else {
        for (int base = 0; base < nsl; base += 64) {
            chunk(base);
            float t4[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) t4[h] = al[h] * da[h];
            wave_sum4(t4);
#pragma unroll
            for (int h = 0; h < 4; ++h) S[h] += t4[h];
        }
        for (int base = 0; base < nsl; base += 64) { chunk(base); finish(); }
    }
    if (lane == 0) *reinterpret_cast<float4*>(dadst + (size_t)i * K) = make_float4(rowsum[0], rowsum[1], rowsum[2], rowsum[3]);
}

// Backward over the by-source CSR: dz[j] = sum_slots alpha~ g[dst] + dadst[j] att_dst + dasrc[j] att_src, dasrc[j] = sum_slots d(raw logit)
template <bool DROP>
__global__ void __launch_bounds__(256) k_gat_bwd_src_w(const int* __restrict__ rowptr, const int* __restrict__ nbr, const int* __restrict__ eid,
                                                       const float* __restrict__ att, const float* __restrict__ adst, const float* __restrict__ asrc,
                                                       const float* __restrict__ mx, const float* __restrict__ den, const float* __restrict__ gout,
                                                       const float* __restrict__ dadst, const float* __restrict__ draw, float* __restrict__ dasrc,
                                                       float slope, float p, uint64_t seed, int64_t E, float* __restrict__ dz, int N,
                                                       const uint64_t* __restrict__ ctr) {
    using namespace gw;
    using V = Vec<4>;
    seed = step_seed(seed, ctr);
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(xcd_block() * 4 + (int)(threadIdx.x >> 6));
    if (j >= N) return;
    const int c = lane * 4, k = lane >> 4, d = c & (D - 1);
    const int s0 = rowptr[j], s1 = rowptr[j + 1];
    const int deg = s1 - s0, nsl = deg + 1;
    const float4 as4 = ld4(asrc + (size_t)j * K);
    const float inv_keep = DROP ? 1.f / (1.f - p) : 1.f;
    V acc = V::zero();
    float das[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < nsl; base += 64) {
        const int t = base + lane;
        const bool valid = t < nsl;
        int il = j, el = 0;
        if (deg > 0) {
            const int se = s0 + min(t, deg - 1);
            il = nbr[se]; el = eid[se];
            asm volatile("" : "+v"(il), "+v"(el));
            if (t >= deg) il = j;
        }
        const int idl = t < deg ? el : (int)E + j;
        const float4 ad4 = ld4(adst + (size_t)il * K), m4 = ld4(mx + (size_t)il * K), dn4 = ld4(den + (size_t)il * K);
        const float4 dr4 = ld4(draw + (size_t)idl * K);
        float at[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float a = __builtin_amdgcn_exp2f((lrelu(comp(ad4, h) + comp(as4, h), slope) - comp(m4, h)) * GAT_LOG2E) * __builtin_amdgcn_rcpf(comp(dn4, h));   // (v_rcp_f32: 1 ulp)
            at[h] = valid ? a * (DROP ? keep_scale(seed, idl, h, K, p, inv_keep) : 1.f) : 0.f;
            das[h] += valid ? comp(dr4, h) : 0.f;
            if (DROP) __builtin_amdgcn_sched_barrier(0);
        }
        const int cnt = min(64, nsl - base);
        auto batch = [&](auto nbc, int q) {
            constexpr int NB = decltype(nbc)::value;
            V gv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) gv[u] = V::ld(gout + (size_t)__builtin_amdgcn_readlane(il, q + u) * H + c);
#pragma unroll
            for (int u = 0; u < NB; ++u) gv[u].pin();
#pragma unroll
            for (int u = 0; u < NB; ++u) acc.fma(sel4(rdl(at[0], q + u), rdl(at[1], q + u), rdl(at[2], q + u), rdl(at[3], q + u), k), gv[u]);
        };
        int q = 0;
        for (; q + 8 <= cnt; q += 8) batch(std::integral_constant<int, 8>(), q);
#define GW_CALL(NB) batch(std::integral_constant<int, NB>(), q)
        GW_SWITCH(cnt - q, GW_CALL)
#undef GW_CALL
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) das[h] = wave_sum(das[h]);
    if (lane == 0) *reinterpret_cast<float4*>(dasrc + (size_t)j * K) = make_float4(das[0], das[1], das[2], das[3]);
    acc.fma(dadst[(size_t)j * K + k], V::ld(att + k * 2 * D + d));
    acc.fma(sel4(das[0], das[1], das[2], das[3], k), V::ld(att + k * 2 * D + D + d));
    acc.st(dz + (size_t)j * H + c);
}
#undef GW_SWITCH

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

// The same partial rows at H = 256 / four heads of 64: a wave per row, 16 B of the row per lane (the kernel above reads 4 B per
// lane and spends one address-path slot per 256 B: 45 us for config 5's 164 MB), eight rows in flight, the four waves' sums
// folded through LDS.
__global__ void __launch_bounds__(256) k_gat_datt_part_w(const float* __restrict__ z, const float* __restrict__ dadst,
                                                         const float* __restrict__ dasrc, float* __restrict__ part,
                                                         int N, int rows_per_block) {
    constexpr int K = 4, D = 64, H = 256, UR = 8;
    __shared__ float4 red[4][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane * 4, k = lane >> 4, d = c & (D - 1);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    for (int r = r0 + w; r < r1; r += 4 * UR) {
        float4 zv[UR];
        float da[UR], ds[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const size_t rr = (size_t)min(r + 4 * u, r1 - 1);
            zv[u] = *reinterpret_cast<const float4*>(z + rr * H + c);
            da[u] = dadst[rr * K + k];
            ds[u] = dasrc[rr * K + k];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) asm volatile("" : "+v"(zv[u].x), "+v"(zv[u].y), "+v"(zv[u].z), "+v"(zv[u].w), "+v"(da[u]), "+v"(ds[u]));
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const bool live = r + 4 * u < r1;
            const float x = live ? da[u] : 0.f, y = live ? ds[u] : 0.f;
            a.x = fmaf(x, zv[u].x, a.x); a.y = fmaf(x, zv[u].y, a.y); a.z = fmaf(x, zv[u].z, a.z); a.w = fmaf(x, zv[u].w, a.w);
            b.x = fmaf(y, zv[u].x, b.x); b.y = fmaf(y, zv[u].y, b.y); b.z = fmaf(y, zv[u].z, b.z); b.w = fmaf(y, zv[u].w, b.w);
        }
    }
    red[w][0][lane] = a; red[w][1][lane] = b;
    __syncthreads();
    if (w < 2) {            // wave 0: the target halves, wave 1: the source halves
        float4 t = red[0][w][lane];
#pragma unroll
        for (int q = 1; q < 4; ++q) { const float4 v = red[q][w][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * H + (size_t)k * 2 * D + w * D + d) = t;
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
template <int G, int LH>
static void gat_fused_one(hipStream_t stream, const int32_t* rowptr, const int32_t* nbr, const int32_t* eid, const float* z, const float* att,
                          const float* bias, int relu, float slope, float p, uint64_t seed, int64_t E, float* out, float* adst,
                          float* asrc, float* mx, float* den, int N, int K, const uint64_t* ctr) {
    if constexpr (LH <= G)
        hipLaunchKernelGGL((k_gat_fwd_fused<4, G, LH>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr, nbr, eid, z, att, bias, relu,
                           slope, p, seed, E, out, adst, asrc, mx, den, N, K, ctr);
}
template <int G, class... A>
static void gat_fused_launch(int lh, hipStream_t stream, A... a) {
    switch (lh) {
        case 1: gat_fused_one<G, 1>(stream, a...); break;
        case 2: gat_fused_one<G, 2>(stream, a...); break;
        case 4: gat_fused_one<G, 4>(stream, a...); break;
        case 8: gat_fused_one<G, 8>(stream, a...); break;
        case 16: gat_fused_one<G, 16>(stream, a...); break;
        case 32: gat_fused_one<G, 32>(stream, a...); break;
        default: gat_fused_one<G, 64>(stream, a...); break;
    }
}
}  // namespace cal

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
    if (vec_ok && K == 4 && D == 64 && aligned16(att) && aligned16(adst) && aligned16(asrc) && aligned16(mx) && aligned16(den)) {
        // H = 256, four heads: one wave per row, slots in lanes (k_gat_fwd_w)
        if (p > 0.f) hipLaunchKernelGGL((k_gat_fwd_w<true, 8>), dim3(cdiv(N, 4)), dim3(256), 0, stream, rowptr_dst, nbr_dst, eid_dst, z, att, bias, relu,
                                        slope, p, seed, E, out, adst, asrc, mx, den, (int)N, ctr);
        else hipLaunchKernelGGL((k_gat_fwd_w<false, 8>), dim3(cdiv(N, 4)), dim3(256), 0, stream, rowptr_dst, nbr_dst, eid_dst, z, att, bias, relu,
                                slope, p, seed, E, out, adst, asrc, mx, den, (int)N, ctr);
        CAL_CHECK_LAUNCH("k_gat_fwd_w");
        return 0;
    }
    if (vec_ok && pow2(D / 4) && aligned16(att) && D / 4 <= 64) {
        // scores + online edge softmax + aggregation in one pass (a head's lanes sit inside one row group)
        CAL_DISPATCH_VG((int)H, true, {
            if (G >= D / 4) {
                gat_fused_launch<G>((int)D / 4, stream, rowptr_dst, nbr_dst, eid_dst, z, att, bias, relu, slope, p, seed, E, out, adst, asrc,
                                    mx, den, (int)N, (int)K, ctr);
                CAL_CHECK_LAUNCH("k_gat_fwd_fused");
                return 0;
            }
        });
    }
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
        const bool wave_rows = vec_ok && K == 4 && D == 64 && aligned16(adst) && aligned16(asrc) && aligned16(mx) && aligned16(den) &&
                               aligned16(draw) && E + N < (1ll << 31);
        if (wave_rows) {
            const dim3 grid(cdiv(N, 4));
            if (p > 0.f) {
                hipLaunchKernelGGL((k_gat_bwd_dst_w<true, 7>), grid, dim3(256), 0, stream, rowptr_dst, nbr_dst, eid_dst, z, adst, asrc, mx, den, gout,
                                   slope, p, seed, E, draw, dadst, (int)N, ctr);
                hipLaunchKernelGGL((k_gat_bwd_src_w<true>), grid, dim3(256), 0, stream, rowptr_src, nbr_src, eid_src, att, adst, asrc, mx, den, gout,
                                   dadst, draw, dasrc, slope, p, seed, E, dz, (int)N, ctr);
            } else {
                hipLaunchKernelGGL((k_gat_bwd_dst_w<false, 7>), grid, dim3(256), 0, stream, rowptr_dst, nbr_dst, eid_dst, z, adst, asrc, mx, den, gout,
                                   slope, p, seed, E, draw, dadst, (int)N, ctr);
                hipLaunchKernelGGL((k_gat_bwd_src_w<false>), grid, dim3(256), 0, stream, rowptr_src, nbr_src, eid_src, att, adst, asrc, mx, den, gout,
                                   dadst, draw, dasrc, slope, p, seed, E, dz, (int)N, ctr);
            }
            CAL_CHECK_LAUNCH("k_gat_bwd_w");
        } else {
        CAL_DISPATCH_VG((int)H, vec_ok, {
            hipLaunchKernelGGL((k_gat_bwd_dst_c<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr_dst, nbr_dst,
                               eid_dst, z, adst, asrc, mx, den, gout, slope, p, seed, E, draw, dadst, (int)N, (int)K, (int)D, ctr);
        });
        CAL_CHECK_LAUNCH("k_gat_bwd_dst");
        CAL_DISPATCH_VG((int)H, vec_ok, {
            hipLaunchKernelGGL((k_gat_bwd_src<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr_src, nbr_src,
                               eid_src, att, adst, asrc, mx, den, gout, dadst, draw, dasrc, slope, p, seed, E, dz, (int)N, (int)K, (int)D, ctr);
        });
        CAL_CHECK_LAUNCH("k_gat_bwd_src");
        }
        int threads = (int)(H > 256 ? 256 : ((H + 63) / 64) * 64);
        if (K == 4 && D == 64 && aligned16(z) && aligned16(part))
            hipLaunchKernelGGL(k_gat_datt_part_w, dim3(nb), dim3(256), 0, stream, z, dadst, dasrc, part, (int)N, rpb);
        else
        hipLaunchKernelGGL(k_gat_datt_part, dim3(nb), dim3(threads), 0, stream, z, dadst, dasrc, part, (int)N, (int)K, (int)D, rpb);
        // (launch-site name of the whole backward, as the engine's stage list sees it: which dst / src kernels ran in front)
        CAL_CHECK_LAUNCH(wave_rows ? "k_gat_bwd_w+k_gat_datt_part" : "k_gat_bwd_dst+k_gat_bwd_src+k_gat_datt_part");
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
