// GCN message passing on the GraphPlan CSR (fp32).
//
//   reference                                   here
//   GCNConv.norm      gcn_conv.py:44-70   ->    cal_gcn_norm_fwd / cal_gcn_norm_bwd
//   propagate/message gcn_conv.py:92-99   ->    cal_spmm_fwd  (also its own transpose for d/dh)
//   update (+bias)    gcn_conv.py:101-104 ->    fused into cal_spmm_fwd (optionally + ReLU, model.py:95)
//
// Roofline: HBM-bound gather.  Algorithmic bytes per cal_spmm_fwd launch =
// 2*N*H*4 (read features once, write output once) + E'*8 (neighbour id + edge
// weight per slot) + (N+1)*4 (row pointers), E' = E + N  (SURVEY.md section 8d).
#include "common.hpp"

namespace cal {

// deg[v] = sum_{e: row_e = v} w_e + loop_w   (edges in id order, loop last: the order of a
// sequential scatter_add over [edges..., loops...], gcn_conv.py:63-66); dis = deg^-1/2, inf -> 0.
__global__ void k_gcn_deg(const int* __restrict__ ptr_src, const int* __restrict__ eid_src,
                          const float* __restrict__ w, float loop_w, float* __restrict__ dis, int N) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    int s0 = ptr_src[v], s1 = ptr_src[v + 1];
    float d = 0.f;
    if (w) {
        for (int s = s0; s < s1; ++s) d += w[eid_src[s]];
    } else {
        d = (float)(s1 - s0);
    }
    d += loop_w;
    float r = 1.0f / sqrtf(d);            // d < 0 -> NaN like pow(-0.5)
    dis[v] = d == 0.f ? 0.f : r;          // gcn_conv.py:68
}

__global__ void k_gcn_norm_e(const int* __restrict__ row32, const int* __restrict__ col32,
                             const float* __restrict__ w, const float* __restrict__ dis,
                             float* __restrict__ norm_e, int64_t E) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int r = row32[e], c = col32[e];
    float we = w ? w[e] : 1.f;
    norm_e[e] = r != c ? dis[r] * we * dis[c] : 0.f;    // gcn_conv.py:70
}

// out[i,:] = act( sum_{slots s of row i} norm_e[eid[s]] * h[nbr[s],:] + dis[i]*loop_w*dis[i] * h[i,:] + bias )
// G lanes share one row (G*VEC consecutive floats per sweep -> one coalesced 16B/lane gather per
// neighbour); 256/G rows per workgroup.  Used with the by-destination CSR for the forward and the
// by-source CSR for d/dh (the transpose), where bias == nullptr and relu == 0.
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_spmm(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                              const int* __restrict__ eid, const float* __restrict__ norm_e,
                                              const float* __restrict__ dis, float loop_w,
                                              const float* __restrict__ h, const float* __restrict__ bias,
                                              int relu, float* __restrict__ out, int N, int H) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = blockIdx.x * RPB + g;
    if (i >= N) return;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    const float d = dis[i];
    const float nself = d * loop_w * d;
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        V acc = V::zero();
        int s = s0;
        for (; s + 4 <= s1; s += 4) {
            int j0 = nbr[s], j1 = nbr[s + 1], j2 = nbr[s + 2], j3 = nbr[s + 3];
            float n0 = norm_e[eid[s]], n1 = norm_e[eid[s + 1]], n2 = norm_e[eid[s + 2]], n3 = norm_e[eid[s + 3]];
            V h0 = V::ld(h + (size_t)j0 * H + c), h1 = V::ld(h + (size_t)j1 * H + c);
            V h2 = V::ld(h + (size_t)j2 * H + c), h3 = V::ld(h + (size_t)j3 * H + c);
            acc.fma(n0, h0); acc.fma(n1, h1); acc.fma(n2, h2); acc.fma(n3, h3);
        }
        for (; s < s1; ++s) {
            V hv = V::ld(h + (size_t)nbr[s] * H + c);
            acc.fma(norm_e[eid[s]], hv);
        }
        acc.fma(nself, V::ld(h + (size_t)i * H + c));
        if (bias) acc.add(V::ld(bias + c));
        if (relu) acc.relu();
        acc.st(out + (size_t)i * H + c);
    }
}

// dz = dout * (y > 0)  (ReLU backward, model.py:95/112-113) and per-block column partial sums of
// dz (-> d bias, gcn_conv.py:103).  part is [gridDim.x, H]; finished by k_colsum_finish.
template <int VEC>
__global__ void __launch_bounds__(256) k_relu_bwd_colsum(const float* __restrict__ dout, const float* __restrict__ y,
                                                         float* __restrict__ dz, float* __restrict__ part,
                                                         int N, int H, int rows_per_block) {
    using V = Vec<VEC>;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(N, r0 + rows_per_block);
    for (int c = threadIdx.x * VEC; c < H; c += blockDim.x * VEC) {
        V acc = V::zero();
        for (int r = r0; r < r1; ++r) {
            V g = V::ld(dout + (size_t)r * H + c);
            if (y) {
                V yy = V::ld(y + (size_t)r * H + c);
                if constexpr (VEC == 4) {
                    g.v.x = yy.v.x > 0.f ? g.v.x : 0.f; g.v.y = yy.v.y > 0.f ? g.v.y : 0.f;
                    g.v.z = yy.v.z > 0.f ? g.v.z : 0.f; g.v.w = yy.v.w > 0.f ? g.v.w : 0.f;
                } else {
                    g.v = yy.v > 0.f ? g.v : 0.f;
                }
            }
            if (dz) g.st(dz + (size_t)r * H + c);
            acc.add(g);
        }
        if (part) acc.st(part + (size_t)blockIdx.x * H + c);
    }
}

__global__ void __launch_bounds__(256) k_colsum_finish(const float* __restrict__ part, float* __restrict__ out,
                                                       int nparts, int H, int accumulate) {
    __shared__ float red[256];
    int c = blockIdx.x * 16 + (threadIdx.x & 15);
    float s = finish_colsum(part, nparts, H, c, c < H, red);
    if ((threadIdx.x >> 4) == 0 && c < H) out[c] = accumulate ? out[c] + s : s;
}

// gn_e[eid] = <dz[i,:], h[nbr,:]> for every by-destination slot of row i; gself[i] = <dz[i,:], h[i,:]>.
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_sddmm(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                               const int* __restrict__ eid, const float* __restrict__ dz,
                                               const float* __restrict__ h, float* __restrict__ gn_e,
                                               float* __restrict__ gself, int N, int H) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = blockIdx.x * RPB + g;
    if (i >= N) return;
    using V = Vec<VEC>;
    const int s0 = rowptr[i], s1 = rowptr[i + 1];
    for (int s = s0; s <= s1; ++s) {            // s == s1: the self term
        const int j = s < s1 ? nbr[s] : i;
        float p = 0.f;
        for (int c = l * VEC; c < H; c += G * VEC)
            p += V::ld(dz + (size_t)i * H + c).dot(V::ld(h + (size_t)j * H + c));
        p = group_sum<G>(p);
        if (l == 0) {
            if (s < s1) gn_e[eid[s]] = p; else gself[i] = p;
        }
    }
}

// d deg[v] from d norm (see DESIGN.md "backward through the weighted normalisation").
__global__ void k_gcn_norm_bwd_node(const int* __restrict__ ptr_src, const int* __restrict__ nbr_src,
                                    const int* __restrict__ eid_src, const int* __restrict__ ptr_dst,
                                    const int* __restrict__ nbr_dst, const int* __restrict__ eid_dst,
                                    const float* __restrict__ w, const float* __restrict__ dis, float loop_w,
                                    const float* __restrict__ gn_e, const float* __restrict__ gself,
                                    float* __restrict__ ddeg, int N) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    float acc = 0.f;
    for (int s = ptr_src[v]; s < ptr_src[v + 1]; ++s) {
        int e = eid_src[s];
        acc += gn_e[e] * (w ? w[e] : 1.f) * dis[nbr_src[s]];
    }
    for (int s = ptr_dst[v]; s < ptr_dst[v + 1]; ++s) {
        int e = eid_dst[s];
        acc += gn_e[e] * (w ? w[e] : 1.f) * dis[nbr_dst[s]];
    }
    float d = dis[v];
    acc += 2.f * gself[v] * d * loop_w;
    ddeg[v] = -0.5f * d * d * d * acc;
}

__global__ void k_gcn_norm_bwd_edge(const int* __restrict__ row32, const int* __restrict__ col32,
                                    const float* __restrict__ dis, const float* __restrict__ gn_e,
                                    const float* __restrict__ ddeg, float* __restrict__ dw, int64_t E) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int r = row32[e], c = col32[e];
    dw[e] = r != c ? gn_e[e] * dis[r] * dis[c] + ddeg[r] : 0.f;
}

}  // namespace cal

using namespace cal;

// GCNConv.norm (gcn_conv.py:44-70): dis[N] = deg^-1/2 over the source index incl. the added loop,
// norm_e[E] in ORIGINAL edge order (0 for dropped self-loop edges).  w may be null (all ones).
CAL_EXPORT int cal_gcn_norm_fwd(const int32_t* rowptr_src, const int32_t* eid_src, const int32_t* row32,
                                const int32_t* col32, const float* w, float loop_w, int64_t N, int64_t E,
                                float* dis, float* norm_e, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N > 0) {
        hipLaunchKernelGGL(k_gcn_deg, dim3(cdiv(N, 256)), dim3(256), 0, stream, rowptr_src, eid_src, w, loop_w, dis, (int)N);
        CAL_CHECK_LAUNCH("k_gcn_deg");
    }
    if (E > 0) {
        hipLaunchKernelGGL(k_gcn_norm_e, dim3(cdiv(E, 256)), dim3(256), 0, stream, row32, col32, w, dis, norm_e, E);
        CAL_CHECK_LAUNCH("k_gcn_norm_e");
    }
    return 0;
}

// Weighted aggregation + self loop + bias (+ReLU).  (rowptr, nbr, eid) is the by-destination CSR for
// the forward (gcn_conv.py:92-104) or the by-source CSR for the gradient w.r.t. h.
CAL_EXPORT int cal_spmm_fwd(const int32_t* rowptr, const int32_t* nbr, const int32_t* eid, const float* norm_e,
                            const float* dis, float loop_w, const float* h, const float* bias, int relu,
                            float* out, int64_t N, int64_t H, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0 || H == 0) return 0;
    CAL_REQUIRE(h != out, "in-place aggregation is not supported");
    bool vec_ok = (H % 4 == 0) && aligned16(h) && aligned16(out) && (!bias || aligned16(bias));
    CAL_DISPATCH_VG((int)H, vec_ok, {
        hipLaunchKernelGGL((k_spmm<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr, nbr, eid, norm_e,
                           dis, loop_w, h, bias, relu, out, (int)N, (int)H);
    });
    CAL_CHECK_LAUNCH("k_spmm");
    return 0;
}

static inline int colsum_rows_per_block(int64_t N) {
    // <= 512 partial rows so the finishing pass stays tiny; >= 32 rows of work per block
    int64_t rpb = (N + 511) / 512;
    return (int)(rpb < 32 ? 32 : rpb);
}

// dz = dout * (y > 0) (y may be null: plain copy/skip), dbias[H] = column sums of dz.
// dz and/or dbias may be null.  `part` needs cal_colsum_parts(N) * H floats.
CAL_EXPORT int64_t cal_colsum_parts(int64_t N) { return N == 0 ? 1 : cdiv(N, colsum_rows_per_block(N)); }

CAL_EXPORT int cal_relu_bwd_colsum(const float* dout, const float* y, float* dz, float* dbias, float* part,
                                   int64_t N, int64_t H, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (H == 0) return 0;
    CAL_REQUIRE(dbias == nullptr || part != nullptr, "partials workspace missing");
    int rpb = colsum_rows_per_block(N);
    int nb = (int)cal_colsum_parts(N);
    bool vec_ok = (H % 4 == 0) && aligned16(dout) && (!y || aligned16(y)) && (!dz || aligned16(dz)) && (!part || aligned16(part));
    int threads = (int)(vec_ok ? (H / 4) : H);
    threads = threads > 256 ? 256 : ((threads + 63) / 64) * 64;
    if (N > 0) {
        if (vec_ok)
            hipLaunchKernelGGL((k_relu_bwd_colsum<4>), dim3(nb), dim3(threads), 0, stream, dout, y, dz, dbias ? part : nullptr, (int)N, (int)H, rpb);
        else
            hipLaunchKernelGGL((k_relu_bwd_colsum<1>), dim3(nb), dim3(threads), 0, stream, dout, y, dz, dbias ? part : nullptr, (int)N, (int)H, rpb);
        CAL_CHECK_LAUNCH("k_relu_bwd_colsum");
    }
    if (dbias) {
        hipLaunchKernelGGL(k_colsum_finish, dim3(cdiv(H, 16)), dim3(256), 0, stream, part, dbias, N > 0 ? nb : 0, (int)H, 0);
        CAL_CHECK_LAUNCH("k_colsum_finish");
    }
    return 0;
}

// Gradient of the loss w.r.t. the edge weights through propagate AND through the normalisation
// (gcn_conv.py:63-70).  h is the aggregation input (x @ W), dz the gradient at the aggregation output.
// Workspaces: gn_e[E], gself[N], ddeg[N].  dw[E] in original edge order (0 on self-loop edges).
CAL_EXPORT int cal_gcn_norm_bwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                                const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src,
                                const int32_t* row32, const int32_t* col32, const float* w, const float* dis,
                                float loop_w, const float* h, const float* dz, float* gn_e, float* gself,
                                float* ddeg, float* dw, int64_t N, int64_t E, int64_t H, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) return 0;
    bool vec_ok = (H % 4 == 0) && aligned16(h) && aligned16(dz);
    CAL_DISPATCH_VG((int)H, vec_ok, {
        hipLaunchKernelGGL((k_sddmm<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, rowptr_dst, nbr_dst,
                           eid_dst, dz, h, gn_e, gself, (int)N, (int)H);
    });
    CAL_CHECK_LAUNCH("k_sddmm");
    hipLaunchKernelGGL(k_gcn_norm_bwd_node, dim3(cdiv(N, 256)), dim3(256), 0, stream, rowptr_src, nbr_src, eid_src,
                       rowptr_dst, nbr_dst, eid_dst, w, dis, loop_w, gn_e, gself, ddeg, (int)N);
    CAL_CHECK_LAUNCH("k_gcn_norm_bwd_node");
    if (E > 0) {
        hipLaunchKernelGGL(k_gcn_norm_bwd_edge, dim3(cdiv(E, 256)), dim3(256), 0, stream, row32, col32, dis, gn_e, ddeg, dw, E);
        CAL_CHECK_LAUNCH("k_gcn_norm_bwd_edge");
    }
    return 0;
}
