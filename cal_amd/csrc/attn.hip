// Causal/trivial soft masks (model.py:97-111) and global add pool (model.py:115-116).
//
//   edge attention  model.py:97-104 : softmax(Linear([x[row] || x[col]])) over 2 classes.
//       The reference materialises the [E, 2H] edge representation; here the Linear is split
//       into per-node projections P = x W[:, :H]^T, Q = x W[:, H:]^T ([N,2] each) and the edge
//       logit is P[row] + Q[col] + b  (same sum, re-associated).
//   node attention  model.py:106-111: a = softmax(Linear(x)); xc = a0 * x; xo = a1 * x.
//   add pool        model.py:115-116: out[b] = sum_{i in graph b} x[i]  (batch sorted -> segments).
#include "common.hpp"

namespace cal {

// pq[v] = (x[v].W[0,:H], x[v].W[1,:H], x[v].W[0,H:], x[v].W[1,H:])
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_edge_proj(const float* __restrict__ x, const float* __restrict__ W,
                                                   float* __restrict__ pq, int N, int H) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int v = blockIdx.x * RPB + g;
    if (v >= N) return;
    using V = Vec<VEC>;
    float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
    for (int c = l * VEC; c < H; c += G * VEC) {
        V xv = V::ld(x + (size_t)v * H + c);
        p0 += xv.dot(V::ld(W + c));
        p1 += xv.dot(V::ld(W + 2 * H + c));
        q0 += xv.dot(V::ld(W + H + c));
        q1 += xv.dot(V::ld(W + 3 * H + c));
    }
    p0 = group_sum<G>(p0); p1 = group_sum<G>(p1); q0 = group_sum<G>(q0); q1 = group_sum<G>(q1);
    if (l == 0) *reinterpret_cast<float4*>(pq + (size_t)v * 4) = make_float4(p0, p1, q0, q1);
}

// att[0*E + e] = w_c, att[1*E + e] = w_o
__global__ void k_edge_softmax2(const int* __restrict__ row32, const int* __restrict__ col32,
                                const float* __restrict__ pq, const float* __restrict__ b,
                                float* __restrict__ att, int64_t E) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float4 pr = *reinterpret_cast<const float4*>(pq + (size_t)row32[e] * 4);
    float4 qc = *reinterpret_cast<const float4*>(pq + (size_t)col32[e] * 4);
    float l0 = pr.x + qc.z + b[0], l1 = pr.y + qc.w + b[1];
    float m = fmaxf(l0, l1);
    float e0 = expf(l0 - m), e1 = expf(l1 - m);
    float inv = 1.f / (e0 + e1);
    att[e] = e0 * inv;
    att[E + e] = e1 * inv;
}

// dl[e] = d logit_0 = a0*a1*(dA0 - dA1)   (d logit_1 = -dl[e])
__global__ void k_edge_softmax2_bwd(const float* __restrict__ att, const float* __restrict__ datt,
                                    float* __restrict__ dl, int64_t E) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    dl[e] = att[e] * att[E + e] * (datt[e] - datt[E + e]);
}

// sp[v] = sum_{e: row_e = v} dl[e], sq[v] = sum_{e: col_e = v} dl[e]  (segment sums in edge-id order).
// Self-loop edges of the input are absent from the CSR; their weights never reach a conv
// (gcn_conv.py:56), so their dl is zero anyway.
__global__ void k_edge_att_bwd_node(const int* __restrict__ ptr_src, const int* __restrict__ eid_src,
                                    const int* __restrict__ ptr_dst, const int* __restrict__ eid_dst,
                                    const float* __restrict__ dl, float* __restrict__ spq, int N) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    float sp = 0.f, sq = 0.f;
    for (int s = ptr_src[v]; s < ptr_src[v + 1]; ++s) sp += dl[eid_src[s]];
    for (int s = ptr_dst[v]; s < ptr_dst[v + 1]; ++s) sq += dl[eid_dst[s]];
    spq[2 * (size_t)v] = sp;
    spq[2 * (size_t)v + 1] = sq;
}

// dx[v,:] (+)= sp[v]*(W[0,:H]-W[1,:H]) + sq[v]*(W[0,H:]-W[1,H:]); per-block partial sums of
// sp[v]*x[v,:] (-> dW[0,:H] = -dW[1,:H]), sq[v]*x[v,:] (-> dW[0,H:]), and of sp (-> db0 = -db1).
// part: [nblocks, 2H + 4] floats (last slot: sum sp).
template <int VEC>
__global__ void __launch_bounds__(256) k_edge_att_bwd_x(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ spq, float* __restrict__ dx,
                                                        int accumulate, float* __restrict__ part, int N, int H,
                                                        int rows_per_block) {
    using V = Vec<VEC>;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    const int P = 2 * H + 4;
    for (int c = threadIdx.x * VEC; c < H; c += blockDim.x * VEC) {
        V wp = V::ld(W + c), wp1 = V::ld(W + 2 * H + c), wq = V::ld(W + H + c), wq1 = V::ld(W + 3 * H + c);
        wp1.scale(-1.f); wp.add(wp1);
        wq1.scale(-1.f); wq.add(wq1);
        V ap = V::zero(), aq = V::zero();
        for (int r = r0; r < r1; ++r) {
            float sp = spq[2 * (size_t)r], sq = spq[2 * (size_t)r + 1];
            V xv = V::ld(x + (size_t)r * H + c);
            ap.fma(sp, xv);
            aq.fma(sq, xv);
            V d = accumulate ? V::ld(dx + (size_t)r * H + c) : V::zero();
            d.fma(sp, wp);
            d.fma(sq, wq);
            d.st(dx + (size_t)r * H + c);
        }
        ap.st(part + (size_t)blockIdx.x * P + c);
        aq.st(part + (size_t)blockIdx.x * P + H + c);
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += spq[2 * (size_t)r];
        part[(size_t)blockIdx.x * P + 2 * H] = s;
    }
}

// dW [2, 2H], db [2] from the partials
__global__ void __launch_bounds__(256) k_edge_att_bwd_finish(const float* __restrict__ part, int nparts, int H,
                                                             float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float red[256];
    const int P = 2 * H + 4;
    int c = blockIdx.x * 16 + (threadIdx.x & 15);
    float s = finish_colsum(part, nparts, P, c, c <= 2 * H, red);
    if ((threadIdx.x >> 4) != 0 || c > 2 * H) return;
    if (c < 2 * H) {
        dW[c] = s;
        dW[2 * H + c] = -s;
    } else {
        db[0] = s;
        db[1] = -s;
    }
}

// a = softmax2(x Wn^T + bn) ; xc = a0 x ; xo = a1 x.  att_n: [N, 2].
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_node_att_split(const float* __restrict__ x, const float* __restrict__ Wn,
                                                        const float* __restrict__ bn, float* __restrict__ att_n,
                                                        float* __restrict__ xc, float* __restrict__ xo,
                                                        int N, int H) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int v = blockIdx.x * RPB + g;
    if (v >= N) return;
    using V = Vec<VEC>;
    float l0 = 0.f, l1 = 0.f;
    for (int c = l * VEC; c < H; c += G * VEC) {
        V xv = V::ld(x + (size_t)v * H + c);
        l0 += xv.dot(V::ld(Wn + c));
        l1 += xv.dot(V::ld(Wn + H + c));
    }
    l0 = group_sum<G>(l0) + bn[0];
    l1 = group_sum<G>(l1) + bn[1];
    float m = fmaxf(l0, l1);
    float e0 = expf(l0 - m), e1 = expf(l1 - m);
    float inv = 1.f / (e0 + e1);
    float a0 = e0 * inv, a1 = e1 * inv;
    if (l == 0) { att_n[2 * (size_t)v] = a0; att_n[2 * (size_t)v + 1] = a1; }
    for (int c = l * VEC; c < H; c += G * VEC) {
        V xv = V::ld(x + (size_t)v * H + c);
        V a = xv, b = xv;
        a.scale(a0); b.scale(a1);
        a.st(xc + (size_t)v * H + c);
        b.st(xo + (size_t)v * H + c);
    }
}

// dx = a0 dxc + a1 dxo + dl0 (Wn[0]-Wn[1]),  dl0 = a0 a1 (<dxc,x> - <dxo,x>); dlv[v] = dl0.
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_node_att_split_bwd(const float* __restrict__ x, const float* __restrict__ Wn,
                                                            const float* __restrict__ att_n,
                                                            const float* __restrict__ dxc, const float* __restrict__ dxo,
                                                            float* __restrict__ dx, float* __restrict__ dlv,
                                                            int N, int H) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int v = blockIdx.x * RPB + g;
    if (v >= N) return;
    using V = Vec<VEC>;
    float d0 = 0.f, d1 = 0.f;
    for (int c = l * VEC; c < H; c += G * VEC) {
        V xv = V::ld(x + (size_t)v * H + c);
        d0 += xv.dot(V::ld(dxc + (size_t)v * H + c));
        d1 += xv.dot(V::ld(dxo + (size_t)v * H + c));
    }
    d0 = group_sum<G>(d0);
    d1 = group_sum<G>(d1);
    const float a0 = att_n[2 * (size_t)v], a1 = att_n[2 * (size_t)v + 1];
    const float dl0 = a0 * a1 * (d0 - d1);
    if (l == 0) dlv[v] = dl0;
    for (int c = l * VEC; c < H; c += G * VEC) {
        V w0 = V::ld(Wn + c), w1 = V::ld(Wn + H + c);
        w1.scale(-1.f); w0.add(w1);
        V o = V::zero();
        o.fma(a0, V::ld(dxc + (size_t)v * H + c));
        o.fma(a1, V::ld(dxo + (size_t)v * H + c));
        o.fma(dl0, w0);
        o.st(dx + (size_t)v * H + c);
    }
}

// partial sums of coef[v] * x[v,:] and of coef[v]; part [nblocks, H + 4]
template <int VEC>
__global__ void __launch_bounds__(256) k_wcolsum1(const float* __restrict__ x, const float* __restrict__ coef,
                                                  float* __restrict__ part, int N, int H, int rows_per_block) {
    using V = Vec<VEC>;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    const int P = H + 4;
    for (int c = threadIdx.x * VEC; c < H; c += blockDim.x * VEC) {
        V a = V::zero();
        for (int r = r0; r < r1; ++r) a.fma(coef[r], V::ld(x + (size_t)r * H + c));
        a.st(part + (size_t)blockIdx.x * P + c);
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += coef[r];
        part[(size_t)blockIdx.x * P + H] = s;
    }
}

__global__ void __launch_bounds__(256) k_node_att_bwd_finish(const float* __restrict__ part, int nparts, int H,
                                                             float* __restrict__ dWn, float* __restrict__ dbn) {
    __shared__ float red[256];
    const int P = H + 4;
    int c = blockIdx.x * 16 + (threadIdx.x & 15);
    float s = finish_colsum(part, nparts, P, c, c <= H, red);
    if ((threadIdx.x >> 4) != 0 || c > H) return;
    if (c < H) { dWn[c] = s; dWn[H + c] = -s; }
    else { dbn[0] = s; dbn[1] = -s; }
}

// out[b,:] = sum_{i in [gptr[b], gptr[b+1])} x[i,:]; rows in index order (= sequential index_add_).
// grid (B, S): split s handles a contiguous slice of the graph's rows -> part[s, b, :].
template <int VEC>
__global__ void __launch_bounds__(256) k_add_pool(const float* __restrict__ x, const int* __restrict__ gptr,
                                                  float* __restrict__ out, int B, int H, int S) {
    using V = Vec<VEC>;
    const int b = blockIdx.x, sp = blockIdx.y;
    const int n0 = gptr[b], n1 = gptr[b + 1];
    const int len = n1 - n0;
    const int chunk = (len + S - 1) / S;
    const int r0 = n0 + sp * chunk, r1 = min(n1, r0 + chunk);
    for (int c = threadIdx.x * VEC; c < H; c += blockDim.x * VEC) {
        V a = V::zero();
        for (int r = r0; r < r1; ++r) a.add(V::ld(x + (size_t)r * H + c));
        a.st(out + ((size_t)sp * B + b) * H + c);
    }
}

__global__ void k_pool_finish(const float* __restrict__ part, float* __restrict__ out, int BH, int S) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BH) return;
    float s = 0.f;
    for (int p = 0; p < S; ++p) s += part[(size_t)p * BH + i];
    out[i] = s;
}

// dx[i,:] = dout[batch[i],:]
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_add_pool_bwd(const float* __restrict__ dout, const int64_t* __restrict__ batch,
                                                      float* __restrict__ dx, int N, int H) {
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int i = blockIdx.x * RPB + g;
    if (i >= N) return;
    const int64_t b = batch[i];
    for (int c = l * VEC; c < H; c += G * VEC)
        Vec<VEC>::ld(dout + (size_t)b * H + c).st(dx + (size_t)i * H + c);
}

}  // namespace cal

using namespace cal;

static inline int rows_per_block_for(int64_t N) {
    int64_t rpb = (N + 511) / 512;
    return (int)(rpb < 32 ? 32 : rpb);
}
static inline int col_threads(int64_t H, bool vec) {
    int t = (int)(vec ? H / 4 : H);
    return t > 256 ? 256 : ((t + 63) / 64) * 64;
}

// model.py:97-104.  W: edge_att_mlp.weight [2, 2H] row-major, b: [2].  pq: [N,4] workspace (kept for
// nothing in backward; recomputed cheaply), att: [2, E] (row 0 = edge_weight_c, row 1 = edge_weight_o).
CAL_EXPORT int cal_edge_att_fwd(const float* x, const float* W, const float* b, const int32_t* row32,
                                const int32_t* col32, float* pq, float* att, int64_t N, int64_t E, int64_t H,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CAL_REQUIRE(aligned16(pq), "pq must be 16B aligned");
    if (N > 0) {
        bool vec_ok = (H % 4 == 0) && aligned16(x) && aligned16(W);
        CAL_DISPATCH_VG((int)H, vec_ok, {
            hipLaunchKernelGGL((k_edge_proj<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, x, W, pq, (int)N, (int)H);
        });
        CAL_CHECK_LAUNCH("k_edge_proj");
    }
    if (E > 0) {
        hipLaunchKernelGGL(k_edge_softmax2, dim3(cdiv(E, 256)), dim3(256), 0, stream, row32, col32, pq, b, att, E);
        CAL_CHECK_LAUNCH("k_edge_softmax2");
    }
    return 0;
}

CAL_EXPORT int64_t cal_edge_att_bwd_ws(int64_t N, int64_t E, int64_t H) {
    // dl[E] + spq[2N] + partials
    int64_t nb = N == 0 ? 1 : cdiv(N, rows_per_block_for(N));
    return E + 2 * N + nb * (2 * H + 4) + 16;
}

// Backward of cal_edge_att_fwd.  datt [2,E]; dx [N,H] is written (accumulate=0) or added to (=1);
// dW [2,2H], db [2].  ws: cal_edge_att_bwd_ws floats, 16B aligned.
CAL_EXPORT int cal_edge_att_bwd(const float* x, const float* W, const float* att, const float* datt,
                                const int32_t* rowptr_src, const int32_t* eid_src, const int32_t* rowptr_dst,
                                const int32_t* eid_dst, float* dx, int accumulate, float* dW, float* db,
                                float* ws, int64_t N, int64_t E, int64_t H, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rpb = rows_per_block_for(N);
    int nb = N == 0 ? 0 : cdiv(N, rpb);
    // 16B-aligned carve
    int64_t offE = (E + 3) / 4 * 4, offN = (2 * N + 3) / 4 * 4;
    float* dl = ws;
    float* spq = ws + offE;
    float* part = spq + offN;
    if (E > 0) {
        hipLaunchKernelGGL(k_edge_softmax2_bwd, dim3(cdiv(E, 256)), dim3(256), 0, stream, att, datt, dl, E);
        CAL_CHECK_LAUNCH("k_edge_softmax2_bwd");
    }
    if (N > 0) {
        hipLaunchKernelGGL(k_edge_att_bwd_node, dim3(cdiv(N, 256)), dim3(256), 0, stream, rowptr_src, eid_src,
                           rowptr_dst, eid_dst, dl, spq, (int)N);
        CAL_CHECK_LAUNCH("k_edge_att_bwd_node");
        bool vec_ok = (H % 4 == 0) && aligned16(x) && aligned16(W) && aligned16(dx) && aligned16(part);
        if (vec_ok)
            hipLaunchKernelGGL((k_edge_att_bwd_x<4>), dim3(nb), dim3(col_threads(H, true)), 0, stream, x, W, spq, dx,
                               accumulate, part, (int)N, (int)H, rpb);
        else
            hipLaunchKernelGGL((k_edge_att_bwd_x<1>), dim3(nb), dim3(col_threads(H, false)), 0, stream, x, W, spq, dx,
                               accumulate, part, (int)N, (int)H, rpb);
        CAL_CHECK_LAUNCH("k_edge_att_bwd_x");
    }
    hipLaunchKernelGGL(k_edge_att_bwd_finish, dim3(cdiv(2 * H + 1, 16)), dim3(256), 0, stream, part, nb, (int)H, dW, db);
    CAL_CHECK_LAUNCH("k_edge_att_bwd_finish");
    return 0;
}

// model.py:106-111.  Wn: node_att_mlp.weight [2,H], bn [2]; att_n [N,2]; xc, xo [N,H].
CAL_EXPORT int cal_node_att_split_fwd(const float* x, const float* Wn, const float* bn, float* att_n, float* xc,
                                      float* xo, int64_t N, int64_t H, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) return 0;
    bool vec_ok = (H % 4 == 0) && aligned16(x) && aligned16(Wn) && aligned16(xc) && aligned16(xo);
    CAL_DISPATCH_VG((int)H, vec_ok, {
        hipLaunchKernelGGL((k_node_att_split<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, x, Wn, bn, att_n,
                           xc, xo, (int)N, (int)H);
    });
    CAL_CHECK_LAUNCH("k_node_att_split");
    return 0;
}

CAL_EXPORT int64_t cal_node_att_bwd_ws(int64_t N, int64_t H) {
    int64_t nb = N == 0 ? 1 : cdiv(N, rows_per_block_for(N));
    return N + nb * (H + 4) + 16;
}

CAL_EXPORT int cal_node_att_split_bwd(const float* x, const float* Wn, const float* att_n, const float* dxc,
                                      const float* dxo, float* dx, float* dWn, float* dbn, float* ws, int64_t N,
                                      int64_t H, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rpb = rows_per_block_for(N);
    int nb = N == 0 ? 0 : cdiv(N, rpb);
    float* dlv = ws;
    float* part = ws + (N + 3) / 4 * 4;
    if (N > 0) {
        bool vec_ok = (H % 4 == 0) && aligned16(x) && aligned16(Wn) && aligned16(dxc) && aligned16(dxo) && aligned16(dx) && aligned16(part);
        CAL_DISPATCH_VG((int)H, vec_ok, {
            hipLaunchKernelGGL((k_node_att_split_bwd<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, x, Wn, att_n,
                               dxc, dxo, dx, dlv, (int)N, (int)H);
        });
        CAL_CHECK_LAUNCH("k_node_att_split_bwd");
        if (vec_ok)
            hipLaunchKernelGGL((k_wcolsum1<4>), dim3(nb), dim3(col_threads(H, true)), 0, stream, x, dlv, part, (int)N, (int)H, rpb);
        else
            hipLaunchKernelGGL((k_wcolsum1<1>), dim3(nb), dim3(col_threads(H, false)), 0, stream, x, dlv, part, (int)N, (int)H, rpb);
        CAL_CHECK_LAUNCH("k_wcolsum1");
    }
    hipLaunchKernelGGL(k_node_att_bwd_finish, dim3(cdiv(H + 1, 16)), dim3(256), 0, stream, part, nb, (int)H, dWn, dbn);
    CAL_CHECK_LAUNCH("k_node_att_bwd_finish");
    return 0;
}

// global_add_pool.  splits S >= 1; part needs S*B*H floats when S > 1 (else may be null).
CAL_EXPORT int cal_add_pool_fwd(const float* x, const int32_t* gptr, float* out, float* part, int64_t B, int64_t H,
                                int64_t S, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B == 0 || H == 0) return 0;
    CAL_REQUIRE(S >= 1 && (S == 1 || part != nullptr), "bad split workspace");
    float* dst = S == 1 ? out : part;
    bool vec_ok = (H % 4 == 0) && aligned16(x) && aligned16(dst);
    if (vec_ok)
        hipLaunchKernelGGL((k_add_pool<4>), dim3((unsigned)B, (unsigned)S), dim3(col_threads(H, true)), 0, stream, x, gptr, dst, (int)B, (int)H, (int)S);
    else
        hipLaunchKernelGGL((k_add_pool<1>), dim3((unsigned)B, (unsigned)S), dim3(col_threads(H, false)), 0, stream, x, gptr, dst, (int)B, (int)H, (int)S);
    CAL_CHECK_LAUNCH("k_add_pool");
    if (S > 1) {
        hipLaunchKernelGGL(k_pool_finish, dim3(cdiv(B * H, 256)), dim3(256), 0, stream, part, out, (int)(B * H), (int)S);
        CAL_CHECK_LAUNCH("k_pool_finish");
    }
    return 0;
}

CAL_EXPORT int cal_add_pool_bwd(const float* dout, const int64_t* batch, float* dx, int64_t N, int64_t H,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0 || H == 0) return 0;
    bool vec_ok = (H % 4 == 0) && aligned16(dout) && aligned16(dx);
    CAL_DISPATCH_VG((int)H, vec_ok, {
        hipLaunchKernelGGL((k_add_pool_bwd<VEC, G>), dim3(cdiv(N, 256 / G)), dim3(256), 0, stream, dout, batch, dx, (int)N, (int)H);
    });
    CAL_CHECK_LAUNCH("k_add_pool_bwd");
    return 0;
}
