// libcalhost -- HOST (CPU) implementation of the operator-level entry points of include/cal_hip.h, behind the same
// symbol names and argument lists ("each also has a host implementation behind the same symbol set
// so the boundary is testable in a GPU-less container"; BASELINE.json configs[0] is the reference's CPU plumbing run).
//
// Plain C++17 loops over the same GraphPlan structures the HIP kernels consume -- no HIP, no torch, nothing from
// oracle/.  Pointers are HOST pointers, `stream` is ignored, `ws` / `part` workspaces are accepted and left untouched.
// cal_amd selects this library by tensor residency (CPU tensors -> libcalhost, CUDA tensors -> libcalhip); data on the
// GPU never comes here and a missing libcalhip.so is an error, not a reason to run on the host.
//
// Every function cites the reference call site it stands for, like its HIP twin (cal_amd/csrc/*.hip): the formulas,
// the slot order of the plan (by edge id inside a row) and the summation order of every segment reduction are the
// same, so host and device agree to fp32 rounding.
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "cal_hip.h"

#define HOST_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define HOST_REQUIRE(cond, msg) do { if (!(cond)) { set_error("%s: %s", __func__, msg); return 2; } } while (0)

inline float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// counter-based attention-dropout keep decision: bit-identical to cal_amd/csrc/gat_common.hpp (32-bit finaliser, round 4)
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
inline float keep_scale(uint64_t seed, int64_t id, int k, int K, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    const uint32_t key = (uint32_t)(id * K + k);
    const uint32_t r = mix32(mix32(key + (uint32_t)seed) ^ (uint32_t)(seed >> 32));
    return ((float)r * (1.0f / 4294967296.0f)) >= p ? inv_keep : 0.f;
}

}  // namespace

HOST_EXPORT const char* cal_last_error() { return g_err; }
HOST_EXPORT int cal_version() { return 100; }

// ---- GraphPlan (remove_self_loops / add_self_loops of GCNConv.norm, gcn_conv.py:56-63; the scatter index of
// MessagePassing.propagate, gcn_conv.py:92): counting sort of the edges by destination and by source, slots of a
// row in edge-id order, explicit self loops dropped, the N added loops implicit.
HOST_EXPORT int cal_plan_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* nbr_dst,
                               int32_t* eid_dst, int32_t* rowptr_src, int32_t* nbr_src, int32_t* eid_src, int32_t* row32,
                               int32_t* col32, int32_t* work, int32_t* status, void*) {
    HOST_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) && E < (1ll << 31), "N/E out of int32 range");
    (void)work;
    *status = 0;
    std::fill(rowptr_dst, rowptr_dst + N + 1, 0);
    std::fill(rowptr_src, rowptr_src + N + 1, 0);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t r = edge_index[e], c = edge_index[E + e];
        if (r < 0 || r >= N || c < 0 || c >= N) {        // the reference would raise an index error
            *status |= 1;
            row32[e] = 0; col32[e] = 0;                     // treated as a dropped self loop
            continue;
        }
        row32[e] = (int32_t)r; col32[e] = (int32_t)c;
        if (r != c) { rowptr_dst[c + 1]++; rowptr_src[r + 1]++; }
    }
    for (int64_t v = 0; v < N; ++v) { rowptr_dst[v + 1] += rowptr_dst[v]; rowptr_src[v + 1] += rowptr_src[v]; }
    std::vector<int32_t> cd(rowptr_dst, rowptr_dst + N), cs(rowptr_src, rowptr_src + N);
    for (int64_t e = 0; e < E; ++e) {                       // ascending edge id = the slot order inside every row
        const int32_t r = row32[e], c = col32[e];
        if (r == c) continue;
        const int32_t p = cd[c]++, q = cs[r]++;
        nbr_dst[p] = r; eid_dst[p] = (int32_t)e;
        nbr_src[q] = c; eid_src[q] = (int32_t)e;
    }
    return 0;
}

// node offsets of every graph from the sorted `batch` vector (the Batch object of train_causal.py:174)
HOST_EXPORT int cal_graph_ptr(const int64_t* batch, int64_t N, int64_t B, int32_t* gptr, int32_t* status, void*) {
    HOST_REQUIRE(N >= 0 && B >= 0 && N < (1ll << 31), "bad sizes");
    int64_t prev = -1;
    for (int64_t i = 0; i <= N; ++i) {
        const int64_t cur = i == N ? B : batch[i];
        if (i < N && (cur < prev || cur >= B || cur < 0)) { *status |= 2; continue; }
        for (int64_t b = prev + 1; b <= cur && b <= B; ++b) gptr[b] = (int32_t)i;
        prev = cur;
    }
    return 0;
}

// ---- GCNConv.norm (gcn_conv.py:44-70): deg over the source index incl. the added loop, dis = deg^-1/2 (inf -> 0),
// norm_e = dis[row] w dis[col] in original edge order (0 on dropped self-loop edges)
HOST_EXPORT int cal_gcn_norm_fwd(const int32_t* rowptr_src, const int32_t* eid_src, const int32_t* row32, const int32_t* col32,
                                 const float* w, float loop_w, int64_t N, int64_t E, float* dis, float* norm_e, void*) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        float d = 0.f;
        if (w) for (int s = rowptr_src[v]; s < rowptr_src[v + 1]; ++s) d += w[eid_src[s]];
        else d = (float)(rowptr_src[v + 1] - rowptr_src[v]);
        d += loop_w;
        dis[v] = d == 0.f ? 0.f : 1.0f / sqrtf(d);
    }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        const int r = row32[e], c = col32[e];
        norm_e[e] = r != c ? dis[r] * (w ? w[e] : 1.f) * dis[c] : 0.f;
    }
    return 0;
}

// ---- propagate + message + update (+ fused ReLU), gcn_conv.py:92-104 / model.py:95: by-destination CSR for the
// forward, by-source CSR for the gradient w.r.t. h (its transpose)
HOST_EXPORT int cal_spmm_fwd(const int32_t* rowptr, const int32_t* nbr, const int32_t* eid, const float* norm_e,
                             const float* dis, float loop_w, const float* h, const float* bias, int relu, float* out,
                             int64_t N, int64_t H, void*) {
    if (N == 0 || H == 0) return 0;
    HOST_REQUIRE(h != out, "in-place aggregation is not supported");
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
        float* o = out + i * H;
        for (int64_t c = 0; c < H; ++c) o[c] = 0.f;
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
            const float n = norm_e[eid[s]];
            const float* hj = h + (int64_t)nbr[s] * H;
            for (int64_t c = 0; c < H; ++c) o[c] = fmaf(n, hj[c], o[c]);
        }
        const float nself = dis[i] * loop_w * dis[i];
        const float* hi = h + i * H;
        for (int64_t c = 0; c < H; ++c) {
            float v = fmaf(nself, hi[c], o[c]);
            if (bias) v += bias[c];
            if (relu) v = v > 0.f ? v : 0.f;
            o[c] = v;
        }
    }
    return 0;
}

HOST_EXPORT int64_t cal_colsum_parts(int64_t) { return 1; }
// ReLU backward + bias gradient (gcn_conv.py:103, model.py:95): dz = dout * (y > 0), dbias = column sums of dz
HOST_EXPORT int cal_relu_bwd_colsum(const float* dout, const float* y, float* dz, float* dbias, float* part, int64_t N,
                                    int64_t H, void*) {
    (void)part;
    std::vector<double> acc(dbias ? H : 0, 0.0);
    for (int64_t r = 0; r < N; ++r)
        for (int64_t c = 0; c < H; ++c) {
            float g = dout[r * H + c];
            if (y && !(y[r * H + c] > 0.f)) g = 0.f;
            if (dz) dz[r * H + c] = g;
            if (dbias) acc[c] += g;
        }
    if (dbias) for (int64_t c = 0; c < H; ++c) dbias[c] = (float)acc[c];
    return 0;
}

// ---- gradient w.r.t. the edge weights through propagate AND through the normalisation (gcn_conv.py:63-70,97
// differentiated): gn_e[e] = <dz[col_e], h[row_e]>, gself[v] = <dz[v], h[v]>, d deg, dw in original edge order
HOST_EXPORT int cal_gcn_norm_bwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                                 const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src, const int32_t* row32,
                                 const int32_t* col32, const float* w, const float* dis, float loop_w, const float* h,
                                 const float* dz, float* gn_e, float* gself, float* ddeg, float* dw, int64_t N, int64_t E,
                                 int64_t H, void*) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
        const float* di = dz + i * H;
        for (int s = rowptr_dst[i]; s <= rowptr_dst[i + 1]; ++s) {
            const bool self = s == rowptr_dst[i + 1];
            const float* hj = h + (int64_t)(self ? i : nbr_dst[s]) * H;
            float p = 0.f;
            for (int64_t c = 0; c < H; ++c) p = fmaf(di[c], hj[c], p);
            if (self) gself[i] = p; else gn_e[eid_dst[s]] = p;
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        float acc = 0.f;
        for (int s = rowptr_src[v]; s < rowptr_src[v + 1]; ++s) { const int e = eid_src[s]; acc += gn_e[e] * (w ? w[e] : 1.f) * dis[nbr_src[s]]; }
        for (int s = rowptr_dst[v]; s < rowptr_dst[v + 1]; ++s) { const int e = eid_dst[s]; acc += gn_e[e] * (w ? w[e] : 1.f) * dis[nbr_dst[s]]; }
        const float d = dis[v];
        acc += 2.f * gself[v] * d * loop_w;
        ddeg[v] = -0.5f * d * d * d * acc;
    }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        const int r = row32[e], c = col32[e];
        dw[e] = r != c ? gn_e[e] * dis[r] * dis[c] + ddeg[r] : 0.f;
    }
    return 0;
}

// ---- dense layers: x @ W (gcn_conv.py:75), torch.nn.Linear (model.py:46-75) and their gradients.
// C[M,N] = op(A) op(B) (+ bias[N]) (ReLU); transA: A stored [K,M]; transB: B stored [N,K]
HOST_EXPORT int64_t cal_gemm_ws(int64_t, int64_t, int64_t) { return 4; }
HOST_EXPORT int cal_gemm(int transA, int transB, const float* A, const float* B, float* C, const float* bias, int relu, float* ws,
                         int64_t M, int64_t N, int64_t K, void*) {
    (void)ws;
    if (M == 0 || N == 0) return 0;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float* c = C + m * N;
        if (transB) {                                       // B[n][k]: both operands contiguous along k
            for (int64_t n = 0; n < N; ++n) {
                const float* b = B + n * K;
                float acc = 0.f;
                if (transA) for (int64_t k = 0; k < K; ++k) acc = fmaf(A[k * M + m], b[k], acc);
                else { const float* a = A + m * K; for (int64_t k = 0; k < K; ++k) acc = fmaf(a[k], b[k], acc); }
                c[n] = acc;
            }
        } else {                                            // B[k][n]: axpy over rows of B
            for (int64_t n = 0; n < N; ++n) c[n] = 0.f;
            for (int64_t k = 0; k < K; ++k) {
                const float a = transA ? A[k * M + m] : A[m * K + k];
                const float* b = B + k * N;
                for (int64_t n = 0; n < N; ++n) c[n] = fmaf(a, b[n], c[n]);
            }
        }
        for (int64_t n = 0; n < N; ++n) {
            float v = c[n] + (bias ? bias[n] : 0.f);
            c[n] = relu ? (v > 0.f ? v : 0.f) : v;
        }
    }
    return 0;
}
HOST_EXPORT int cal_gemm_ks(int transB, const float* A, const float* B, float* C, const float* bias, int relu, int64_t M, int64_t N,
                            int64_t K, void* stream) {
    return cal_gemm(0, transB, A, B, C, bias, relu, nullptr, M, N, K, stream);
}

// ---- edge attention (model.py:97-104): softmax over 2 classes of Linear([x[row] || x[col]]) = P[row] + Q[col] + b with
// the per-node projections P = x W[:, :H]^T, Q = x W[:, H:]^T; att [2,E] (row 0 = edge_weight_c, row 1 = edge_weight_o)
HOST_EXPORT int cal_edge_att_fwd(const float* x, const float* W, const float* b, const int32_t* row32, const int32_t* col32,
                                 float* pq, float* att, int64_t N, int64_t E, int64_t H, void*) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        const float* xv = x + v * H;
        float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
        for (int64_t c = 0; c < H; ++c) {
            p0 = fmaf(xv[c], W[c], p0); p1 = fmaf(xv[c], W[2 * H + c], p1);
            q0 = fmaf(xv[c], W[H + c], q0); q1 = fmaf(xv[c], W[3 * H + c], q1);
        }
        pq[4 * v] = p0; pq[4 * v + 1] = p1; pq[4 * v + 2] = q0; pq[4 * v + 3] = q1;
    }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        const float* pr = pq + 4 * (int64_t)row32[e];
        const float* qc = pq + 4 * (int64_t)col32[e];
        const float l0 = pr[0] + qc[2] + b[0], l1 = pr[1] + qc[3] + b[1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.f / (e0 + e1);
        att[e] = e0 * inv; att[E + e] = e1 * inv;
    }
    return 0;
}
HOST_EXPORT int64_t cal_edge_att_bwd_ws(int64_t, int64_t, int64_t) { return 4; }
HOST_EXPORT int cal_edge_att_bwd(const float* x, const float* W, const float* att, const float* datt, const int32_t* rowptr_src,
                                 const int32_t* eid_src, const int32_t* rowptr_dst, const int32_t* eid_dst, float* dx, int accumulate,
                                 float* dW, float* db, float* ws, int64_t N, int64_t E, int64_t H, void*) {
    (void)ws;
    // d logit_0 = a0 a1 (dA0 - dA1) per edge (d logit_1 = -d logit_0); self-loop edges are absent from the CSR and
    // their weights never reach a conv (gcn_conv.py:56)
    std::vector<float> dl(E > 0 ? E : 1);
    for (int64_t e = 0; e < E; ++e) dl[e] = att[e] * att[E + e] * (datt[e] - datt[E + e]);
    std::vector<double> aw(2 * H, 0.0);
    double sb = 0.0;
    for (int64_t v = 0; v < N; ++v) {
        float sp = 0.f, sq = 0.f;
        for (int s = rowptr_src[v]; s < rowptr_src[v + 1]; ++s) sp += dl[eid_src[s]];
        for (int s = rowptr_dst[v]; s < rowptr_dst[v + 1]; ++s) sq += dl[eid_dst[s]];
        const float* xv = x + v * H;
        float* d = dx + v * H;
        for (int64_t c = 0; c < H; ++c) {
            const float wp = W[c] - W[2 * H + c], wq = W[H + c] - W[3 * H + c];
            const float base = accumulate ? d[c] : 0.f;
            d[c] = fmaf(sq, wq, fmaf(sp, wp, base));
            aw[c] += (double)sp * xv[c];
            aw[H + c] += (double)sq * xv[c];
        }
        sb += sp;
    }
    for (int64_t c = 0; c < 2 * H; ++c) { dW[c] = (float)aw[c]; dW[2 * H + c] = -(float)aw[c]; }
    db[0] = (float)sb; db[1] = -(float)sb;
    return 0;
}

// ---- node attention + split (model.py:106-111): a = softmax2(x Wn^T + bn), xc = a0 x, xo = a1 x
HOST_EXPORT int cal_node_att_split_fwd(const float* x, const float* Wn, const float* bn, float* att_n, float* xc, float* xo, int64_t N,
                                       int64_t H, void*) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        const float* xv = x + v * H;
        float l0 = 0.f, l1 = 0.f;
        for (int64_t c = 0; c < H; ++c) { l0 = fmaf(xv[c], Wn[c], l0); l1 = fmaf(xv[c], Wn[H + c], l1); }
        l0 += bn[0]; l1 += bn[1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.f / (e0 + e1), a0 = e0 * inv, a1 = e1 * inv;
        att_n[2 * v] = a0; att_n[2 * v + 1] = a1;
        for (int64_t c = 0; c < H; ++c) { xc[v * H + c] = a0 * xv[c]; xo[v * H + c] = a1 * xv[c]; }
    }
    return 0;
}
HOST_EXPORT int64_t cal_node_att_bwd_ws(int64_t, int64_t) { return 4; }
HOST_EXPORT int cal_node_att_split_bwd(const float* x, const float* Wn, const float* att_n, const float* dxc, const float* dxo, float* dx,
                                       float* dWn, float* dbn, float* ws, int64_t N, int64_t H, void*) {
    (void)ws;
    std::vector<double> aw(H, 0.0);
    double sb = 0.0;
    for (int64_t v = 0; v < N; ++v) {
        const float a0 = att_n[2 * v], a1 = att_n[2 * v + 1];
        const float* xv = x + v * H;
        float d0 = 0.f, d1 = 0.f;
        for (int64_t c = 0; c < H; ++c) { d0 = fmaf(dxc[v * H + c], xv[c], d0); d1 = fmaf(dxo[v * H + c], xv[c], d1); }
        const float dl0 = a0 * a1 * (d0 - d1);              // d logit_0 (softmax2 backward); d logit_1 = -dl0
        for (int64_t c = 0; c < H; ++c) {
            dx[v * H + c] = a0 * dxc[v * H + c] + a1 * dxo[v * H + c] + dl0 * (Wn[c] - Wn[H + c]);
            aw[c] += (double)dl0 * xv[c];
        }
        sb += dl0;
    }
    for (int64_t c = 0; c < H; ++c) { dWn[c] = (float)aw[c]; dWn[H + c] = -(float)aw[c]; }
    dbn[0] = (float)sb; dbn[1] = -(float)sb;
    return 0;
}

// ---- global_add_pool (model.py:115-116, 403-404): batch is sorted, so a graph is the node segment [gptr[b], gptr[b+1])
HOST_EXPORT int cal_add_pool_fwd(const float* x, const int32_t* gptr, float* out, float* part, int64_t B, int64_t H, int64_t S, void*) {
    (void)part; (void)S;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float* o = out + b * H;
        for (int64_t c = 0; c < H; ++c) o[c] = 0.f;
        for (int v = gptr[b]; v < gptr[b + 1]; ++v)
            for (int64_t c = 0; c < H; ++c) o[c] += x[(int64_t)v * H + c];
    }
    return 0;
}
HOST_EXPORT int cal_add_pool_bwd(const float* dout, const int64_t* batch, float* dx, int64_t N, int64_t H, void*) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) memcpy(dx + v * H, dout + batch[v] * H, sizeof(float) * H);
    return 0;
}

// ---- GATConv(hidden, hidden / heads, heads, dropout = p) of PyG (call sites model.py:340,390), after z = x W:
//   a_dst[i,k] = <z[i,k,:], att[k,:D]>, a_src[j,k] = <z[j,k,:], att[k,D:]>, e = LeakyReLU(a_dst[i] + a_src[j]),
//   alpha = softmax over the incoming edges of i (input self loops dropped, one loop per node added),
//   exp(e - max) / (sum + 1e-16); alpha~ = alpha keep / (1 - p) in training; out[i,k,:] = sum_j alpha~ z[j,k,:] + bias
HOST_EXPORT int cal_gat_fwd(const int32_t* rowptr, const int32_t* nbr, const int32_t* eid, const float* z, const float* att,
                            const float* bias, int relu, float slope, float p, uint64_t seed, float* out, float* adst, float* asrc,
                            float* mx, float* den, int64_t N, int64_t E, int64_t K, int64_t D, void*) {
    HOST_REQUIRE(K > 0 && D > 0, "bad head shape");
    HOST_REQUIRE(p >= 0.f && p < 1.f, "dropout p must be in [0,1)");
    const int64_t H = K * D;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < N * K; ++t) {
        const int64_t k = t % K;
        const float* zp = z + t * D;
        float sd = 0.f, ss = 0.f;
        for (int64_t d = 0; d < D; ++d) { sd = fmaf(zp[d], att[k * 2 * D + d], sd); ss = fmaf(zp[d], att[k * 2 * D + D + d], ss); }
        adst[t] = sd; asrc[t] = ss;
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i)
        for (int64_t k = 0; k < K; ++k) {
            const float ad = adst[i * K + k];
            const float eself = lrelu(ad + asrc[i * K + k], slope);
            float m = eself;
            for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) m = fmaxf(m, lrelu(ad + asrc[(int64_t)nbr[s] * K + k], slope));
            float* o = out + i * H + k * D;
            for (int64_t d = 0; d < D; ++d) o[d] = 0.f;
            float lsum = 0.f;
            for (int s = rowptr[i]; s <= rowptr[i + 1]; ++s) {
                const bool self = s == rowptr[i + 1];
                const int64_t j = self ? i : nbr[s];
                const float pe = expf((self ? eself : lrelu(ad + asrc[j * K + k], slope)) - m);
                lsum += pe;
                const float wgt = pe * keep_scale(seed, self ? E + i : (int64_t)eid[s], (int)k, (int)K, p, inv_keep);
                const float* zj = z + j * H + k * D;
                for (int64_t d = 0; d < D; ++d) o[d] = fmaf(wgt, zj[d], o[d]);
            }
            const float dn = lsum + 1e-16f;
            for (int64_t d = 0; d < D; ++d) {
                float v = o[d] / dn + (bias ? bias[k * D + d] : 0.f);
                o[d] = relu ? (v > 0.f ? v : 0.f) : v;
            }
            mx[i * K + k] = m; den[i * K + k] = dn;
        }
    return 0;
}
HOST_EXPORT int64_t cal_gat_bwd_ws(int64_t, int64_t, int64_t, int64_t) { return 4; }
// gout: gradient at the layer output (already masked by the ReLU if one was fused).  dz [N,K*D], datt [K,2D].
HOST_EXPORT int cal_gat_bwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst, const int32_t* rowptr_src,
                            const int32_t* nbr_src, const int32_t* eid_src, const float* z, const float* att, const float* adst,
                            const float* asrc, const float* mx, const float* den, const float* gout, float slope, float p,
                            uint64_t seed, float* dz, float* datt, float* ws, int64_t N, int64_t E, int64_t K, int64_t D, void*) {
    (void)ws;
    const int64_t H = K * D;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    std::vector<float> draw((size_t)(E + N) * K, 0.f), dadst((size_t)N * K, 0.f), dasrc((size_t)N * K, 0.f);
    // by destination: d(raw logit) of every incoming slot (softmax + LeakyReLU backward), its row sum = d a_dst
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i)
        for (int64_t k = 0; k < K; ++k) {
            const float ad = adst[i * K + k], m = mx[i * K + k], dn = den[i * K + k];
            const float* gi = gout + i * H + k * D;
            const int s0 = rowptr_dst[i], s1 = rowptr_dst[i + 1];
            float S = 0.f;
            for (int s = s0; s <= s1; ++s) {
                const bool self = s == s1;
                const int64_t j = self ? i : nbr_dst[s], id = self ? E + i : (int64_t)eid_dst[s];
                const float* zj = z + j * H + k * D;
                float dot = 0.f;
                for (int64_t d = 0; d < D; ++d) dot = fmaf(gi[d], zj[d], dot);
                const float alpha = expf(lrelu(ad + asrc[j * K + k], slope) - m) / dn;
                const float dalpha = dot * keep_scale(seed, id, (int)k, (int)K, p, inv_keep);
                S = fmaf(alpha, dalpha, S);
                draw[id * K + k] = dalpha;
            }
            float rowsum = 0.f;
            for (int s = s0; s <= s1; ++s) {
                const bool self = s == s1;
                const int64_t j = self ? i : nbr_dst[s], id = self ? E + i : (int64_t)eid_dst[s];
                const float raw = ad + asrc[j * K + k];
                const float alpha = expf(lrelu(raw, slope) - m) / dn;
                const float dr = alpha * (draw[id * K + k] - S) * (raw > 0.f ? 1.f : slope);
                rowsum += dr;
                draw[id * K + k] = dr;
            }
            dadst[i * K + k] = rowsum;
        }
    // by source: dz[j] = sum over the edges leaving j (and its loop) of alpha~ g[dst] + d a_dst att[:D] + d a_src att[D:]
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t j = 0; j < N; ++j)
        for (int64_t k = 0; k < K; ++k) {
            const float as = asrc[j * K + k];
            float* o = dz + j * H + k * D;
            for (int64_t d = 0; d < D; ++d) o[d] = 0.f;
            float das = 0.f;
            const int s0 = rowptr_src[j], s1 = rowptr_src[j + 1];
            for (int s = s0; s <= s1; ++s) {
                const bool self = s == s1;
                const int64_t i = self ? j : nbr_src[s], id = self ? E + j : (int64_t)eid_src[s];
                const float alpha = expf(lrelu(adst[i * K + k] + as, slope) - mx[i * K + k]) / den[i * K + k];
                const float wgt = alpha * keep_scale(seed, id, (int)k, (int)K, p, inv_keep);
                const float* gi = gout + i * H + k * D;
                for (int64_t d = 0; d < D; ++d) o[d] = fmaf(wgt, gi[d], o[d]);
                das += draw[id * K + k];
            }
            dasrc[j * K + k] = das;
            const float dad = dadst[j * K + k];
            for (int64_t d = 0; d < D; ++d) o[d] = fmaf(das, att[k * 2 * D + D + d], fmaf(dad, att[k * 2 * D + d], o[d]));
        }
    // d att[k,:D] = sum_v d a_dst[v,k] z[v,k,:], d att[k,D:] = sum_v d a_src[v,k] z[v,k,:]
    std::vector<double> acc((size_t)2 * H, 0.0);
    for (int64_t v = 0; v < N; ++v)
        for (int64_t k = 0; k < K; ++k) {
            const double da = dadst[v * K + k], ds = dasrc[v * K + k];
            const float* zv = z + v * H + k * D;
            for (int64_t d = 0; d < D; ++d) { acc[k * 2 * D + d] += da * zv[d]; acc[k * 2 * D + D + d] += ds * zv[d]; }
        }
    for (int64_t c = 0; c < 2 * H; ++c) datt[c] = (float)acc[c];
    return 0;
}
// keep mask (1/0) as floats, [E + N, K]: row e < E for original edge e, row E + i for node i's loop
HOST_EXPORT int cal_gat_dropout_mask(uint64_t seed, int64_t E, int64_t N, int64_t K, float p, float* mask, void*) {
    for (int64_t t = 0; t < (E + N) * K; ++t) mask[t] = keep_scale(seed, t / K, (int)(t % K), (int)K, p, 1.f);
    return 0;
}

// Batch.from_data_list for graphs that live in ONE host-side concatenation (train_causal.py:13-15; cal_amd/data.py::_HostConcat):
// features copied, edge_index rebased by the running node offset, batch vector and labels written.  Same entry point as
// libcalhip.so (collate.hip) -- it only ever sees host pointers, so a machine with the host library alone serves DataLoader too.
HOST_EXPORT int cal_collate_host(const float* X, const int64_t* EI, int64_t Etot, int64_t F, const int64_t* node_ptr,
                                 const int64_t* edge_ptr, const int64_t* Y, const int64_t* idx, int64_t B, float* xo,
                                 int64_t* eio, int64_t Eout, int64_t* batcho, int64_t* yo) {
    HOST_REQUIRE(X && EI && node_ptr && edge_ptr && Y && idx && xo && eio && batcho && yo && B >= 0 && F > 0, "bad arguments");
    // offsets of every member graph first (serial, B additions), then the copies -- 128 scattered graphs of a 10 MB dataset are
    // ~0.7 MB of cache-missing reads: 120 us on one core, the largest piece of a host-collated step (0.34 ms against 0.22 ms on
    // the GPU) -- over a few threads (a fixed small team: the work is 100 us, the caller's loop runs it every 0.3 ms)
    std::vector<int64_t> noff((size_t)B + 1), eoff((size_t)B + 1);
    noff[0] = 0; eoff[0] = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t g = idx[b];
        const int64_t nn = node_ptr[g + 1] - node_ptr[g], ne = edge_ptr[g + 1] - edge_ptr[g];
        HOST_REQUIRE(nn >= 0 && ne >= 0 && eoff[b] + ne <= Eout, "offsets out of range");
        noff[b + 1] = noff[b] + nn; eoff[b + 1] = eoff[b] + ne;
    }
    const int64_t edges = eoff[B];
    static const int team_max = [] { const char* v = getenv("CAL_HOST_COLLATE_THREADS"); const int n = v ? atoi(v) : 4; return n < 1 ? 1 : n; }();
    const int team = B >= 32 ? team_max : 1;
#pragma omp parallel for schedule(static) num_threads(team) if (team > 1)
    for (int64_t b = 0; b < B; ++b) {
        const int64_t g = idx[b];
        const int64_t nb = node_ptr[g], nn = node_ptr[g + 1] - nb, eb = edge_ptr[g], ne = edge_ptr[g + 1] - eb;
        const int64_t nodes = noff[b], e0 = eoff[b];
        std::copy(X + nb * F, X + (nb + nn) * F, xo + nodes * F);
        const int64_t* src = EI + eb;
        const int64_t* dst = EI + Etot + eb;
        for (int64_t k = 0; k < ne; ++k) { eio[e0 + k] = src[k] + nodes; eio[Eout + e0 + k] = dst[k] + nodes; }
        std::fill(batcho + nodes, batcho + nodes + nn, b);
        yo[b] = Y[g];
    }
    HOST_REQUIRE(edges == Eout, "edge count mismatch");
    return 0;
}
