// Per-graph fused backward of everything between the last backbone convolution and the two causal convolutions for WIDE graphs
// (129 .. 256 nodes, <= 2048 edges: the reference's default SPMotif shape, opts.py:18) -- k_att_bwd_graph (engine_attbwd.hpp) with
// its edge phase restated SPARSELY (round 6).  The 64-node kernel sums by source and by destination through dense 64 x 64 LDS
// blocks; at 256 nodes those are 3 x 263 KB.  Here every per-edge quantity lives in LDS BY LOCAL EDGE ID (a graph's edges are a
// contiguous run of edge_index columns, [eptr[b], eptr[b + 1])), and the per-node sums walk the node's rows of the two CSR views,
// eight lanes per node (SPMotif's wheel bases have hubs of ~230 edges), slots -> local edge id -> value:
//   S_k[e] = g_k[e] w_k[e] dis_k[col_e]   (what edge e adds to d deg_k of its ROW node),   D_k[e] = g_k[e] w_k[e] dis_k[row_e]
//   d deg_k[v] = -1/2 dis_k[v]^3 (sum_{row_e = v} S_k[e] + sum_{col_e = v} D_k[e] + 2 gself_k[v] dis_k[v] loop_w)        (gcn_conv.py:63-70)
//   dl[e]      = w_c w_o ((g_c dis_c[r] dis_c[c] + d deg_c[r]) - (g_o dis_o[r] dis_o[c] + d deg_o[r])),  r != c          (edge softmax2)
//   sp[v] = sum_{row_e = v} dl[e],  sq[v] = sum_{col_e = v} dl[e]
// then the row phase of k_att_bwd_graph unchanged.  Replaces k_normbwd_node2 -> k_normbwd_edge -> k_att_bwd (three node- / edge-
// parallel launches with ddeg / dl through HBM: 48 us of the 500 us step at 30 k rows).  Inputs in EDGE-ID order (att, gn as the
// wide convolution backward leaves them): one round of loads, no slot-order twins.
//   grid (nsplit B), 512 threads: the nsplit workgroups of a graph all run the (cheap) edge phase, each takes 1 / nsplit of the rows
//   (nsplit = CUs / graphs, at most 8: 32 graphs of 235 nodes -> one row-loop iteration per workgroup).
// Sums are added in a fixed order (slots of a row strided over 8 lanes, DPP sums): bit-reproducible.
#pragma once
#include "engine_attbwd.hpp"

namespace cal {

constexpr int AW_T = 256;                 // nodes per graph
constexpr int AW_E = 2048;                // edges per graph

struct AttBwdWideArgs {
    AttBwdArgs a;                              // a.dl unused; a.dbias / a.dWn / a.dWe are per-workgroup partial rows
    const int* gptr; const int* eptr;          // node / edge range of every graph
    const int* row32; const int* col32;        // [E] endpoints (global node ids)
    const float* att;                          // [2,E] attention weights (context, objects), edge-id order
    const float* dis;                          // [2,N] deg^-1/2 of the weighted degrees
    const float* gn; const float* gn2;         // [2,E] <dOut[col_e], z[row_e]> of column slice 0 / 1 (gn2 null: one slice), edge-id order
    const float* gself; const float* gself2;   // [2,N]
    float loop_w;
    int64_t E; int N;
    int* status;
};

template <int VEC, int G>
__global__ void __launch_bounds__(512) k_att_bwd_wide(const AttBwdWideArgs ga, int relu, int H, int nsplit) {
    constexpr int RPB = 512 / G, UR = 2, EU = AW_E / 512;
    warm_kernargs<sizeof(AttBwdWideArgs) + 16>();
    const AttBwdArgs& a = ga.a;
    __shared__ float S_c[AW_E], D_c[AW_E], S_o[AW_E], D_o[AW_E];          // per local edge id; S_c doubles as dl after the d deg phase
    __shared__ unsigned short es_l[AW_E], ed_l[AW_E];                     // local edge id of every by-source / by-destination slot
    __shared__ double red[8][4][G * VEC];                                 // column sums: [wave][quantity][column]
    __shared__ double sc_lds[2][8];
    __shared__ float bnk_s[8][G * VEC];
    __shared__ int ps_s[AW_T + 1], pd_s[AW_T + 1];
    __shared__ float dis_c_s[AW_T], dis_o_s[AW_T], dd_c_s[AW_T], dd_o_s[AW_T], spv_s[AW_T], sqv_s[AW_T], gs_c_s[AW_T], gs_o_s[AW_T];
    const int b = (int)blockIdx.x / nsplit, half = (int)blockIdx.x - b * nsplit, t = threadIdx.x, grp = t / G, l = t % G;     // half: this workgroup's part of the rows
    const int g0 = ga.gptr[b], rows = ga.gptr[b + 1] - g0, e0 = ga.eptr[b], ne = ga.eptr[b + 1] - e0;
    const int64_t E = ga.E;
    const int N = ga.N;
    using V = Vec<VEC>;
    const int c = l * VEC;
    const bool cok = c < H;
    const int cc = min(c, H - VEC);
    double cs[4][VEC];                       // d bias, d Wn, d We (source half), d We (destination half)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) cs[q][j] = 0.0;
    double sdl = 0.0, ssp = 0.0;
    const bool live = rows > 0 && rows <= AW_T && ne >= 0 && ne <= AW_E;
    if (rows > 0 && !live && t == 0) atomicOr(ga.status, 8);
    if (live) {
        // ---- round 1 (independent of the graph): per-column constants; the eight BatchNorm sums per column by a STRIPED reader -----
        float mc[VEC], rc[VEC], gc[VEC], m1c[VEC], m2c[VEC], mo[VEC], ro[VEC], go[VEC], m1o[VEC], m2o[VEC];
        float wn[VEC], wp[VEC], wq[VEC];
        float w0[VEC], w1[VEC], w2[VEC], w3[VEC], w4[VEC], w5[VEC];
        const int oc = min(t, H - 1);
        StripeVal sv[8];
        sv[0] = stripe_load(a.bnc.sum, oc, a.bnc.ss); sv[1] = stripe_load(a.bnc.sq, oc, a.bnc.ss);
        sv[2] = stripe_load(a.bno.sum, oc, a.bno.ss); sv[3] = stripe_load(a.bno.sq, oc, a.bno.ss);
        sv[4] = stripe_load(a.dsc, oc, a.dss); sv[5] = stripe_load(a.dpc, oc, a.dss);
        sv[6] = stripe_load(a.dso, oc, a.dss); sv[7] = stripe_load(a.dpo, oc, a.dss);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            gc[j] = a.bnc.gamma[cc + j]; go[j] = a.bno.gamma[cc + j];
            w0[j] = a.Wn[cc + j]; w1[j] = a.Wn[H + cc + j];
            w2[j] = a.We[cc + j]; w3[j] = a.We[2 * H + cc + j]; w4[j] = a.We[H + cc + j]; w5[j] = a.We[3 * H + cc + j];
        }
        // ---- round 2 (needs g0 / e0): the graph's edges by id, both CSR views' rows, per-node values, this group's first rows ---------
        const float* gself2 = ga.gself2 ? ga.gself2 : ga.gself;
        const float* gn2 = ga.gn2 ? ga.gn2 : ga.gn;
        const float* dxhc2 = a.dxhc2 ? a.dxhc2 : a.dxhc;
        const float* dxho2 = a.dxho2 ? a.dxho2 : a.dxho;
        const float f2 = ga.gn2 ? 1.f : 0.f;                  // weight of the slice-1 partials
        const int tn = min(t, rows);
        int psv = a.gs.ptr[g0 + tn], pdv = a.gd.ptr[g0 + tn];
        const int vn = g0 + min(t, rows - 1);
        float dcv = ga.dis[vn], dov = ga.dis[(size_t)N + vn];
        float gsc = ga.gself[vn], gso = ga.gself[(size_t)N + vn], gsc2 = gself2[vn], gso2 = gself2[(size_t)N + vn];
        int er[EU], ec[EU], sse[EU], sde[EU];
        float egc[EU], ego[EU], ewc[EU], ewo[EU], egc2[EU], ego2[EU];
        const int64_t ehi = max(E - 1, (int64_t)0);
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int64_t e = min((int64_t)e0 + max(min(t + u * 512, ne - 1), 0), ehi);
            er[u] = ga.row32[e]; ec[u] = ga.col32[e];
            ewc[u] = ga.att[e]; ewo[u] = ga.att[E + e];
            egc[u] = ga.gn[e]; ego[u] = ga.gn[E + e]; egc2[u] = gn2[e]; ego2[u] = gn2[E + e];
            sse[u] = a.gs.eid[e]; sde[u] = a.gd.eid[e];          // (slot ranges = edge ranges: no self loops, one slot per edge in either view)
        }
        float a0[UR], a1[UR];
        V x4[UR], hc4[UR], ho4[UR], hc2[UR], ho2[UR];
        auto load_rows = [&](int i0) {
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const size_t v = (size_t)(g0 + min(i0 + u * RPB, rows - 1));
                a0[u] = a.anode[2 * v]; a1[u] = a.anode[2 * v + 1];
                x4[u] = V::ld(a.x + v * H + cc); hc4[u] = V::ld(a.dxhc + v * H + cc); ho4[u] = V::ld(a.dxho + v * H + cc);
                hc2[u] = V::ld(dxhc2 + v * H + cc); ho2[u] = V::ld(dxho2 + v * H + cc);
            }
        };
        const int hrows = (rows + nsplit - 1) / nsplit;
        const int rbeg = half * hrows, rend = min(rows, rbeg + hrows);      // this workgroup's rows
        load_rows(rbeg + grp);
        // pins: nothing above may sink below this point
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            asm volatile("" : "+v"(gc[j]), "+v"(go[j]), "+v"(w0[j]), "+v"(w1[j]), "+v"(w2[j]), "+v"(w3[j]), "+v"(w4[j]), "+v"(w5[j]));
#pragma unroll
        for (int q = 0; q < 8; ++q) stripe_pin(sv[q]);
        asm volatile("" : "+v"(psv), "+v"(pdv), "+v"(dcv), "+v"(dov), "+v"(gsc), "+v"(gso), "+v"(gsc2), "+v"(gso2));
#pragma unroll
        for (int u = 0; u < EU; ++u)
            asm volatile("" : "+v"(er[u]), "+v"(ec[u]), "+v"(sse[u]), "+v"(sde[u]), "+v"(egc[u]), "+v"(ego[u]), "+v"(egc2[u]), "+v"(ego2[u]),
                         "+v"(ewc[u]), "+v"(ewo[u]));
#pragma unroll
        for (int u = 0; u < EU; ++u) { egc[u] = fmaf(f2, egc2[u], egc[u]); ego[u] = fmaf(f2, ego2[u], ego[u]); }
        gsc = fmaf(f2, gsc2, gsc); gso = fmaf(f2, gso2, gso);
        if (t < H) {   // BatchNorm mean / rstd of bnc and bno from their batch statistics (training-mode backward: never running stats)
            const double inv = (double)a.bnc.inv_n;
            const double bsc = stripe_total(sv[0], a.bnc.ss), bqc = stripe_total(sv[1], a.bnc.ss);
            const double bso = stripe_total(sv[2], a.bno.ss), bqo = stripe_total(sv[3], a.bno.ss);
            const double m_c = bsc * inv, v_c = bqc * inv - m_c * m_c, m_o = bso * inv, v_o = bqo * inv - m_o * m_o;
            bnk_s[0][t] = (float)m_c; bnk_s[1][t] = 1.0f / sqrtf((float)(v_c > 0.0 ? v_c : 0.0) + a.bnc.eps);
            bnk_s[2][t] = (float)m_o; bnk_s[3][t] = 1.0f / sqrtf((float)(v_o > 0.0 ? v_o : 0.0) + a.bno.eps);
            bnk_s[4][t] = (float)(stripe_total(sv[4], a.dss) * inv); bnk_s[5][t] = (float)(stripe_total(sv[5], a.dss) * inv);
            bnk_s[6][t] = (float)(stripe_total(sv[6], a.dss) * inv); bnk_s[7][t] = (float)(stripe_total(sv[7], a.dss) * inv);
        }
        if (t <= rows) { ps_s[t] = psv - e0; pd_s[t] = pdv - e0; }
        if (t < rows) { dis_c_s[t] = dcv; dis_o_s[t] = dov; gs_c_s[t] = gsc; gs_o_s[t] = gso; }
        int lr[EU], lc[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int s = t + u * 512;
            lr[u] = er[u] - g0; lc[u] = ec[u] - g0;
            if (s < ne) {
                if (lr[u] < 0 || lr[u] >= rows || lc[u] < 0 || lc[u] >= rows) atomicOr(ga.status, 16);     // an edge that leaves its graph
                const int le_s = sse[u] - e0, le_d = sde[u] - e0;
                if (le_s < 0 || le_s >= ne || le_d < 0 || le_d >= ne) atomicOr(ga.status, 16);
                es_l[s] = (unsigned short)min(max(le_s, 0), ne - 1);
                ed_l[s] = (unsigned short)min(max(le_d, 0), ne - 1);
            }
            lr[u] = min(max(lr[u], 0), rows - 1); lc[u] = min(max(lc[u], 0), rows - 1);
        }
        __syncthreads();
        // ---- d deg: per edge the two contributions, then per node its rows of both views (8 lanes per node, 64 nodes per pass) ----------
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int s = t + u * 512;
            if (s < ne) {
                const float tc = egc[u] * ewc[u], to = ego[u] * ewo[u];
                S_c[s] = tc * dis_c_s[lc[u]]; D_c[s] = tc * dis_c_s[lr[u]];
                S_o[s] = to * dis_o_s[lc[u]]; D_o[s] = to * dis_o_s[lr[u]];
            }
        }
        __syncthreads();
        for (int v0 = 0; v0 < rows; v0 += 64) {
            const int v = v0 + (t >> 3), p = t & 7, vc = min(v, rows - 1);
            float ac = 0.f, ao = 0.f;
            const int s0 = ps_s[vc], s1 = v < rows ? ps_s[vc + 1] : s0, d0 = pd_s[vc], d1 = v < rows ? pd_s[vc + 1] : d0;
            for (int s = s0 + p; s < s1; s += 8) { const int le = es_l[s]; ac += S_c[le]; ao += S_o[le]; }
            for (int s = d0 + p; s < d1; s += 8) { const int le = ed_l[s]; ac += D_c[le]; ao += D_o[le]; }
            ac = group_sum<8>(ac); ao = group_sum<8>(ao);
            if (v < rows && p == 0) {
                const float dc = dis_c_s[v], dq = dis_o_s[v];
                dd_c_s[v] = -0.5f * dc * dc * dc * (ac + 2.f * gs_c_s[v] * dc * ga.loop_w);
                dd_o_s[v] = -0.5f * dq * dq * dq * (ao + 2.f * gs_o_s[v] * dq * ga.loop_w);
            }
        }
        __syncthreads();
        // ---- dl per edge (an input self loop carries no gradient: k_normbwd_edge), into S_c; summed by source / by destination ----------
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int s = t + u * 512;
            if (s < ne) {
                const int r = lr[u], q = lc[u];
                const float xc = egc[u] * dis_c_s[r] * dis_c_s[q] + dd_c_s[r];
                const float xo = ego[u] * dis_o_s[r] * dis_o_s[q] + dd_o_s[r];
                S_c[s] = r != q ? a.fedge * ewc[u] * ewo[u] * (xc - xo) : 0.f;
            }
        }
        __syncthreads();
        for (int v0 = 0; v0 < rows; v0 += 64) {
            const int v = v0 + (t >> 3), p = t & 7, vc = min(v, rows - 1);
            float asp = 0.f, asq = 0.f;
            const int s0 = ps_s[vc], s1 = v < rows ? ps_s[vc + 1] : s0, d0 = pd_s[vc], d1 = v < rows ? pd_s[vc + 1] : d0;
            for (int s = s0 + p; s < s1; s += 8) asp += S_c[es_l[s]];
            for (int s = d0 + p; s < d1; s += 8) asq += S_c[ed_l[s]];
            asp = group_sum<8>(asp); asq = group_sum<8>(asq);
            if (v < rows && p == 0) { spv_s[v] = asp; sqv_s[v] = asq; }
        }
        __syncthreads();
        // ---- row phase (k_att_bwd_graph's) ---------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool on = cok && c + j < H;
            mc[j] = bnk_s[0][cc + j]; rc[j] = bnk_s[1][cc + j]; mo[j] = bnk_s[2][cc + j]; ro[j] = bnk_s[3][cc + j];
            gc[j] = on ? gc[j] * rc[j] : 0.f; go[j] = on ? go[j] * ro[j] : 0.f;
            m1c[j] = bnk_s[4][cc + j]; m2c[j] = bnk_s[5][cc + j];
            m1o[j] = bnk_s[6][cc + j]; m2o[j] = bnk_s[7][cc + j];
            wn[j] = on ? w0[j] - w1[j] : 0.f; wp[j] = on ? w2[j] - w3[j] : 0.f; wq[j] = on ? w4[j] - w5[j] : 0.f;
        }
        for (int i0 = rbeg + grp; i0 < rend; i0 += RPB * UR) {
            if (i0 != rbeg + grp) load_rows(i0);
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                x4[u].pin(); hc4[u].pin(); ho4[u].pin(); hc2[u].pin(); ho2[u].pin();
                hc4[u].fma(f2, hc2[u]); ho4[u].fma(f2, ho2[u]);
                asm volatile("" : "+v"(a0[u]), "+v"(a1[u]));
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int i = i0 + u * RPB;
                float xv[VEC], dxc[VEC], dxo[VEC];
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    xv[j] = cok ? x4[u].get(j) : 0.f;
                    const float xcn = (a0[u] * xv[j] - mc[j]) * rc[j], xon = (a1[u] * xv[j] - mo[j]) * ro[j];
                    dxc[j] = cok ? gc[j] * (hc4[u].get(j) - m1c[j] - xcn * m2c[j]) : 0.f;
                    dxo[j] = cok ? go[j] * (ho4[u].get(j) - m1o[j] - xon * m2o[j]) : 0.f;
                    d0 = fmaf(dxc[j], xv[j], d0);
                    d1 = fmaf(dxo[j], xv[j], d1);
                }
                d0 = group_sum<G>(d0); d1 = group_sum<G>(d1);
                const float dl0 = a.fnode * a0[u] * a1[u] * (d0 - d1);
                if (i < rend) {
                    const float spv = spv_s[i], sqv = sqv_s[i];
                    if (l == 0) { sdl += (double)dl0; ssp += (double)spv; }
                    if (cok) {
                        float o[VEC];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            float dx = a0[u] * dxc[j] + a1[u] * dxo[j] + dl0 * wn[j] + spv * wp[j] + sqv * wq[j];
                            if (relu && !(xv[j] > 0.f)) dx = 0.f;
                            o[j] = dx;
                            cs[0][j] += (double)dx;
                            cs[1][j] += (double)(dl0 * xv[j]);
                            cs[2][j] += (double)(spv * xv[j]);
                            cs[3][j] += (double)(sqv * xv[j]);
                        }
                        V ov;
                        if constexpr (VEC == 4) ov.v = make_float4(o[0], o[1], o[2], o[3]); else ov.v = o[0];
                        ov.st(a.dZ + (size_t)(g0 + i) * H + c);
                    }
                }
            }
        }
    }
    // ---- this workgroup's partial rows (zeros for an empty / rejected graph: the rows must exist) ----------------------------------
    const int wv = t >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double v = cs[q][j];
            if (G < 64) for (int off = G; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if ((t & 63) < G) red[wv][q][l * VEC + j] = v;
        }
    if (l == 0) {
        double s0 = sdl, s1 = ssp;
        if (G < 64) for (int off = G; off < 64; off <<= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
        if ((t & 63) == 0) { sc_lds[0][wv] = s0; sc_lds[1][wv] = s1; }
    }
    __syncthreads();
    for (int idx = t; idx < 4 * H; idx += 512) {
        const int q = idx / H, col = idx - q * H;
        double tot = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) tot += red[w8][q][col];
        if (q == 0) { if (a.dbias.on()) a.dbias.add(col, tot); }
        else if (q == 1) a.dWn.add(col, tot);
        else if (q == 2) a.dWe.add(col, tot);
        else a.dWe.add(H + col, tot);
    }
    if (t == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < 8; ++k) { t0 += sc_lds[0][k]; t1 += sc_lds[1][k]; }
        a.dWn.add(H, t0);
        a.dWe.add(2 * H, t1);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// k_att_fwd_graph (engine_plan.hpp) for graphs of up to AW_T nodes / AW_E edges: node attention + edge attention + weighted degrees
// of one graph in one kernel (model.py:97-111, gcn_conv.py:63-68) -- the rows in passes of 4 x (512 / G) (four passes at H = 128),
// four CSR slots per lane instead of two; otherwise the same kernel.  Replaces k_node_att_fwd + k_edge_att_deg (+ a finishing launch)
// of the node-level chain behind the wide convolutions.
// ------------------------------------------------------------------------------------------------------------------
template <int VEC, int G>
__global__ void __launch_bounds__(512) k_att_fwd_wide(const int* __restrict__ gptr, const CSR gs, const float* __restrict__ x,
                                                       const float* __restrict__ Wn, const float* __restrict__ bn,
                                                       const float* __restrict__ We, const float* __restrict__ be,
                                                       float* __restrict__ anode, float* __restrict__ pq, float* __restrict__ att,
                                                       float* __restrict__ dis_c, float* __restrict__ dis_o, const Acc stc_sum,
                                                       const Acc stc_sq, const Acc sto_sum, const Acc sto_sq, float loop_w, int H,
                                                       int64_t E, int* __restrict__ status, float fnode, float fedge,
                                                       const int* __restrict__ eptr) {
    // fnode / fedge: 1, or 0 for without_node_attention / without_edge_attention (equal logits -> constant 0.5 masks)
    constexpr int RPB = 512 / G, NB = AW_T / (4 * RPB), MAXR = 4 * RPB * NB, SU = AW_E / 512, GE = AW_E;
    static_assert(NB >= 1 && MAXR == AW_T, "row batches");
    __shared__ double lds[4 * 512 * (VEC == 4 ? 4 : 1)];
    __shared__ float4 pq_s[MAXR];
    warm_kernargs<320>();
    const int b = blockIdx.x, t = threadIdx.x, grp = t / G, l = t % G;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0;
    // the graph's by-source CSR rows for the edge phase: requested NOW, with the row loads (its slot range is its edge range,
    // eptr; read after the row phase they were two more dependent rounds of global loads in the middle of the kernel)
    const int e0 = eptr[b], ne = eptr[b + 1] - e0;
    const int rcl = max(rows, 0);
    int pv = gs.ptr[g0 + min(t, rcl)], pn = gs.ptr[g0 + min(t + 1, rcl)];
    int nd[SU], ed[SU];
    const int slot_hi = max(gs.nnz - 1, 0);
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = min(e0 + max(min(t + u * 512, ne - 1), 0), slot_hi);
        nd[u] = gs.nbr[s];
        ed[u] = gs.eid[s];
    }
    using V = Vec<VEC>;
    const int c = l * VEC, cc = min(c, H - VEC);
    const bool cok = c < H;
    double sc1[VEC], sc2[VEC], so1[VEC], so2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sc1[j] = sc2[j] = so1[j] = so2[j] = 0.0; }
    if (rows > MAXR) { if (t == 0) atomicOr(status, 8); }
    else if (rows > 0) {
        V w[6], xv[4];
        w[0] = V::ld(Wn + cc); w[1] = V::ld(Wn + H + cc); w[2] = V::ld(We + cc);
        w[3] = V::ld(We + 2 * H + cc); w[4] = V::ld(We + H + cc); w[5] = V::ld(We + 3 * H + cc);
        const float b0 = bn[0], b1 = bn[1];
        V xn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xn[u] = V::ld(x + (size_t)(g0 + min(grp + u * RPB, rows - 1)) * H + cc);
#pragma unroll
        for (int u = 0; u < 6; ++u) w[u].pin();
        for (int nb = 0; nb < NB && nb * 4 * RPB < rows; ++nb) {          // four rows per lane group and pass; the next pass's rows are on their way
#pragma unroll
        for (int u = 0; u < 4; ++u) { xn[u].pin(); xv[u] = xn[u]; if (!cok) xv[u] = V::zero(); }
        if ((nb + 1) * 4 * RPB < rows) {
#pragma unroll
            for (int u = 0; u < 4; ++u) xn[u] = V::ld(x + (size_t)(g0 + min(grp + ((nb + 1) * 4 + u) * RPB, rows - 1)) * H + cc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = grp + (nb * 4 + u) * RPB;
            const float l0 = fnode * (group_sum<G>(xv[u].dot(w[0])) + b0), l1 = fnode * (group_sum<G>(xv[u].dot(w[1])) + b1);
            const float p0 = group_sum<G>(xv[u].dot(w[2])), p1 = group_sum<G>(xv[u].dot(w[3]));
            const float q0 = group_sum<G>(xv[u].dot(w[4])), q1 = group_sum<G>(xv[u].dot(w[5]));
            const float m = fmaxf(l0, l1);
            const float e0 = expf(l0 - m), e1 = expf(l1 - m);
            const float inv = 1.f / (e0 + e1), a0 = e0 * inv, a1 = e1 * inv;
            if (i < rows) {
                if (l == 0) {
                    const size_t v = (size_t)(g0 + i);
                    anode[2 * v] = a0;
                    anode[2 * v + 1] = a1;
                    const float4 pv = make_float4(p0, p1, q0, q1);
                    *reinterpret_cast<float4*>(pq + 4 * v) = pv;
                    pq_s[i] = pv;
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const double xc = (double)(a0 * xv[u].get(j)), xo = (double)(a1 * xv[u].get(j));
                    sc1[j] += xc; sc2[j] += xc * xc; so1[j] += xo; so2[j] += xo * xo;
                }
            }
        }
        }       // row batches
    }
    // Column sums over the row groups: every lane parks its 4 x VEC partials, ONE barrier (it also publishes pq_s), then one
    // lane per (statistic, column) adds the RPB partials in group order -- as VEC rounds of "groups park, the G lanes of
    // group 0 add RPB x 4 values each" this was 8 barriers and 256 serial fp64 LDS adds on 32 lanes with 480 lanes idle
    {
        constexpr int NC = G * VEC;                      // columns covered by one row group (>= H)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int cs = l * VEC + j;
            lds[(0 * RPB + grp) * NC + cs] = cok ? sc1[j] : 0.0;
            lds[(1 * RPB + grp) * NC + cs] = cok ? sc2[j] : 0.0;
            lds[(2 * RPB + grp) * NC + cs] = cok ? so1[j] : 0.0;
            lds[(3 * RPB + grp) * NC + cs] = cok ? so2[j] : 0.0;
        }
        __syncthreads();
        for (int o = t; o < 4 * NC; o += 512) {
            const int q = o / NC, col = o % NC;
            constexpr int CH = RPB < 16 ? RPB : 16;
            double tot = 0.0;
#pragma unroll
            for (int k0 = 0; k0 < RPB; k0 += CH) {
                double v[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) v[k] = lds[(q * RPB + k0 + k) * NC + col];
#pragma unroll
                for (int k = 0; k < CH; ++k) tot += v[k];
            }
            if (col < H) {
                if (q == 0) stc_sum.add(col, tot);
                else if (q == 1) stc_sq.add(col, tot);
                else if (q == 2) sto_sum.add(col, tot);
                else sto_sq.add(col, tot);
            }
        }
    }
    if (rows <= 0 || rows > MAXR) return;
    // edge softmax (model.py:102-104) + weighted degrees.  The graph's by-source CSR rows (pointers, targets, edge ids) are
    // fetched in ONE round of loads and staged in LDS; then one lane per slot computes the two attention weights (no
    // dependent global round trips per out-edge, hubs do not serialise), and one lane per node adds its slots in order.
    const float e_b0 = be[0], e_b1 = be[1];
    __shared__ int sp_s[MAXR + 1];
    __shared__ short sd_s[GE], sr_s[GE];
    __shared__ float a0_s[GE], a1_s[GE];
    if (ne > GE || ne < 0) { if (t == 0) atomicOr(status, 8); return; }
    asm volatile("" : "+v"(pv), "+v"(pn));
#pragma unroll
    for (int u = 0; u < SU; ++u) { asm volatile("" : "+v"(nd[u]), "+v"(ed[u])); if (ne <= 0) { nd[u] = g0; ed[u] = 0; } }      // no slot of this graph exists: the clamped loads fetched no index
    if (t <= rows) sp_s[t] = pv - e0;
    if (t < rows) for (int s = pv - e0; s < pn - e0; ++s) sr_s[s] = (short)t;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = t + u * 512;
        if (s < ne) sd_s[s] = (short)min(max(nd[u] - g0, 0), rows - 1);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = t + u * 512;
        if (s < ne) {
            const float4 pvv = pq_s[sr_s[s]], qd = pq_s[sd_s[s]];
            const float l0 = fedge * (pvv.x + qd.z + e_b0), l1 = fedge * (pvv.y + qd.w + e_b1);
            const float m = fmaxf(l0, l1);
            const float x0 = expf(l0 - m), x1 = expf(l1 - m);
            const float inv = 1.f / (x0 + x1);
            const float a0 = x0 * inv, a1 = x1 * inv;
            att[ed[u]] = a0;
            att[E + ed[u]] = a1;
            a0_s[s] = a0; a1_s[s] = a1;
        }
    }
    __syncthreads();
    if (t < rows) {
        float dc = loop_w, dq = loop_w;
        const int s1 = sp_s[t + 1];
        for (int s = sp_s[t]; s < s1; s += 8) {
            float x[8], y[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int sq = min(s + q, s1 - 1); x[q] = a0_s[sq]; y[q] = a1_s[sq]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { dc += s + q < s1 ? x[q] : 0.f; dq += s + q < s1 ? y[q] : 0.f; }
        }
        dis_c[g0 + t] = dc == 0.f ? 0.f : 1.0f / sqrtf(dc);
        dis_o[g0 + t] = dq == 0.f ? 0.f : 1.0f / sqrtf(dq);
    }
}


}  // namespace cal
