// Per-graph fused backward of everything between the last backbone convolution and the two causal convolutions
// (model.py:97-113 differentiated; the transpose of k_att_fwd_graph in engine_plan.hpp): for one graph
//   d deg_k[v] = -1/2 dis_k[v]^3 (sum_{e touching v} g_k[e] w_k[e] dis_k[other end] + 2 gself_k[v] dis_k[v] loop_w)      (gcn_conv.py:63-70)
//   dl[e]      = w_c[e] w_o[e] ((g_c[e] dis_c[r] dis_c[c] + d deg_c[r]) - (g_o[e] dis_o[r] dis_o[c] + d deg_o[r]))        (edge softmax2)
//   sp[v] = sum_{row_e = v} dl[e],  sq[v] = sum_{col_e = v} dl[e]                                                          (edge projections)
//   dxc = BNbwd_c(dXc_hat), dxo = BNbwd_o(dXo_hat),  dl0 = a0 a1 (<dxc,x> - <dxo,x>)                                      (node softmax2)
//   dZ  = relu'(x) (a0 dxc + a1 dxo + dl0 (Wn0-Wn1) + sp (We0[:H]-We1[:H]) + sq (We0[H:]-We1[H:]))
// plus this graph's partial rows of the column sums behind d bias_L, d node_att_mlp, d edge_att_mlp -- what
// k_normbwd_node2 -> k_normbwd_edge -> k_att_bwd do in three node- / edge-parallel launches with the per-edge `dl`
// and per-node `ddeg` going through HBM.  Both endpoints of an edge are in the same graph, so one workgroup that
// owns the graph keeps the per-slot terms, ddeg, dl, sp and sq in LDS.
//
//   grid (nsplit B), 512 threads: nsplit = 2 workgroups per graph (same edge phase, one half of the rows each) for B <= 256,
//   else one; graphs of at most
//   64 nodes and GP_E stored edges -- the bounds of the per-graph fused backward (use_gcb in engine.hip).
//
// Edge phase: only the by-destination CSR rows of the graph are read, and everything per edge arrives in SLOT order
// (g from k_gconv_bwd<POOL> with gn_slot, the attention weights from k_gconv_fwd's w_out): one round of loads after
// the graph extents, no edge-id indirection.  Sums by SOURCE
// node come from dense [source][destination] blocks in LDS (64 x 64 floats each) that every slot adds its term to:
// row sums = by source, column sums = by destination, 8 lanes per node -- no second CSR view, no per-node slot walk.
// Row phase: as k_att_bwd, its column sums combined by wave shuffles + one LDS pass.
// (First version, kept both CSR views and per-slot lists: four dependent load rounds, 21 us; the three launches it
// replaces took 24.5 us.)
#pragma once
#include "engine_plan.hpp"

namespace cal {

struct AttBwdGraphArgs {
    AttBwdArgs a;                              // a.dl / a.gs are unused; a.dbias / a.dWn / a.dWe are per-graph partial rows
    const int* gptr; const int* eptr;          // node / CSR-slot range of every graph
    const float* att;                          // [2,E] attention weights (context, objects) in CSR-by-destination SLOT order (k_gconv_fwd's w_out)
    const float* dis;                          // [2,N] deg^-1/2 of the weighted degrees
    const float* gn; const float* gn2;         // [2,E] <dOut[col_e], z[row_e]> of column slice 0 / 1 (gn2 null: one slice), slot order (gn_slot)
    const float* gself; const float* gself2;   // [2,N] <dOut[v], z[v]>
    float loop_w;
    int64_t E; int N;
    int* status;
};

constexpr int AG_T = 64;                  // nodes per graph
constexpr int AG_LD = AG_T + 1;

template <int VEC, int G>
__global__ void __launch_bounds__(512) k_att_bwd_graph(const AttBwdGraphArgs ga, int relu, int H, int nsplit) {
    constexpr int RPB = 512 / G, UR = 2, HALF = AG_T / 2;
    warm_kernargs<sizeof(AttBwdGraphArgs) + 16>();
    const AttBwdArgs& a = ga.a;
    // dense per-graph blocks, [source r][destination c]: Tc / To = sum over the edges r -> c of g_k w_k (d deg terms),
    // D = sum of dl.  Duplicate edges accumulate through the LDS atomic (commutative for the usual <= 2 copies).
    __shared__ float Tc[AG_T * AG_LD], To[AG_T * AG_LD], Dm[AG_T * AG_LD];
    __shared__ double red[8][4][G * VEC];                                 // column sums: [wave][quantity][column]
    __shared__ double sc_lds[2][8];
    __shared__ float bnk_s[8][G * VEC];                                   // per column: mean / rstd of bnc, bno; their backward means m1, m2
    __shared__ int dp_s[AG_T + 1];
    __shared__ short d_oth[GP_E], d_own[GP_E];                            // source / destination row of every by-destination slot
    __shared__ float dis_c_s[AG_T], dis_o_s[AG_T], dd_c_s[AG_T], dd_o_s[AG_T], spv_s[AG_T], sqv_s[AG_T], gs_c_s[AG_T], gs_o_s[AG_T];
    BLK_CLK(0);
    // two workgroups per graph: both run the (cheap) edge phase, each takes one half of the rows -- the row phase streams
    // five [rows, H] tensors the previous kernel wrote on other XCDs, and 128 workgroups leave half the chip's CUs (and
    // their share of the fabric bandwidth) idle: 8.3 us until the first barrier with one workgroup per graph
    // (nsplit = 2 while 2 B workgroups still fit the chip in one wave; bigger batches keep one workgroup per graph: the
    //  duplicated edge phase would only add to a launch that already covers every CU several times)
    const int b = nsplit == 2 ? blockIdx.x >> 1 : blockIdx.x, half = nsplit == 2 ? blockIdx.x & 1 : 0, t = threadIdx.x, grp = t / G, l = t % G;
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
    const bool live = rows > 0 && rows <= AG_T && ne >= 0 && ne <= GP_E;
    if (rows > 0 && !live && t == 0) atomicOr(ga.status, 8);
    if (live) {
        // Every global load of the kernel is issued here, unconditionally (null partials point at their slice-0 twin with a
        // zero factor, indices are clamped), and pinned before the first use: with `p ? p[i] : 0` selects and the two
        // bn_mean_rstd_v calls hipcc built a branch per optional pointer and waited for each BatchNorm's statistics
        // before requesting anything else -- five serial round trips, 8.3 us until the first barrier.
        // ---- round 1 (independent of the graph): per-column constants --------------------------------------------------
        float mc[VEC], rc[VEC], gc[VEC], m1c[VEC], m2c[VEC], mo[VEC], ro[VEC], go[VEC], m1o[VEC], m2o[VEC];
        float wn[VEC], wp[VEC], wq[VEC];
        float w0[VEC], w1[VEC], w2[VEC], w3[VEC], w4[VEC], w5[VEC];
        // the eight BatchNorm sums per column (statistics and backward sums of bnc / bno): a STRIPED reader (engine.hpp) -- thread
        // t takes column t's NSTRIPE accumulator planes of each (as many registers as the VEC columns x one row it held before),
        // and the column constants reach the lanes through LDS behind the first barrier
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
        // ---- round 2 (needs g0 / e0): the by-destination CSR rows of the graph, per-node values, this group's rows -------
        const float* gself2 = ga.gself2 ? ga.gself2 : ga.gself;
        const float* gn2 = ga.gn2 ? ga.gn2 : ga.gn;
        const float* dxhc2 = a.dxhc2 ? a.dxhc2 : a.dxhc;
        const float* dxho2 = a.dxho2 ? a.dxho2 : a.dxho;
        const float f2 = ga.gn2 ? 1.f : 0.f;                  // weight of the slice-1 partials
        int pdv = a.gd.ptr[g0 + min(t, rows)], pdn = a.gd.ptr[g0 + min(t + 1, rows)];
        const int vn = g0 + min(t, rows - 1);
        float dcv = ga.dis[vn], dov = ga.dis[(size_t)N + vn];
        float gsc = ga.gself[vn], gso = ga.gself[(size_t)N + vn], gsc2 = gself2[vn], gso2 = gself2[(size_t)N + vn];
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
        const int rbeg = half * HALF, rend = nsplit == 2 ? min(rows, rbeg + HALF) : rows;      // this workgroup's rows
        load_rows(rbeg + grp);
        int dn[2];
        float dgc[2], dgo[2], dwc[2], dwo[2], dgc2[2], dgo2[2];            // per-slot values: everything is in slot order, no edge-id round
        const int64_t slot_hi = max(E - 1, (int64_t)0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t s = min((int64_t)e0 + max(min(t + u * 512, ne - 1), 0), slot_hi);
            dn[u] = a.gd.nbr[s];
            dgc[u] = ga.gn[s]; dgo[u] = ga.gn[E + s]; dgc2[u] = gn2[s]; dgo2[u] = gn2[E + s];
            dwc[u] = ga.att[s]; dwo[u] = ga.att[E + s];
        }
        // pins: nothing above may sink below this point
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            asm volatile("" : "+v"(gc[j]), "+v"(go[j]), "+v"(w0[j]), "+v"(w1[j]), "+v"(w2[j]), "+v"(w3[j]), "+v"(w4[j]), "+v"(w5[j]));
#pragma unroll
        for (int q = 0; q < 8; ++q) stripe_pin(sv[q]);
        asm volatile("" : "+v"(pdv), "+v"(pdn), "+v"(dcv), "+v"(dov), "+v"(gsc), "+v"(gso), "+v"(gsc2), "+v"(gso2));
#pragma unroll
        for (int u = 0; u < 2; ++u)
            asm volatile("" : "+v"(dn[u]), "+v"(dgc[u]), "+v"(dgo[u]), "+v"(dgc2[u]), "+v"(dgo2[u]), "+v"(dwc[u]), "+v"(dwo[u]));
#pragma unroll
        for (int u = 0; u < 2; ++u) { dgc[u] = fmaf(f2, dgc2[u], dgc[u]); dgo[u] = fmaf(f2, dgo2[u], dgo[u]); }
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
        // the dense blocks are cleared while the loads are in flight
        for (int i = t; i < AG_T * AG_LD; i += 512) { Tc[i] = 0.f; To[i] = 0.f; Dm[i] = 0.f; }
        if (t <= rows) dp_s[t] = pdv - e0;
        if (t < AG_T) {                                   // (rows past the graph: zero, the block sums below run over all 64)
            dis_c_s[t] = t < rows ? dcv : 0.f; dis_o_s[t] = t < rows ? dov : 0.f;
        }
        if (t < rows) {
            gs_c_s[t] = gsc; gs_o_s[t] = gso;
            for (int s = pdv - e0; s < pdn - e0; ++s) d_own[s] = (short)t;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = t + u * 512;
            if (s < ne) {
                const int ld = dn[u] - g0;
                if (ld < 0 || ld >= rows) atomicOr(ga.status, 16);          // an edge that leaves its graph
                d_oth[s] = (short)min(max(ld, 0), rows - 1);
            }
        }
        __syncthreads();
        BLK_CLK(2);
        // ---- d deg -------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = t + u * 512;
            if (s < ne) {
                const int r = d_oth[s], q = d_own[s];
                atomicAdd(&Tc[r * AG_LD + q], dgc[u] * dwc[u]);
                atomicAdd(&To[r * AG_LD + q], dgo[u] * dwo[u]);
            }
        }
        __syncthreads();
        {   // node v, branch k, quarter p of the other endpoints: out-edges v -> j (row v) and in-edges j -> v (column v)
            const int v = t >> 3, k = (t >> 2) & 1, p = t & 3;
            const float* T = k ? To : Tc;
            const float* dsv = k ? dis_o_s : dis_c_s;
            // (unconditional over the lane's 16 columns: entries past the graph are zero, and so is their deg^-1/2; with
            //  `if (j < rows)` inside, every iteration was a branch with its own LDS round trip)
            float acc = 0.f;
            float tv[16], tw[16], dj[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = p * 16 + jj;
                tv[jj] = T[v * AG_LD + j]; tw[jj] = T[j * AG_LD + v]; dj[jj] = dsv[j];
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) acc = fmaf(tv[jj] + tw[jj], dj[jj], acc);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (v < rows && p == 0) {                     // + the self loop; d deg = d(deg^-1/2) chain
                const float d = dsv[v], gsv = (k ? gs_o_s : gs_c_s)[v];
                (k ? dd_o_s : dd_c_s)[v] = -0.5f * d * d * d * (acc + 2.f * gsv * d * ga.loop_w);
            }
        }
        __syncthreads();
        // ---- dl per edge r -> q (an input self loop carries no gradient: k_normbwd_edge), summed by source / by destination --
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = t + u * 512;
            if (s < ne) {
                const int r = d_oth[s], q = d_own[s];
                const float xc = dgc[u] * dis_c_s[r] * dis_c_s[q] + dd_c_s[r];
                const float xo = dgo[u] * dis_o_s[r] * dis_o_s[q] + dd_o_s[r];
                if (r != q) atomicAdd(&Dm[r * AG_LD + q], a.fedge * dwc[u] * dwo[u] * (xc - xo));
            }
        }
        __syncthreads();
        {
            const int v = t >> 3, k = (t >> 2) & 1, p = t & 3;
            float acc = 0.f;
            float dv[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) { const int j = p * 16 + jj; dv[jj] = Dm[k ? j * AG_LD + v : v * AG_LD + j]; }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) acc += dv[jj];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (v < rows && p == 0) (k ? sqv_s : spv_s)[v] = acc;
        }
        __syncthreads();
        BLK_CLK(3);
        // ---- row phase (k_att_bwd) -----------------------------------------------------------------------------------
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
    // ---- this graph's partial rows (zeros for an empty / rejected graph: the rows must exist) ----------------------------
    // lanes of a wave that hold the same column (64 / G groups per wave) combine by shuffle, the 8 waves through LDS
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
        // one value per group: groups of a wave sit G lanes apart
        double s0 = sdl, s1 = ssp;
        if (G < 64) for (int off = G; off < 64; off <<= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
        if ((t & 63) == 0) { sc_lds[0][wv] = s0; sc_lds[1][wv] = s1; }
    }
    __syncthreads();
    constexpr int NW = G == 64 ? 8 : 8;      // waves that hold rows: all 8 (512 threads)
    for (int idx = t; idx < 4 * H; idx += 512) {
        const int q = idx / H, col = idx - q * H;
        double tot = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) tot += red[w8][q][col];
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
    BLK_CLK(1);
}

}  // namespace cal
