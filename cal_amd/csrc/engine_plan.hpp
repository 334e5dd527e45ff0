// Per-graph GraphPlan of the step engine: the work of plan.hip's five launches (zero, count, scan, fill,
// rank: gcn_conv.py:56-57 remove/add self loops, :92 the scatter index) plus k_gptr_dis in ONE kernel,
// when the host can vouch for the shape of the mini-batch -- what every collate knows for free:
//   * node_ptr / edge_ptr [B+1]: graph b owns nodes [node_ptr[b], node_ptr[b+1]) and the CONTIGUOUS block
//     of edge_index columns [edge_ptr[b], edge_ptr[b+1]) (PyG's Batch, cal_collate and Batch.from_data_list
//     all build batches this way),
//   * no self loops in the input (so every edge keeps its slot and the slot range of a graph is its edge
//     range: no cross-graph prefix sum), at most GP_T nodes and GP_E edges per graph.
// A workgroup builds both CSR views of its graph in LDS: degrees with LDS atomics, one wave scans them,
// edges are scattered with LDS cursors and then ranked by edge id inside their row (rows are short), which
// gives the same deterministic slot order as plan.hip.  Violations are flagged in the status word
// (bit 0: endpoint outside the graph / index range, bit 1: batch vector disagrees with node_ptr, bit 5: self
// loop, bit 3: bound exceeded); the generic path remains for everything else.
#pragma once
#include "engine_kernels.hpp"

namespace cal {

constexpr int GP_T = 128;                 // nodes per graph (the per-graph attention / conv kernels' LDS arrays)
constexpr int GP_E = 1024;                // edges per graph
constexpr int GP_T2 = 256;                // the wider instantiation of k_plan_graph (SPMotif at the reference's default
constexpr int GP_E2 = 2048;               // node_num = 15: up to ~250 nodes per graph)

// The step's FIRST-kernel duties riding in k_plan_graph (round 6; they were a launch of their own, k_zero_f64: 4.9 us of the
// 224 us headline step).  Taken by steps that END WITH k_finish in the same call (forward + backward), on the per-graph plan:
//   * every workgroup zeroes its share of the fp64 arena EXCEPT bn_feat's statistics [skip_lo, skip_hi) -- this kernel adds the
//     raw features' column sums there while other workgroups are still zeroing, so that range is zeroed by the PREVIOUS step's
//     last kernel instead (k_finish: nothing reads it there) and is clean at entry by invariant;
//   * thread 0 of workgroup 0 advances the attention-dropout counter and the Adam step counter;
//   * the workgroups behind the B planning ones rank-sort the in-step permutation draw (randperm_slice, NT / 4 elements each).
// `dirty` is the device word of that invariant: the stand-alone k_zero_f64 of a training forward raises it (the range is about to
// be used and nothing in that call cleans it), k_finish clears it.  A captured folded step replayed behind such an orphan forward
// would add onto stale sums: it flags status bit 512 instead and updates nothing (the host-side twin of the word keeps EAGER
// steps off the fold in that state, so only a graph replay can meet it).
struct PlanFold {
    double* arena; int64_t n; int skip_lo, skip_hi;
    unsigned long long* gat_tick; float* adam_step;
    int64_t* perm; int permB; unsigned long long seed; const unsigned long long* perm_ctr;
    const int* dirty;
    int on;
};

template <int GT, int GE, int NT = 256>      // NT threads: 1024 for the wide instantiation (8 edges per lane and 8 ranking passes of 256 lanes were 14 us at 240-node graphs)
__global__ void __launch_bounds__(NT) k_plan_graph(const int64_t* __restrict__ ei, int64_t E, int N, int B,
                                                    const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                                                    const int64_t* __restrict__ batch, const int64_t* __restrict__ tile_gptr,
                                                    int Bgraphs, float loop_w,
                                                    int* __restrict__ ptr_dst, int* __restrict__ nbr_dst, int* __restrict__ eid_dst,
                                                    int* __restrict__ ptr_src, int* __restrict__ nbr_src, int* __restrict__ eid_src,
                                                    int* __restrict__ row32, int* __restrict__ col32, int* __restrict__ gptr,
                                                    int* __restrict__ eptr, float* __restrict__ dis_unit, int* __restrict__ status,
                                                    const float* __restrict__ x0, int F, double* __restrict__ st_sum,
                                                    double* __restrict__ st_sq, float* __restrict__ coef_dst, float* __restrict__ coef_src,
                                                    const PlanFold pf) {
    if (pf.on) {
        if ((int)blockIdx.x >= B) {                      // permutation draw (blocks B ..)
            __shared__ unsigned long long pkey[1024];
            randperm_slice<NT>(pf.perm, pf.permB, pf.seed, pf.perm_ctr, pkey, (int)blockIdx.x - B);
            return;
        }
        const int64_t per = (pf.n + B - 1) / B, lo = (int64_t)blockIdx.x * per, hi = lo + per < pf.n ? lo + per : pf.n;
        for (int64_t i = lo + threadIdx.x; i < hi; i += NT)
            if (i < pf.skip_lo || i >= pf.skip_hi) pf.arena[i] = 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (pf.gat_tick) *pf.gat_tick += 1;
            if (pf.adam_step) pf.adam_step[0] += 1.f;
            if (*pf.dirty) atomicOr(status, 512);
        }
    }
    // coef_dst / coef_src != null: the unit-weight edge coefficient deg^-1/2 of the slot's neighbour, in slot order of either view (the
    // wide per-graph convolutions read them with their first loads instead of chasing nbr -> dis in a second round)
    __shared__ float dis_l[GT];
    // x0 != null: also the column sums / sums of squares of the raw features (bn_feat's batch statistics, model.py:90;
    // F <= 64), pre-reduced per graph in LDS and added to the zeroed accumulators with F fp64 atomics per workgroup
    __shared__ double fs[2][64];
    constexpr int EU = GE / NT;                          // edges per lane
    static_assert(GE % NT == 0 && GT <= NT && NT >= 256, "block shape");
    constexpr int SU = GT / 64;                          // scan elements per lane
    __shared__ int deg_in[GT], deg_out[GT], off_in[GT + 1], off_out[GT + 1], cur_in[GT], cur_out[GT];
    __shared__ short rl[GE], cl[GE];                     // local endpoints of the graph's edges (edge-id order)
    __shared__ short tn_d[GE], te_d[GE], tn_s[GE], te_s[GE];             // unordered row contents: neighbour, local edge id
    __shared__ int gid_s[GT];                            // packed batch: graph of every row of the tile
    BLK_CLK(0);
    const int b = blockIdx.x, t = threadIdx.x;
    // packed batch (cal_engine_set_tiles): this workgroup's unit is a TILE of the consecutive graphs [tg0, tg1); node_ptr /
    // edge_ptr are the tiles' offsets.  The batch vector must stay inside [tg0, tg1), sorted, and no edge may join two graphs
    const int64_t tg0 = tile_gptr ? tile_gptr[b] : (int64_t)b, tg1 = tile_gptr ? tile_gptr[b + 1] : (int64_t)b + 1;
    if (tile_gptr && t == 0 && ((b == 0 && tg0 != 0) || (b == B - 1 && tg1 != Bgraphs) || tg1 < tg0)) atomicOr(status, 2);
    const int g0 = (int)node_ptr[b], rows = (int)node_ptr[b + 1] - g0;
    const int64_t e0 = edge_ptr[b];
    const int m = (int)(edge_ptr[b + 1] - e0);
    if (t == 0) {
        gptr[b] = g0; eptr[b] = (int)e0;
        if (b == B - 1) { gptr[B] = g0 + rows; eptr[B] = (int)(e0 + m); ptr_dst[N] = (int)(e0 + m); ptr_src[N] = (int)(e0 + m); }
        if (b == B - 1 && (g0 + rows != N || e0 + m != E)) atomicOr(status, 2);
        if (b == 0 && (g0 != 0 || e0 != 0)) atomicOr(status, 2);
    }
    if (rows < 0 || rows > GT || m < 0 || m > GE) { if (t == 0) atomicOr(status, 8); return; }
    // edges: EU per lane, both endpoints requested before anything waits
    int64_t rv[EU], cv[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
        const int64_t e = e0 + max(min(t + u * NT, m - 1), 0);
        rv[u] = m > 0 ? ei[e] : 0;
        cv[u] = m > 0 ? ei[E + e] : 0;
    }
    const int64_t bv = rows > 0 ? batch[g0 + min(t, rows - 1)] : (int64_t)b;
    if (t < GT) { deg_in[t] = 0; deg_out[t] = 0; cur_in[t] = 0; cur_out[t] = 0; gid_s[t] = (int)(bv - tg0); }
    if (t < 128) fs[t >> 6][t & 63] = 0.0;
    __syncthreads();
    if (x0) {
        // lane t of the first (256 / F) F lanes walks the elements t, t + lanes, ..: its column (t % F) is fixed, so it sums
        // in registers (<= ~10 terms, fp32) and adds ONE pair of values to the LDS accumulators -- one fp64 LDS atomic per
        // element was 480 serialised updates per address at 240-node graphs (5 us of this kernel)
        const int lanes = (NT / F) * F, tot = rows * F;
        float s1 = 0.f, s2 = 0.f;
        if (t < lanes) {
            for (int i0 = t; i0 < tot; i0 += 4 * lanes) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = x0[(size_t)g0 * F + min(i0 + u * lanes, tot - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float vv = i0 + u * lanes < tot ? v[u] : 0.f;
                    s1 += vv; s2 = fmaf(vv, vv, s2);
                }
            }
            atomicAdd(&fs[0][t % F], (double)s1);
            atomicAdd(&fs[1][t % F], (double)s2);
        }
    }
    if (t < rows && (bv < tg0 || bv >= tg1 || (t > 0 && gid_s[t - 1] > gid_s[t]))) atomicOr(status, 2);
#pragma unroll
    for (int u = 0; u < EU; ++u) {
        const int s = t + u * NT;
        if (s < m) {
            int r = (int)(rv[u] - g0), c = (int)(cv[u] - g0);
            const bool bad = r < 0 || r >= rows || c < 0 || c >= rows;
            if (bad) { atomicOr(status, 1); r = 0; c = 0; }
            if (r == c) atomicOr(status, bad ? 1 : 32);
            if (gid_s[r] != gid_s[c]) atomicOr(status, 16);              // an edge between two graphs of the tile: not a mini-batch
            rl[s] = (short)r; cl[s] = (short)c;
            row32[e0 + s] = g0 + r; col32[e0 + s] = g0 + c;
            atomicAdd(&deg_out[r], 1);
            atomicAdd(&deg_in[c], 1);
        }
    }
    __syncthreads();
    BLK_CLK(2);
    // exclusive scans of the two degree arrays: waves 0 / 1, SU consecutive elements per lane
    if (t < 128) {
        const int* deg = t < 64 ? deg_in : deg_out;
        int* off = t < 64 ? off_in : off_out;
        const int l = t & 63;
        int a[SU], x = 0;
#pragma unroll
        for (int u = 0; u < SU; ++u) { a[u] = SU * l + u < rows ? deg[SU * l + u] : 0; x += a[u]; }
        const int own = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (l >= o) x += y;
        }
        int ex = x - own;
#pragma unroll
        for (int u = 0; u < SU; ++u) { if (SU * l + u <= rows) off[SU * l + u] = ex; ex += a[u]; }
        if (l == 63) off[rows] = x;                      // the total (rows == GT has no lane for it above)
    }
    __syncthreads();
    if (x0 && t < F) { atomicAdd(st_sum + t, fs[0][t]); atomicAdd(st_sq + t, fs[1][t]); }
    if (t < rows) {
        ptr_dst[g0 + t] = (int)e0 + off_in[t];
        ptr_src[g0 + t] = (int)e0 + off_out[t];
        const float d = (float)deg_out[t] + loop_w;
        const float dd = d == 0.f ? 0.f : 1.0f / sqrtf(d);
        dis_unit[g0 + t] = dd;
        dis_l[t] = dd;
    }
    // scatter into the rows (arbitrary order inside a row) ...
#pragma unroll
    for (int u = 0; u < EU; ++u) {
        const int s = t + u * NT;
        if (s < m) {
            const int r = rl[s], c = cl[s];
            const int p = off_in[c] + atomicAdd(&cur_in[c], 1);
            tn_d[p] = (short)r; te_d[p] = (short)s;
            const int q = off_out[r] + atomicAdd(&cur_out[r], 1);
            tn_s[q] = (short)c; te_s[q] = (short)s;
        }
    }
    __syncthreads();
    BLK_CLK(3);
    // ... then every slot moves to its rank by edge id inside its row
    for (int p = t; p < 2 * m; p += NT) {
        const bool d = p < m;
        const int q = d ? p : p - m;
        const short* te = d ? te_d : te_s;
        const short* tn = d ? tn_d : tn_s;
        const int s = te[q];
        const int v = d ? cl[s] : rl[s];
        const int* off = d ? off_in : off_out;
        const int s0 = off[v], s1 = off[v + 1];
        int rank = 0;
        for (int k = s0; k < s1; k += 8) {               // eight independent LDS reads per round (hub rows: 30+ slots)
            int x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = te[min(k + u, s1 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (k + u < s1 && x[u] < s) ? 1 : 0;
        }
        const int64_t slot = e0 + s0 + rank;
        if (d) { nbr_dst[slot] = g0 + tn[q]; eid_dst[slot] = (int)(e0 + s); if (coef_dst) coef_dst[slot] = dis_l[tn[q]]; }
        else { nbr_src[slot] = g0 + tn[q]; eid_src[slot] = (int)(e0 + s); if (coef_src) coef_src[slot] = dis_l[tn[q]]; }
    }
    BLK_CLK(1);
}

// ------------------------------------------------------------------------------------------------------------------
// The same per-graph GraphPlan for LARGE graphs (round 4; BASELINE config 5: 32 BA graphs of 5000 nodes / 20 000 edges per GPU),
// where the LDS copies of k_plan_graph (six node arrays, six edge arrays) do not fit: one 1024-thread workgroup per (graph, CSR view)
// keeps only its degree / cursor array in LDS (8192 ints), re-reads the endpoints it wrote to row32 / col32 and parks the
// unordered rows in the global scratch of plan.hip's k_plan_rank, which then moves every slot to its rank by edge id inside its
// row with 2 E threads.  Ranking inside this kernel was tried three ways and lost on the 32 CUs it occupies at config 5 (per
// view: one lane per slot with the row looked up in col32 26 us; rows held in LDS as 16-bit ids and ordered in place by a lane
// per row / a wave per hub row 17 + 93 us; by a lane per slot with a bisection of the row ends ~50 us) against 29 us for the
// separate launch over all CUs.
// Together they replace plan.hip's count -> scan -> fill (global atomics on 2 x 640 k counters: 62 + 10 + 105 us at config 5) +
// k_gptr_dis.  Same outputs, same slot order, same status bits as k_plan_graph.
// ------------------------------------------------------------------------------------------------------------------
constexpr int GPB_T = 8192;               // nodes per graph
__device__ __forceinline__ void plan_block_scan(int* __restrict__ a, int n, int* wave_tot) {
    // exclusive scan of a[0..n) in place, a[n] = total; 1024 threads x 8 consecutive elements (n <= 8192), three barriers
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, i0 = t * 8;
    int v[8], tsum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = i0 + j < n ? a[i0 + j] : 0; tsum += v[j]; }
    int x = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int wt = wave_tot[w]; woff += w < wid ? wt : 0; tot += wt; }
    int run = woff + x - tsum;
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (i0 + j < n) a[i0 + j] = run; run += v[j]; }
    if (t == 0) a[n] = tot;
    __syncthreads();
}
__global__ void __launch_bounds__(1024) k_plan_big(const int64_t* __restrict__ ei, int64_t E, int N, int B,
                                                   const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                                                   const int64_t* __restrict__ batch, float loop_w,
                                                   int* __restrict__ ptr_dst, int* __restrict__ ptr_src,
                                                   int* __restrict__ row32, int* __restrict__ col32, int* __restrict__ gptr,
                                                   int* __restrict__ eptr, float* __restrict__ dis_unit, int* __restrict__ status,
                                                   int* __restrict__ tmp, const float* __restrict__ x0, int F,
                                                   double* __restrict__ st_sum, double* __restrict__ st_sq) {
    // grid (graphs, 2 views): workgroup (b, 0) builds the by-destination rows of graph b (and row32 / col32, gptr / eptr, the
    // checks), workgroup (b, 1) the by-source rows (and deg^-1/2, bn_feat's statistics); each reads the graph's edges itself
    __shared__ int cnt[GPB_T + 1];                       // degree -> offsets -> cursors of this view
    __shared__ int wave_tot[16];
    __shared__ double part[2][1024];                     // per-lane partial column sums of the raw features
    const int b = blockIdx.x, view = blockIdx.y, t = threadIdx.x;
    const int g0 = (int)node_ptr[b], rows = (int)node_ptr[b + 1] - g0;
    const int64_t e0 = edge_ptr[b];
    const int m = (int)(edge_ptr[b + 1] - e0);
    if (t == 0 && view == 0) {
        gptr[b] = g0; eptr[b] = (int)e0;
        if (b == B - 1) { gptr[B] = g0 + rows; eptr[B] = (int)(e0 + m); ptr_dst[N] = (int)(e0 + m); ptr_src[N] = (int)(e0 + m); }
        if (b == B - 1 && (g0 + rows != N || e0 + m != E)) atomicOr(status, 2);
        if (b == 0 && (g0 != 0 || e0 != 0)) atomicOr(status, 2);
    }
    if (rows < 0 || rows > GPB_T || m < 0) { if (t == 0) atomicOr(status, 8); return; }
    int* tn = tmp + (view == 0 ? 0 : 2 * E);             // scratch of k_plan_rank: {tn_d, te_d, tn_s, te_s}
    int* te = tn + E;
    int* ptr_o = view == 0 ? ptr_dst : ptr_src;
    for (int v = t; v <= rows; v += 1024) cnt[v] = 0;
    if (view == 0) for (int v = t; v < rows; v += 1024) if (batch[g0 + v] != (int64_t)b) atomicOr(status, 2);
    __syncthreads();
    // degrees of this view (LDS atomics); view 0 also leaves the 32-bit endpoints
    for (int s0 = t; s0 < m; s0 += 8 * 1024) {
        int64_t rv[8], cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t e = e0 + min(s0 + u * 1024, m - 1); rv[u] = ei[e]; cv[u] = ei[E + e]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u * 1024;
            if (s < m) {
                int r = (int)(rv[u] - g0), c = (int)(cv[u] - g0);
                const bool bad = r < 0 || r >= rows || c < 0 || c >= rows;
                if (bad) { atomicOr(status, 1); r = 0; c = 0; }
                if (r == c) atomicOr(status, bad ? 1 : 32);
                if (view == 0) { row32[e0 + s] = g0 + r; col32[e0 + s] = g0 + c; }
                atomicAdd(&cnt[view == 0 ? c : r], 1);
            }
        }
    }
    if (x0 && view == 1) {
        // bn_feat's batch statistics (model.py:90; F <= 64): a lane's column is fixed (lanes a multiple of F), fp32 inside a
        // round of eight, fp64 across rounds; the lanes park their pair in LDS and lane f < F adds the lanes of column f in
        // order (fp64 LDS atomics from 1020 lanes onto 2 F addresses are ~100 serialised updates per address)
        const int lanes = (1024 / F) * F, tot = rows * F;
        double s1 = 0.0, s2 = 0.0;
        if (t < lanes) {
            for (int i0 = t; i0 < tot; i0 += 8 * lanes) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x0[(size_t)g0 * F + min(i0 + u * lanes, tot - 1)];
                float p1 = 0.f, p2 = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float vv = i0 + u * lanes < tot ? v[u] : 0.f; p1 += vv; p2 = fmaf(vv, vv, p2); }
                s1 += (double)p1; s2 += (double)p2;
            }
        }
        part[0][t] = s1; part[1][t] = s2;
    }
    __syncthreads();
    if (x0 && view == 1 && t < 2 * F) {
        const int f = t % F, which = t / F, lanes = (1024 / F) * F;
        double tot = 0.0;
        for (int k0 = f; k0 < lanes; k0 += 8 * F) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + u * F < lanes ? part[which][k0 + u * F] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += v[u];
        }
        atomicAdd((which ? st_sq : st_sum) + f, tot);
    }
    if (view == 1)
        for (int v = t; v < rows; v += 1024) {
            const float d = (float)cnt[v] + loop_w;
            dis_unit[g0 + v] = d == 0.f ? 0.f : 1.0f / sqrtf(d);
        }
    __syncthreads();
    plan_block_scan(cnt, rows, wave_tot);
    for (int v = t; v < rows; v += 1024) ptr_o[g0 + v] = (int)e0 + cnt[v];
    __syncthreads();
    // scatter into the rows (arbitrary order inside a row): the offsets become cursors; the unordered rows go to plan.hip's scratch
    // and k_plan_rank (2 E threads) moves every slot to its rank by edge id inside its row
    for (int s0 = t; s0 < m; s0 += 8 * 1024) {
        int64_t rv[8], cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t e = e0 + min(s0 + u * 1024, m - 1); rv[u] = ei[e]; cv[u] = ei[E + e]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = s0 + u * 1024;
            if (sl < m) {
                int r = (int)(rv[u] - g0), c = (int)(cv[u] - g0);
                if (r < 0 || r >= rows || c < 0 || c >= rows) { r = 0; c = 0; }
                const int p = atomicAdd(&cnt[view == 0 ? c : r], 1);
                tn[e0 + p] = g0 + (view == 0 ? r : c); te[e0 + p] = (int)(e0 + sl);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Node attention + edge attention + weighted degrees of one graph in one kernel (k_node_att_fwd's fast path followed
// by k_edge_att_deg, model.py:97-111, gcn_conv.py:63-68): an edge's two endpoints are in the same graph, so the edge
// softmax reads the projections P[row] / Q[col] from LDS right after they are computed.
//   grid (B), 512 threads = 512/G row groups of G lanes x VEC columns; at most 4 rows per group (64 nodes at H = 128).
//   Column statistics of a0 x / a1 x: one partial row per graph (Acc.parts indexed by blockIdx.x).
// ------------------------------------------------------------------------------------------------------------------
template <int VEC, int G>
__global__ void __launch_bounds__(512) k_att_fwd_graph(const int* __restrict__ gptr, const CSR gs, const float* __restrict__ x,
                                                       const float* __restrict__ Wn, const float* __restrict__ bn,
                                                       const float* __restrict__ We, const float* __restrict__ be,
                                                       float* __restrict__ anode, float* __restrict__ pq, float* __restrict__ att,
                                                       float* __restrict__ dis_c, float* __restrict__ dis_o, const Acc stc_sum,
                                                       const Acc stc_sq, const Acc sto_sum, const Acc sto_sq, float loop_w, int H,
                                                       int64_t E, int* __restrict__ status, float fnode, float fedge,
                                                       const int* __restrict__ eptr) {
    // fnode / fedge: 1, or 0 for without_node_attention / without_edge_attention (equal logits -> constant 0.5 masks)
    constexpr int RPB = 512 / G, MAXR = 4 * RPB;
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
    int nd[2], ed[2];
    const int slot_hi = max(gs.nnz - 1, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
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
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = V::ld(x + (size_t)(g0 + min(grp + u * RPB, rows - 1)) * H + cc);
        const float b0 = bn[0], b1 = bn[1];
#pragma unroll
        for (int u = 0; u < 6; ++u) w[u].pin();
#pragma unroll
        for (int u = 0; u < 4; ++u) { xv[u].pin(); if (!cok) xv[u] = V::zero(); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = grp + u * RPB;
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
    __shared__ short sd_s[GP_E], sr_s[GP_E];
    __shared__ float a0_s[GP_E], a1_s[GP_E];
    if (ne > GP_E || ne < 0) { if (t == 0) atomicOr(status, 8); return; }
    asm volatile("" : "+v"(pv), "+v"(pn), "+v"(nd[0]), "+v"(nd[1]), "+v"(ed[0]), "+v"(ed[1]));
    if (ne <= 0) { nd[0] = nd[1] = g0; ed[0] = ed[1] = 0; }      // no slot of this graph exists: the clamped loads fetched no index
    if (t <= rows) sp_s[t] = pv - e0;
    if (t < rows) for (int s = pv - e0; s < pn - e0; ++s) sr_s[s] = (short)t;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int s = t + u * 512;
        if (s < ne) sd_s[s] = (short)min(max(nd[u] - g0, 0), rows - 1);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
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
