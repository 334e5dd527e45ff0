// Per-graph fused GCN convolution for WIDE graphs (65 .. 256 nodes, <= 2048 stored edges): the reference's DEFAULT SPMotif shape
// (opts.py:18 node_num = 15 -> 225-247-node graphs, utils.py:62-63) and BASELINE.json configs[0].
//     forward   out = relu(A_hat (BN(rs * x) @ W) + b)                    (gcn_conv.py:72-104 behind model.py:93-95, 112-113)
//     backward  dz = A_hat^T dOut,  dX' = dz W^T (+ BatchNorm-backward sums),  dW = x'^T dz
// The per-graph kernels of engine_gconv.hpp / engine_gconv_bwd.hpp aggregate through a DENSE adjacency block on the matrix
// cores: at 256 nodes that block is 256 KB and the product 50 x the useful flops.  Here a workgroup owns (graph, 64- or
// 32-column slice) and keeps only the slice of z = x' W (resp. of dOut / dz) for ALL the graph's nodes in LDS (256 x 64 x 4 B =
// 64 KB); the dense products run on v_mfma_f32_32x32x2_f32 with the node operand read STRAIGHT FROM GLOBAL MEMORY into MFMA
// registers (a 256 x 128 x' stage would be 128 KB), and the aggregation is SPARSE from LDS: one wave per row, one lane per
// column, the row's CSR slots preloaded lane-parallel and handed to the gathers by v_readlane (no dependent LDS chain per
// slot).  One workgroup still owns (graph, slice): no exchange between workgroups.
//
//   forward   grid (B, H / NC, branches), 512 threads, NC = 64 (124 KB of LDS) or 32 (75 KB: two workgroups per CU -- taken
//             when a launch would otherwise leave CUs idle)
//   backward  grid (B, (H / 64) * nsplit, branches), 512 threads; nsplit = 2: the dX' product and the dW product of a
//             (graph, slice) -- both only need dz -- go to two workgroups that each redo the (cheap) sparse dz
#pragma once
#include "engine_gconv_bwd.hpp"

namespace cal {

constexpr int GW_T = 256;                 // nodes per graph
constexpr int GW_E = 2048;                // stored edges per graph
constexpr int GW_NT = 512;                // threads per workgroup (8 waves: one 32-row tile of the node operand each)
constexpr int GW_K = 128;                 // reduction width (= hidden)

// ---- sparse rows from LDS -------------------------------------------------------------------------------------------------
// One wave works on FOUR rows at a time: the 16 lanes of a DPP row own one graph row and VW = 4 (2) consecutive columns each
// (ds_read_b128 / _b64 gathers).  The CSR slots of the row are preloaded lane-parallel (lane c of the group holds slot c: source
// node + coefficient) and handed to the group's 16 lanes by v_mov_b32_dpp row_newbcast:c -- a register move, no LDS round trip and
// no scalar traffic per slot; a row of more than 16 slots reloads the chunk.  (First version, round 6: one row per wave, one lane
// per column, slots by v_readlane: ~1000 cycles per row -- every 4-slot batch, the chunk preload and the self-loop term were
// LDS round trips of their own -- 12-17 us of a 29 us kernel at 240-node graphs; profiles/r6/micro_gw_phases.txt.)
template <int S> __device__ __forceinline__ int gw_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + S, 0xF, 0xF, true); }      // row_newbcast:S
template <int S> __device__ __forceinline__ float gw_bcast(float v) { return __int_as_float(gw_bcast<S>(__float_as_int(v))); }
template <int VW> __device__ __forceinline__ void gw_ld(const float* p, float (&v)[VW]) {
    if constexpr (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
}
template <int VW> __device__ __forceinline__ void gw_st(float* p, const float (&v)[VW]) {
    if constexpr (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}
// extents of the row a lane group owns in iteration `it` (wave w, group grp): row (it * 8 + w) * 4 + grp
struct GwRow { int r, p0, cnt; float di; };
__device__ __forceinline__ GwRow gw_row(const int* ptr_s, const float* dis_s, int rows, int it, int w, int grp) {
    GwRow o;
    o.r = ((it * 8 + w) << 2) + grp;
    const int rc = min(o.r, rows - 1);
    o.p0 = ptr_s[rc];
    const int p1 = ptr_s[rc + 1];
    o.cnt = o.r < rows ? p1 - o.p0 : 0;
    o.di = dis_s[rc];
    return o;
}
// raw chunk load: slot p0 + base + lc of the group's row (clamped: the select against cnt happens at the use, an iteration later)
struct GwRaw { int j; float c; };
__device__ __forceinline__ GwRaw gw_raw(const unsigned char* en, const float* ec, const GwRow& rw, int base, int lc) {
    GwRaw v;
    const int s = rw.p0 + min(base + lc, max(rw.cnt - 1, 0));
    v.j = en[s]; v.c = ec[s];
    return v;
}
// four gather steps SB * 4 .. + 3 of a chunk: a[k] += c_s * src[j_s][col + k]
template <int SB, int VW, int LD>
__device__ __forceinline__ void gw_steps4(const float* srcl, int cj, float cc, float (&a)[VW]) {
    float z[4][VW], cs[4];
    { const int j = gw_bcast<SB * 4 + 0>(cj); cs[0] = gw_bcast<SB * 4 + 0>(cc); gw_ld<VW>(srcl + j * LD, z[0]); }
    { const int j = gw_bcast<SB * 4 + 1>(cj); cs[1] = gw_bcast<SB * 4 + 1>(cc); gw_ld<VW>(srcl + j * LD, z[1]); }
    { const int j = gw_bcast<SB * 4 + 2>(cj); cs[2] = gw_bcast<SB * 4 + 2>(cc); gw_ld<VW>(srcl + j * LD, z[2]); }
    { const int j = gw_bcast<SB * 4 + 3>(cj); cs[3] = gw_bcast<SB * 4 + 3>(cc); gw_ld<VW>(srcl + j * LD, z[3]); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < VW; ++k) a[k] = fmaf(cs[u], z[u][k], a[k]);
}
// All rows of a graph: emit(row, di, a) receives a[k] = sum_s c_s src[j_s][col + k] + di * loop_w * src[row][col + k] for the lane's
// VW columns col = VW * lc (called for the rows < rows only, every lane of the owning group).
template <int VW, int LD, class Emit>
__device__ __forceinline__ void gw_sparse_rows(const float* __restrict__ src, const int* ptr_s, const float* dis_s, const unsigned char* en,
                                               const float* ec, int rows, int w, int lane, float loop_w, Emit emit) {
    const int grp = lane >> 4, lc = lane & 15;
    const float* srcl = src + VW * lc;
    const int nit = (((rows + 3) >> 2) - w + 7) >> 3;    // row quads w, w + 8, ..
    if (nit <= 0) return;
    GwRow cur = gw_row(ptr_s, dis_s, rows, 0, w, grp);
    GwRow nx1 = gw_row(ptr_s, dis_s, rows, 1, w, grp);   // (rows past the graph: clamped reads, cnt 0)
    GwRaw craw = gw_raw(en, ec, cur, 0, lc);
    for (int it = 0; it < nit; ++it) {
        // requests for the coming iterations first: the next row's chunk (its extents arrived an iteration ago), the extents after it
        const GwRaw nraw = gw_raw(en, ec, nx1, 0, lc);
        const GwRow nx2 = gw_row(ptr_s, dis_s, rows, it + 2, w, grp);
        float zs[VW];
        gw_ld<VW>(srcl + min(cur.r, rows - 1) * LD, zs);
        float a[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) a[k] = 0.f;
        const int cm = max(max(__builtin_amdgcn_readlane(cur.cnt, 0), __builtin_amdgcn_readlane(cur.cnt, 16)),
                           max(__builtin_amdgcn_readlane(cur.cnt, 32), __builtin_amdgcn_readlane(cur.cnt, 48)));
        const int rc = min(cur.r, rows - 1);
        for (int base = 0; base < cm; base += 16) {
            GwRaw ch = craw;
            if (base > 0) ch = gw_raw(en, ec, cur, base, lc);
            const bool ok = base + lc < cur.cnt;
            const int cj = ok ? ch.j : rc;
            const float cc = ok ? ch.c : 0.f;
            const int m = cm - base;
            gw_steps4<0, VW, LD>(srcl, cj, cc, a);
            if (m > 4) gw_steps4<1, VW, LD>(srcl, cj, cc, a);
            if (m > 8) gw_steps4<2, VW, LD>(srcl, cj, cc, a);
            if (m > 12) gw_steps4<3, VW, LD>(srcl, cj, cc, a);
        }
        const float sl = cur.di * loop_w;
#pragma unroll
        for (int k = 0; k < VW; ++k) a[k] = fmaf(sl, zs[k], a[k]);
        if (cur.r < rows) emit(cur.r, cur.di, a);
        cur = nx1; nx1 = nx2; craw = nraw;
    }
}

template <bool RS, int NC>
__global__ void __launch_bounds__(GW_NT, 2) k_gw_fwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                                      const GconvBranch2 bb, int relu, float loop_w, int H, int K,
                                                                      int* __restrict__ status) {
    static_assert(NC == 64 || NC == 32, "column slice of 64 or 32");
    constexpr int CT = NC / 32;                          // 32-column MFMA tiles per slice
    constexpr int LDZ = NC + 4, LDB = NC + 4;
    constexpr int WU = GW_K * (NC / 4) / GW_NT;          // W float4s per lane
    constexpr int SU = GW_E / GW_NT;                     // CSR slots per lane
    __shared__ __attribute__((aligned(16))) float Zs[GW_T * LDZ];          // z slice, all rows: Zs[row][col]
    __shared__ __attribute__((aligned(16))) float Ws[GW_K * LDB];          // W slice [k][col]; after the product: reduction scratch
    __shared__ __attribute__((aligned(16))) float sc_s[GW_K], sh_s[GW_K];
    __shared__ int ptr_s[GW_T + 4];
    __shared__ float dis_s[GW_T];
    __shared__ unsigned char en[GW_E];                   // source node (local) of every CSR-by-destination slot
    __shared__ float ec[GW_E];                           // its coefficient dis_j * w_e
    BLK_CLK(0);
    warm_kernargs<sizeof(CSR) + 2 * sizeof(void*) + sizeof(GconvBranch2) + 32>();
    const GconvBranch& br = bb.b[blockIdx.z];
    const int b = blockIdx.x, n0 = blockIdx.y * NC, t = threadIdx.x;
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    // the W slice does not depend on the graph: requested before the graph's extents (a scalar round trip) are known
    float4 vb[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int idx = t + u * GW_NT, k = min(idx / (NC / 4), K - 1), j4 = idx % (NC / 4);
        vb[u] = *reinterpret_cast<const float4*>(br.W + (size_t)k * H + n0 + 4 * j4);
    }
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const bool want = br.st_sum.on();
    if (rows <= 0) {                                     // empty graph: its partial rows still have to exist
        if (br.bn.update && blockIdx.x == 0 && blockIdx.y == 0 && t < K) { const BNRaw r0 = bn_raw_load_st(br.bn, t); bn_raw_update_running(br.bn, r0, t); }
        if (t < NC) {
            if (want) { br.st_sum.add(n0 + t, 0.0); br.st_sq.add(n0 + t, 0.0); }
            if (br.pooled) br.pooled[(size_t)b * H + n0 + t] = 0.f;
        }
        return;
    }
    if (rows > GW_T || ne > GW_E || ne < 0) {            // the host's bounds were wrong: flag it, write nothing
        if (t == 0) atomicOr(status, 8);
        return;
    }
    const bool hasw = br.ew != nullptr;
    const int nk8 = K >> 3;
    // ---- every global load of the kernel, issued before the first wait -------------------------------------------------
    // node operand: wave w owns rows w * 32 .. + 31; lane (li, lk) takes the four consecutive k of every eight of ITS row
    // (any bijection of k onto (MFMA step, lk) is a valid reduction order when both operands share it: gconv_mma_arow)
    const int arow = min(w * 32 + li, rows - 1);
    float4 xa[GW_K / 8];
    {
        const float* xp = br.x + (size_t)(g0 + arow) * K + 4 * lk;
#pragma unroll
        for (int i = 0; i < GW_K / 8; ++i) xa[i] = *reinterpret_cast<const float4*>(xp + 8 * min(i, nk8 - 1));
    }
    float rsv = 1.f;
    if (RS) rsv = br.rs[(size_t)(g0 + arow) * br.rs_stride];
    const int pv = g.ptr[g0 + min(t, rows)];
    const float dv = br.dis[g0 + min(t, rows - 1)];
    int nv[SU], ev[SU];
    const int slot_hi = max(g.nnz - 1, 0);
    const float* coefp = br.coef_in ? br.coef_in : br.dis;
    const int coef_hi = br.coef_in ? slot_hi : 0;
    float cin[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = min(e0 + max(min(t + u * GW_NT, ne - 1), 0), slot_hi);
        nv[u] = g.nbr[s];
        ev[u] = g.eid[s];
        cin[u] = coefp[min(s, coef_hi)];
    }
    const float* biasp = br.bias ? br.bias : br.W;       // W: any valid [>= H] float array; the value is masked below
    constexpr int VW = NC / 16;                          // columns per lane in the sparse phase (16 lanes per row)
    const int lc = lane & 15;
    float bias4[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) bias4[k] = biasp[n0 + VW * lc + k];
    BNRawS braws = bn_raws_load(br.bn, min(t, K - 1));   // (striped reader: the producer may be a per-graph kernel)
#pragma unroll
    for (int i = 0; i < GW_K / 8; ++i) ro_pin(xa[i]);
#pragma unroll
    for (int u = 0; u < WU; ++u) ro_pin(vb[u]);
    bn_raws_pin(braws);
#pragma unroll
    for (int u = 0; u < SU; ++u) asm volatile("" : "+v"(nv[u]), "+v"(ev[u]), "+v"(cin[u]));
    asm volatile("" : "+v"(rsv));
#pragma unroll
    for (int k = 0; k < VW; ++k) { asm volatile("" : "+v"(bias4[k])); if (!br.bias) bias4[k] = 0.f; }
    const BNRaw braw = bn_raws_sum(br.bn, braws);
    if (ne <= 0) {                                       // no slot of this graph exists: what the clamped loads fetched is not an index
#pragma unroll
        for (int u = 0; u < SU; ++u) { nv[u] = g0; ev[u] = 0; }
    }
    if (t < K) {
        bn_raw_scale_shift(br.bn, braw, sc_s[t], sh_s[t]);
        if (br.bn.update && blockIdx.x == 0 && blockIdx.y == 0) bn_raw_update_running(br.bn, braw, t);
    }
    // second round: edge coefficients dis_j * w_e (needs the neighbour / edge ids)
    float cv[SU];
    if (br.coef_in) {
#pragma unroll
        for (int u = 0; u < SU; ++u) cv[u] = cin[u];
    } else {
        const float* ewp = hasw ? br.ew : br.dis;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const float c = br.dis[nv[u]];
            const float wl = ewp[hasw ? ev[u] : 0];
            cv[u] = hasw ? c * wl : c;
        }
    }
    // ---- stage in LDS ------------------------------------------------------------------------------------------------------
    if (t <= rows) ptr_s[t] = pv - e0;
    if (t < rows) dis_s[t] = dv;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = t + u * GW_NT;
        if (s < ne) {
            const int loc = nv[u] - g0;
            const bool inb = loc >= 0 && loc < rows;        // an edge that leaves its graph is not a mini-batch: flag it
            en[s] = (unsigned char)(inb ? loc : 0); ec[s] = inb ? cv[u] : 0.f;
            if (br.coef_out && blockIdx.y == 0) br.coef_out[e0 + s] = cv[u];
            if (!inb) atomicOr(status, 16);
        }
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int idx = t + u * GW_NT, k = idx / (NC / 4), j4 = idx % (NC / 4);
        if (k < K) *reinterpret_cast<float4*>(Ws + k * LDB + 4 * j4) = vb[u];
    }
    __syncthreads();
    BLK_CLK(2);
    // ---- z tile = BN(rs x) W on the matrix cores: wave w owns row tile w, all CT column tiles -----------------------------------
    gc_f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    if (w * 32 < rows) {
        const float* bp = Ws + 4 * lk * LDB + li;
        float bv[2][4 * CT];
        auto read_b = [&](int i, int s) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < CT; ++c) bv[s][j * CT + c] = bp[(8 * i + j) * LDB + 32 * c];
        };
        read_b(0, 0);
#pragma unroll
        for (int i = 0; i < GW_K / 8; ++i) {
            if (i < nk8) {
                const int s = i & 1;
                if (i + 1 < nk8) read_b(i + 1, s ^ 1);
                const float4 sc4 = *reinterpret_cast<const float4*>(sc_s + 8 * i + 4 * lk);
                const float4 sh4 = *reinterpret_cast<const float4*>(sh_s + 8 * i + 4 * lk);
                const float a[4] = {fmaf(xa[i].x * rsv, sc4.x, sh4.x), fmaf(xa[i].y * rsv, sc4.y, sh4.y),
                                    fmaf(xa[i].z * rsv, sc4.z, sh4.z), fmaf(xa[i].w * rsv, sc4.w, sh4.w)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[s][j * CT + c], acc[c], 0, 0, 0);
            }
        }
        // z rows -> LDS (row-major: the gathers below read one row by consecutive lanes), and to HBM for the weighted convs' backward
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                Zs[row * LDZ + c * 32 + li] = acc[c][r];
            }
            if (br.z) gc_store_tile(acc[c], br.z + (size_t)(g0 + w * 32) * H + n0 + c * 32, H, rows - w * 32, li, lk);
        }
    }
    __syncthreads();
    BLK_CLK(3);
    // ---- out rows = A_hat z, sparse from LDS (gw_sparse_rows: four rows per wave at a time, 16 lanes x VW columns per row) ----------
    float f1[VW], f2[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { f1[k] = 0.f; f2[k] = 0.f; }
    gw_sparse_rows<VW, LDZ>(Zs, ptr_s, dis_s, en, ec, rows, w, lane, loop_w, [&](int row, float di, float (&a)[VW]) {
        float v[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            v[k] = fmaf(di, a[k], bias4[k]);
            if (relu) v[k] = fmaxf(v[k], 0.f);
            f1[k] += v[k]; f2[k] = fmaf(v[k], v[k], f2[k]);
        }
        gw_st<VW>(br.out + (size_t)(g0 + row) * H + n0 + VW * lc, v);
    });
    // ---- column sums of this graph: across the groups / waves in fp64 through LDS (over the W stage: every wave is past the product) ----
    double* red = reinterpret_cast<double*>(Ws);         // [32 row groups][2][NC]
    {
        const int rg = w * 4 + (lane >> 4);
#pragma unroll
        for (int k = 0; k < VW; ++k) { red[(rg * 2 + 0) * NC + VW * lc + k] = (double)f1[k]; red[(rg * 2 + 1) * NC + VW * lc + k] = (double)f2[k]; }
    }
    __syncthreads();
    if (t < NC) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) { s1 += red[(k * 2 + 0) * NC + t]; s2 += red[(k * 2 + 1) * NC + t]; }
        if (want) { br.st_sum.add(n0 + t, s1); br.st_sq.add(n0 + t, s2); }
        if (br.pooled) br.pooled[(size_t)b * H + n0 + t] = (float)s1;
    }
    BLK_CLK(1);
}

// ----------------------------------------------------------------------------------------------------------------------
// Backward.  grid (B, (H / 64) * nsplit, branches), 512 threads.  A workgroup owns one graph and the 64-column slice `ns`
// of the OUTPUT features, like k_gconv_bwd: dz[:, ns] = A_hat^T dOut[:, ns] (sparse from LDS, one wave per row, CSR BY
// SOURCE), then
//     P2  partial dX'[:, :] = dz[:, ns] W[:, ns]^T   (+ the BatchNorm-backward column sums  sum dX', sum dX' * x_hat)
//     P3  dW[:, ns] (this graph's slab) = x'^T dz[:, ns]
// on the matrix cores.  Operand sources: dz rows from LDS (16-byte reads, k-contiguous), the W rows of P2 in registers for the
// whole kernel (one lane = one row of W[:, ns], requested with the first loads), x' of P3 and x_hat of P2's sums straight from
// global memory (4-byte loads, 128 B per half-wave; the BatchNorm transform is a per-lane constant: a lane's k is fixed).
// nsplit = 2: workgroup 2 sl does P1 + P2, workgroup 2 sl + 1 does P1 + P3 (launches that would leave CUs idle).
//   MODE 0: dOut given;  1: UP -- dOut = BatchNorm-backward (+ ReLU mask) of the layer above, from its partial dX' (see
//   GconvBwdBranch);  2: POOL -- dOut[v] = relu'(y[v]) * g_b is never built: LDS holds one 64-bit ReLU mask per row and
//   Zg = z * g_b, the aggregation adds coefficients under the mask, and gn[e] = <dOut[col_e], z[row_e]> / gself are masked sums
//   of a Zg row (the mask of the destination row is the v_cndmask condition).
// ----------------------------------------------------------------------------------------------------------------------
constexpr int GW_LDD = GC_N + 4;

// P3 inner product for NCT column tiles: acc[c] += sum_i x'[i][k] dz[i][c * 32 + li], i < rowsP (steps of two rows: lk).  Batches of
// 16 steps; the operands of batch b + 1 -- x from global memory, dz (and the row scales) from LDS -- are requested before the
// MFMAs of batch b.
template <int NCT, bool RS>
__device__ __forceinline__ void gw_p3(const float* __restrict__ xk, int K, const float* rs_s, const float* dzp, int rowsP, int rows, int lk,
                                      float sc, float sh, gc_f32x16 (&acc)[2], int sb0, int sb1) {
    constexpr int BS = 16;
    float xg[2][BS], bz[2][BS][NCT], rsv[2][RS ? BS : 1];
    auto ld = [&](int sb, float (&x)[BS], float (&bzz)[BS][NCT], float (&r)[RS ? BS : 1]) {
#pragma unroll
        for (int u = 0; u < BS; ++u) x[u] = xk[(size_t)min(2 * (sb * BS + u) + lk, rows - 1) * K];
#pragma unroll
        for (int u = 0; u < BS; ++u) {
            const int i = 2 * (sb * BS + u) + lk;
            if (RS) r[u] = rs_s[i];
#pragma unroll
            for (int c = 0; c < NCT; ++c) bzz[u][c] = dzp[i * GW_LDD + 32 * c];
        }
    };
    auto mul = [&](float (&x)[BS], float (&bzz)[BS][NCT], float (&r)[RS ? BS : 1]) {
#pragma unroll
        for (int u = 0; u < BS; ++u) {
            const float av = RS ? fmaf(x[u] * r[u], sc, sh) : fmaf(x[u], sc, sh);
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bzz[u][c], acc[c], 0, 0, 0);
        }
    };
    const int nsb = sb1;                                 // batches [sb0, sb1) of the graph's rowsP / 32
    if (sb0 >= sb1) return;
    ld(sb0, xg[0], bz[0], rsv[0]);
    for (int sb = sb0; sb < nsb; sb += 2) {
        if (sb + 1 < nsb) ld(sb + 1, xg[1], bz[1], rsv[1]);
        __builtin_amdgcn_sched_barrier(0);
        mul(xg[0], bz[0], rsv[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (sb + 1 < nsb) {
            if (sb + 2 < nsb) ld(sb + 2, xg[0], bz[0], rsv[0]);
            __builtin_amdgcn_sched_barrier(0);
            mul(xg[1], bz[1], rsv[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <bool RS, int MODE>
__global__ void __launch_bounds__(GW_NT, 2) k_gw_bwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                     const GconvBwdBranch2 bb, float loop_w, int N, int H, int K, int nsplit,
                                                     int* __restrict__ status) {
    constexpr bool UP = MODE == 1, POOL = MODE == 2;
    constexpr int SU = GW_E / GW_NT;                     // CSR slots per lane
    __shared__ __attribute__((aligned(16))) float Ds[GW_T * GW_LDD];       // dOut slice rows [j][n] (POOL: Zg = z * g_b); after P1: fp64 scratch
    __shared__ __attribute__((aligned(16))) float Dz[GW_T * GW_LDD];       // dz slice rows [i][n]
    __shared__ unsigned mask_s[POOL ? GW_T * 2 : 2];     // POOL: ReLU mask of every row, this slice's 64 columns
    __shared__ unsigned char en[GW_E];                   // destination node (local) of every CSR-by-source slot
    __shared__ float ec[GW_E];                           // its coefficient w_e * dis_dst
    __shared__ unsigned short ee[POOL ? GW_E : 2];       // POOL: local edge id of the slot
    __shared__ int ptr_s[GW_T + 4];
    __shared__ float dis_s[GW_T], rs_s[GW_T];
    __shared__ float mean_s[GW_K], rstd_s[GW_K], gam_s[GW_K], bet_s[GW_K];
    __shared__ float um_s[UP ? GC_N : 1], ur_s[UP ? GC_N : 1], ug_s[UP ? GC_N : 1], u1_s[UP ? GC_N : 1], u2_s[UP ? GC_N : 1];
    __shared__ float gv_s[POOL ? GC_N : 1];
    __shared__ float bs_s[GW_NT / 64][16][4];
    BLK_CLK(0);
    warm_kernargs<sizeof(CSR) + 2 * sizeof(void*) + sizeof(GconvBwdBranch2) + 32>();
    const GconvBwdBranch& br = bb.b[blockIdx.z];
    const int b = blockIdx.x, sl = blockIdx.y / nsplit, part = blockIdx.y - sl * nsplit, ns0 = sl * GC_N, t = threadIdx.x;
    // nsplit 1: one workgroup does P2 and P3; 2: part 0 = P2, part 1 = P3; 4: parts 0 / 1 = P2 on row tiles 0-3 / 4-7, parts 2 / 3 = P3 on
    // column tile 0 / 1 (the node range halved over its waves)
    const int np2 = nsplit == 4 ? 2 : 1;
    const bool do_p2 = nsplit == 1 || part < np2, do_p3 = nsplit == 1 || part >= np2, first = part == 0;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const int pb = (POOL && br.iperm) ? br.iperm[b] : b;
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* parts = br.dot_parts + (((size_t)sl * np2 + (do_p2 ? part : 0)) * gridDim.x + b) * (2 * K);
    float* slab = br.slab + (size_t)b * K * H;
    if (rows <= 0 || rows > GW_T || ne > GW_E || ne < 0) {
        // empty graph (or a violated bound, flagged): its partial row and its slab slice must still exist
        if (rows > 0 && t == 0) atomicOr(status, 8);
        if (do_p2 && !br.dacc_sum) for (int i = t; i < 2 * K; i += GW_NT) parts[i] = 0.0;
        if ((UP || POOL) && first && t < GC_N) br.bias_parts[(size_t)b * H + ns0 + t] = 0.0;
        if (do_p3) for (int i = t; i < K * GC_N; i += GW_NT) slab[(size_t)(i / GC_N) * H + ns0 + i % GC_N] = 0.f;
        return;
    }
    const bool hasw = br.ew != nullptr;
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5;
    // which product this wave runs, and on which tiles
    const bool p2w = do_p2 && (nsplit >= 2 || w < 4), p3w = do_p3 && (nsplit >= 2 || w >= 4);
    const int wk = w & 3;                                // 32-wide tile of the K input columns
    const int kq = min(wk * 32 + li, K - 1);             // this lane's input column in P2's epilogue / P3
    // ---- every global load of the kernel, issued before the first wait ----------------------------------------------------
    RoBatch<float4, 8> bd, bd1, by;
    float gv = 0.f, gv1 = 0.f;
    if (POOL) {
        ro_issue<GW_NT>(by, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(br.y + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        ro_issue<GW_NT>(bd, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(br.z + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        const float* gp1 = br.gp1 ? br.gp1 : br.gp0;
        gv = br.gp0[(size_t)b * H + ns0 + (t & (GC_N - 1))];
        gv1 = gp1[(size_t)pb * H + ns0 + (t & (GC_N - 1))];
    } else {
        const float* d0 = UP ? br.dy0 : br.dout;
        ro_issue<GW_NT>(bd, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d0 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        if (UP) {
            const float* d1 = br.dy1 ? br.dy1 : br.dy0;
            ro_issue<GW_NT>(bd1, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d1 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
            ro_issue<GW_NT>(by, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(br.y + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        }
    }
    float4 wb[8];                                        // P2's B operand: row kq of W[:, ns], the four n of every eight
    {
        const float* wrow = br.W + (size_t)kq * H + ns0 + 4 * lk;
#pragma unroll
        for (int i = 0; i < 8; ++i) wb[i] = *reinterpret_cast<const float4*>(wrow + 8 * i);
    }
    const int pv = g.ptr[g0 + min(t, rows)];
    const float dv = br.dis[g0 + min(t, rows - 1)];
    const float rv = RS ? br.rs[(size_t)(g0 + min(t, rows - 1)) * br.rs_stride] : 1.f;
    int nv[SU], ev[SU];
    float cin[SU];                                       // coefficients in this slot order from an earlier kernel (k_plan_graph), if any
    const int slot_hi = max(g.nnz - 1, 0);
    const float* coefp = br.coef_in ? br.coef_in : br.dis;
    const int coef_hi = br.coef_in ? slot_hi : 0;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = min(e0 + max(min(t + u * GW_NT, ne - 1), 0), slot_hi);
        nv[u] = g.nbr[s];
        ev[u] = g.eid[s];
        cin[u] = coefp[min(s, coef_hi)];
    }
    const bool ulane = UP && t >= 256;
    BNRawS braws = UP ? bn_raws_load2(br.bn, min(t, K - 1), br.ubn, ns0 + (t & (GC_N - 1)), ulane) : bn_raws_load(br.bn, min(t, K - 1));
    StripeVal ud1s, ud2s;
    if (UP) {
        const int c = ns0 + (t & (GC_N - 1));
        ud1s = stripe_load(br.udot_sum, c, br.ubn.ss); ud2s = stripe_load(br.udot_prod, c, br.ubn.ss);
    }
    bn_raws_pin(braws);
    if (UP) { stripe_pin(ud1s); stripe_pin(ud2s); }
#pragma unroll
    for (int u = 0; u < SU; ++u) asm volatile("" : "+v"(nv[u]), "+v"(ev[u]), "+v"(cin[u]));
#pragma unroll
    for (int i = 0; i < 8; ++i) ro_pin(wb[i]);
    if (POOL) { asm volatile("" : "+v"(gv), "+v"(gv1)); gv += br.gp1 ? gv1 : 0.f; }
    if (ne <= 0) {
#pragma unroll
        for (int u = 0; u < SU; ++u) { nv[u] = g0; ev[u] = 0; }
    }
    if (UP && t >= 256 && t < 256 + GC_N) {
        float m1, r1;
        const BNRaw uraw = bn_raws_sum(br.ubn, braws);
        const double ud1 = stripe_total(ud1s, br.ubn.ss), ud2 = stripe_total(ud2s, br.ubn.ss);
        bn_raw_mean_rstd(br.ubn, uraw, m1, r1);
        um_s[t - 256] = m1; ur_s[t - 256] = r1;
        ug_s[t - 256] = uraw.g * r1;
        u1_s[t - 256] = (float)(ud1 * (double)br.ubn.inv_n);
        u2_s[t - 256] = (float)(ud2 * (double)br.ubn.inv_n);
    }
    if (t < K) {
        float m1, r1;
        const BNRaw braw = bn_raws_sum(br.bn, braws);
        bn_raw_mean_rstd(br.bn, braw, m1, r1);
        mean_s[t] = m1; rstd_s[t] = r1;
        gam_s[t] = braw.g;
        bet_s[t] = braw.b;
    }
    // second round: coefficient of the out-edge i -> j: w_e * dis_j
    float cv[SU];
    if (br.coef_in) {
#pragma unroll
        for (int u = 0; u < SU; ++u) cv[u] = cin[u];
    } else {
        const float* ewp = hasw ? br.ew : br.dis;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const float c = br.dis[nv[u]];
            const float wl = ewp[hasw ? ev[u] : 0];
            cv[u] = hasw ? c * wl : c;
        }
    }
    // ---- stage in LDS ------------------------------------------------------------------------------------------------------
    if (t <= rows) ptr_s[t] = pv - e0;
    if (t < GW_T) { dis_s[t] = t < rows ? dv : 0.f; rs_s[t] = t < rows ? rv : 0.f; }
    if (POOL && t < GW_T * 2) mask_s[t] = 0u;
    if (POOL && t < GC_N) gv_s[t] = gv;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int s = t + u * GW_NT;
        if (s < ne) {
            const int loc = nv[u] - g0;
            const bool inb = loc >= 0 && loc < rows;
            en[s] = (unsigned char)(inb ? loc : 0); ec[s] = inb ? cv[u] : 0.f;
            if (POOL) ee[s] = (unsigned short)min(max(ev[u] - e0, 0), GW_E - 1);
            if (!inb) atomicOr(status, 16);
        }
    }
    if (MODE == 0) ro_commit<GW_NT>(bd, rows, 16, [&](int j, int n4, const float4 v) { *reinterpret_cast<float4*>(Ds + j * GW_LDD + 4 * n4) = v; });
    __syncthreads();                                     // BatchNorm constants, row scales, zeroed masks, CSR
    if (POOL) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) { ro_pin(by.v[u]); ro_pin(bd.v[u]); }
        const int c = 4 * (t & 15);
        const float g4[4] = {gv_s[c], gv_s[c + 1], gv_s[c + 2], gv_s[c + 3]};
#pragma unroll
        for (int u = 0; u < 8; ++u) {                   // item (u, t) = row t / 16 + 32 u, column group t % 16
            const int j = (t >> 4) + u * (GW_NT / 16);
            if (j < rows) {
                const float4 yv = by.v[u], zv = bd.v[u];
                const unsigned bits = (yv.x > 0.f ? 1u : 0u) | (yv.y > 0.f ? 2u : 0u) | (yv.z > 0.f ? 4u : 0u) | (yv.w > 0.f ? 8u : 0u);
                cs[0] += yv.x > 0.f ? g4[0] : 0.f; cs[1] += yv.y > 0.f ? g4[1] : 0.f;
                cs[2] += yv.z > 0.f ? g4[2] : 0.f; cs[3] += yv.w > 0.f ? g4[3] : 0.f;
                atomicOr(&mask_s[2 * j + ((t & 15) >> 3)], bits << (4 * (t & 7)));
                *reinterpret_cast<float4*>(Ds + j * GW_LDD + c) = make_float4(zv.x * g4[0], zv.y * g4[1], zv.z * g4[2], zv.w * g4[3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cs[q] += __shfl_xor(cs[q], 16, 64);
            cs[q] += __shfl_xor(cs[q], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bs_s[t >> 6][lane][q] = cs[q];
        }
    }
    if (UP) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        const bool two = br.dy1 != nullptr;
#pragma unroll
        for (int u = 0; u < 8; ++u) { ro_pin(bd.v[u]); ro_pin(bd1.v[u]); ro_pin(by.v[u]); }
        const int c = 4 * (t & 15);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = (t >> 4) + u * (GW_NT / 16);
            if (j < rows) {
                const float4 v0 = bd.v[u], v1 = bd1.v[u], yv = by.v[u];
                const float d[4] = {v0.x + (two ? v1.x : 0.f), v0.y + (two ? v1.y : 0.f), v0.z + (two ? v1.z : 0.f), v0.w + (two ? v1.w : 0.f)};
                const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float yn = (yy[q] - um_s[c + q]) * ur_s[c + q];
                    const float g1 = ug_s[c + q] * (d[q] - u1_s[c + q] - yn * u2_s[c + q]);
                    o[q] = yy[q] > 0.f ? g1 : 0.f;
                    cs[q] += o[q];
                }
                *reinterpret_cast<float4*>(Ds + j * GW_LDD + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cs[q] += __shfl_xor(cs[q], 16, 64);
            cs[q] += __shfl_xor(cs[q], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bs_s[t >> 6][lane][q] = cs[q];
        }
    }
    // rows rows .. rowsP of dz: zero (they are reduced over in P3; P2 computes and drops them)
    for (int i = t; i < (rowsP - rows) * GW_LDD; i += GW_NT) Dz[rows * GW_LDD + i] = 0.f;
    __syncthreads();
    if ((UP || POOL) && first && t < GC_N) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < GW_NT / 64; ++k) tot += (double)bs_s[k][t >> 2][t & 3];
        br.bias_parts[(size_t)b * H + ns0 + t] = tot;
    }
    BLK_CLK(2);
    // ---- P1: dz rows = A_hat^T dOut, sparse from LDS (four rows per wave at a time, 16 lanes x 4 columns per row) ---------------
    if (!POOL) {
        gw_sparse_rows<4, GW_LDD>(Ds, ptr_s, dis_s, en, ec, rows, w, lane, loop_w, [&](int row, float di, float (&a)[4]) {
            *reinterpret_cast<float4*>(Dz + row * GW_LDD + 4 * (lane & 15)) = make_float4(di * a[0], di * a[1], di * a[2], di * a[3]);
        });
    } else {
        // POOL: dOut[j][n] = mask[j][n] * g[n] is not in LDS -- the aggregation adds the coefficients of the slots whose source row has the
        // column's bit set, and the SDDMM  gn[e: i -> j] = <dOut[j], z[i]> = sum_n mask[j][n] Zg[i][n]  is a masked sum of the row's OWN
        // Zg values, reduced over the group's 16 lanes (DPP) and kept by the lane that holds slot e -- which then stores it by edge id
        const int grp = lane >> 4, lc = lane & 15;
        const int mw = lc >> 3, msh = 4 * (lc & 7);      // the lane's four mask bits: word mw of the row's two, shifted by msh
        const int nit = (((rows + 3) >> 2) - w + 7) >> 3;
        float* gnp = br.gn + (size_t)sl * br.gn_stride;
        float* gsp = br.gself + (size_t)sl * br.gself_stride;
        const bool emit = first;
        const float4 g4 = *reinterpret_cast<const float4*>(gv_s + 4 * lc);
        GwRow cur = gw_row(ptr_s, dis_s, rows, 0, w, grp);
        GwRow nx1 = gw_row(ptr_s, dis_s, rows, 1, w, grp);
        GwRaw craw = gw_raw(en, ec, cur, 0, lc);
        int ceid = ee[cur.p0 + min(lc, max(cur.cnt - 1, 0))];
        for (int it = 0; it < nit; ++it) {
            const GwRaw nraw = gw_raw(en, ec, nx1, 0, lc);
            const int neid = ee[nx1.p0 + min(lc, max(nx1.cnt - 1, 0))];
            const GwRow nx2 = gw_row(ptr_s, dis_s, rows, it + 2, w, grp);
            const int rc = min(cur.r, rows - 1);
            const float4 zg = *reinterpret_cast<const float4*>(Ds + rc * GW_LDD + 4 * lc);
            const unsigned mself = (mask_s[2 * rc + mw] >> msh) & 15u;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            const int cm = max(max(__builtin_amdgcn_readlane(cur.cnt, 0), __builtin_amdgcn_readlane(cur.cnt, 16)),
                               max(__builtin_amdgcn_readlane(cur.cnt, 32), __builtin_amdgcn_readlane(cur.cnt, 48)));
            auto msum = [&](unsigned m) {                // sum over the group's 64 columns of mask * Zg[row]
                float p = ((m & 1u) ? zg.x : 0.f) + ((m & 2u) ? zg.y : 0.f) + ((m & 4u) ? zg.z : 0.f) + ((m & 8u) ? zg.w : 0.f);
                return group_sum<16>(p);
            };
            for (int base = 0; base < cm; base += 16) {
                GwRaw ch = craw;
                int eid = ceid;
                if (base > 0) { ch = gw_raw(en, ec, cur, base, lc); eid = ee[cur.p0 + min(base + lc, max(cur.cnt - 1, 0))]; }
                const bool ok = base + lc < cur.cnt;
                const int cj = ok ? ch.j : rc;
                const float cc = ok ? ch.c : 0.f;
                const int m = cm - base;
                float gnv = 0.f;
                auto steps4 = [&](auto sbt) {
                    constexpr int SB = decltype(sbt)::value;
                    unsigned mk[4]; float cs[4];
                    { const int j = gw_bcast<SB * 4 + 0>(cj); cs[0] = gw_bcast<SB * 4 + 0>(cc); mk[0] = mask_s[2 * j + mw]; }
                    { const int j = gw_bcast<SB * 4 + 1>(cj); cs[1] = gw_bcast<SB * 4 + 1>(cc); mk[1] = mask_s[2 * j + mw]; }
                    { const int j = gw_bcast<SB * 4 + 2>(cj); cs[2] = gw_bcast<SB * 4 + 2>(cc); mk[2] = mask_s[2 * j + mw]; }
                    { const int j = gw_bcast<SB * 4 + 3>(cj); cs[3] = gw_bcast<SB * 4 + 3>(cc); mk[3] = mask_s[2 * j + mw]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned mm = (mk[u] >> msh) & 15u;
                        a[0] += (mm & 1u) ? cs[u] : 0.f; a[1] += (mm & 2u) ? cs[u] : 0.f;
                        a[2] += (mm & 4u) ? cs[u] : 0.f; a[3] += (mm & 8u) ? cs[u] : 0.f;
                        if (emit) { const float pv = msum(mm); gnv = lc == SB * 4 + u ? pv : gnv; }
                    }
                };
                steps4(std::integral_constant<int, 0>());
                if (m > 4) steps4(std::integral_constant<int, 1>());
                if (m > 8) steps4(std::integral_constant<int, 2>());
                if (m > 12) steps4(std::integral_constant<int, 3>());
                if (emit && ok) gnp[e0 + eid] = gnv;
            }
            const float sw = cur.di * loop_w;
            a[0] += (mself & 1u) ? sw : 0.f; a[1] += (mself & 2u) ? sw : 0.f;
            a[2] += (mself & 4u) ? sw : 0.f; a[3] += (mself & 8u) ? sw : 0.f;
            if (emit) { const float pv = msum(mself); if (lc == 0 && cur.r < rows) gsp[g0 + cur.r] = pv; }
            if (cur.r < rows)
                *reinterpret_cast<float4*>(Dz + cur.r * GW_LDD + 4 * lc) =
                    make_float4(g4.x * cur.di * a[0], g4.y * cur.di * a[1], g4.z * cur.di * a[2], g4.w * cur.di * a[3]);
            cur = nx1; nx1 = nx2; craw = nraw; ceid = neid;
        }
    }
    __syncthreads();                                     // dz complete; the dOut stage is free
    BLK_CLK(3);
    double* red = reinterpret_cast<double*>(Ds);         // [8][2][32]
    // ---- P2: partial dX' = dz[:, ns] W[:, ns]^T, row tile by row tile; the BatchNorm-backward sums ride on its epilogue -------------
    if (p2w && wk * 32 < K) {
        const int rt0 = nsplit == 4 ? part * 4 + (w >> 2) * 2 : nsplit == 2 ? (w >> 2) * 4 : 0;
        const int rt1 = min(R, nsplit == 4 ? rt0 + 2 : nsplit == 2 ? rt0 + 4 : 8);
        const float mean = mean_s[kq], rstd = rstd_s[kq];
        float* dxp = sl ? br.dxp1 : br.dxp0;
        const float* xk = br.x + (size_t)g0 * K + kq;
        float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
        // operands of tile rt + 1 (dz rows from LDS, this lane's x column from L2) are requested before the MFMAs of tile rt
        auto rd = [&](int rt, float4 (&av)[8], float (&xh)[16], float (&rr)[RS ? 16 : 1]) {
            const float* ap = Dz + (rt * 32 + li) * GW_LDD + 4 * lk;
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = *reinterpret_cast<const float4*>(ap + 8 * i);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                xh[r] = xk[(size_t)min(i, rows - 1) * K];
                if (RS) rr[r] = rs_s[i];
            }
        };
        auto tile = [&](int rt, float4 (&av)[8], float (&xh)[16], float (&rr)[RS ? 16 : 1]) {
            gc_f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, wb[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, wb[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, wb[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, wb[i].w, acc, 0, 0, 0);
            }
            // (rows past the graph: their dz rows are zero, so is the product -- no mask; their x_hat is a clamped, finite read)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float xs = RS ? xh[r] * rr[r] : xh[r];
                const float xn = (xs - mean) * rstd;
                f1[r & 3] += acc[r];
                f2[r & 3] = fmaf(acc[r], xn, f2[r & 3]);
            }
            gc_store_tile(acc, dxp + (size_t)(g0 + rt * 32) * K + wk * 32, K, rows - rt * 32, li, lk);
        };
        float4 avA[8], avB[8];
        float xhA[16], xhB[16], rrA[RS ? 16 : 1], rrB[RS ? 16 : 1];
        if (rt0 < rt1) rd(rt0, avA, xhA, rrA);
        for (int rt = rt0; rt < rt1; rt += 2) {
            if (rt + 1 < rt1) rd(rt + 1, avB, xhB, rrB);
            __builtin_amdgcn_sched_barrier(0);
            tile(rt, avA, xhA, rrA);
            __builtin_amdgcn_sched_barrier(0);
            if (rt + 1 < rt1) {
                if (rt + 2 < rt1) rd(rt + 2, avA, xhA, rrA);
                __builtin_amdgcn_sched_barrier(0);
                tile(rt + 1, avB, xhB, rrB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double s1 = ((double)f1[0] + (double)f1[1]) + ((double)f1[2] + (double)f1[3]);
        double s2 = ((double)f2[0] + (double)f2[1]) + ((double)f2[2] + (double)f2[3]);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) { red[(w * 2 + 0) * 32 + li] = s1; red[(w * 2 + 1) * 32 + li] = s2; }
    }
    gc_f32x16 acc_p3;                                    // nsplit 4: this wave's half of the node range
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_p3[i] = 0.f;
    // ---- P3: dW[:, ns] (this graph's slab) = x'^T dz[:, ns] ---------------------------------------------------------------------
    if (p3w && wk * 32 < K) {
        const float sc = rstd_s[kq] * gam_s[kq], sh = bet_s[kq] - mean_s[kq] * sc;
        const float* xk = br.x + (size_t)g0 * K + kq;
        gc_f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        const int nsb = rowsP >> 5;
        if (nsplit == 4) {
            // this workgroup's column tile; waves w and w + 4 share the k tile and halve the node range, combined through LDS below
            const int ct = part - np2, hsb = (nsb + 1) >> 1;
            gw_p3<1, RS>(xk, K, rs_s, Dz + ct * 32 + li, rowsP, rows, lk, sc, sh, acc, w < 4 ? 0 : hsb, w < 4 ? hsb : nsb);
            acc_p3 = acc[0];
        } else if (nsplit == 2) {
            const int ct = w >> 2;
            gw_p3<1, RS>(xk, K, rs_s, Dz + ct * 32 + li, rowsP, rows, lk, sc, sh, acc, 0, nsb);
            gc_store_tile(acc[0], slab + (size_t)(wk * 32) * H + ns0 + ct * 32, H, 32, li, lk);
        } else {
            gw_p3<2, RS>(xk, K, rs_s, Dz + li, rowsP, rows, lk, sc, sh, acc, 0, nsb);
#pragma unroll
            for (int q = 0; q < 2; ++q) gc_store_tile(acc[q], slab + (size_t)(wk * 32) * H + ns0 + q * 32, H, 32, li, lk);
        }
    }
    if (nsplit == 4 && do_p3) {
        // (P3-only workgroup: the P2 scratch is idle, every wave reaches this barrier)
        float* sc3 = reinterpret_cast<float*>(Ds);       // [4 k tiles][16][64]
        gc_f32x16 acc3;
        const bool live = wk * 32 < K;
        if (w >= 4 && live) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc3[((w - 4) * 16 + i) * 64 + lane] = acc_p3[i];
        }
        __syncthreads();
        if (w < 4 && live) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc3[i] = acc_p3[i] + sc3[(w * 16 + i) * 64 + lane];
            gc_store_tile(acc3, slab + (size_t)(wk * 32) * H + ns0 + (part - np2) * 32, H, 32, li, lk);
        }
    }
    if (!do_p2) { BLK_CLK(1); return; }
    __syncthreads();
    BLK_CLK(1);
    if (t < K) {
        // column t of the two sums: P2's wave(s) of k tile t / 32
        const int kt = t >> 5, l = t & 31;
        double s1 = red[(kt * 2 + 0) * 32 + l], s2 = red[(kt * 2 + 1) * 32 + l];
        if (nsplit >= 2) { s1 += red[((kt + 4) * 2 + 0) * 32 + l]; s2 += red[((kt + 4) * 2 + 1) * 32 + l]; }
        if (br.dacc_sum) {
            const size_t po = (size_t)stripe_of_block() * br.dacc_ss + t;
            atomicAdd(br.dacc_sum + po, s1); atomicAdd(br.dacc_prod + po, s2);
        } else { parts[t] = s1; parts[K + t] = s2; }
    }
}

}  // namespace cal
