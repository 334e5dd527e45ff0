// Per-graph fused GCN convolution of the step engine, forward:
//     out = relu(A_hat (BN(rs * x) @ W) + b)          (gcn_conv.py:75-104 behind model.py:93-95, 112-113)
// as ONE kernel instead of GEMM -> aggregation.  A mini-batch's adjacency is block diagonal (one block
// per graph, ~57 nodes for SPMotif), so a workgroup that owns a whole graph and a 64-column slice of the
// output keeps z = BN(x) W in LDS, builds the graph's dense normalised adjacency block next to it and
// aggregates with a second matrix product: z never goes to HBM, there is no neighbour gather at all, and
// the per-graph column sums of the output (add-pool, model.py:115-116) and the next BatchNorm's batch
// statistics fall out of the epilogue.
//
//   grid (B graphs, H / 64 column slices, branches), 256 threads; requires per graph <= GC_T nodes and
//   <= GC_E stored edges (the host passes the batch's bounds, cal_engine_set_graph_bounds; otherwise the
//   unfused kernels run), H % 64 == 0 and K = hidden <= GC_K.
//
// Timeline of a workgroup: graph extents (gptr/eptr) -> every global load of the kernel issued at once
// (x rows, W slice, CSR rows and edges, BN statistics) -> edge coefficients dis_j * w_e -> operands
// staged in LDS (x transposed to k-major with BN / row scale applied) -> z = x' W on 32x32x2 f32 MFMAs ->
// z tile to LDS (over the W stage), adjacency block At[j][i] = dis_i * coef_ij (+ the self loop) built
// over the x stage by one lane per row -> out tile = A z on MFMAs -> bias, ReLU, store, column sums.
// (Aggregating from the CSR rows with LDS reads instead was a chain of dependent LDS accesses per edge
// with one wave per SIMD to hide it: 4.5-9 us per graph; the dense product is ~1 us.)
#pragma once
#include "engine_readout.hpp"     // RO_CLK profiling aid

namespace cal {

constexpr int GC_T = 128;                 // nodes per graph (T = 128 instantiation; T = 64 for small graphs)
constexpr int GC_N = 64;                  // output columns per workgroup
constexpr int GC_K = 128;                 // reduction width (= hidden)
constexpr int GC_E = 2048;                // stored edges per graph (T = 128; half of it for T = 64)
constexpr int GC_LDB = GC_N + 4, GC_LDZ = GC_N + 1;
// T = 64 keeps a workgroup under 80 KB of LDS, so two of them share a CU: the two-branch launch (512
// workgroups) then needs one pass over the chip instead of two, and one workgroup's loads overlap the
// other's MFMAs.
constexpr int gc_edge_cap(int T) { return T == 64 ? 1024 : 2048; }

struct GconvBranch {
    const float* x;          // [N,K] layer input (raw)
    const float* W;          // [K,H]
    const float* bias;       // [H]
    const float* ew;         // per-edge weight in edge-id order, or null (all ones)
    const float* dis;        // [N] deg^-1/2 (of the weighted degrees when ew is set)
    const float* rs;         // per-row scale of x (node attention), or null
    int rs_stride;
    BNRef bn;                // BatchNorm applied to rs * x
    float* out;              // [N,H]
    float* z;                // [N,H] BN(rs x) W, kept for the backward of the weighted convs, or null
    float* pooled;           // [B,H] per-graph column sums of out (global_add_pool), or null
    Acc st_sum, st_sq;       // column statistics of out (one partial row per graph), or off
    // edge coefficients dis_j * w_e in CSR-slot order: the first kernel of a step that needs them writes them
    // (coef_out), every later one (deeper layers, the backward) reads them in its first round of loads (coef_in)
    // instead of chasing nbr -> dis / eid -> w in a second one
    const float* coef_in;
    float* coef_out;
    float* w_out;            // with coef_out: the raw edge weights w_e in CSR-slot order (read by the per-graph attention backward), or null
    // packed batch (TILED instantiation, cal_engine_set_tiles): the workgroup's unit is a tile of the consecutive graphs
    // [tile_gptr[b], tile_gptr[b + 1]) and `pooled` is per GRAPH: batch [N] names the graph of every row
    const int64_t* batch;
    const int64_t* tile_gptr;
};
constexpr int GC_TILE_GRAPHS = 8;         // graphs per tile at most (the POOL backward keeps one pooled-gradient row per graph in LDS)

struct GconvBranch2 { GconvBranch b[2]; };

typedef float gc_f32x16 __attribute__((ext_vector_type(16)));

// ---- storing a 32 x 32 MFMA accumulator tile --------------------------------------------------------------------------
// Lane (li, lk) of a 32x32 tile holds ONE column (li) of the rows (r & 3) + 8 (r >> 2) + 4 lk: stored as it lies that is 16
// 4-byte store instructions per tile.  The four registers of a row group and the four lanes of a quad form a 4 x 4 block of
// (row, column): transposed inside the quad (two DPP exchanges, no LDS) every lane holds four CONSECUTIVE columns of one
// row, i.e. one 16-byte store -- 4 instructions per tile instead of 16, same bytes, same addresses.  Measured on
// k_gconv_bwd (profiles/r3/store_burst.txt): the store phase of a workgroup is 3.4 us of its 12 us and stays 2.8 us with
// the wide stores -- it is bound by BYTES (a workgroup's 64 KB leave its CU at ~20 GB/s), not by instruction issue; the wide
// form is kept for the 0.5 us.
__device__ __forceinline__ float gc_dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float gc_dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
// in: lane q of the quad holds (v0..v3) = column q of rows 0..3; out: row q of columns 0..3
__device__ __forceinline__ void gc_quad_transpose(float& v0, float& v1, float& v2, float& v3, int q) {
    const bool o1 = q & 1, o2 = q & 2;
    float r = gc_dpp_xor1(o1 ? v0 : v1);
    if (o1) v0 = r; else v1 = r;
    r = gc_dpp_xor1(o1 ? v2 : v3);
    if (o1) v2 = r; else v3 = r;
    r = gc_dpp_xor2(o2 ? v0 : v2);
    if (o2) v0 = r; else v2 = r;
    r = gc_dpp_xor2(o2 ? v1 : v3);
    if (o2) v1 = r; else v3 = r;
}
// tile(row, col) -> base[row * ld + col] for the rows with row < nrow (base, ld: 16-byte aligned / a multiple of 4 floats);
// f(v): applied to every element before the store (bias, ReLU ..)
template <typename F>
__device__ __forceinline__ void gc_store_tile(const gc_f32x16& acc, float* base, size_t ld, int nrow, int li, int lk, F f) {
    const int q = li & 3, c4 = li & ~3;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v0 = f(acc[4 * g]), v1 = f(acc[4 * g + 1]), v2 = f(acc[4 * g + 2]), v3 = f(acc[4 * g + 3]);
        gc_quad_transpose(v0, v1, v2, v3, q);
        const int row = 8 * g + 4 * lk + q;
        if (row < nrow) *reinterpret_cast<float4*>(base + (size_t)row * ld + c4) = make_float4(v0, v1, v2, v3);
    }
}
__device__ __forceinline__ void gc_store_tile(const gc_f32x16& acc, float* base, size_t ld, int nrow, int li, int lk) {
    gc_store_tile(acc, base, ld, nrow, li, lk, [](float v) { return v; });
}

// kred/32 blocks of 16 MFMA steps over k-major LDS operands A[k][row] (stride LDA) and B[k][col]
// (stride LDB); TWO = this wave also owns row tile r0 + 2 (graphs with more than 64 nodes)
template <bool TWO, int LDA, int LDB>
__device__ __forceinline__ void gconv_mma(const float* As, const float* Bs, int kred, int r0, int ct, int li, int lk,
                                          gc_f32x16& acc0, gc_f32x16& acc1) {
    const float* a0p = As + r0 * 32 + li;
    const float* a1p = As + (r0 + 2) * 32 + li;
    const float* bp = Bs + ct * 32 + li;
    float a0[2][16], a1[2][16], bv[2][16];
    auto read_ops = [&](int kb, int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kb * 32 + 2 * i + lk;
            a0[s][i] = a0p[k * LDA];
            if (TWO) a1[s][i] = a1p[k * LDA];
            bv[s][i] = bp[k * LDB];
        }
    };
    const int nkb = kred / 32;
    read_ops(0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        if (kb + 1 < nkb) read_ops(kb + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0][i], bv[0][i], acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][i], bv[0][i], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) {
            if (kb + 2 < nkb) read_ops(kb + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1][i], bv[1][i], acc0, 0, 0, 0);
                if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][i], bv[1][i], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// z = x' W with the A operand ROW-MAJOR in k (x' rows as loaded: Xr[row * GC_LDX + k], stride 132 = 4 mod 32: conflict-free
// 16 B reads) and B k-major (the W slice as loaded).  Lane (li, lk) takes the four consecutive k of every eight from its
// row with one ds_read_b128 and the matching four B values with 4 B reads -- any bijection of k onto (MFMA step, lk) is a
// valid reduction order when both operands share it.  Against the k-major x stage: no transposing scalar stores while
// staging (8 float4 stores per lane instead of 32 scalar ones) and a third fewer LDS reads in the product.
constexpr int GC_LDX = GC_K + 4;
template <bool TWO, int LDB>
__device__ __forceinline__ void gconv_mma_arow(const float* Xr, const float* Bs, int kred, int r0, int ct, int li, int lk,
                                               gc_f32x16& acc0, gc_f32x16& acc1) {
    const float* a0p = Xr + (r0 * 32 + li) * GC_LDX + 4 * lk;
    const float* a1p = Xr + ((r0 + 2) * 32 + li) * GC_LDX + 4 * lk;
    const float* bp = Bs + ct * 32 + li + 4 * lk * LDB;
    float4 a0[2][4], a1[2][4];
    float bv[2][16];
    auto read_ops = [&](int kb, int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[s][i] = *reinterpret_cast<const float4*>(a0p + kb * 32 + 8 * i);
            if (TWO) a1[s][i] = *reinterpret_cast<const float4*>(a1p + kb * 32 + 8 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[s][4 * i + j] = bp[(kb * 32 + 8 * i + j) * LDB];
        }
    };
    auto mul = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x[4] = {a0[s][i].x, a0[s][i].y, a0[s][i].z, a0[s][i].w};
            const float y[4] = {TWO ? a1[s][i].x : 0.f, TWO ? a1[s][i].y : 0.f, TWO ? a1[s][i].z : 0.f, TWO ? a1[s][i].w : 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], bv[s][4 * i + j], acc0, 0, 0, 0);
                if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y[j], bv[s][4 * i + j], acc1, 0, 0, 0);
            }
        }
    };
    const int nkb = kred / 32;
    read_ops(0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        if (kb + 1 < nkb) read_ops(kb + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mul(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) {
            if (kb + 2 < nkb) read_ops(kb + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mul(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// out = A z with BOTH operands row-major in the reduction index j (the adjacency block as At[i * LD + j], the z tile
// transposed as Zt[col * LD + j], LD = 4 mod 32): 16 B reads, four MFMA steps per read (same k mapping as above)
template <bool TWO, int LD>
__device__ __forceinline__ void gconv_mma_rowk(const float* At, const float* Zt, int kred, int r0, int ct, int li, int lk,
                                               gc_f32x16& acc0, gc_f32x16& acc1) {
    const float* a0p = At + (r0 * 32 + li) * LD + 4 * lk;
    const float* a1p = At + ((r0 + 2) * 32 + li) * LD + 4 * lk;
    const float* bp = Zt + (ct * 32 + li) * LD + 4 * lk;
    for (int k0 = 0; k0 < kred; k0 += 32) {
        float4 a0[4], a1[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = *reinterpret_cast<const float4*>(a0p + k0 + 8 * i);
            if (TWO) a1[i] = *reinterpret_cast<const float4*>(a1p + k0 + 8 * i);
            bv[i] = *reinterpret_cast<const float4*>(bp + k0 + 8 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x[4] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w}, b[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
            const float y[4] = {TWO ? a1[i].x : 0.f, TWO ? a1[i].y : 0.f, TWO ? a1[i].z : 0.f, TWO ? a1[i].w : 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], b[j], acc0, 0, 0, 0);
                if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y[j], b[j], acc1, 0, 0, 0);
            }
        }
    }
}

template <bool RS, int T, int NT = 256, bool TILED = false>
__global__ void __launch_bounds__(NT, (T == 64 ? 2 : 1)) k_gconv_fwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                   const GconvBranch2 bb, int relu, float loop_w, int H,
                                                   int K, int* __restrict__ status) {
    // NT = 256: one 32 x 32 tile pair per wave.  NT = 512 (T = 64): eight waves -- the first product's reduction range is
    // split over two waves per tile (partial z tiles combined through LDS), twice the lanes stage the operands, and waves
    // 4-7 are free for the adjacency block while waves 0-3 finish the z tile
    constexpr int LDA = T + 1, ECAP = gc_edge_cap(T);
    constexpr int WU = 2048 / NT, CU = (ECAP + NT - 1) / NT, RPP = NT / 8, NRS = T / RPP > 0 ? T / RPP : 1;
    static_assert(NT == 256 || (NT == 512 && T == 64), "512 threads: 64-node variant only");
    __shared__ __attribute__((aligned(16))) float As[(GC_K * LDA > T * GC_LDX) ? GC_K * LDA : T * GC_LDX];   // x' rows [row][k] (stride GC_LDX); later the adjacency block [j][i] (stride LDA)
    __shared__ __attribute__((aligned(16))) float Bs[GC_K * GC_LDB];       // W slice [k][col]; later the z tile [row][col]
    __shared__ float sc_s[GC_K], sh_s[GC_K];
    __shared__ int ptr_s[T + 4];
    __shared__ float dis_s[T];
    __shared__ short en[ECAP];                          // (local node index < T: 2 bytes keep the T = 64 instantiation at two workgroups per CU)
    __shared__ float ec[ECAP];
    __shared__ signed char er[ECAP];
    __shared__ double red[4][2][32];
    __shared__ float pool_s[4][32];
    __shared__ unsigned char bg_s[TILED ? T : 4];        // TILED: graph (inside the tile) of every row
    BLK_CLK(0);
    warm_kernargs<sizeof(CSR) + 2 * sizeof(void*) + sizeof(GconvBranch2) + 32>();
    const GconvBranch& br = bb.b[blockIdx.z];           // indexed in the kernel-argument segment: one set of scalar loads (b0 / b1 as two parameters were loaded both and selected field by field)
    const int b = blockIdx.x, n0 = blockIdx.y * GC_N, t = threadIdx.x;
    // the W slice does not depend on the graph: requested before the graph's extents (a scalar round trip) are known
    float4 vb[WU];                                       // W[k][n0 + 4 j4 ..]: 16 lanes per k row
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int idx = t + u * NT, k = min(idx >> 4, K - 1), j4 = idx & 15;
        vb[u] = *reinterpret_cast<const float4*>(br.W + (size_t)k * H + n0 + 4 * j4);
    }
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const int tg0 = TILED ? (int)br.tile_gptr[b] : b, ng = TILED ? (int)br.tile_gptr[b + 1] - tg0 : 1;
    const bool want = br.st_sum.on();
    if (rows <= 0) {                                     // empty graph: its partial rows still have to exist
        if (br.bn.update && blockIdx.x == 0 && blockIdx.y == 0 && t < K) { const BNRaw r0 = bn_raw_load_st(br.bn, t); bn_raw_update_running(br.bn, r0, t); }
        if (t < GC_N) {
            if (want) { br.st_sum.add(n0 + t, 0.0); br.st_sq.add(n0 + t, 0.0); }
            if (br.pooled) for (int q = 0; q < ng; ++q) br.pooled[(size_t)(tg0 + q) * H + n0 + t] = 0.f;
        }
        return;
    }
    if (TILED && (ng < 1 || ng > GC_TILE_GRAPHS)) { if (t == 0) atomicOr(status, 8); return; }
    if (rows > T || ne > ECAP || ne < 0) {            // the host's bounds were wrong: flag it, write nothing
        if (t == 0) atomicOr(status, 8);
        return;
    }
    RO_CLK(32);
    BLK_CLK(2);
    const bool hasw = br.ew != nullptr;
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5, nkc = K >> 5, RB = (rowsP + RPP - 1) / RPP;
    // ---- every global load of the kernel, issued before the first wait -------------------------------------
    // x rows: item (u, t) -> 32-wide k chunk kc, row block rr, row (t >> 3), float4 (t & 7) of the chunk:
    // 8 lanes x 16 B per row (coalesced), and the transposing LDS stores below see only 2-way bank conflicts
    constexpr int UA = T * 32 / NT;                    // x float4s per lane: T rows x GC_K / 4 over NT lanes
    float4 va[UA];
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const bool ok = kc < nkc;
            const int r = min((ok ? rr : 0) * RPP + (t >> 3), rows - 1), k = ((ok ? kc : 0) << 5) + ((t & 7) << 2);
            va[u] = *reinterpret_cast<const float4*>(br.x + (size_t)(g0 + r) * K + k);
            if (++rr == RB) { rr = 0; ++kc; }
        }
    }
    const int pv = g.ptr[g0 + min(t, rows)];
    const float dv = br.dis[g0 + min(t, rows - 1)];
    long long bgv = 0;
    if (TILED) bgv = br.batch[g0 + min(t, rows - 1)];
    float rsv[4] = {1.f, 1.f, 1.f, 1.f};                 // row scale of this lane's x row in each RPP-row block
    if (RS) {
#pragma unroll
        for (int q = 0; q < NRS; ++q) rsv[q] = br.rs[(size_t)(g0 + min(q * RPP + (t >> 3), rows - 1)) * br.rs_stride];
    }
    // CSR slots, BatchNorm constants, bias, coefficients: all unconditional on clamped indices / substituted pointers
    // (see BNRaw in engine.hpp: guarded loads here cost four serial round trips behind the tile loads)
    int nv[CU], ev[CU];
    const int slot_hi = max(g.nnz - 1, 0);
#pragma unroll
    for (int u = 0; u < CU; ++u) {
        const int s = min(e0 + max(min(t + u * NT, ne - 1), 0), slot_hi);
        nv[u] = g.nbr[s];
        ev[u] = g.eid[s];
    }
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = w >> 2, ct = w & 1, r0 = (w & 3) >> 1;     // kh: half of the first product's reduction range (NT = 512)
    const float* biasp = br.bias ? br.bias : br.W;       // W: any valid [>= H] float array; the value is masked below
    float bias = biasp[n0 + ct * 32 + li];
    BNRawS braws = bn_raws_load(br.bn, min(t, K - 1));       // (striped reader: the producer may be a per-graph kernel)
    const float* coefp = br.coef_in ? br.coef_in : br.dis;
    const int coef_hi = br.coef_in ? slot_hi : 0;
    float cin[CU];                                       // coefficients of an earlier kernel of this step, if any
#pragma unroll
    for (int u = 0; u < CU; ++u) cin[u] = coefp[min(e0 + max(min(t + u * NT, ne - 1), 0), coef_hi)];
    // all of the above stay in flight together: without the pins hipcc pairs every W load with its LDS store
    // ("load, s_waitcnt vmcnt(0), ds_write" x 8: eight serial round trips, 5-30 us under 256-way contention)
#pragma unroll
    for (int u = 0; u < UA; ++u) ro_pin(va[u]);
#pragma unroll
    for (int u = 0; u < WU; ++u) ro_pin(vb[u]);
    bn_raws_pin(braws);
#pragma unroll
    for (int u = 0; u < CU; ++u) asm volatile("" : "+v"(nv[u]), "+v"(ev[u]), "+v"(cin[u]));
    const BNRaw braw = bn_raws_sum(br.bn, braws);
    asm volatile("" : "+v"(bias));
    if (!br.bias) bias = 0.f;
    if (ne <= 0) {                                       // no slot of this graph exists: what the clamped loads fetched is not an index
#pragma unroll
        for (int u = 0; u < CU; ++u) { nv[u] = g0; ev[u] = 0; }
    }
    if (t < K) {
        bn_raw_scale_shift(br.bn, braw, sc_s[t], sh_s[t]);
        if (br.bn.update && blockIdx.x == 0 && blockIdx.y == 0) bn_raw_update_running(br.bn, braw, t);
    }
    // second round: edge coefficients dis_j * w_e (needs the neighbour / edge ids)
    float cv[CU], wv[CU];
    if (br.coef_in) {
#pragma unroll
        for (int u = 0; u < CU; ++u) { cv[u] = cin[u]; wv[u] = 1.f; }
    } else {
        const float* ewp = hasw ? br.ew : br.dis;        // (masked when there are no edge weights)
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const float c = br.dis[nv[u]];
            const float wl = ewp[hasw ? ev[u] : 0];
            wv[u] = hasw ? wl : 1.f;
            cv[u] = c * wv[u];
        }
    }
    RO_CLK(33);
    // ---- stage everything in LDS ---------------------------------------------------------------------------
    if (t <= rows) ptr_s[t] = pv - e0;
    if (t < rows) dis_s[t] = dv;
    if (TILED && t < rows) bg_s[t] = (unsigned char)min(max((int)(bgv - tg0), 0), ng - 1);
#pragma unroll
    for (int u = 0; u < CU; ++u) {
        const int s = t + u * NT;
        if (s < ne) {
            const int loc = nv[u] - g0;
            const bool inb = loc >= 0 && loc < rows;        // an edge that leaves its graph is not a mini-batch: flag it
            en[s] = (short)(inb ? loc : 0); ec[s] = inb ? cv[u] : 0.f;
            if (br.coef_out && blockIdx.y == 0) { br.coef_out[e0 + s] = cv[u]; if (br.w_out) br.w_out[e0 + s] = wv[u]; }
            if (!inb) atomicOr(status, 16);
        }
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int idx = t + u * NT, k = idx >> 4, j4 = idx & 15;
        if (k < K) *reinterpret_cast<float4*>(Bs + k * GC_LDB + 4 * j4) = vb[u];
    }
    RO_CLK(34);
    __syncthreads();                                     // BN tables
    if (t < rows) {                                      // destination row of every CSR slot (stores only: no LDS latency chain)
        const int s1 = ptr_s[t + 1];
        for (int s = ptr_s[t]; s < s1; ++s) er[s] = (signed char)t;
    }
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            if (kc < nkc) {
                const int r = rr * RPP + (t >> 3), k = (kc << 5) + ((t & 7) << 2);
                float x0 = va[u].x, x1 = va[u].y, x2 = va[u].z, x3 = va[u].w;
                if (RS) {
                    const float s = (NRS == 1 || rr == 0) ? rsv[0] : ((NRS == 2 || rr == 1) ? rsv[1] : (rr == 2 ? rsv[2] : rsv[3]));
                    x0 *= s; x1 *= s; x2 *= s; x3 *= s;
                }
                *reinterpret_cast<float4*>(As + r * GC_LDX + k) =
                    make_float4(fmaf(x0, sc_s[k], sh_s[k]), fmaf(x1, sc_s[k + 1], sh_s[k + 1]), fmaf(x2, sc_s[k + 2], sh_s[k + 2]),
                                fmaf(x3, sc_s[k + 3], sh_s[k + 3]));
            }
            if (++rr == RB) { rr = 0; ++kc; }
        }
    }
    __syncthreads();
    RO_CLK(35);
    BLK_CLK(3);
    // ---- z tile = BN(x) W on the matrix cores: wave w owns column tile w & 1 and row tiles w >> 1 (, + 2) ---
    gc_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (r0 < R) {
        if (NT == 512) {                                 // this wave's half of the reduction range
            const int kofs = kh * (K >> 1);
            gconv_mma_arow<false, GC_LDB>(As + kofs, Bs + kofs * GC_LDB, K >> 1, r0, ct, li, lk, acc0, acc1);
        } else if (r0 + 2 < R) gconv_mma_arow<true, GC_LDB>(As, Bs, K, r0, ct, li, lk, acc0, acc1);
        else gconv_mma_arow<false, GC_LDB>(As, Bs, K, r0, ct, li, lk, acc0, acc1);
    }
    RO_CLK(36);
    __syncthreads();                                     // every wave is done reading both stages
    // ---- z tile -> LDS (over the W stage); zero the adjacency block (over the x stage) ------------------------
    constexpr int LDT = T + 4;                           // row stride of the two j-major tiles below (4 mod 32)
    float* Zt = Bs;                                      // Zt[col * LDT + j] = z[j][col]   (over the W stage)
    float* At = As;                                      // At[i * LDT + j] = weight of edge j -> i, times dis_i   (over the x stage)
    const bool own = NT == 256 || kh == 0;               // the wave that finishes its tile (NT = 512: adds its partner's partial below)
    if (r0 < R && (NT == 256 || kh == 1)) {
        // an accumulator holds rows 8 g + 4 lk .. + 3 of its tile in elements 4 g .. 4 g + 3: four consecutive j of one column
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int row = r0 * 32 + 8 * gq + 4 * lk;
            *reinterpret_cast<float4*>(Zt + (ct * 32 + li) * LDT + row) = make_float4(acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]);
        }
        if (NT == 256 && br.z) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < rows) br.z[(size_t)(g0 + row) * H + n0 + ct * 32 + li] = acc0[r];
            }
        }
        if (NT == 256 && r0 + 2 < R) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row = (r0 + 2) * 32 + 8 * gq + 4 * lk;
                *reinterpret_cast<float4*>(Zt + (ct * 32 + li) * LDT + row) = make_float4(acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]);
            }
            if (br.z) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r0 + 2) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < rows) br.z[(size_t)(g0 + row) * H + n0 + ct * 32 + li] = acc1[r];
                }
            }
        }
    }
    {
        const int nz4 = (rowsP * LDT) >> 2;           // rows i < rowsP of the block (contiguous), as float4s
        float4* z4 = reinterpret_cast<float4*>(At);
        for (int idx = t; idx < nz4; idx += NT) z4[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (NT == 512 && kh == 0 && r0 < R) {                // z tile = this wave's half + the partner's (already in Zt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float4* zp = reinterpret_cast<float4*>(Zt + (ct * 32 + li) * LDT + r0 * 32 + 8 * gq + 4 * lk);
            const float4 p = *zp;
            acc0[4 * gq] += p.x; acc0[4 * gq + 1] += p.y; acc0[4 * gq + 2] += p.z; acc0[4 * gq + 3] += p.w;
            *zp = make_float4(acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]);
        }
        if (br.z) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < rows) br.z[(size_t)(g0 + row) * H + n0 + ct * 32 + li] = acc0[r];
            }
        }
    }
    // one lane per CSR slot (then one per self loop): duplicate edges accumulate through the LDS atomic.  One lane per
    // destination ROW walked a hub's 30 slots as 30 dependent LDS round trips (read source, read coefficient,
    // read-modify-write the block) while the other lanes idled -- the slowest row was the phase.
    for (int s = t; s < ne; s += NT) {
        const int i = er[s];
        atomicAdd(&At[i * LDT + en[s]], dis_s[i] * ec[s]);
    }
    if (t < rows) atomicAdd(&At[t * LDT + t], dis_s[t] * dis_s[t] * loop_w);
    __syncthreads();
    RO_CLK(37);
    // ---- out tile = A z on the matrix cores (reduction over the graph's rowsP nodes) ---------------------------
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (own && r0 < R) {
        if (r0 + 2 < R) gconv_mma_rowk<true, LDT>(At, Zt, rowsP, r0, ct, li, lk, acc0, acc1);
        else gconv_mma_rowk<false, LDT>(At, Zt, rowsP, r0, ct, li, lk, acc0, acc1);
    }
    RO_CLK(38);
    // ---- epilogue: bias, ReLU, store, column sums of this graph ---------------------------------------------------
    // a lane's <= 32 terms of the column sums in fp32 (four chains, masked, no guards: inside the row guard every element
    // was a branch with two fp64 conversions and two dependent fp64 adds), everything across lanes / graphs in fp64
    float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
    float psum = 0.f;
    const int col = n0 + ct * 32 + li;
    asm volatile("" :: "v"(bias));                       // consume the bias load before the guarded stores (see gemm.hip)
    // TILED: the add-pool is per graph, several graphs share the tile: the output tile is parked in LDS (over the adjacency
    // block, once every wave has finished reading it) and summed per graph below, rows in order
    constexpr int LDO = GC_N + 1;
    float* Ot = As;
    if (TILED) __syncthreads();
    if (own && r0 < R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            float v = acc0[r] + bias;
            if (relu) v = fmaxf(v, 0.f);
            acc0[r] = v;                                 // (stored below, four columns per lane)
            const float vm = row < rows ? v : 0.f;
            if (TILED) Ot[row * LDO + ct * 32 + li] = vm;
            f1[r & 3] += vm; f2[r & 3] = fmaf(vm, vm, f2[r & 3]);
        }
        if (r0 + 2 < R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r0 + 2) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float v = acc1[r] + bias;
                if (relu) v = fmaxf(v, 0.f);
                acc1[r] = v;
                const float vm = row < rows ? v : 0.f;
                if (TILED) Ot[row * LDO + ct * 32 + li] = vm;
                f1[r & 3] += vm; f2[r & 3] = fmaf(vm, vm, f2[r & 3]);
            }
        }
        gc_store_tile(acc0, br.out + (size_t)(g0 + r0 * 32) * H + n0 + ct * 32, H, rows - r0 * 32, li, lk);
        if (r0 + 2 < R) gc_store_tile(acc1, br.out + (size_t)(g0 + (r0 + 2) * 32) * H + n0 + ct * 32, H, rows - (r0 + 2) * 32, li, lk);
    }
    psum = (f1[0] + f1[1]) + (f1[2] + f1[3]);
    double s1 = ((double)f1[0] + (double)f1[1]) + ((double)f1[2] + (double)f1[3]);
    double s2 = ((double)f2[0] + (double)f2[1]) + ((double)f2[2] + (double)f2[3]);
    // lanes lk = 0 / 1 hold different rows of the same column; waves w and w ^ 2 hold the other row tiles
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    psum += __shfl_xor(psum, 32, 64);
    if (own && lk == 0) { red[w & 3][0][li] = s1; red[w & 3][1][li] = s2; pool_s[w & 3][li] = psum; }
    __syncthreads();
    if (w < 2 && lk == 0) {
        if (want) {
            br.st_sum.add(col, red[w][0][li] + red[w + 2][0][li]);
            br.st_sq.add(col, red[w][1][li] + red[w + 2][1][li]);
        }
        if (!TILED && br.pooled) br.pooled[(size_t)b * H + col] = pool_s[w][li] + pool_s[w + 2][li];
    }
    if (TILED && br.pooled && t < GC_N) {
        // one lane per column walks the tile's rows in order and closes a pooled row whenever the graph changes; graphs
        // without nodes keep the zero written first (same lane, same address: program order)
        float* pp = br.pooled + (size_t)tg0 * H + n0 + t;
        for (int q = 0; q < ng; ++q) pp[(size_t)q * H] = 0.f;
        float sum = 0.f;
        int cur = bg_s[0];
        for (int r = 0; r < rows; ++r) {
            const int gq = bg_s[r];
            if (gq != cur) { pp[(size_t)cur * H] = sum; sum = 0.f; cur = gq; }
            sum += Ot[r * LDO + t];
        }
        pp[(size_t)cur * H] = sum;
    }
    RO_CLK(39);
    BLK_CLK(1);
}

}  // namespace cal
