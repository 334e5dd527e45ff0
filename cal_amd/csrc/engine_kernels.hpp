// Node-/edge-/graph-level kernels of the native CausalGCN step engine (included by engine.hip only).
//
// Conventions: fp32 activations, fp64 accumulation of every cross-row reduction (BatchNorm batch
// statistics, BatchNorm-backward sums, bias / attention-parameter gradients) with one atomic per
// column per workgroup into the accumulator arena; G lanes cooperate on one feature row (VEC = 4
// floats per lane -> one coalesced 16 B access per lane); a workgroup walks `rows_per_block` rows
// so the per-column pre-reduction in LDS keeps the atomic count at (#blocks x #columns).
#pragma once
#include "engine.hpp"

namespace cal {

#ifdef CAL_BLK_CLOCKS                     // profiling aid: start / end timestamp (100 MHz) of every workgroup of one kernel
__device__ long long g_blk_clk[4 * 2048];     // 4 timestamps per workgroup: entry, two free marks, exit
#define BLK_CLK(which) do { const int b_ = blockIdx.x + gridDim.x * blockIdx.y; \
                            if (threadIdx.x == 0 && b_ < 2048) g_blk_clk[4 * b_ + ((which) == 1 ? 3 : (which) == 0 ? 0 : (which) - 1)] = wall_clock64(); } while (0)
// the same from lane 0 of wave `wv`, after everything the wave has in flight (memory counters, and the MFMA chain that
// ends in accumulator element `dep`: reading it with a VALU instruction waits for the matrix pipe)
#define BLK_CLK_W(which, wv, dep) do { const int b_ = blockIdx.x + gridDim.x * blockIdx.y; int d_; \
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "=s"(d_) : "v"(dep)); \
                            if ((int)threadIdx.x == 64 * (wv) && b_ < 2048) g_blk_clk[4 * b_ + ((which) == 1 ? 3 : (which) == 0 ? 0 : (which) - 1)] = wall_clock64() + (d_ & 0); } while (0)
#else
#define BLK_CLK(which) do {} while (0)
#define BLK_CLK_W(which, wv, dep) do {} while (0)
#endif

struct CSR {
    const int* ptr;
    const int* nbr;
    const int* eid;
    int nnz;               // stored entries (edges without self loops)
};

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// Destination of a per-column cross-row sum.  Two modes:
//   atomic : dst[col] += t with one fp64 atomic per (block, column) -- fine for small launches; with stride != 0 the
//            workgroup adds into ITS plane of the site's NSTRIPE accumulator planes, `stride` doubles apart (engine.hpp:
//            stripe_sum -- the per-graph kernels, whose consumers add the planes themselves: no finishing launch);
//   partial: parts[blockIdx.x * stride + col] = t, one plain store; k_stats_final (or the final
//            commit) sums the rows.  Hot-address fp64 atomics run at only ~1.5 per ns chip-wide on
//            MI355X, so the ~230-block node-level kernels use partial rows.
struct Acc {
    double* dst;
    double* parts;
    int stride;
    __host__ __device__ Acc() : dst(nullptr), parts(nullptr), stride(0) {}
    __host__ __device__ Acc(double* d) : dst(d), parts(nullptr), stride(0) {}
    __host__ __device__ Acc(double* d, double* p, int s) : dst(d), parts(p), stride(s) {}
    __host__ __device__ bool on() const { return dst != nullptr || parts != nullptr; }
    __device__ __forceinline__ void add(int col, double t) const {
        if (parts) parts[(size_t)blockIdx.x * stride + col] = t;
        else atomicAdd(dst + (size_t)stripe_of_block() * stride + col, t);
    }
};

// Sum `v` over the row-lanes of a block for column slot `cslot` (0..ncols-1) and emit it for `col`.
// lds: at least nrl * ncols doubles.  Must be called by every thread of the block.
__device__ __forceinline__ void block_col_atomic(double v, int cslot, int rlane, int nrl, int ncols, bool valid,
                                                 const Acc& dst, int col, double* lds) {
    lds[rlane * ncols + cslot] = valid ? v : 0.0;
    __syncthreads();
    if (rlane == 0 && valid) {
        double t = 0.0;
        for (int k = 0; k < nrl; ++k) t += lds[k * ncols + cslot];
        dst.add(col, t);
    }
    __syncthreads();
}

// NV column sums at once (two barriers in total instead of two per sum): out[q] is valid on the lanes
// with rlane == 0 && valid.  lds: at least NV * nrl * ncols doubles.
template <int NV>
__device__ __forceinline__ void block_col_sums(const double (&v)[NV], int cslot, int rlane, int nrl, int ncols, bool valid,
                                               double* lds, double (&out)[NV]) {
#pragma unroll
    for (int q = 0; q < NV; ++q) lds[(q * nrl + rlane) * ncols + cslot] = valid ? v[q] : 0.0;
    __syncthreads();
    if (rlane == 0 && valid) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double t = 0.0;
            for (int k = 0; k < nrl; ++k) t += lds[(q * nrl + k) * ncols + cslot];
            out[q] = t;
        }
    }
    __syncthreads();
}

// Sum the rows of partial buffers into their final destination.  grid (ceil(n/8), ntasks): a block
// owns 8 columns, 32 lanes walk the P rows of each.  The row loop keeps 8 loads in flight per lane
// (unconditional, clamped, pinned): as a plain dependent loop over P = 458 rows this kernel took
// 6.3 us -- the loop is nothing but global-load latency.
struct FinalTask { const double* parts; int P; int stride; int n; double* dst; };
struct FinalArgs { FinalTask t[8]; int nt; };
__global__ void __launch_bounds__(256) k_stats_final(const FinalArgs fa) {
    __shared__ double red[256];
    const FinalTask t = fa.t[blockIdx.y];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7), pl = threadIdx.x >> 3;
    const int cc = min(c, t.n - 1);
    double s0 = 0.0, s1 = 0.0;
    for (int p0 = pl; p0 < t.P; p0 += 32 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t.parts[(size_t)min(p0 + 32 * u, t.P - 1) * t.stride + cc];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            s0 += p0 + 32 * u < t.P ? v[u] : 0.0;
            s1 += p0 + 32 * (u + 1) < t.P ? v[u + 1] : 0.0;
        }
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (pl == 0 && c < t.n) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += red[k * 8 + (threadIdx.x & 7)];
        t.dst[c] = tot;
    }
}

// First kernel of a step: zero the fp64 arena, the GraphPlan degree counters / cursors and the status word (the previous
// step's status bits are folded into the sticky word status[1] first, read by StepEngine.check_status once per epoch);
// for a GAT backbone in training, advance the attention-dropout step counter (one fresh mask per step); and, when the
// step draws its own random-intervention permutation (mode bit 16), the LAST cdiv(B, 64) workgroups rank-sort it (B <= ZP_CAP;
// randperm_slice: 64 elements each) -- it is consumed only by the readout, so no launch of its own in front of the step.  The
// permutation counter is advanced by the step's last kernel (k_finish).
constexpr int ZP_CAP = 1024;
constexpr int ZP_EPB = 64;                 // elements ranked per workgroup
__global__ void __launch_bounds__(256) k_zero_f64(double* __restrict__ a, int64_t n, int* __restrict__ ints, int64_t ni,
                                                  int* __restrict__ status, unsigned long long* __restrict__ tick,
                                                  int64_t* __restrict__ perm, int B, unsigned long long seed,
                                                  unsigned long long* __restrict__ counter, float* __restrict__ adam_step,
                                                  int* __restrict__ dirty) {
    // dirty != null (a training forward): bn_feat's statistics range is about to be used; only a k_finish cleans it (PlanFold)
    __shared__ unsigned long long key[ZP_CAP];
    const int pblocks = perm ? (B + ZP_EPB - 1) / ZP_EPB : 0;
    if (perm && (int)blockIdx.x >= (int)gridDim.x - pblocks) {
        randperm_slice<256>(perm, B, seed, counter, key, (int)blockIdx.x - ((int)gridDim.x - pblocks));
        return;
    }
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 0.0;
    if (i < ni) ints[i] = 0;
    if (i == 0 && status) { status[1] |= status[0]; status[0] = 0; }
    if (i == 0 && tick) *tick += 1;
    if (i == 0 && adam_step) adam_step[0] += 1.f;       // the step ends with Adam inside k_finish, which reads the counter
    if (i == 0 && dirty) *dirty = 1;
}

// ------------------------------------------------------------------------------------------------
// column statistics of a raw matrix (the bn_feat input, model.py:90)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_colstats(const float* __restrict__ x, int N, int W, int tc, int rows_per_block,
                                                  const Acc sum, const Acc sq) {
    __shared__ double lds[256];
    const int nrl = 256 / tc, c = threadIdx.x % tc, rl = threadIdx.x / tc;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    for (int cc = c; cc - c < W; cc += tc) {
        const bool ok = cc < W;
        // eight rows in flight per lane (clamped, masked when added): as a loop of single dependent-latency loads this kernel
        // took 52 us over a [15k, 139] matrix
        double s = 0.0, q = 0.0;
        if (ok && r0 + rl < r1)
            for (int r = r0 + rl; r < r1; r += 8 * nrl) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x[(size_t)min(r + u * nrl, r1 - 1) * W + cc];
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double d = r + u * nrl < r1 ? (double)v[u] : 0.0;
                    s += d; q += d * d;
                }
            }
        block_col_atomic(s, c, rl, nrl, tc, ok, sum, cc, lds);
        block_col_atomic(q, c, rl, nrl, tc, ok, sq, cc, lds);
    }
}

// The same for W = 256 (the GATConv layer outputs of config 5, model.py:388-390: the next BatchNorm's statistics): 16 B per lane,
// a wave per row, eight rows in flight per wave -- the 4 B-per-lane form above keeps 32 KB in flight per CU with the four
// workgroups a 1020-block grid leaves on it, and ran at 4 TB/s (41 us for 164 MB).  Column sums: the four waves through LDS,
// then one fp64 atomic pair per column and block.
__global__ void __launch_bounds__(256) k_colstats4(const float* __restrict__ x, int N, int rows_per_block, const Acc sum, const Acc sq) {
    constexpr int W = 256;
    __shared__ double lds[2 * 4 * W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane * 4;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = r0 + wv; r < r1; r += 32) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(x + (size_t)min(r + 4 * u, r1 - 1) * W + c);
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = r + 4 * u < r1;
            const double d0 = on ? (double)v[u].x : 0.0, d1 = on ? (double)v[u].y : 0.0, d2 = on ? (double)v[u].z : 0.0, d3 = on ? (double)v[u].w : 0.0;
            s[0] += d0; q[0] += d0 * d0; s[1] += d1; q[1] += d1 * d1;
            s[2] += d2; q[2] += d2 * d2; s[3] += d3; q[3] += d3 * d3;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { lds[wv * W + c + j] = s[j]; lds[4 * W + wv * W + c + j] = q[j]; }
    __syncthreads();
    const int col = threadIdx.x;
    const double ts = (lds[col] + lds[W + col]) + (lds[2 * W + col] + lds[3 * W + col]);
    const double tq = (lds[4 * W + col] + lds[5 * W + col]) + (lds[6 * W + col] + lds[7 * W + col]);
    sum.add(col, ts);
    sq.add(col, tq);
}

// ------------------------------------------------------------------------------------------------
// gptr + unweighted deg^-1/2 (gcn_conv.py:65-68 with edge_weight = 1)
// ------------------------------------------------------------------------------------------------
// eptr[b] = first CSR-by-destination slot of graph b (the per-graph kernels' edge range)
__global__ void k_gptr_dis(const int64_t* __restrict__ batch, int N, int B, int* __restrict__ gptr,
                           const int* __restrict__ ptr_src, float loop_w, float* __restrict__ dis_unit,
                           int* __restrict__ status, const int* __restrict__ ptr_dst, int* __restrict__ eptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    if (i < N) {
        float d = (float)(ptr_src[i + 1] - ptr_src[i]) + loop_w;
        dis_unit[i] = d == 0.f ? 0.f : 1.0f / sqrtf(d);
    }
    int64_t prev = i == 0 ? -1 : batch[i - 1];
    int64_t cur = i == N ? (int64_t)B : batch[i];
    if (i < N && (cur < prev || cur >= B || cur < 0)) { atomicOr(status, 2); return; }
    if (prev + 1 <= cur) {
        const int pd = ptr_dst[i];
        for (int64_t b = prev + 1; b <= cur && b <= B; ++b) { gptr[b] = i; eptr[b] = pd; }
    }
}

// ------------------------------------------------------------------------------------------------
// aggregation  out = act(A_hat h + bias)  (gcn_conv.py:92-104) with optional output statistics
// ------------------------------------------------------------------------------------------------
struct SpmmBranch {
    const float* h;
    float* out;
    const float* bias;     // null -> none
    const float* w;        // per-edge weight (edge-id order) or null (all ones)
    const float* dis;      // deg^-1/2 per node
    Acc st_sum;            // column statistics of the output (next BatchNorm) or off
    Acc st_sq;
    // SDDMM riding on the TRANSPOSED aggregation (k_espmm<.., SD = true>, by-source CSR: the gathered rows are dOut[dst]):
    // gn[eid] = <dOut[dst_e], z[src_e]>, gself[v] = <dOut[v], z[v]> -- the gradient w.r.t. the edge weights (gcn_conv.py:63-70,97)
    const float* sd_z;     // z = x' W of this branch (the forward aggregation's input), or null
    float* sd_gn;          // [E] by edge id
    float* sd_gself;       // [N]
    // add-pool + ReLU backward folded into the gathers (k_espmm<.., PB = true>): `h` is the branch's ACTIVATION relu(A_hat z + b),
    // and the row the aggregation uses for node v is dOut[v] = (h[v] > 0) * pb_g[pb_batch[v]] -- pb_g[b] the gradient of graph b's
    // pooled row (model.py:115-116,153-156; combined per graph by k_pool_bias_grad).  Every neighbour of a row lies in the row's
    // own graph (block-diagonal batch), so the pooled-row gradient is fetched once per row.
    const float* pb_g;     // [B, H] or null (plain feature rows)
    const int64_t* pb_batch;
};

struct SpmmBranch2 { SpmmBranch b[2]; };

// Round 4: one WAVE per row.  The row's CSR slots are fetched by ONE coalesced instruction (lane l <- slot base + l: neighbour
// id, then deg^-1/2 of the neighbour and the edge weight in the same lanes), and every gather address / coefficient is
// broadcast out of those registers (v_readlane -> SGPR base addresses when a wave's 64 lanes cover the row, G = 64; ds_bpermute
// when 64 / G lane groups each take another neighbour of the SAME row, partial sums added across the groups at the end): a row
// costs rowptr -> {nbr, eid} -> {dis, w} -> EXACTLY deg feature gathers issued back to back, eight (x 64 / G) in flight.  The
// per-lane form it replaces (four clamped-free neighbours at a time, then a serial remainder loop in which every neighbour was a
// dependent nbr -> {dis, h} chain) left the degree-2..3 rows of BA / molecule graphs on ~8 dependent round trips:
// scripts/micro/gather_lds.hip, profiles/r4/micro_gather_lds.txt: 128 -> 101 us at config 5 (32 BA graphs of 5000 nodes, H = 256).
// sum over the G lanes (8..64, a power of two) of a lane group; every lane of the group gets the total
template <int G>
__device__ __forceinline__ float group_dot_sum(float d) { return group_sum<G>(d); }     // (common.hpp: fused DPP steps + row totals by v_readlane)
// relu'(h) * g, elementwise: the row of d(conv output) that belongs to activation row h under pooled-row gradient g
__device__ __forceinline__ void pool_bwd_row(Vec<4>& h, const Vec<4>& g) {
    h.v.x = h.v.x > 0.f ? g.v.x : 0.f; h.v.y = h.v.y > 0.f ? g.v.y : 0.f;
    h.v.z = h.v.z > 0.f ? g.v.z : 0.f; h.v.w = h.v.w > 0.f ? g.v.w : 0.f;
}
template <int NB, int G, bool SD = false, bool PB = false>
__device__ __forceinline__ void espmm_batch(Vec<4>& acc, const float* __restrict__ h, int H, int jl, float cl, int q, int cnt, int gi, int c,
                                            const Vec<4>* zj = nullptr, float* gdot = nullptr, int lane = 0, const Vec<4>* gp = nullptr) {
    constexpr int SPLIT = 64 / G;
    Vec<4> v[NB];
    float cf[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        int j;
        if constexpr (SPLIT == 1) {
            j = __builtin_amdgcn_readlane(jl, q + u);
            cf[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cl), q + u));
        } else {
            const int slot = q + u * SPLIT + gi;             // this lane group's neighbour of the batch
            const int src = min(slot, cnt - 1) << 2;
            j = __builtin_amdgcn_ds_bpermute(src, jl);
            const float cc = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, cl)));
            cf[u] = slot < cnt ? cc : 0.f;
        }
        v[u] = Vec<4>::ld(h + (size_t)j * H + c);          // (unconditional: a guarded load is its own basic block behind a vmcnt(0))
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u].pin();
    if constexpr (PB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) pool_bwd_row(v[u], *gp);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) acc.fma(cf[u], v[u]);
    if constexpr (SD) {
        // <gathered row, own z row>: the G lanes' partial dots summed (DPP row sums, cross-row exchanges), parked in the
        // slot's lane; the lanes write gn[eid] once per 64 slots
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            float d = group_dot_sum<G>(v[u].dot(*zj));
            if constexpr (SPLIT == 1) *gdot = lane == q + u ? d : *gdot;
            else {
                // lane L holds slot L: in this batch that is element (L - q) / SPLIT of lane group (L - q) % SPLIT
                const int rel = lane - q;
                const float dv = __shfl(d, (rel & (SPLIT - 1)) * G, 64);
                *gdot = (rel >= 0 && rel / SPLIT == u) ? dv : *gdot;
            }
        }
    }
}

// WT: per-edge weights (the two causal branches; their load rides with deg^-1/2 of the neighbour), ST: column statistics of the
// output for the next BatchNorm (8 fp64 accumulators per lane: without them the kernel keeps 8 waves per SIMD)
template <int VEC, int G, bool WT, bool ST, bool SD = false, bool PB = false>
__global__ void __launch_bounds__(256) k_espmm(const CSR g, const SpmmBranch2 bb, int relu,
                                               float loop_w, int N, int H, int rows_per_block) {
    static_assert(VEC == 4 && G >= 8 && G <= 64, "16 B per lane, 8..64 lanes per row");
    __shared__ double lds[ST ? 2 : 1][ST ? 256 * 4 : 1];
    constexpr int SPLIT = 64 / G;
    warm_kernargs<sizeof(CSR) + sizeof(SpmmBranch2) + 32>();
    const SpmmBranch& br = bb.b[blockIdx.y];            // indexed in the kernel-argument segment: one set of scalar loads
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gi = lane / G, l = lane % G, c = l * VEC;
    // XCD-contiguous row blocks: workgroups are dealt to the 8 XCDs round-robin, and a row's neighbours live in its
    // own graph (block-diagonal batch).  Workgroup w therefore takes row block (w % 8) * (blocks / 8) + w / 8 -- each
    // XCD walks one contiguous eighth of the rows, so the ~5 gathers of every feature row hit ONE L2 instead of being
    // spread over eight (big batches: the gather volume E'*H*4 is 2.4x the algorithmic bytes and it was all fabric traffic)
    const int per = gridDim.x >> 3;
    const int bxr = (int)blockIdx.x < 8 * per ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int rbeg = bxr * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<VEC>;
    const bool cok = c < H;
    const int cld = cok ? c : 0;                       // lanes beyond the row width read column 0 and store nothing
    BLK_CLK(0);
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
    for (int iw = rbeg + wv; iw < rend; iw += 4) {
        const int i = __builtin_amdgcn_readfirstlane(iw);
        const int p0 = g.ptr[i], p1 = g.ptr[i + 1];
        const float di = br.dis[i];
        V hs = V::ld(br.h + (size_t)i * H + cld);            // the row's own features go out with the first round
        V gp = V::zero();
        if constexpr (PB) {
            gp = V::ld(br.pb_g + (size_t)br.pb_batch[i] * H + cld);
            pool_bwd_row(hs, gp);
        }
        V zj = V::zero();
        if constexpr (SD) { zj = V::ld(br.sd_z + (size_t)i * H + cld); if (!cok) zj = V::zero(); }
        V acc = V::zero();
        for (int base = p0; base < p1; base += 64) {
            const int s = min(base + lane, p1 - 1);
            int jl = g.nbr[s], el = (WT || SD) ? g.eid[s] : 0;
            float gdot = 0.f;
            asm volatile("" : "+v"(jl), "+v"(el));            // both ids requested before either is used
            float cl = br.dis[jl];
            if constexpr (WT) cl *= br.w[el];
            const int cnt = min(64, p1 - base);
            int q = 0;
            for (; q + 8 * SPLIT <= cnt; q += 8 * SPLIT) espmm_batch<8, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp);
            switch ((cnt - q + SPLIT - 1) / SPLIT) {          // (8 only when 64 / G > 1: 7 * SPLIT < cnt - q < 8 * SPLIT)
                case 8: espmm_batch<8, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 7: espmm_batch<7, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 6: espmm_batch<6, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 5: espmm_batch<5, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 4: espmm_batch<4, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 3: espmm_batch<3, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 2: espmm_batch<2, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                case 1: espmm_batch<1, G, SD, PB>(acc, br.h, H, jl, cl, q, cnt, gi, cld, &zj, &gdot, lane, &gp); break;
                default: break;
            }
            if constexpr (SD) { if (lane < cnt) br.sd_gn[el] = gdot; }
        }
        if constexpr (SD) {
            const float d = group_dot_sum<G>(cok ? hs.dot(zj) : 0.f);      // gself: the row's own loop
            if (lane == 0) br.sd_gself[i] = d;
        }
        if constexpr (SPLIT > 1) {
#pragma unroll
            for (int off = G; off < 64; off <<= 1) {
                acc.v.x += __shfl_xor(acc.v.x, off, 64); acc.v.y += __shfl_xor(acc.v.y, off, 64);
                acc.v.z += __shfl_xor(acc.v.z, off, 64); acc.v.w += __shfl_xor(acc.v.w, off, 64);
            }
        }
        acc.fma(di * loop_w, hs);
        acc.scale(di);
        if (br.bias) acc.add(V::ld(br.bias + cld));
        if (relu) acc.relu();
        if (gi == 0 && cok) {
            acc.st(br.out + (size_t)i * H + c);
            if constexpr (ST) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { double v = acc.get(j); s1[j] += v; s2[j] += v * v; }
            }
        }
    }
    if constexpr (ST) {
        // the block's column sums: the four waves through LDS, one lane per column (2 x 2 barriers)
        double* L = &lds[0][0];
        const int ncol = G * VEC;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (gi == 0) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) L[wv * ncol + l * VEC + j] = cok ? (pass ? s2[j] : s1[j]) : 0.0;
            }
            __syncthreads();
            if ((int)threadIdx.x < ncol && (int)threadIdx.x < H) {
                const int qq = threadIdx.x;
                const double t = (L[qq] + L[ncol + qq]) + (L[2 * ncol + qq] + L[3 * ncol + qq]);
                (pass ? br.st_sq : br.st_sum).add(qq, t);
            }
            __syncthreads();
        }
    }
    BLK_CLK(1);
}

// ------------------------------------------------------------------------------------------------
// node attention + edge projections (model.py:97-111):  a = softmax2(x Wn^T + bn),
// pq = (x.We[0,:H], x.We[1,:H], x.We[0,H:], x.We[1,H:]), and the batch statistics of
// xc = a0 x and xo = a1 x (inputs of bnc / bno) -- xc / xo themselves are never stored.
// ------------------------------------------------------------------------------------------------
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_node_att_fwd(const float* __restrict__ x, const float* __restrict__ Wn,
                                                      const float* __restrict__ bn, const float* __restrict__ We,
                                                      float* __restrict__ anode, float* __restrict__ pq,
                                                      const Acc stc_sum, const Acc stc_sq, const Acc sto_sum,
                                                      const Acc sto_sq, int N, int H, int rows_per_block, float fnode) {
    // fnode: 1, or 0 for without_node_attention (model.py:106-107): equal logits -> the constant 0.5 / 0.5 split
    __shared__ double lds[4 * 256 * (VEC == 4 ? 4 : 1)];
    constexpr int RPB = 256 / G;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<VEC>;
    // Fast path (one column chunk per lane): the six weight vectors are loaded once and stay in registers, the group's rows
    // come four at a time (all four loads up front) and serve both the projections and the statistics -- the generic code
    // below reloads the weights per row and reads anode / x back from memory for the statistics (config 5, 157 rows per
    // workgroup, ran there until round 4: 98 us for one pass over 164 MB).
    if (H <= G * VEC) {
        const int c = l * VEC, cc = min(c, H - VEC);
        const bool cok = c < H;
        V w[6], xv[4];
        w[0] = V::ld(Wn + cc); w[1] = V::ld(Wn + H + cc); w[2] = V::ld(We + cc);
        w[3] = V::ld(We + 2 * H + cc); w[4] = V::ld(We + H + cc); w[5] = V::ld(We + 3 * H + cc);
        const float b0 = bn[0], b1 = bn[1];
#pragma unroll
        for (int u = 0; u < 6; ++u) w[u].pin();
        double sc1[VEC], sc2[VEC], so1[VEC], so2[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { sc1[j] = sc2[j] = so1[j] = so2[j] = 0.0; }
        for (int rb = rbeg + grp; rb < rend; rb += 4 * RPB) {
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = V::ld(x + (size_t)min(rb + u * RPB, rend - 1) * H + cc);
#pragma unroll
        for (int u = 0; u < 4; ++u) { xv[u].pin(); if (!cok) xv[u] = V::zero(); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = rb + u * RPB;
            const float l0 = fnode * (group_sum<G>(xv[u].dot(w[0])) + b0), l1 = fnode * (group_sum<G>(xv[u].dot(w[1])) + b1);
            const float p0 = group_sum<G>(xv[u].dot(w[2])), p1 = group_sum<G>(xv[u].dot(w[3]));
            const float q0 = group_sum<G>(xv[u].dot(w[4])), q1 = group_sum<G>(xv[u].dot(w[5]));
            const float m = fmaxf(l0, l1);
            const float e0 = expf(l0 - m), e1 = expf(l1 - m);
            const float inv = 1.f / (e0 + e1), a0 = e0 * inv, a1 = e1 * inv;
            if (v < rend) {
                if (l == 0) {
                    anode[2 * (size_t)v] = a0;
                    anode[2 * (size_t)v + 1] = a1;
                    *reinterpret_cast<float4*>(pq + 4 * (size_t)v) = make_float4(p0, p1, q0, q1);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const double xc = (double)(a0 * xv[u].get(j)), xo = (double)(a1 * xv[u].get(j));
                    sc1[j] += xc; sc2[j] += xc * xc; so1[j] += xo; so2[j] += xo * xo;
                }
            }
        }
        }
        // the 4 VEC column sums of this lane through LDS in one go (G lanes x VEC columns = one slot per column)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double v4[4] = {sc1[j], sc2[j], so1[j], so2[j]};
            double o4[4];
            block_col_sums<4>(v4, l * VEC + j, grp, RPB, G * VEC, cok, lds, o4);
            if (grp == 0 && cok) { stc_sum.add(c + j, o4[0]); stc_sq.add(c + j, o4[1]); sto_sum.add(c + j, o4[2]); sto_sq.add(c + j, o4[3]); }
        }
        return;
    }
    // pass 1: logits / projections per row (full-row dot products)
    for (int v = rbeg + grp; v < rend; v += RPB) {
        float l0 = 0.f, l1 = 0.f, p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
        for (int c = l * VEC; c < H; c += G * VEC) {
            V xv = V::ld(x + (size_t)v * H + c);
            l0 += xv.dot(V::ld(Wn + c));
            l1 += xv.dot(V::ld(Wn + H + c));
            p0 += xv.dot(V::ld(We + c));
            p1 += xv.dot(V::ld(We + 2 * H + c));
            q0 += xv.dot(V::ld(We + H + c));
            q1 += xv.dot(V::ld(We + 3 * H + c));
        }
        l0 = fnode * (group_sum<G>(l0) + bn[0]); l1 = fnode * (group_sum<G>(l1) + bn[1]);
        p0 = group_sum<G>(p0); p1 = group_sum<G>(p1); q0 = group_sum<G>(q0); q1 = group_sum<G>(q1);
        if (l == 0) {
            float m = fmaxf(l0, l1);
            float e0 = expf(l0 - m), e1 = expf(l1 - m);
            float inv = 1.f / (e0 + e1);
            anode[2 * (size_t)v] = e0 * inv;
            anode[2 * (size_t)v + 1] = e1 * inv;
            *reinterpret_cast<float4*>(pq + 4 * (size_t)v) = make_float4(p0, p1, q0, q1);
        }
    }
    __syncthreads();   // anode of this block's rows is visible to the block (same-CU L1 write-through + barrier)
    // pass 2: statistics of a0*x and a1*x
    for (int c = l * VEC; c - l * VEC < H; c += G * VEC) {
        const bool cok = c < H;
        double sc1[VEC], sc2[VEC], so1[VEC], so2[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { sc1[j] = sc2[j] = so1[j] = so2[j] = 0.0; }
        if (cok)
            for (int v = rbeg + grp; v < rend; v += RPB) {
                // recompute a0/a1 from the logits written by lane 0 of this very group: same thread
                // group, ordered by the barrier above
                const float a0 = anode[2 * (size_t)v], a1 = anode[2 * (size_t)v + 1];
                V xv = V::ld(x + (size_t)v * H + c);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    double xc = (double)(a0 * xv.get(j)), xo = (double)(a1 * xv.get(j));
                    sc1[j] += xc; sc2[j] += xc * xc; so1[j] += xo; so2[j] += xo * xo;
                }
            }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            block_col_atomic(sc1[j], l * VEC + j, grp, RPB, G * VEC, cok, stc_sum, c + j, lds);
            block_col_atomic(sc2[j], l * VEC + j, grp, RPB, G * VEC, cok, stc_sq, c + j, lds);
            block_col_atomic(so1[j], l * VEC + j, grp, RPB, G * VEC, cok, sto_sum, c + j, lds);
            block_col_atomic(so2[j], l * VEC + j, grp, RPB, G * VEC, cok, sto_sq, c + j, lds);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// edge softmax (model.py:102-104) + weighted degrees of both branches (gcn_conv.py:63-68).
// 8 lanes per source node walk its out-edges; att[0,E] = edge_weight_c, att[1,E] = edge_weight_o.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edge_att_deg(const CSR gs, const float* __restrict__ pq, const float* __restrict__ be,
                                                      float* __restrict__ att, float* __restrict__ dis_c,
                                                      float* __restrict__ dis_o, float loop_w, int N, int64_t E, float fedge) {
    // fedge: 1, or 0 for without_edge_attention (model.py:99-100): both edge weights are the constant 0.5
    const int v = blockIdx.x * 32 + threadIdx.x / 8, l = threadIdx.x % 8;
    if (v >= N) return;
    const float4 pv = *reinterpret_cast<const float4*>(pq + 4 * (size_t)v);
    const float b0 = be[0], b1 = be[1];
    float dc = 0.f, dq = 0.f;
    for (int s = gs.ptr[v] + l; s < gs.ptr[v + 1]; s += 8) {
        const int d = gs.nbr[s], e = gs.eid[s];
        const float4 qd = *reinterpret_cast<const float4*>(pq + 4 * (size_t)d);
        const float l0 = fedge * (pv.x + qd.z + b0), l1 = fedge * (pv.y + qd.w + b1);
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.f / (e0 + e1);
        const float a0 = e0 * inv, a1 = e1 * inv;
        att[e] = a0;
        att[E + e] = a1;
        dc += a0; dq += a1;
    }
    dc = group_sum<8>(dc); dq = group_sum<8>(dq);
    if (l == 0) {
        dc += loop_w; dq += loop_w;
        dis_c[v] = dc == 0.f ? 0.f : 1.0f / sqrtf(dc);
        dis_o[v] = dq == 0.f ? 0.f : 1.0f / sqrtf(dq);
    }
}

// ------------------------------------------------------------------------------------------------
// global_add_pool for both branches (model.py:115-116): grid (B, 2)
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_pool2(const float* __restrict__ hc, const float* __restrict__ ho,
                                               const int* __restrict__ gptr, float* __restrict__ pc,
                                               float* __restrict__ po, int H, int tc, float* __restrict__ slices,
                                               float* __restrict__ cnt) {
    __shared__ float lds[2][256 * 4];
    const float* h = blockIdx.y ? ho : hc;
    const int b = blockIdx.x;
    // gridDim.z > 1 (graphs of thousands of nodes, few graphs): row slice z of the graph goes to slices[z][branch][b][:]
    // and k_pool2_sum adds the slices in a fixed order -- B x 2 workgroups alone read config 5's 328 MB at 1 TB/s.
    // cnt (training steps of the node-level path): how many rows of the graph are POSITIVE per column -- the ReLU mask the
    // backward needs only as this count (the bias gradient is count x pooled-row gradient, k_pool_bias_grad) and per gathered
    // row (k_espmm<.., PB>); [2, B, H] after the pooled rows in both layouts.
    const size_t BH = (size_t)gridDim.x * H;
    float* out = gridDim.z > 1 ? slices + ((size_t)blockIdx.z * 4 + blockIdx.y) * BH : (blockIdx.y ? po : pc);
    float* outc = gridDim.z > 1 ? slices + ((size_t)blockIdx.z * 4 + 2 + blockIdx.y) * BH : (cnt ? cnt + blockIdx.y * BH : nullptr);
    const int nrl = 256 / tc, cl = threadIdx.x % tc, rl = threadIdx.x / tc;
    int n0 = gptr[b], n1 = gptr[b + 1];
    if (gridDim.z > 1) {
        const int len = (n1 - n0 + gridDim.z - 1) / gridDim.z;
        n0 = min(n1, n0 + (int)blockIdx.z * len);
        n1 = min(n1, n0 + len);
    }
    using V = Vec<VEC>;
    for (int c = cl * VEC; c - cl * VEC < H; c += tc * VEC) {
        const bool cok = c < H;
        V a = V::zero(), a2 = V::zero();
        float pos[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) pos[j] = 0.f;
        if (cok) {
            int r = n0 + rl;
            for (; r + nrl < n1; r += 2 * nrl) {
                const V x0 = V::ld(h + (size_t)r * H + c), x1 = V::ld(h + (size_t)(r + nrl) * H + c);
                a.add(x0); a2.add(x1);
#pragma unroll
                for (int j = 0; j < VEC; ++j) pos[j] += (x0.get(j) > 0.f ? 1.f : 0.f) + (x1.get(j) > 0.f ? 1.f : 0.f);
            }
            if (r < n1) {
                const V x0 = V::ld(h + (size_t)r * H + c);
                a.add(x0);
#pragma unroll
                for (int j = 0; j < VEC; ++j) pos[j] += x0.get(j) > 0.f ? 1.f : 0.f;
            }
            a.add(a2);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) { lds[0][(rl * tc + cl) * VEC + j] = a.get(j); lds[1][(rl * tc + cl) * VEC + j] = pos[j]; }
        __syncthreads();
        if (rl == 0 && cok) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float t = 0.f, n = 0.f;
                for (int k = 0; k < nrl; ++k) { t += lds[0][(k * tc + cl) * VEC + j]; n += lds[1][(k * tc + cl) * VEC + j]; }
                out[(size_t)b * H + c + j] = t;
                if (outc) outc[(size_t)b * H + c + j] = n;
            }
        }
        __syncthreads();
    }
}

// {pooled, cnt}[branch][b][:] = sum over the S row slices written by k_pool2 (a slice: [4][B][H] = two pooled branches, two counts;
// n = 2 * B * H floats per half)
__global__ void k_pool2_sum(const float* __restrict__ slices, int S, int64_t n, float* __restrict__ pooled, float* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n) return;
    float s = 0.f;
#pragma unroll 8
    for (int z = 0; z < S; ++z) s += slices[(size_t)z * 2 * n + i];
    if (i < n) pooled[i] = s;
    else if (cnt) cnt[i - n] = s;
}

// The positive counts alone, from the node -> graph map (any batch layout: packed tiles, graphs without nodes): for a batch whose
// FORWARD pooled inside the per-graph convolution kernel while its backward runs node-level (65..128-node graphs, > 512 units,
// edge lists beyond the per-graph kernels' capacity).  Integer-valued float atomics: exact, so order-independent.
// cnt [2, B, H] zeroed by the caller; grid (row blocks, 2)
template <int VEC, int G>
__global__ void __launch_bounds__(256) k_pool_cnt(const float* __restrict__ hc, const float* __restrict__ ho, const int64_t* __restrict__ batch,
                                                  float* __restrict__ cnt, int N, int B, int H, int rows_per_block) {
    constexpr int RPB = 256 / G;
    const float* h = blockIdx.y ? ho : hc;
    float* out = cnt + (size_t)blockIdx.y * B * H;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<VEC>;
    for (int c = l * VEC; c < H; c += G * VEC) {
        float n[VEC];
        int cur = -1;
        auto flush = [&]() {
            if (cur < 0) return;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (n[j] != 0.f) atomicAdd(out + (size_t)cur * H + c + j, n[j]);
        };
        for (int v = rbeg + grp; v < rend; v += RPB) {
            const int b = (int)batch[v];
            if (b != cur) {
                flush();
                cur = b;
#pragma unroll
                for (int j = 0; j < VEC; ++j) n[j] = 0.f;
            }
            const V hv = V::ld(h + (size_t)v * H + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) n[j] += hv.get(j) > 0.f ? 1.f : 0.f;
        }
        flush();
    }
}
__global__ void k_zero_f32(float* __restrict__ a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 0.f;
}

// Node-level backward of the add-pool: per graph, (1) the gradient of its two pooled rows, combined from the readout-input
// gradients when the readout backward left them apart (model.py:153-156: d pooled_c[b] = dxin_c[b] + dxin_co[iperm[b]],
// d pooled_o[b] = dxin_o[b] + dxin_co[b]) -> g [2, B, H], the rows k_espmm<.., PB> masks per gathered activation row; (2) the
// bias gradient of the two causal convs from the positive counts of k_pool2: d b_k = sum_v relu'(h_k[v]) * g_k[batch[v]] =
// sum_b cnt_k[b] * g_k[b] (exact in fp64: every term of the left sum is the same fp32 number).  One partial row per graph,
// summed by k_finish.  grid (B, 2)
__global__ void __launch_bounds__(256) k_pool_bias_grad(const float* __restrict__ cnt, const float* __restrict__ g0, const float* __restrict__ g1,
                                                        const int* __restrict__ iperm, float* __restrict__ g, double* __restrict__ parts_c,
                                                        double* __restrict__ parts_o, int B, int H) {
    const int b = blockIdx.x, k = blockIdx.y;
    const size_t BH = (size_t)B * H;
    double* out = (k ? parts_o : parts_c) + (size_t)b * H;
    for (int c = threadIdx.x; c < H; c += 256) {
        float v = g0[k * BH + (size_t)b * H + c];
        if (g1) {
            v += g1[(size_t)(k == 0 ? iperm[b] : b) * H + c];
            g[k * BH + (size_t)b * H + c] = v;
        }
        out[c] = (double)cnt[k * BH + (size_t)b * H + c] * (double)v;
    }
}

// ------------------------------------------------------------------------------------------------
// readout inputs (model.py:145-156): x_co = xc[perm] + xo, inverse permutation, and the batch
// statistics of the three readout inputs (fc1_bn_c / _o / _co).  pooled: [2,B,H]; xin: [3,B,H].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_readout_prep(const float* __restrict__ pooled, const int64_t* __restrict__ perm,
                                                      int* __restrict__ iperm, float* __restrict__ xco, int B, int H,
                                                      int tc, int rows_per_block, const Acc s_c, const Acc q_c,
                                                      const Acc s_o, const Acc q_o, const Acc s_co, const Acc q_co, int cat) {
    // cat (model.py:153-154): x_co = [xc[perm] | xo], [B, 2H]; its batch statistics are those of xc[perm] (columns < H)
    // and of xo (columns >= H)
    __shared__ double lds[256];
    const int nrl = 256 / tc, cl = threadIdx.x % tc, rl = threadIdx.x / tc;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(B, r0 + rows_per_block);
    const float* pc = pooled;
    const float* po = pooled + (size_t)B * H;
    const int W = cat ? 2 * H : H;
    if (cl == 0)
        for (int r = r0 + rl; r < r1; r += nrl) iperm[perm[r]] = r;
    for (int c = cl; c - cl < H; c += tc) {
        const bool cok = c < H;
        double a1 = 0, a2 = 0, b1 = 0, b2 = 0, c1 = 0, c2 = 0;
        // eight rows per pass: their perm entries in one round of loads, then the 3 x 8 row values in a second one (as a loop
        // of "perm[r], then the gathered row" per row this kernel was 2 dependent round trips per row: 14.8 us at B = 512)
        if (cok)
            for (int rb = r0 + rl; rb < r1; rb += 8 * nrl) {
                int pr[8];
                float vc[8], vo[8], vcp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) pr[u] = (int)perm[min(rb + u * nrl, r1 - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = min(rb + u * nrl, r1 - 1);
                    vc[u] = pc[(size_t)r * H + c]; vo[u] = po[(size_t)r * H + c];
                    vcp[u] = pc[(size_t)pr[u] * H + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + u * nrl;
                    if (r < r1) {
                        const float vco = cat ? vcp[u] : vcp[u] + vo[u];
                        xco[(size_t)r * W + c] = vco;
                        if (cat) xco[(size_t)r * W + H + c] = vo[u];
                        a1 += vc[u]; a2 += (double)vc[u] * vc[u]; b1 += vo[u]; b2 += (double)vo[u] * vo[u];
                        c1 += vco; c2 += (double)vco * vco;
                    }
                }
            }
        block_col_atomic(a1, cl, rl, nrl, tc, cok, s_c, c, lds);
        block_col_atomic(a2, cl, rl, nrl, tc, cok, q_c, c, lds);
        block_col_atomic(b1, cl, rl, nrl, tc, cok, s_o, c, lds);
        block_col_atomic(b2, cl, rl, nrl, tc, cok, q_o, c, lds);
        block_col_atomic(c1, cl, rl, nrl, tc, cok, s_co, c, lds);
        block_col_atomic(c2, cl, rl, nrl, tc, cok, q_co, c, lds);
        if (cat) {
            block_col_atomic(b1, cl, rl, nrl, tc, cok, s_co, H + c, lds);
            block_col_atomic(b2, cl, rl, nrl, tc, cok, q_co, H + c, lds);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// log_softmax + 3-term loss (train_causal.py:176-183) + gradient w.r.t. the pre-softmax scores.
// One workgroup (NT = 256 threads, 1024 for B > 256: 19.7 -> us at B = 512).  z, logp, dz: [3,B,C] (heads c, o, co).  stats: [loss, c, o, co, correct_o].
// db2[h*C + k] = column sums of dz (fc2 bias gradients), written to the fp64 arena.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(NT) k_loss(const float* __restrict__ z, const int64_t* __restrict__ y,
                                              float* __restrict__ logp, float* __restrict__ dz,
                                              float* __restrict__ stats, double* __restrict__ db2, int B, int C,
                                              float wc, float wo, float wco, int want_grad) {
    __shared__ double red[NT];
    __shared__ double tot[6];
    const float w[3] = {wc, wo, wco};
    double part[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // losses c, o, co; hits o, c, co
    const float u = 1.0f / (float)C, invB = 1.0f / (float)B;
    for (int t = threadIdx.x; t < 3 * B; t += NT) {
        const int hd = t / B, b = t % B;
        const float* zr = z + (size_t)t * C;
        float m = -INFINITY;
        for (int k = 0; k < C; ++k) m = fmaxf(m, zr[k]);
        float se = 0.f;
        for (int k = 0; k < C; ++k) se += expf(zr[k] - m);
        const float lse = m + logf(se);
        const int yy = (int)y[b];
        int arg = 0; float best = -INFINITY;
        for (int k = 0; k < C; ++k) {
            const float lp = zr[k] - lse;
            logp[(size_t)t * C + k] = lp;
            if (lp > best) { best = lp; arg = k; }
            if (hd == 0) part[0] += (double)(u * (logf(u) - lp));           // KL(c || uniform), batchmean
            if (want_grad) {
                const float p = expf(lp);
                const float g = hd == 0 ? (p - u) : (p - (k == yy ? 1.f : 0.f));
                dz[(size_t)t * C + k] = w[hd] * invB * g;
            }
        }
        if (hd == 1) { part[1] += (double)(-(zr[yy] - lse)); part[3] += (arg == yy) ? 1.0 : 0.0; }
        if (hd == 2) { part[2] += (double)(-(zr[yy] - lse)); part[5] += (arg == yy) ? 1.0 : 0.0; }
        if (hd == 0) part[4] += (arg == yy) ? 1.0 : 0.0;
    }
    // the six sums in two levels behind two barriers (six trees of log2(NT) barrier-separated stages were ~8 us of this kernel
    // at NT = 1024): wave sums by DPP-free LDS strips -- thread (q, j) adds the 64 values of statistic q that wave j parked
    constexpr int NW = NT / 64;
    static_assert(6 * NW <= NT, "one thread per (statistic, wave)");
    {
        __shared__ double park[6][NT + 8];
        __shared__ double wsum[6][NW];
#pragma unroll
        for (int q = 0; q < 6; ++q) park[q][threadIdx.x] = part[q];
        __syncthreads();
        if (threadIdx.x < 6 * NW) {
            const int q = threadIdx.x / NW, j = threadIdx.x % NW;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                a0 += park[q][j * 64 + k]; a1 += park[q][j * 64 + k + 1];
                a2 += park[q][j * 64 + k + 2]; a3 += park[q][j * 64 + k + 3];
            }
            wsum[q][j] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NW; ++j) a += wsum[threadIdx.x][j];
            tot[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float lc = (float)(tot[0] * invB), lo = (float)(tot[1] * invB), lco = (float)(tot[2] * invB);
        stats[0] = wc * lc + wo * lo + wco * lco;
        stats[1] = lc; stats[2] = lo; stats[3] = lco; stats[4] = (float)tot[3]; stats[5] = (float)tot[4]; stats[6] = (float)tot[5];
    }
    if (want_grad) {
        __syncthreads();
        // column sums of dz: (3C columns) x (NT / 3C row lanes), then a serial LDS tail per column
        const int nc = 3 * C, nl = NT / nc;
        const int cidx = threadIdx.x % nc, pl = threadIdx.x / nc;
        double sacc = 0.0;
        if (pl < nl) {
            const int hd = cidx / C, k = cidx % C;
            for (int b = pl; b < B; b += nl) sacc += (double)dz[((size_t)hd * B + b) * C + k];
        }
        red[threadIdx.x] = sacc;
        __syncthreads();
        if (threadIdx.x < nc) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;      // (four chains: the serial tail was nl dependent fp64 adds)
            int q = 0;
            for (; q + 3 < nl; q += 4) {
                t0 += red[q * nc + threadIdx.x]; t1 += red[(q + 1) * nc + threadIdx.x];
                t2 += red[(q + 2) * nc + threadIdx.x]; t3 += red[(q + 3) * nc + threadIdx.x];
            }
            for (; q < nl; ++q) t0 += red[q * nc + threadIdx.x];
            db2[threadIdx.x] = (t0 + t1) + (t2 + t3);
        }
    }
}

// log_softmax backward for an EXTERNAL loss (the nn.Module / autograd surface): given d loss / d logp
// for the three heads, dz = g - exp(logp) * rowsum(g), plus the fc2 bias gradients.  One workgroup.
__global__ void __launch_bounds__(256) k_logsoftmax_bwd(const float* __restrict__ logp, const float* __restrict__ g,
                                                        float* __restrict__ dz, double* __restrict__ db2, int B, int C) {
    __shared__ double red[256];
    for (int t = threadIdx.x; t < 3 * B; t += 256) {
        float rs = 0.f;
        for (int k = 0; k < C; ++k) rs += g[(size_t)t * C + k];
        for (int k = 0; k < C; ++k) dz[(size_t)t * C + k] = g[(size_t)t * C + k] - expf(logp[(size_t)t * C + k]) * rs;
    }
    __syncthreads();
    const int nc = 3 * C, nl = 256 / nc;
    const int cidx = threadIdx.x % nc, pl = threadIdx.x / nc;
    double sacc = 0.0;
    if (pl < nl) {
        const int hd = cidx / C, k = cidx % C;
        for (int b = pl; b < B; b += nl) sacc += (double)dz[((size_t)hd * B + b) * C + k];
    }
    red[threadIdx.x] = sacc;
    __syncthreads();
    if (threadIdx.x < nc) {
        double tsum = 0.0;
        for (int q = 0; q < nl; ++q) tsum += red[q * nc + threadIdx.x];
        db2[threadIdx.x] = tsum;
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward (elementwise part) + ReLU mask of the producer + bias-gradient column sums.
//   dy = gamma*rstd*(dyh - m1 - x_n*m2) * [relu ? (x > 0) : 1],  x_n = (x - mean)*rstd,
//   m1 = dot_sum/n, m2 = dot_prod/n (accumulated by the GEMM that produced dyh).
// Batched over blockIdx.y (readout heads).  x is the BN input (= previous post-ReLU activation).
// ------------------------------------------------------------------------------------------------
struct BnBwdProb {
    const float* dyh;
    const float* x;
    float* dy;
    BNRef bn;
    const double* dot_sum;
    const double* dot_prod;
    Acc colsum;          // column sums of dy (bias gradient of the producing layer) or off
    const float* dyh2;   // second partial of dyh (per-graph fused backward: one per output-column slice) or null
};

struct BnBwdProb3 { BnBwdProb p[3]; };

// ST: striped reader (engine.hpp: the statistics / backward sums may sit in the site's NSTRIPE accumulator planes -- small batches,
// whose GEMM epilogues add into them instead of leaving partial rows for k_stats_final)
template <int VEC, int G, bool ST = false>
__global__ void __launch_bounds__(256) k_bn_bwd(const BnBwdProb3 pp, int relu,
                                                int N, int W, int rows_per_block) {
    __shared__ double lds[256 * (VEC == 4 ? 4 : 1)];
    __shared__ float ck_s[ST ? 4 : 1][ST ? 256 : 1];
    constexpr int RPB = 256 / G;
    warm_kernargs<sizeof(BnBwdProb3) + 32>();
    const BnBwdProb& p = pp.p[blockIdx.y];              // indexed in the kernel-argument segment (see k_gconv_fwd)
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<VEC>;
    // ST with one column chunk per lane (W <= 256): thread t adds the planes of column t's four sums (16 loads in one round, no
    // arithmetic between them) and the column constants reach the lanes through LDS -- per lane and VEC column that is 64 loads
    // with the adds of one column in front of the next column's loads (7.4 vs 5.5 us at 3.8 k rows)
    const bool st_lds = ST && W <= 256 && !p.bn.use_running;
    if (st_lds) {
        const int oc = min((int)threadIdx.x, W - 1);
        StripeVal q0 = stripe_load(p.bn.sum, oc, p.bn.ss), q1 = stripe_load(p.bn.sq, oc, p.bn.ss);
        StripeVal q2 = stripe_load(p.dot_sum, oc, p.bn.ss), q3 = stripe_load(p.dot_prod, oc, p.bn.ss);
        stripe_pin(q0); stripe_pin(q1); stripe_pin(q2); stripe_pin(q3);
        const double inv = (double)p.bn.inv_n;
        const double m = stripe_total(q0, p.bn.ss) * inv, v = stripe_total(q1, p.bn.ss) * inv - m * m;
        ck_s[0][threadIdx.x] = (float)m;
        ck_s[1][threadIdx.x] = 1.0f / sqrtf((float)(v > 0.0 ? v : 0.0) + p.bn.eps);
        ck_s[2][threadIdx.x] = (float)(stripe_total(q2, p.bn.ss) * inv);
        ck_s[3][threadIdx.x] = (float)(stripe_total(q3, p.bn.ss) * inv);
        __syncthreads();
    }
    for (int c = l * VEC; c - l * VEC < W; c += G * VEC) {
        const bool cok = c < W;
        float mean[VEC], rstd[VEC], gs[VEC], m1[VEC], m2[VEC];
        double cs[VEC];
        if (st_lds) {
            const int cb = min(c, W - VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                cs[j] = 0.0;
                mean[j] = ck_s[0][cb + j]; rstd[j] = ck_s[1][cb + j]; m1[j] = ck_s[2][cb + j]; m2[j] = ck_s[3][cb + j];
                gs[j] = (p.bn.gamma ? p.bn.gamma[cb + j] : 1.f) * rstd[j];
            }
        } else
        {   // per-column constants, all loads up front on a clamped column (W % VEC == 0)
            const int cb = min(c, W - VEC);
            double ds[VEC], dp[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                cs[j] = 0.0;
                gs[j] = p.bn.gamma ? p.bn.gamma[cb + j] : 1.f;
                ds[j] = ST ? stripe_sum(p.dot_sum, cb + j, p.bn.ss) : p.dot_sum[cb + j];
                dp[j] = ST ? stripe_sum(p.dot_prod, cb + j, p.bn.ss) : p.dot_prod[cb + j];
            }
            bn_mean_rstd_v<VEC, ST>(p.bn, cb, mean, rstd);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                gs[j] *= rstd[j];
                m1[j] = (float)(ds[j] * (double)p.bn.inv_n);
                m2[j] = (float)(dp[j] * (double)p.bn.inv_n);
            }
        }
        // UR rows per pass with all their loads issued first: one row per iteration (load, wait, store --
        // the store may alias the next row's loads as far as hipcc knows) is one memory round trip per row
        constexpr int UR = 4;
        const int cc = min(c, W - VEC);
        for (int r0 = rbeg + grp; r0 < rend; r0 += RPB * UR) {
            V d[UR], xv[UR], d2[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const size_t r = (size_t)min(r0 + u * RPB, rend - 1);
                d[u] = V::ld(p.dyh + r * W + cc); xv[u] = V::ld(p.x + r * W + cc);
                d2[u] = p.dyh2 ? V::ld(p.dyh2 + r * W + cc) : V::zero();
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) { d[u].pin(); xv[u].pin(); d2[u].pin(); d[u].add(d2[u]); }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * RPB;
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xn = (xv[u].get(j) - mean[j]) * rstd[j];
                    float t = gs[j] * (d[u].get(j) - m1[j] - xn * m2[j]);
                    if (relu && !(xv[u].get(j) > 0.f)) t = 0.f;
                    o[j] = t;
                }
                if (r < rend && cok) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) cs[j] += (double)o[j];
                    V ov;
                    if constexpr (VEC == 4) ov.v = make_float4(o[0], o[1], o[2], o[3]); else ov.v = o[0];
                    ov.st(p.dy + (size_t)r * W + c);
                }
            }
        }
        if (p.colsum.on()) {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                block_col_atomic(cs[j], l * VEC + j, grp, RPB, G * VEC, cok, p.colsum, c + j, lds);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// readout tail of the backward: BN-backward of the three fc1_bn inputs and the scatter of the
// random-intervention sum back to the two pooled vectors (model.py:153-156):
//   dpool_c[j] = dxc[j] + dxco[iperm[j]],   dpool_o[b] = dxo[b] + dxco[b].
// ------------------------------------------------------------------------------------------------
struct BnIn {
    const float* dyh;   // [B,H]
    const float* x;     // [B,H]
    BNRef bn;
    const double* dot_sum;
    const double* dot_prod;
};
__device__ __forceinline__ float bn_bwd_elem(const BnIn& p, int r, int c, int H) {      // H = row stride of dyh / x
    float mean, rstd;
    bn_mean_rstd(p.bn, c, mean, rstd);
    const float gs = (p.bn.gamma ? p.bn.gamma[c] : 1.f) * rstd;
    const float m1 = (float)(p.dot_sum[c] * (double)p.bn.inv_n), m2 = (float)(p.dot_prod[c] * (double)p.bn.inv_n);
    const float xn = (p.x[(size_t)r * H + c] - mean) * rstd;
    return gs * (p.dyh[(size_t)r * H + c] - m1 - xn * m2);
}
__global__ void __launch_bounds__(256) k_readout_bwd_tail(const BnIn hc, const BnIn ho, const BnIn hco,
                                                          const int* __restrict__ iperm, float* __restrict__ dpool,
                                                          int B, int H, int cat) {
    // cat: the co input is [xc[perm] | xo] ([B, 2H]): its first H columns flow back to xc (un-permuted), the last H to xo
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * H) return;
    const int r = (int)(t / H), c = (int)(t % H);
    const int W = cat ? 2 * H : H;
    dpool[t] = bn_bwd_elem(hc, r, c, H) + bn_bwd_elem(hco, iperm[r], c, W);
    dpool[(size_t)B * H + t] = bn_bwd_elem(ho, r, c, H) + bn_bwd_elem(hco, r, cat ? H + c : c, W);
}

// d deg for both branches (gcn_conv.py:63-70 differentiated); 8 lanes per node
__global__ void __launch_bounds__(256) k_normbwd_node2(const CSR gs, const CSR gd, const float* __restrict__ att,
                                                       const float* __restrict__ dis, const float* __restrict__ gn,
                                                       const float* __restrict__ gself, float* __restrict__ ddeg,
                                                       float loop_w, int N, int64_t E, const float* __restrict__ gn2,
                                                       const float* __restrict__ gself2) {
    // gn2 / gself2: second partial of gn / gself (one per 64-column slice of the per-graph fused backward) or null
    const int brn = blockIdx.y;
    const float* w = att + (size_t)brn * E;
    const float* di = dis + (size_t)brn * N;
    const float* g = gn + (size_t)brn * E;
    const float* g2 = gn2 ? gn2 + (size_t)brn * E : nullptr;
    const int v = blockIdx.x * 32 + threadIdx.x / 8, l = threadIdx.x % 8;
    if (v >= N) return;
    float acc = 0.f;
    for (int s = gs.ptr[v] + l; s < gs.ptr[v + 1]; s += 8) { const int e = gs.eid[s]; acc += (g[e] + (g2 ? g2[e] : 0.f)) * w[e] * di[gs.nbr[s]]; }
    for (int s = gd.ptr[v] + l; s < gd.ptr[v + 1]; s += 8) { const int e = gd.eid[s]; acc += (g[e] + (g2 ? g2[e] : 0.f)) * w[e] * di[gd.nbr[s]]; }
    acc = group_sum<8>(acc);
    if (l == 0) {
        const float d = di[v];
        acc += 2.f * (gself[(size_t)brn * N + v] + (gself2 ? gself2[(size_t)brn * N + v] : 0.f)) * d * loop_w;
        ddeg[(size_t)brn * N + v] = -0.5f * d * d * d * acc;
    }
}

// d edge_weight of both branches -> d(edge logit 0): dl[e] = a0 a1 (dw_c - dw_o)  (softmax2 backward)
__global__ void k_normbwd_edge(const int* __restrict__ row32, const int* __restrict__ col32, const float* __restrict__ att,
                               const float* __restrict__ dis, const float* __restrict__ gn, const float* __restrict__ ddeg,
                               float* __restrict__ dl, int N, int64_t E, const float* __restrict__ gn2, float fedge) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int r = row32[e], c = col32[e];
    if (r == c) { dl[e] = 0.f; return; }
    const float gc = gn[e] + (gn2 ? gn2[e] : 0.f), go = gn[E + e] + (gn2 ? gn2[E + e] : 0.f);
    const float dwc = gc * dis[r] * dis[c] + ddeg[r];
    const float dwo = go * dis[(size_t)N + r] * dis[(size_t)N + c] + ddeg[(size_t)N + r];
    dl[e] = fedge * att[e] * att[E + e] * (dwc - dwo);       // fedge = 0: constant edge masks, nothing flows into the edge MLP
}

// ------------------------------------------------------------------------------------------------
// backward through bnc/bno, the node attention split, the node/edge attention MLPs and the ReLU of
// the last backbone layer, in one pass over the nodes (model.py:97-113 differentiated):
//   dxc = BNbwd_c(dXc_hat), dxo = BNbwd_o(dXo_hat)            (inputs were a0*x, a1*x)
//   dl0 = a0 a1 (<dxc,x> - <dxo,x>)                            (node softmax2 backward)
//   sp = sum_{row_e=v} dl[e], sq = sum_{col_e=v} dl[e]         (edge projections backward)
//   dx = a0 dxc + a1 dxo + dl0 (Wn0-Wn1) + sp (We0[:H]-We1[:H]) + sq (We0[H:]-We1[H:])
//   dZ = dx * (x > 0)                                           (x = relu output of the last conv)
// and the column sums for d bias_L, d Wn, d bn, d We, d be.
// ------------------------------------------------------------------------------------------------
struct AttBwdArgs {
    const float* x; const float* anode; const float* dxhc; const float* dxho;
    BNRef bnc, bno;
    const double *dsc, *dpc, *dso, *dpo;    // dot sums of bnc / bno (plane 0 of the NSTRIPE planes dss doubles apart: k_att_bwd_graph adds them)
    int dss;
    const float* Wn; const float* We; const float* dl;
    CSR gs, gd;
    float* dZ;
    Acc dbias;         // [H] bias gradient of the last backbone conv (off when there is none)
    Acc dWn;           // [H] (+1: d bn0 at [H])
    Acc dWe;           // [2H] (+1: d be0 at [2H])
    const float* dxhc2; const float* dxho2;   // second partials of dxhc / dxho (per-graph fused backward) or null
    float fnode, fedge;                       // 0 for without_node_attention / without_edge_attention (constant masks), else 1
};

// ST: striped reader of the four backward sums (a.dss: they may sit in the sites' NSTRIPE accumulator planes, engine.hpp) -- thread t
// adds the planes of column t, the per-column means reach the lanes through LDS
template <int VEC, int G, bool ST = false>
__global__ void __launch_bounds__(256) k_att_bwd(const AttBwdArgs a, int relu, int N, int H, int rows_per_block) {
    warm_kernargs<sizeof(AttBwdArgs) + 16>();
    __shared__ double lds[4 * 256 * (VEC == 4 ? 4 : 1)];
    __shared__ float mk_s[ST ? 4 : 1][ST ? 256 : 1];
    __shared__ double sc_lds[2][256 / G];
    constexpr int RPB = 256 / G;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(N, rbeg + rows_per_block);
    using V = Vec<VEC>;
    const double inv_n = a.bnc.inv_n;
    double sdl = 0.0, ssp = 0.0;
    // the kernel supports one column chunk per lane (H <= G*VEC), enforced by the host
    const int c = l * VEC;
    const bool cok = c < H;
    float mc[VEC], rc[VEC], gc[VEC], m1c[VEC], m2c[VEC], mo[VEC], ro[VEC], go[VEC], m1o[VEC], m2o[VEC];
    float wn[VEC], wp[VEC], wq[VEC];
    double cs_b[VEC], cs_n[VEC], cs_p[VEC], cs_q[VEC];
    {   // per-column constants: every load issued up front on a clamped column (H % VEC == 0); per column and
        // under `if (cok)` this prologue was ~20 dependent load groups
        const int cb = min(c, H - VEC);
        double d1c[VEC], d2c[VEC], d1o[VEC], d2o[VEC];
        float w0[VEC], w1[VEC], w2[VEC], w3[VEC], w4[VEC], w5[VEC];
        StripeVal sq[4];
        const int oc = min((int)threadIdx.x, H - 1);
        if constexpr (ST) {
            sq[0] = stripe_load(a.dsc, oc, a.dss); sq[1] = stripe_load(a.dpc, oc, a.dss);
            sq[2] = stripe_load(a.dso, oc, a.dss); sq[3] = stripe_load(a.dpo, oc, a.dss);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            cs_b[j] = cs_n[j] = cs_p[j] = cs_q[j] = 0.0;
            gc[j] = a.bnc.gamma ? a.bnc.gamma[cb + j] : 1.f;
            go[j] = a.bno.gamma ? a.bno.gamma[cb + j] : 1.f;
            if constexpr (!ST) { d1c[j] = a.dsc[cb + j]; d2c[j] = a.dpc[cb + j]; d1o[j] = a.dso[cb + j]; d2o[j] = a.dpo[cb + j]; }
            w0[j] = a.Wn[cb + j]; w1[j] = a.Wn[H + cb + j];
            w2[j] = a.We[cb + j]; w3[j] = a.We[2 * H + cb + j]; w4[j] = a.We[H + cb + j]; w5[j] = a.We[3 * H + cb + j];
        }
        bn_mean_rstd_v<VEC>(a.bnc, cb, mc, rc);
        bn_mean_rstd_v<VEC>(a.bno, cb, mo, ro);
        if constexpr (ST) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { stripe_pin(sq[q]); mk_s[q][threadIdx.x] = (float)(stripe_total(sq[q], a.dss) * (double)inv_n); }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool on = cok && c + j < H;
            gc[j] = on ? gc[j] * rc[j] : 0.f; go[j] = on ? go[j] * ro[j] : 0.f;
            if constexpr (ST) { m1c[j] = mk_s[0][cb + j]; m2c[j] = mk_s[1][cb + j]; m1o[j] = mk_s[2][cb + j]; m2o[j] = mk_s[3][cb + j]; }
            else {
            m1c[j] = (float)(d1c[j] * (double)inv_n); m2c[j] = (float)(d2c[j] * (double)inv_n);
            m1o[j] = (float)(d1o[j] * (double)inv_n); m2o[j] = (float)(d2o[j] * (double)inv_n);
            }
            wn[j] = on ? w0[j] - w1[j] : 0.f; wp[j] = on ? w2[j] - w3[j] : 0.f; wq[j] = on ? w4[j] - w5[j] : 0.f;
        }
    }
    // UR rows per pass, each dependent round of loads (row data + CSR extents -> edge ids -> edge
    // gradients) issued for all of them at once: a row at a time this loop was 8 rows x 4 round trips
    constexpr int UR = 4;
    const int cc = min(c, H - VEC);
    for (int v0 = rbeg + grp; v0 < rend; v0 += RPB * UR) {
        float a0[UR], a1[UR];
        V x4[UR], hc4[UR], ho4[UR], hc2[UR], ho2[UR];
        int ps0[UR], ps1[UR], pd0[UR], pd1[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const size_t v = (size_t)min(v0 + u * RPB, rend - 1);
            a0[u] = a.anode[2 * v]; a1[u] = a.anode[2 * v + 1];
            x4[u] = V::ld(a.x + v * H + cc); hc4[u] = V::ld(a.dxhc + v * H + cc); ho4[u] = V::ld(a.dxho + v * H + cc);
            hc2[u] = a.dxhc2 ? V::ld(a.dxhc2 + v * H + cc) : V::zero();
            ho2[u] = a.dxho2 ? V::ld(a.dxho2 + v * H + cc) : V::zero();
            ps0[u] = a.gs.ptr[v]; ps1[u] = a.gs.ptr[v + 1]; pd0[u] = a.gd.ptr[v]; pd1[u] = a.gd.ptr[v + 1];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            x4[u].pin(); hc4[u].pin(); ho4[u].pin(); hc2[u].pin(); ho2[u].pin();
            hc4[u].add(hc2[u]); ho4[u].add(ho2[u]);
            asm volatile("" : "+v"(a0[u]), "+v"(a1[u]), "+v"(ps0[u]), "+v"(ps1[u]), "+v"(pd0[u]), "+v"(pd1[u]));
        }
        int es[UR], ed[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {          // this lane's first edge of either list (clamped index, masked below)
            es[u] = a.gs.eid[ps0[u] + l < ps1[u] ? ps0[u] + l : 0];
            ed[u] = a.gd.eid[pd0[u] + l < pd1[u] ? pd0[u] + l : 0];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) asm volatile("" : "+v"(es[u]), "+v"(ed[u]));
        float sp[UR], sq[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const float vs = a.dl[ps0[u] + l < ps1[u] ? es[u] : 0], vd = a.dl[pd0[u] + l < pd1[u] ? ed[u] : 0];
            sp[u] = ps0[u] + l < ps1[u] ? vs : 0.f;
            sq[u] = pd0[u] + l < pd1[u] ? vd : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int v = v0 + u * RPB;
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
            float spv = sp[u], sqv = sq[u];
            for (int s = ps0[u] + l + G; s < ps1[u]; s += G) spv += a.dl[a.gs.eid[s]];      // rare: degree > G
            for (int s = pd0[u] + l + G; s < pd1[u]; s += G) sqv += a.dl[a.gd.eid[s]];
            spv = group_sum<G>(spv); sqv = group_sum<G>(sqv);
            if (v < rend) {
                if (l == 0) { sdl += (double)dl0; ssp += (double)spv; }
                if (cok) {
                    float o[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float dx = a0[u] * dxc[j] + a1[u] * dxo[j] + dl0 * wn[j] + spv * wp[j] + sqv * wq[j];
                        if (relu && !(xv[j] > 0.f)) dx = 0.f;
                        o[j] = dx;
                        cs_b[j] += (double)dx;
                        cs_n[j] += (double)(dl0 * xv[j]);
                        cs_p[j] += (double)(spv * xv[j]);
                        cs_q[j] += (double)(sqv * xv[j]);
                    }
                    V ov;
                    if constexpr (VEC == 4) ov.v = make_float4(o[0], o[1], o[2], o[3]); else ov.v = o[0];
                    ov.st(a.dZ + (size_t)v * H + c);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const bool on = cok && c + j < H;
        const double v4[4] = {cs_b[j], cs_n[j], cs_p[j], cs_q[j]};
        double o4[4];
        block_col_sums<4>(v4, l * VEC + j, grp, RPB, G * VEC, on, lds, o4);
        if (grp == 0 && on) {
            if (a.dbias.on()) a.dbias.add(c + j, o4[0]);
            a.dWn.add(c + j, o4[1]); a.dWe.add(c + j, o4[2]); a.dWe.add(H + c + j, o4[3]);
        }
    }
    if (l == 0) { sc_lds[0][grp] = sdl; sc_lds[1][grp] = ssp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < RPB; ++k) { t0 += sc_lds[0][k]; t1 += sc_lds[1][k]; }
        a.dWn.add(H, t0);
        a.dWe.add(2 * H, t1);
    }
}

// ------------------------------------------------------------------------------------------------
// deterministic reduction of split-K weight-gradient slabs + commit of the fp64 arena to the fp32
// gradient buffer + Adam (torch.optim.Adam semantics: bias-corrected, eps outside the sqrt of the
// corrected second moment, optional L2 weight decay added to the gradient).
// ------------------------------------------------------------------------------------------------
struct SlabTask { const float* slabs; float* dst; int n; int S; };
struct CommitTask { const double* src; int P; int stride; int dst; int n; float scale; };   // (partial rows of) fp64 sums -> flat-gradient offset

// the same update for one element, for kernels that finish a gradient and apply it on the spot (k_finish)
constexpr int MAX_ADAM_RANGES = 24;
struct AdamArgs { float* p; float* m; float* v; float* step; const float* lr; float beta1, beta2, eps, wd, gscale; int on; };
struct AdamRange { int64_t begin, end; };
__device__ __forceinline__ void adam_update(const AdamArgs& A, int64_t i, float g, float t, float lr) {
    float gi = g * A.gscale;
    if (A.wd != 0.f) gi = fmaf(A.wd, A.p[i], gi);
    const float mi = A.beta1 * A.m[i] + (1.f - A.beta1) * gi;
    const float vi = A.beta2 * A.v[i] + (1.f - A.beta2) * gi * gi;
    A.m[i] = mi; A.v[i] = vi;
    const float bc1 = 1.f - powf(A.beta1, t), bc2 = 1.f - powf(A.beta2, t);
    const float denom = sqrtf(vi) / sqrtf(bc2) + A.eps;
    A.p[i] -= (lr / bc1) * (mi / denom);
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       float* __restrict__ step, const float* __restrict__ lr_ptr, float beta1, float beta2, float eps,
                       float wd, int64_t n, int ticked, float gscale, const int* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // a flagged step (or an unchecked earlier one) leaves the gradients untrusted: parameters and moments stay as they are
    if (status && (status[0] | status[1]) != 0) return;
    // every thread reads the step counter, so it cannot be advanced in this launch: either the step's
    // k_finish has already done it (ticked = 1) or a separate tiny launch follows (k_adam_tick)
    if (i >= n) return;
    const float t = step[0] + (ticked ? 0.f : 1.f);
    const float lr = lr_ptr[0];
    float gi = g[i] * gscale;
    if (wd != 0.f) gi = fmaf(wd, p[i], gi);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
__global__ void k_adam_tick(float* step) { step[0] += 1.f; }

// ------------------------------------------------------------------------------------------------------------------
// One-shot gradient exchange over peer-mapped memory + Adam (the data-parallel exchange of the ~0.5 MB
// gradient bucket is latency-bound; this is the alternative to the RCCL all-reduce node).  Every rank owns a REGION that all
// ranks have mapped (FINE-GRAINED device memory from cal_p2p_alloc -- coarse-grained hipMalloc memory gives no cross-device
// visibility inside a running kernel -- shared through IPC handles; xGMI peer access between GPUs):
//     stage[2][np] floats (two parities of the gradient bucket) | flags[8] u64 | ctr[0..1] u32 | epoch u64 | ctr[2] u32, abort u32
// and runs, in stream order behind its backward (cal_engine_step mode 1|2|8):
//   1. copy its gradient bucket into its own stage[epoch & 1], system-scope fence; the LAST workgroup to finish writes
//      flags[rank] = epoch into EVERY rank's region (peers poll their own memory);
//   2. every workgroup waits until all `world` flags of its own region have reached `epoch` (bounded: `max_polls`);
//   3. CONSENSUS: a workgroup whose wait ran out sets the region's sticky `abort` word; all workgroups of the launch (fully
//      resident, <= P2P_BLOCKS) meet at a local counter and read `abort` afterwards -- either EVERY workgroup applies the
//      update or NONE does.  After an abort the parameters and moments are never touched again by this kernel (status bit 64
//      and the host-mapped `host_status` word say so; the trainer raises on its next step, re-binding clears it): a rank
//      that lags past the timeout must not make the others step on partial sums (round-3 review);
//   4. read the slice of every rank's stage (system-scope loads), sum in RANK ORDER (all replicas get the same bits), Adam
//      with the 1 / world gradient factor;
//   5. the last workgroup through advances the local epoch.
// Two parities: a rank can only publish epoch e + 1 after its own step e finished, i.e. after it has seen every peer's flag
// e -- every peer has finished step e - 1's reads by then, so stage[(e + 1) & 1] is free.
// ------------------------------------------------------------------------------------------------------------------
constexpr int P2P_MAX_WORLD = 8, P2P_BLOCKS = 448;
struct P2PArgs {
    float* region[P2P_MAX_WORLD];          // every rank's region, as mapped in THIS process
    int rank, world;
    int64_t np;                            // floats per stage parity (nparam rounded up to 64)
    int* host_status;                      // host-mapped word (hipHostMalloc): 64 once an exchange timed out
    int max_polls;                         // bound of the flag wait (cal_engine_p2p_set_timeout)
};
__device__ __forceinline__ unsigned long long* p2p_flags(float* region, int64_t np) { return reinterpret_cast<unsigned long long*>(region + 2 * np); }
__global__ void __launch_bounds__(256) k_p2p_adam(const P2PArgs pa, const float* __restrict__ G, const AdamArgs A, int64_t nparam,
                                                  int* __restrict__ status) {
    float* mine = pa.region[pa.rank];
    unsigned long long* flags = p2p_flags(mine, pa.np);
    unsigned* ctr = reinterpret_cast<unsigned*>(flags + P2P_MAX_WORLD);
    unsigned long long* epoch_p = flags + P2P_MAX_WORLD + 1;
    unsigned* ctr2 = reinterpret_cast<unsigned*>(flags + P2P_MAX_WORLD + 2);
    unsigned* abort_p = ctr2 + 1;
    __shared__ unsigned long long ep_s;
    __shared__ int last_s, abort_s;
    if (threadIdx.x == 0) ep_s = __hip_atomic_load(epoch_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __syncthreads();
    const unsigned long long epoch = ep_s;
    const int par = (int)(epoch & 1);
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    // 1. publish
    float* st = mine + (size_t)par * pa.np;
    for (int64_t i = gid; i < nparam; i += stride) st[i] = G[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ctr + 0, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == (unsigned)(gridDim.x * epoch - 1);
        if (last_s) {
            __threadfence_system();
            for (int r = 0; r < pa.world; ++r)
                __hip_atomic_store(p2p_flags(pa.region[r], pa.np) + pa.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // 2. wait for every rank's flag in OUR region (an earlier abort is final: no wait, no update)
        bool bad = __hip_atomic_load(abort_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        for (int r = 0; r < pa.world && !bad; ++r) {
            int spins = 0;
            while (__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > pa.max_polls) { bad = true; break; }
            }
        }
        if (bad) __hip_atomic_fetch_or(abort_p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // 3. consensus: every workgroup of this launch has decided before anyone reads `abort` (the grid is resident)
        __hip_atomic_fetch_add(ctr2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)(gridDim.x * epoch);
        int spins = 0;
        while (__hip_atomic_load(ctr2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 24)) { __hip_atomic_fetch_or(abort_p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        const int ab = __hip_atomic_load(abort_p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        abort_s = ab;
        if (ab) {
            atomicOr(status, 64);
            if (blockIdx.x == 0 && pa.host_status) __hip_atomic_store(pa.host_status, 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence_system();
    }
    __syncthreads();
    if (!abort_s) {
        const float adam_t = A.step[0], adam_lr = A.lr[0];
        for (int64_t i = gid; i < nparam; i += stride) {
            float g = 0.f;
            for (int r = 0; r < pa.world; ++r)
                g += __hip_atomic_load(pa.region[r] + (size_t)par * pa.np + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            adam_update(A, i, g, adam_t, adam_lr);
        }
    }
    // 5. the last workgroup through closes the epoch
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(gridDim.x * epoch - 1)) __hip_atomic_store(epoch_p, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The reference's 3-term loss on given log-probabilities, and its gradient (train_causal.py:176-183):
//   c_loss = KL(uniform || exp(c)) batchmean = -log C - sum(c) / (B C);  o_loss = NLL(o, y);  co_loss = NLL(co, y)
//   loss = wc c_loss + wo o_loss + wco co_loss;   out = {loss, c_loss, o_loss, co_loss};
//   dlogp [3, B, C] = d loss / d (c, o, co): -wc / (B C) everywhere | -wo / B at the label | -wco / B at the label.
// One workgroup (B C is a few hundred to a few thousand): what a loop that keeps `model(data)` and `loss.backward()` as its
// own statements would otherwise spend on ~20 tiny torch launches (cal_amd.train_causal.causal_loss).  Labels outside
// [0, C) contribute nothing and raise status bit 1 of `flag` (torch's nll_loss asserts on the device).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_causal_loss(const float* __restrict__ logp, const int64_t* __restrict__ y, int B, int C,
                                                     float wc, float wo, float wco, float* __restrict__ out,
                                                     float* __restrict__ dlogp, int* __restrict__ flag) {
    __shared__ double red[3][256];
    const int t = threadIdx.x, BC = B * C;
    double sc = 0.0, so = 0.0, sco = 0.0;
    const float gc = -wc / (float)BC, go = -wo / (float)B, gco = -wco / (float)B;
    for (int i = t; i < BC; i += 256) {
        const int b = i / C, k = i - b * C;
        const int64_t yb = y[b];
        const bool hit = yb == (int64_t)k;
        sc += (double)logp[i];
        if (hit) { so += (double)logp[BC + i]; sco += (double)logp[2 * BC + i]; }
        if (dlogp) { dlogp[i] = gc; dlogp[BC + i] = hit ? go : 0.f; dlogp[2 * BC + i] = hit ? gco : 0.f; }
        if (k == 0 && (yb < 0 || yb >= C) && flag) atomicOr(flag, 1);
    }
    red[0][t] = sc; red[1][t] = so; red[2][t] = sco;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; red[2][t] += red[2][t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        const float cl = (float)(-log((double)C) - red[0][0] / (double)BC);
        const float ol = (float)(-red[1][0] / (double)B), col = (float)(-red[2][0] / (double)B);
        out[0] = wc * cl + wo * ol + wco * col; out[1] = cl; out[2] = ol; out[3] = col;
    }
}

}  // namespace cal
