// GraphPlan: COO edge list -> CSR-by-destination + CSR-by-source (int32), both
// with the original edge id per slot, explicit self loops dropped.
//
// Replaces, once per mini-batch, what the reference recomputes in every layer:
// remove_self_loops + add_self_loops inside GCNConv.norm (gcn_conv.py:56-57)
// and the implicit scatter index of MessagePassing.propagate (gcn_conv.py:92).
// The N added self loops are never materialised: every consumer kernel adds the
// diagonal term itself.  Slots inside a row are sorted by edge id, so every
// segment reduction downstream runs in a fixed order (run-to-run deterministic,
// and the same order as a sequential CPU scatter_add).
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace cal {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// cnt layout: [0, N] in-degree(+1 pad), [N+1, 2N+1] out-degree(+1 pad)
__global__ void k_plan_count(const int64_t* __restrict__ ei, int64_t E, int N,
                             int* __restrict__ cnt_dst, int* __restrict__ cnt_src,
                             int* __restrict__ row32, int* __restrict__ col32,
                             int* __restrict__ status) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t r = ei[e], c = ei[E + e];
    if (r < 0 || r >= N || c < 0 || c >= N) {   // reference would raise an index error
        atomicOr(status, 1);
        row32[e] = 0; col32[e] = 0;             // treated as a dropped self loop
        return;
    }
    row32[e] = (int)r;
    col32[e] = (int)c;
    if (r != c) {
        atomicAdd(&cnt_dst[c], 1);
        atomicAdd(&cnt_src[r], 1);
    }
}

// Single-workgroup exclusive scan of n counters into n+1 offsets; blockIdx.x picks the array.
// Each thread owns 8 consecutive counters (serial prefix in registers), the 1024 thread totals are
// scanned with wave shuffles + one LDS step: one pass (3 barriers) per 8192 elements.
__global__ void __launch_bounds__(1024) k_plan_scan(const int* __restrict__ cnt_a, int* __restrict__ ptr_a,
                                                    const int* __restrict__ cnt_b, int* __restrict__ ptr_b,
                                                    int n) {
    const int* cnt = blockIdx.x == 0 ? cnt_a : cnt_b;
    int* ptr = blockIdx.x == 0 ? ptr_a : ptr_b;
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 8192) {
        const int i0 = base + threadIdx.x * 8;
        int v[8];
        int tsum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (i0 + j < n) ? cnt[i0 + j] : 0; tsum += v[j]; }
        int x = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        const int carry = carry_s;
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wave_tot[w];
        int run = carry + woff + x - tsum;
#pragma unroll
        for (int j = 0; j < 8; ++j) { if (i0 + j < n) ptr[i0 + j] = run; run += v[j]; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) ptr[n] = carry_s;
}

// Large batches (config 5: 160k nodes = 20 serial passes of the single-workgroup scan, 145 us): two launches instead --
// per-chunk totals, then every chunk scans itself on top of the sum of the totals before it.  grid (chunks, 2 arrays).
__global__ void __launch_bounds__(1024) k_plan_scan_tot(const int* __restrict__ cnt_a, const int* __restrict__ cnt_b, int n,
                                                        int* __restrict__ tot) {
    const int* cnt = blockIdx.y == 0 ? cnt_a : cnt_b;
    __shared__ int wave_tot[16];
    const int i0 = blockIdx.x * 8192 + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (i0 + j < n) ? cnt[i0 + j] : 0;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wave_tot[w];
        tot[blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(1024) k_plan_scan_chunk(const int* __restrict__ cnt_a, int* __restrict__ ptr_a,
                                                          const int* __restrict__ cnt_b, int* __restrict__ ptr_b, int n,
                                                          const int* __restrict__ tot) {
    const int* cnt = blockIdx.y == 0 ? cnt_a : cnt_b;
    int* ptr = blockIdx.y == 0 ? ptr_a : ptr_b;
    __shared__ int wave_tot[16];
    __shared__ int red[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // carry = sum of the totals of the chunks before this one
    int c = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 1024) c += tot[blockIdx.y * gridDim.x + b];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) red[wid] = c;
    const int i0 = blockIdx.x * 8192 + threadIdx.x * 8;
    int v[8];
    int tsum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (i0 + j < n) ? cnt[i0 + j] : 0; tsum += v[j]; }
    int x = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    int carry = 0, woff = 0;
    for (int w = 0; w < 16; ++w) carry += red[w];
    for (int w = 0; w < wid; ++w) woff += wave_tot[w];
    int run = carry + woff + x - tsum;
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (i0 + j < n) ptr[i0 + j] = run; run += v[j]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) ptr[n] = carry + woff + x;
}

__global__ void k_plan_fill(const int* __restrict__ row32, const int* __restrict__ col32, int64_t E,
                            const int* __restrict__ ptr_dst, const int* __restrict__ ptr_src,
                            int* __restrict__ cur_dst, int* __restrict__ cur_src,
                            int* __restrict__ nbr_d, int* __restrict__ eid_d,
                            int* __restrict__ nbr_s, int* __restrict__ eid_s) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int r = row32[e], c = col32[e];
    if (r == c) return;
    int p = ptr_dst[c] + atomicAdd(&cur_dst[c], 1);
    nbr_d[p] = r; eid_d[p] = (int)e;
    int q = ptr_src[r] + atomicAdd(&cur_src[r], 1);
    nbr_s[q] = c; eid_s[q] = (int)e;
}

// Stable placement without a sort: one thread per (direction, unordered slot p) counts how many
// slots of the same row hold a smaller edge id and writes its entry at that rank.  O(sum deg^2)
// comparisons, all parallel (a hub of degree d is shared by d threads).
__global__ void k_plan_rank(const int* __restrict__ row32, const int* __restrict__ col32,
                            const int* __restrict__ ptr_dst, const int* __restrict__ tn_d, const int* __restrict__ te_d,
                            int* __restrict__ nbr_d, int* __restrict__ eid_d,
                            const int* __restrict__ ptr_src, const int* __restrict__ tn_s, const int* __restrict__ te_s,
                            int* __restrict__ nbr_s, int* __restrict__ eid_s, int N) {
    const int nnz = ptr_dst[N];
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * nnz) return;
    const bool d = t < nnz;
    const int p = d ? t : t - nnz;
    const int* te = d ? te_d : te_s;
    const int e = te[p];
    const int v = d ? col32[e] : row32[e];
    const int* ptr = d ? ptr_dst : ptr_src;
    const int s0 = ptr[v], s1 = ptr[v + 1];
    int rank = 0;
    for (int q = s0; q < s1; ++q) rank += te[q] < e;
    if (d) { nbr_d[s0 + rank] = tn_d[p]; eid_d[s0 + rank] = e; }
    else { nbr_s[s0 + rank] = tn_s[p]; eid_s[s0 + rank] = e; }
}

__global__ void k_zero_i32(int* __restrict__ a, int64_t n, int* __restrict__ b) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 0;
    if (i == 0 && b) *b = 0;
}

// gptr[b] = first node index whose graph id is >= b; batch must be sorted.
__global__ void k_graph_ptr(const int64_t* __restrict__ batch, int N, int B, int* __restrict__ gptr,
                            int* __restrict__ status) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    int64_t prev = i == 0 ? -1 : batch[i - 1];
    int64_t cur = i == N ? (int64_t)B : batch[i];
    if (i < N && (cur < prev || cur >= B || cur < 0)) { atomicOr(status, 2); return; }
    for (int64_t b = prev + 1; b <= cur && b <= B; ++b) gptr[b] = i;
}

}  // namespace cal

using namespace cal;

CAL_EXPORT const char* cal_last_error() { return g_err; }
CAL_EXPORT int cal_version() { return 100; }

// Build both CSR views.  `work` must hold 4*(N+1) + 4*E ints; `status` one int (bit0: edge index out of
// range, bit1: batch vector not sorted / out of range) -- zeroed here, read by the caller when it
// chooses to validate.
namespace cal {
// prezeroed: the caller has already zeroed work[0 .. 4(N+1)) and *status on this stream (the step engine
// does it in its own arena-zeroing launch: one kernel boundary less per step)
int plan_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* nbr_dst, int32_t* eid_dst,
               int32_t* rowptr_src, int32_t* nbr_src, int32_t* eid_src, int32_t* row32, int32_t* col32, int32_t* work,
               int32_t* status, bool prezeroed, hipStream_t stream) {
    CAL_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) && E < (1ll << 31), "N/E out of int32 range");
    int n = (int)N;
    int* cnt_dst = work;
    int* cnt_src = work + (n + 1);
    int* cur_dst = work + 2 * (n + 1);
    int* cur_src = work + 3 * (n + 1);
    int* tn_d = work + 4 * (n + 1);
    int* te_d = tn_d + E;
    int* tn_s = te_d + E;
    int* te_s = tn_s + E;
    // a kernel, not hipMemsetAsync: memset nodes inside several captured hipGraphs faulted on replay
    // (ROCm 7.2), and one launch is cheaper than two memset nodes anyway
    if (!prezeroed) {
        hipLaunchKernelGGL(k_zero_i32, dim3(cdiv(4 * (int64_t)(n + 1), 256)), dim3(256), 0, stream, work,
                           4 * (int64_t)(n + 1), status);
        CAL_CHECK_LAUNCH("k_zero_i32");
    }
    if (E > 0) {
        hipLaunchKernelGGL(k_plan_count, dim3(cdiv(E, 256)), dim3(256), 0, stream, edge_index, E, n,
                           cnt_dst, cnt_src, row32, col32, status);
        CAL_CHECK_LAUNCH("k_plan_count");
    }
    if (n > 4 * 8192 && E >= 2 * (int64_t)cdiv(n, 8192)) {
        // chunk totals live in the (not yet written) sort scratch: 2 * chunks ints
        const int chunks = cdiv(n, 8192);
        hipLaunchKernelGGL(k_plan_scan_tot, dim3(chunks, 2), dim3(1024), 0, stream, cnt_dst, cnt_src, n, tn_d);
        CAL_CHECK_LAUNCH("k_plan_scan_tot");
        hipLaunchKernelGGL(k_plan_scan_chunk, dim3(chunks, 2), dim3(1024), 0, stream, cnt_dst, rowptr_dst, cnt_src, rowptr_src, n, tn_d);
        CAL_CHECK_LAUNCH("k_plan_scan_chunk");
    } else {
        hipLaunchKernelGGL(k_plan_scan, dim3(2), dim3(1024), 0, stream, cnt_dst, rowptr_dst, cnt_src, rowptr_src, n);
        CAL_CHECK_LAUNCH("k_plan_scan");
    }
    if (E > 0) {
        hipLaunchKernelGGL(k_plan_fill, dim3(cdiv(E, 256)), dim3(256), 0, stream, row32, col32, E,
                           rowptr_dst, rowptr_src, cur_dst, cur_src, tn_d, te_d, tn_s, te_s);
        CAL_CHECK_LAUNCH("k_plan_fill");
        // grid sized for the upper bound 2E (self loops make nnz <= E); surplus threads exit
        hipLaunchKernelGGL(k_plan_rank, dim3(cdiv(2 * E, 256)), dim3(256), 0, stream, row32, col32, rowptr_dst, tn_d,
                           te_d, nbr_dst, eid_dst, rowptr_src, tn_s, te_s, nbr_src, eid_src, n);
        CAL_CHECK_LAUNCH("k_plan_rank");
    }
    return 0;
}
}  // namespace cal

namespace cal {
// the rank pass alone, behind a per-graph fill (engine_plan.hpp k_plan_big): scratch = 4 E ints {tn_d, te_d, tn_s, te_s}
int plan_rank(int64_t E, int64_t N, const int32_t* rowptr_dst, int32_t* nbr_dst, int32_t* eid_dst, const int32_t* rowptr_src,
              int32_t* nbr_src, int32_t* eid_src, const int32_t* row32, const int32_t* col32, const int32_t* scratch,
              hipStream_t stream) {
    if (E <= 0) return 0;
    hipLaunchKernelGGL(k_plan_rank, dim3(cdiv(2 * E, 256)), dim3(256), 0, stream, row32, col32, rowptr_dst, scratch, scratch + E,
                       nbr_dst, eid_dst, rowptr_src, scratch + 2 * E, scratch + 3 * E, nbr_src, eid_src, (int)N);
    CAL_CHECK_LAUNCH("k_plan_rank");
    return 0;
}
}  // namespace cal

CAL_EXPORT int cal_plan_build(const int64_t* edge_index, int64_t E, int64_t N,
                              int32_t* rowptr_dst, int32_t* nbr_dst, int32_t* eid_dst,
                              int32_t* rowptr_src, int32_t* nbr_src, int32_t* eid_src,
                              int32_t* row32, int32_t* col32, int32_t* work, int32_t* status,
                              void* stream_) {
    return cal::plan_build(edge_index, E, N, rowptr_dst, nbr_dst, eid_dst, rowptr_src, nbr_src, eid_src, row32, col32, work,
                           status, false, (hipStream_t)stream_);
}

CAL_EXPORT int cal_graph_ptr(const int64_t* batch, int64_t N, int64_t B, int32_t* gptr, int32_t* status,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CAL_REQUIRE(N >= 0 && B >= 0 && N < (1ll << 31), "bad sizes");
    hipLaunchKernelGGL(k_graph_ptr, dim3(cdiv(N + 1, 256)), dim3(256), 0, stream, batch, (int)N, (int)B, gptr, status);
    CAL_CHECK_LAUNCH("k_graph_ptr");
    return 0;
}
