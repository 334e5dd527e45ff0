// On-device mini-batch assembly: what PyG's DataLoader/Batch collate does on the host per step
// (train_causal.py:13-15,171-174: concatenate node features, offset edge_index by the cumulative
// node count, build the `batch` vector, gather labels) as ONE kernel over a device-resident
// dataset (once the step is ~0.5 ms, Python collation of 128 graphs
// dominates end-to-end graphs/s).
//
// Dataset layout (built once by cal_amd.device_data.DeviceDataset): all graphs concatenated,
//   X [sumN, F] fp32, EI [2, sumE] int64 with node ids LOCAL to their graph, node_ptr/edge_ptr
//   [G+1] int64, Y [G] int64.
// One workgroup per selected graph copies its rows / edges to the output offsets computed on the
// host from the (host-resident) size arrays.
#include "common.hpp"
#include <cstring>

#include <algorithm>
#include <vector>

namespace cal {

__global__ void __launch_bounds__(256) k_collate(const float* __restrict__ X, const int64_t* __restrict__ EI, int64_t sumE,
                                                 int F, const int64_t* __restrict__ node_ptr,
                                                 const int64_t* __restrict__ edge_ptr, const int64_t* __restrict__ Y,
                                                 const int64_t* __restrict__ sel, const int64_t* __restrict__ out_node_off,
                                                 const int64_t* __restrict__ out_edge_off, float* __restrict__ xo,
                                                 int64_t* __restrict__ eio, int64_t Eout, int64_t* __restrict__ batcho,
                                                 int64_t* __restrict__ yo) {
    const int b = blockIdx.x;
    const int64_t g = sel[b];
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1], e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    const int64_t on = out_node_off[b], oe = out_edge_off[b];
    const int64_t nf = (n1 - n0) * F;
    const float* src = X + n0 * F;
    float* dst = xo + on * F;
    for (int64_t i = threadIdx.x; i < nf; i += 256) dst[i] = src[i];
    for (int64_t i = threadIdx.x; i < n1 - n0; i += 256) batcho[on + i] = b;
    for (int64_t i = threadIdx.x; i < e1 - e0; i += 256) {
        eio[oe + i] = EI[e0 + i] + on;
        eio[Eout + oe + i] = EI[sumE + e0 + i] + on;
    }
    if (threadIdx.x == 0) yo[b] = Y[g];
}

// Stand-alone draw of the random-intervention permutation (randperm_block in engine_kernels.hpp): one workgroup, B <= 4096.
constexpr int PERM_MAX = 4096;
__global__ void __launch_bounds__(1024) k_randperm(int64_t* __restrict__ perm, int B, unsigned long long seed,
                                                    unsigned long long* __restrict__ counter) {
    __shared__ unsigned long long key[PERM_MAX];
    __shared__ int idx[PERM_MAX];
    randperm_block<1024>(perm, B, seed, counter, key, idx);
}

}  // namespace cal

using namespace cal;

CAL_EXPORT int cal_collate(const float* X, const int64_t* EI, int64_t sumE, int64_t F, const int64_t* node_ptr,
                           const int64_t* edge_ptr, const int64_t* Y, const int64_t* sel, const int64_t* out_node_off,
                           const int64_t* out_edge_off, float* xo, int64_t* eio, int64_t Eout, int64_t* batcho,
                           int64_t* yo, int64_t B, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_collate, dim3((unsigned)B), dim3(256), 0, stream, X, EI, sumE, (int)F, node_ptr, edge_ptr, Y, sel,
                       out_node_off, out_edge_off, xo, eio, Eout, batcho, yo);
    CAL_CHECK_LAUNCH("k_collate");
    return 0;
}

// perm[0..B) <- a uniformly random permutation keyed by (seed, *counter); *counter += 1.  B <= 4096.
CAL_EXPORT int cal_randperm(int64_t* perm, int64_t B, uint64_t seed, uint64_t* counter, void* stream) {
    if (B == 0) return 0;
    CAL_REQUIRE(B > 0 && B <= cal::PERM_MAX, "cal_randperm supports 1..4096 graphs per batch");
    hipLaunchKernelGGL(cal::k_randperm, dim3(1), dim3(1024), 0, (hipStream_t)stream, perm, (int)B,
                       (unsigned long long)seed, (unsigned long long*)counter);
    CAL_CHECK_LAUNCH("k_randperm");
    return 0;
}


// Batch assembly for HOST-resident datasets (the reference's own feed: DataLoader + Batch collate, train_causal.py:13-15,171-174):
// the dataset is one concatenation (X [Ntot, F], EI [2, Etot] with edge ids LOCAL to their graph, node_ptr / edge_ptr [G + 1],
// Y [G]); the mini-batch `idx` [B] is written into caller-owned (pinned) staging memory -- features, edge_index rebased by
// the batch's node offsets, batch vector, labels -- with one memcpy per graph and attribute.  Python's per-graph torch.cat of
// the same took ~1 ms for 128 SPMotif graphs, four GPU steps.  Host pointers, no device work.
CAL_EXPORT int cal_collate_host(const float* X, const int64_t* EI, int64_t Etot, int64_t F, const int64_t* node_ptr,
                                const int64_t* edge_ptr, const int64_t* Y, const int64_t* idx, int64_t B, float* xo,
                                int64_t* eio, int64_t Eout, int64_t* batcho, int64_t* yo) {
    CAL_REQUIRE(X && EI && node_ptr && edge_ptr && Y && idx && xo && eio && batcho && yo && B >= 0 && F > 0, "bad arguments");
    int64_t no = 0, eo = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t g = idx[b], n0 = node_ptr[g], n = node_ptr[g + 1] - n0, e0 = edge_ptr[g], e = edge_ptr[g + 1] - e0;
        CAL_REQUIRE(n >= 0 && e >= 0 && eo + e <= Eout, "cal_collate_host: offsets out of range");
        memcpy(xo + (size_t)no * F, X + (size_t)n0 * F, (size_t)n * F * sizeof(float));
        for (int64_t k = 0; k < e; ++k) { eio[eo + k] = EI[e0 + k] + no; eio[Eout + eo + k] = EI[Etot + e0 + k] + no; }
        for (int64_t k = 0; k < n; ++k) batcho[no + k] = b;
        yo[b] = Y[g];
        no += n; eo += e;
    }
    CAL_REQUIRE(eo == Eout, "cal_collate_host: edge count mismatch");
    return 0;
}

// Host-side helper of the small-graph packing (cal_engine_set_tiles): order the B graphs of a mini-batch so that consecutive
// runs of them fill 64-node tiles.  A mini-batch is a SET of graphs (the loss is a mean over it, the intervention permutation
// random): their order inside the batch is the loader's to choose.  First-fit decreasing by node count under the three tile
// bounds; graphs are emitted tile by tile.  order_out [B]: positions into the input arrays; first_out [T + 1]: first emitted
// graph of every tile.  Returns T, or -1 when a graph exceeds a bound (the batch is not packed).  No device work.
CAL_EXPORT int64_t cal_pack_order(const int64_t* node_sizes, const int64_t* edge_sizes, int64_t B, int64_t max_nodes,
                                  int64_t max_edges, int64_t max_graphs, int64_t* order_out, int64_t* first_out) {
    if (!node_sizes || !edge_sizes || !order_out || !first_out || B < 0 || max_graphs < 1) return -1;
    std::vector<int64_t> idx((size_t)B);
    for (int64_t i = 0; i < B; ++i) {
        if (node_sizes[i] > max_nodes || edge_sizes[i] > max_edges || node_sizes[i] < 0 || edge_sizes[i] < 0) return -1;
        idx[(size_t)i] = i;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return node_sizes[a] > node_sizes[b]; });
    struct Bin { int64_t n, e, g; std::vector<int64_t> items; };
    std::vector<Bin> bins;
    size_t first_open = 0;                                   // bins before it are full in one of the three dimensions
    for (int64_t k = 0; k < B; ++k) {
        const int64_t i = idx[(size_t)k], n = node_sizes[i], e = edge_sizes[i];
        size_t j = first_open;
        for (; j < bins.size(); ++j)
            if (bins[j].n + n <= max_nodes && bins[j].e + e <= max_edges && bins[j].g < max_graphs) break;
        if (j == bins.size()) bins.push_back(Bin{0, 0, 0, {}});
        bins[j].n += n; bins[j].e += e; bins[j].g += 1; bins[j].items.push_back(i);
        while (first_open < bins.size() && (bins[first_open].n >= max_nodes || bins[first_open].g >= max_graphs)) ++first_open;
    }
    int64_t pos = 0, t = 0;
    for (auto& b : bins) {
        first_out[t++] = pos;
        std::sort(b.items.begin(), b.items.end());           // inside a tile: the loader's order
        for (int64_t i : b.items) order_out[pos++] = i;
    }
    first_out[t] = pos;
    return t;
}
