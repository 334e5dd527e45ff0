// Native CausalGCN step engine: the whole training step of train_causal.py:173-192 on
// model.py:85-164 -- forward (3 heads), 3-term loss, backward, Adam -- as ONE C call that enqueues
// a few dozen hand-written kernels on a stream (hipGraph-capturable: no allocation, no sync, no memset
// nodes).  What is fused, relative to the op sequence of the reference:
//   * every BatchNorm is folded into its consumer: batch statistics are accumulated (fp64) by the
//     producer's epilogue, the normalised tensor is never written -- the consumer GEMM applies
//     scale/shift (and the node-attention row scale) while staging its operand;
//   * bias + ReLU (+ next-BN statistics) ride on the aggregation / GEMM epilogues;
//   * xc = a0*x, xo = a1*x and the [E,2H] edge representation are never materialised;
//   * the GCN normalisation is computed on the fly from deg^-1/2 (no per-layer norm(), no
//     message tensor);
//   * BatchNorm-backward column sums ride on the GEMM that produces the incoming gradient;
//   * all parameter gradients that are column sums go through one fp64 arena and one commit
//     kernel; split-K weight-gradient slabs are reduced deterministically in the same launch;
//   * the three readout heads (BN -> fc1 -> ReLU -> BN -> fc2 -> log_softmax -> loss) and their backward are ONE launch
//     of resident workgroups that meet at two in-kernel barriers per head (engine_ro_step.hpp);
//   * Adam runs inside that last commit kernel for single-process steps (every gradient element is updated by the
//     thread that finishes it), as one kernel over the flat parameter buffer after a gradient exchange.
#include <string.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "engine_kernels.hpp"
#include "engine_readout.hpp"
#include "engine_gconv.hpp"
#include "engine_gconv_bwd.hpp"
#include "engine_ggat.hpp"
#include "engine_plan.hpp"
#include "engine_attbwd.hpp"
#include "engine_ro_step.hpp"
#include "engine_gin.hpp"
#include "engine_ggin.hpp"
#include "engine_feat.hpp"
#include "engine_gwide.hpp"
#include "engine_attwide.hpp"

namespace cal {

constexpr int MAX_LAYERS = 6;
constexpr int MAX_SLABS = 24;
constexpr int MAX_COMMITS = 64;

struct FinishArgs {
    SlabTask st[MAX_SLABS];
    CommitTask ct[MAX_COMMITS];
    int nst, nct;
    float* stats;            // fused readout: stats[0] = wc*stats[1] + wo*stats[2] + wco*stats[3]
    float wc, wo, wco;
    float* tick;             // Adam step counter to advance (the update follows in the same step) or null
    float* ticked;           // Adam step counter the step's FIRST kernel has already advanced (k_zero_f64), or null: taken back here when the
                             // step is flagged -- whether the update rides in this kernel (adam.on) or follows as k_adam (range overflow)
    unsigned long long* perm_ctr;   // counter of the in-step permutation draw to advance (k_zero_f64's rank-sort workgroups only read it), or null
    // Adam inside this kernel (single-process steps, mode bit 4 without bit 8): every gradient element is updated by the
    // thread that finishes it, the ranges no task writes (gradients other kernels stored directly) by `nar` extra tasks
    // of 256 elements per block; the step counter was advanced by the step's FIRST kernel (a ticket of 2700 blocks on one
    // address, each behind a device-scope fence, cost 47 us)
    AdamArgs adam;
    AdamRange ar[MAX_ADAM_RANGES];
    int nar;
    double* bn0z; int bn0z_n;   // bn_feat's statistics range of the fp64 arena: zeroed HERE for the next step (engine_plan.hpp: PlanFold)
    int* dirty;              // the device word of that invariant, cleared here
    int* host_status;        // host-mapped mirror of status[0] | status[1], written by this kernel (cal_engine_peek_status), or null
    const int* status;       // the engine's status words: while any bit is up (this step's or a sticky earlier one) the gradients are
                             // not trusted and NO parameter / moment is updated (check_status raises and clears them)
    int blk0[MAX_SLABS + MAX_COMMITS + MAX_ADAM_RANGES + 1];   // first block of every task in the flattened 1-D grid (filled at launch)
};
// grid: 1-D, task t owns blocks [blk0[t], blk0[t+1]): n/64 per slab task, ceil(n/16) per commit task (a
// (64, tasks) grid spent most of its ~12k blocks on nothing once the slab tasks wanted 256 blocks each)
__global__ void __launch_bounds__(256) k_finish(const FinishArgs fa, float* __restrict__ grad) {
    int task = 0;
    const int ntask = fa.nst + fa.nct + fa.nar;
    AdamArgs A = fa.adam;
    float adam_t = 0.f, adam_lr = 0.f;
    const bool flagged = fa.status && (fa.status[0] | fa.status[1]) != 0;     // every earlier kernel of the step has finished: uniform
    // a gated step applies no update, so it must not count as one either: the step's first kernel has already advanced the
    // Adam step counter (bias correction) -- take that back (advisor, round 4: the counter drifted by the frozen steps)
    // (advisor, round 5: keyed on adam.on the fallback to a separate k_adam -- adam.on = 0, counter already advanced -- kept counting)
    if (fa.ticked && flagged && blockIdx.x == 0 && threadIdx.x == 0) fa.ticked[0] -= 1.f;
    if (A.on && flagged) A.on = 0;
    if (A.on) { adam_t = A.step[0]; adam_lr = A.lr[0]; }      // the step's first kernel has already advanced the counter
    while (task + 1 < ntask && (int)blockIdx.x >= fa.blk0[task + 1]) ++task;
    const int bx = blockIdx.x - fa.blk0[task], nbx = fa.blk0[task + 1] - fa.blk0[task];
    if (fa.stats && blockIdx.x == 0 && threadIdx.x == 0)
        fa.stats[0] = fa.wc * fa.stats[1] + fa.wo * fa.stats[2] + fa.wco * fa.stats[3];
    if (fa.tick && !flagged && blockIdx.x == 0 && threadIdx.x == 0) fa.tick[0] += 1.f;      // (the k_adam that follows skips a flagged step too)
    if (fa.perm_ctr && blockIdx.x == 0 && threadIdx.x == 0) fa.perm_ctr[0] += 1;
    if (fa.bn0z && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < fa.bn0z_n; i += 256) fa.bn0z[i] = 0.0;
        if (threadIdx.x == 0) *fa.dirty = 0;
    }
    if (fa.host_status && fa.status && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(fa.host_status, fa.status[0] | fa.status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // Both task kinds are pure reductions over S slabs / P partial rows: the loops keep 8 loads in flight
    // per lane (unconditional on a clamped index, pinned, masked when added) -- as dependent loops with
    // two loads in flight the 58-slab weight-gradient sums made this kernel 11 us.
    if (task < fa.nst) {
        // 64 elements per block pass, the S slabs split over the block's 4 waves (S = #graphs = 128 with the
        // per-graph fused backward): 4x shorter load chains, combined through LDS in a fixed order
        const SlabTask t = fa.st[task];
        __shared__ float sred[256];
        const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
        const int per = (t.S + 3) / 4, z_lo = grp * per, z_hi = min(t.S, z_lo + per);
        for (int i0 = bx * 64; i0 < t.n; i0 += nbx * 64) {
            const int i = min(i0 + el, t.n - 1);
            float s0 = 0.f, s1 = 0.f;
            for (int z0 = z_lo; z0 < z_hi; z0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t.slabs[(size_t)max(min(z0 + u, z_hi - 1), 0) * t.n + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    s0 += z0 + u < z_hi ? v[u] : 0.f;
                    s1 += z0 + u + 1 < z_hi ? v[u + 1] : 0.f;
                }
            }
            sred[threadIdx.x] = s0 + s1;
            __syncthreads();
            if (grp == 0 && i0 + el < t.n) {
                const float gsum = (sred[el] + sred[64 + el]) + (sred[128 + el] + sred[192 + el]);
                t.dst[i] = gsum;
                if (A.on) adam_update(A, (t.dst - grad) + i, gsum, adam_t, adam_lr);
            }
            __syncthreads();
        }
    } else if (task - fa.nst < fa.nct) {
        const CommitTask t = fa.ct[task - fa.nst];
        // 16 part-lanes x 16 columns per block pass
        __shared__ double red[256];
        const int pl = threadIdx.x >> 4, cl = threadIdx.x & 15;
        for (int c0 = bx * 16; c0 < t.n; c0 += nbx * 16) {
            const int c = c0 + cl, cc = min(c, t.n - 1);
            double sacc = 0.0;
            for (int q0 = pl; q0 < t.P; q0 += 16 * 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t.src[(size_t)max(min(q0 + 16 * u, t.P - 1), 0) * t.stride + cc];
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc += q0 + 16 * u < t.P ? v[u] : 0.0;
            }
            red[threadIdx.x] = sacc;
            __syncthreads();
            if (pl == 0 && c < t.n) {
                double tot = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) tot += red[k * 16 + cl];
                const float gv = t.scale * (float)tot;
                grad[t.dst + c] = gv;
                if (A.on) adam_update(A, t.dst + c, gv, adam_t, adam_lr);
            }
            __syncthreads();
        }
    } else if (A.on) {
        const AdamRange r = fa.ar[task - fa.nst - fa.nct];
        const int64_t i = r.begin + (int64_t)bx * 256 + threadIdx.x;
        if (i < r.end) adam_update(A, i, grad[i], adam_t, adam_lr);
    }
}

struct BNSlot { int gamma, beta; int width; float* rm; float* rv; int64_t* nbt; int arena; };

struct Engine {
    int F, H, C, L;
    int cat;                    // cat_or_add == "cat": the co readout takes cat(xc[perm], xo) [B, 2H] (model.py:65-69,153-154)
    int no_node_att, no_edge_att;   // without_node_attention / without_edge_attention: constant 0.5 masks (model.py:99-107)
    int gin;                    // backbone of GINConv(Linear, BN, ReLU, Linear, ReLU) layers (CausalGIN, model.py:188-194)
    int o_gin_w2[MAX_LAYERS], o_gin_b2[MAX_LAYERS];      // second Linear of every GIN layer (o_conv_w / o_conv_b: the first)
    float *gagg, *gt1, *gy, *ones;   // GIN: per layer aggregation output, pre-BN activations, relu(BN(.)) [L][N,H]; ones [N]
    float loop_w;
    // bound buffers
    float *P, *G, *M1, *M2, *step, *lr;
    int64_t nparam;
    float beta1, beta2, eps, wd;
    unsigned long long perm_seed; unsigned long long* perm_ctr;   // device draw of the random-intervention permutation (mode bit 16)
    int64_t* perm_dev;          // [capB] the permutation drawn by the step itself
    P2PArgs p2p; int p2p_on;    // one-shot peer-memory gradient exchange (cal_engine_p2p_bind)
    int num_cus;                      // compute units of the device (launch-shape decisions)
    int* host_status;                 // host-mapped mirror of the status words, refreshed by every step's last kernel (cal_engine_peek_status)
    int* p2p_host_status;             // host-mapped word k_p2p_adam sets when an exchange timed out (cal_engine_p2p_status)
    int p2p_max_polls;                // bound of k_p2p_adam's flag wait (cal_engine_create: 2^22; cal_engine_p2p_set_timeout)
    float grad_scale;           // gradient factor inside Adam (1 / world_size after a sum all-reduce)
    // parameter offsets (floats into P / G)
    int o_feat_w;
    int o_conv_w[MAX_LAYERS], o_conv_b[MAX_LAYERS];
    int o_eatt_w, o_eatt_b, o_natt_w, o_natt_b;
    int o_cw, o_cb, o_ow, o_ob;
    int o_fc1_w[3], o_fc1_b[3], o_fc2_w[3], o_fc2_b[3];
    // BatchNorms: 0 bn_feat, 1..L bns_conv, L+1 bnc, L+2 bno, L+3+2h fc1_bn_h, L+4+2h fc2_bn_h
    BNSlot bn[MAX_LAYERS + 9];
    int nbn;
    // arena offsets (doubles)
    int a_convb[MAX_LAYERS], a_cb, a_ob, a_dwn, a_dwe, a_db1, a_db2, a_sync, arena_n, bn_plane;
    // workspace
    char* ws; size_t ws_bytes;
    int64_t capN, capE, capB;
    // float regions
    float *h, *z, *zco, *hco, *anode, *pq, *att, *dis_unit, *dis_co, *pooled, *pcnt, *xco, *y1, *zl, *logp, *stats, *zpart;
    int adam_fused;          // 1: mode-4 steps apply Adam inside k_finish; CAL_AMD_ADAM_FUSED=0 keeps the k_adam launch
    int ro_rows;             // 1: ... also for 129 .. 512 graphs, in row blocks (k_ro_step<true>); CAL_AMD_RO_ROWS=0: the GEMM chain there
    int striped;             // 1: the per-graph kernels exchange their BatchNorm sums through NSTRIPE accumulator planes (engine.hpp: stripe_sum)
                             // instead of partial rows + k_stats_final; CAL_AMD_STRIPED=0 keeps the finishing launches
    int attwide;             // 1: the attention block of 129-256-node graphs per graph (engine_attwide.hpp); CAL_AMD_ATTWIDE=0: the node-level kernels
    int rpb_div;             // rows per workgroup of the node-level row kernels = max(32, N / rpb_div); CAL_AMD_RPB_DIV (experiment); 0 = by size
    int fold_zero;           // 1: forward + backward steps on the per-graph plan have no k_zero_f64 launch (PlanFold); CAL_AMD_FOLD_ZERO=0: always the launch
    int bn0_dirty_host;      // host twin of the device word status[3]: a training forward has been enqueued since the last k_finish
    int gw_cols;             // CAL_AMD_GW_COLS (experiment): 32 / 64 forces the wide forward kernels' slice width, 0 = by occupancy
    int gwide;               // 1: graphs of 129 .. 256 nodes run the wide per-graph convolutions (engine_gwide.hpp); CAL_AMD_GWIDE=0: the node-level chain
    int ro_step;             // 1: training steps run the readout as one launch (k_ro_step); CAL_AMD_RO_STEP=0 keeps the four kernels
    float *dzl, *dyh1, *dy1, *dxh, *dpool, *dZco, *gn, *gself, *ddeg, *dl, *dzco, *dXhco, *dZ, *dzi, *dXh, *slabs;
    size_t slab_floats;
    double* arena;
    double* parts; size_t parts_doubles;
    // side stream for the weight-gradient GEMMs (off the critical path until the final commit)
    hipStream_t side; hipEvent_t ev_fork[24], ev_join[24];
    int *rowptr_dst, *nbr_dst, *eid_dst, *rowptr_src, *nbr_src, *eid_src, *row32, *col32, *work, *status, *gptr, *iperm, *eptr;
    float* wslot;               // [2][E] attention edge weights (context, objects) in CSR-by-destination slot order
    float* coef_src;            // [E] unit-weight coefficients deg^-1/2 of the destination in CSR-by-SOURCE slot order (k_plan_graph, for the wide per-graph backward)
    float* coef;                // [3][E] edge coefficients dis_j * w_e in CSR-by-destination slot order: unit, context, objects
    int max_nodes, max_edges;   // per-graph bounds of the coming batches (0 = unknown): cal_engine_set_graph_bounds
    const int64_t *node_ptr, *edge_ptr;   // [B+1] device arrays of the coming batch (null = unknown): cal_engine_set_graph_ptrs
    // small-graph packing (cal_engine_set_tiles): the per-graph kernels run one workgroup per TILE = a run of consecutive
    // graphs; node_ptr / edge_ptr and the bounds above then describe tiles, tile_gptr [ntiles + 1] = first graph of every tile
    const int64_t* tile_gptr; int ntiles;
    // GATConv backbone (CausalGAT, model.py:340,390): K heads (0 = GCNConv backbone), attention dropout p with
    // per-layer seeds and an optional device step counter, att [K, 2D] parameter offsets
    int K; float gat_p, gat_slope;
    uint64_t gat_seed[MAX_LAYERS];
    unsigned long long* gat_ctr;
    int o_conv_att[MAX_LAYERS];
    float *gz, *gsc, *gws;      // per layer: z = BN(h) W [L][N,H]; a_dst, a_src, max, denominator [L][4][N,K]; backward scratch
};

static size_t al(size_t n) { return (n + 63) / 64 * 64; }   // 256 B granules (in floats/ints)

}  // namespace cal

using namespace cal;

namespace cal {
int gat_forward(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst, const float* z,
                const float* att, const float* bias, int relu, float slope, float p, uint64_t seed, const uint64_t* ctr,
                float* out, float* adst, float* asrc, float* mx, float* den, int64_t N, int64_t E,
                int64_t K, int64_t D, hipStream_t stream);
int gat_backward(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                 const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src, const float* z,
                 const float* att, const float* adst, const float* asrc, const float* mx, const float* den,
                 const float* gout, float slope, float p, uint64_t seed, const uint64_t* ctr, float* dz, float* datt,
                 float* ws, int64_t N, int64_t E, int64_t K, int64_t D, hipStream_t stream, float* part_out, int* nparts);
int gat_datt_parts(int64_t N);
int plan_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* nbr_dst, int32_t* eid_dst,
               int32_t* rowptr_src, int32_t* nbr_src, int32_t* eid_src, int32_t* row32, int32_t* col32, int32_t* work,
               int32_t* status, bool prezeroed, hipStream_t stream);
int plan_rank(int64_t E, int64_t N, const int32_t* rowptr_dst, int32_t* nbr_dst, int32_t* eid_dst, const int32_t* rowptr_src,
              int32_t* nbr_src, int32_t* eid_src, const int32_t* row32, const int32_t* col32, const int32_t* scratch,
              hipStream_t stream);
}

extern "C" int64_t cal_gat_bwd_ws(int64_t N, int64_t E, int64_t K, int64_t D);

// cfg: [F, H, C, L].  Returns an opaque handle (0 on failure).
CAL_EXPORT void* cal_engine_create(int64_t F, int64_t H, int64_t C, int64_t L) {
    if (H % 4 != 0 || H > 256 || H < 4 || L < 0 || L > MAX_LAYERS || F < 1 || C < 1 || C > 64) {
        set_error("cal_engine_create: unsupported shape (need hidden %% 4 == 0, hidden <= 256, layers <= %d, classes <= 64)", MAX_LAYERS);
        return nullptr;
    }
    Engine* e = new Engine();
    memset(e, 0, sizeof(Engine));
    if (hipHostMalloc((void**)&e->host_status, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) *e->host_status = 0;
    else { (void)hipGetLastError(); e->host_status = nullptr; }       // (no mirror: cal_engine_peek_status reports 0, check_status still works)
    { const char* v = getenv("CAL_AMD_RO_STEP"); e->ro_step = !(v && v[0] == '0'); }
    { const char* v = getenv("CAL_AMD_RO_ROWS"); e->ro_rows = !(v && v[0] == '0'); }
    { const char* v = getenv("CAL_AMD_STRIPED"); e->striped = !(v && v[0] == '0'); }
    { const char* v = getenv("CAL_AMD_GWIDE"); e->gwide = !(v && v[0] == '0'); }
    { const char* v = getenv("CAL_AMD_GW_COLS"); e->gw_cols = v ? atoi(v) : 0; }
    { const char* v = getenv("CAL_AMD_FOLD_ZERO"); e->fold_zero = !(v && v[0] == '0'); }
    { const char* v = getenv("CAL_AMD_ATTWIDE"); e->attwide = !(v && v[0] == '0'); }
    { const char* v = getenv("CAL_AMD_RPB_DIV"); e->rpb_div = v ? std::max(1, atoi(v)) : 0; }
    e->bn0_dirty_host = 1;                            // 32 / 64 (experiment): forward slice width forced
    {
        // k_ro_step's 3 * H / 16 workgroups (133 KB of LDS each: one per CU) meet at spin barriers: they must all be
        // resident.  A device (or CU mask / partition) with fewer compute units than that takes the four-kernel readout.
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (cus < 3 * ((int)H / RO_CW) + 8) e->ro_step = 0;
        e->num_cus = cus > 0 ? cus : 256;
        if (cus < 3 * ((int)H / RO_CW) * RBK_MAXRB + 8) e->ro_rows = 0;
    }
    { const char* v = getenv("CAL_AMD_ADAM_FUSED"); e->adam_fused = !(v && v[0] == '0'); }
    e->F = (int)F; e->H = (int)H; e->C = (int)C; e->L = (int)L;
    e->loop_w = 1.f;
    e->grad_scale = 1.f;
    e->p2p_max_polls = 1 << 22;
    e->nbn = (int)L + 9;
    if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) { set_error("cal_engine_create: side stream"); delete e; return nullptr; }
    for (int i = 0; i < 24; ++i) {
        hipEventCreateWithFlags(&e->ev_fork[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming);
    }
    return e;
}

CAL_EXPORT void cal_engine_destroy(void* h) {
    Engine* e = (Engine*)h;
    if (!e) return;
    for (int i = 0; i < 24; ++i) { hipEventDestroy(e->ev_fork[i]); hipEventDestroy(e->ev_join[i]); }
    hipStreamDestroy(e->side);
    if (e->p2p_host_status) hipHostFree(e->p2p_host_status);
    if (e->host_status) hipHostFree(e->host_status);
    delete e;
}

CAL_EXPORT int64_t cal_engine_num_param_slots(void* h) { Engine* e = (Engine*)h; return 3 + (e->gin ? 6 : 4) * e->L + 12 + 24; }
CAL_EXPORT int64_t cal_engine_num_bn(void* h) { return ((Engine*)h)->nbn; }

// Bind the flat parameter / gradient / Adam-state buffers.
//   offs[slot]: float offset of every parameter inside P (and G, M1, M2), slots in the order
//     bn_feat.{weight,bias}, conv_feat.weight,
//     {bns_conv.i.weight, bns_conv.i.bias, convs.i.weight, convs.i.bias} for i < L,
//     edge_att_mlp.{weight,bias}, node_att_mlp.{weight,bias}, bnc.{weight,bias}, bno.{weight,bias},
//     context_convs.{weight,bias}, objects_convs.{weight,bias},
//     {fc1_bn_h.weight, .bias, fc1_h.weight, .bias, fc2_bn_h.weight, .bias, fc2_h.weight, .bias} for h in (c, o, co)
//   bn_ptrs[3*k + {0,1,2}]: running_mean / running_var / num_batches_tracked device addresses of
//     BatchNorm k in the order bn_feat, bns_conv.*, bnc, bno, fc1_bn_c, fc2_bn_c, fc1_bn_o, fc2_bn_o,
//     fc1_bn_co, fc2_bn_co.
//   step, lr: device floats (Adam step counter, learning rate).
CAL_EXPORT int cal_engine_bind(void* h, float* P, float* G, float* M1, float* M2, float* step, float* lr,
                               int64_t nparam, const int64_t* offs, const int64_t* bn_ptrs, float beta1, float beta2,
                               float eps, float weight_decay) {
    Engine* e = (Engine*)h;
    e->P = P; e->G = G; e->M1 = M1; e->M2 = M2; e->step = step; e->lr = lr; e->nparam = nparam;
    e->beta1 = beta1; e->beta2 = beta2; e->eps = eps; e->wd = weight_decay;
    const int L = e->L, H = e->H, F = e->F, C = e->C;
    int s = 0;
    int bn_g[MAX_LAYERS + 9], bn_b[MAX_LAYERS + 9], bn_w[MAX_LAYERS + 9];
    int k = 0;
    bn_g[k] = (int)offs[s++]; bn_b[k] = (int)offs[s++]; bn_w[k] = F; ++k;
    e->o_feat_w = (int)offs[s++];
    for (int i = 0; i < L; ++i) {
        bn_g[k] = (int)offs[s++]; bn_b[k] = (int)offs[s++]; bn_w[k] = H; ++k;
        e->o_conv_w[i] = (int)offs[s++]; e->o_conv_b[i] = (int)offs[s++];
        if (e->gin) { e->o_gin_w2[i] = (int)offs[s++]; e->o_gin_b2[i] = (int)offs[s++]; }
    }
    e->o_eatt_w = (int)offs[s++]; e->o_eatt_b = (int)offs[s++];
    e->o_natt_w = (int)offs[s++]; e->o_natt_b = (int)offs[s++];
    bn_g[k] = (int)offs[s++]; bn_b[k] = (int)offs[s++]; bn_w[k] = H; ++k;   // bnc
    bn_g[k] = (int)offs[s++]; bn_b[k] = (int)offs[s++]; bn_w[k] = H; ++k;   // bno
    e->o_cw = (int)offs[s++]; e->o_cb = (int)offs[s++];
    e->o_ow = (int)offs[s++]; e->o_ob = (int)offs[s++];
    for (int hd = 0; hd < 3; ++hd) {
        bn_g[k] = (int)offs[s++]; bn_b[k] = (int)offs[s++]; bn_w[k] = (hd == 2 && e->cat) ? 2 * H : H; ++k;
        e->o_fc1_w[hd] = (int)offs[s++]; e->o_fc1_b[hd] = (int)offs[s++];
        bn_g[k] = (int)offs[s++]; bn_b[k] = (int)offs[s++]; bn_w[k] = H; ++k;
        e->o_fc2_w[hd] = (int)offs[s++]; e->o_fc2_b[hd] = (int)offs[s++];
    }
    int a = 0;
    for (int i = 0; i < L; ++i) { e->a_convb[i] = a; a += H; }
    e->a_cb = a; a += H;
    e->a_ob = a; a += H;
    e->a_dwn = a; a += H + 4;
    e->a_dwe = a; a += 2 * H + 4;
    e->a_db1 = a; a += 3 * H;
    e->a_db2 = a; a += (3 * C + 3) / 4 * 4;
    e->a_sync = a; a += (RBK_SYNC_INTS + 1) / 2 + 2;  // barrier counters of k_ro_step (ints: 6, or RBK_SYNC_INTS row-blocked), zeroed with the arena
    // the BatchNorm sums last: NSTRIPE planes of them, bn_plane doubles apart (engine.hpp: stripe_sum); BNSlot::arena is plane 0
    a = (a + 3) / 4 * 4;
    const int a_bn = a;
    for (int i = 0; i < e->nbn; ++i) {
        BNSlot& b = e->bn[i];
        b.gamma = bn_g[i]; b.beta = bn_b[i]; b.width = bn_w[i];
        b.rm = (float*)bn_ptrs[3 * i]; b.rv = (float*)bn_ptrs[3 * i + 1]; b.nbt = (int64_t*)bn_ptrs[3 * i + 2];
        b.arena = a;
        a += 4 * ((b.width + 3) / 4 * 4);
    }
    e->bn_plane = a - a_bn;
    a = a_bn + NSTRIPE * e->bn_plane;
    e->arena_n = a;
    (void)C;
    return 0;
}

static size_t engine_layout(Engine* e, int64_t N, int64_t E, int64_t B, bool assign) {
    const size_t H = e->H, C = e->C, L = e->L, F = e->F;
    size_t off = 0;   // in 4-byte units
    auto F32 = [&](float*& p, size_t n) { if (assign) p = (float*)(e->ws) + off; off += al(n); };
    auto I32 = [&](int*& p, size_t n) { if (assign) p = (int*)(e->ws) + off; off += al(n); };
    F32(e->h, (L + 1) * N * H); F32(e->z, N * H); F32(e->zco, 2 * N * H); F32(e->hco, 2 * N * H);
    F32(e->anode, 2 * N); F32(e->pq, 4 * N); F32(e->att, 2 * E); F32(e->dis_unit, N); F32(e->dis_co, 2 * N);
    F32(e->pooled, 2 * B * H); F32(e->pcnt, 2 * B * H); F32(e->xco, 2 * B * H); F32(e->y1, 3 * B * H); F32(e->zl, 3 * B * C); F32(e->logp, 3 * B * C);
    F32(e->stats, 8); F32(e->zpart, 3 * ((H + 15) / 16) * B * C);
    F32(e->dzl, 3 * B * C); F32(e->dyh1, 3 * B * H); F32(e->dy1, 3 * B * H); F32(e->dxh, 4 * B * H); F32(e->dpool, 2 * B * H);
    F32(e->dZco, 2 * N * H); F32(e->gn, 4 * E); F32(e->gself, 4 * N); F32(e->ddeg, 2 * N); F32(e->dl, E);
    F32(e->dzco, 2 * N * H); F32(e->dXhco, 2 * N * H); F32(e->dZ, N * H); F32(e->dzi, (L > 0 ? L : 1) * N * H); F32(e->dXh, N * H);
    // split-K slabs: worst case per weight gradient (S <= 256, but S*tiles ~ 512 => S*M*N <= ~512*64*64 + M*N)
    size_t slab = 0;
    // (per-graph fused backward: one [M,N] slab per graph)
    // (big batches: one slab per 1024-node split-K slice of the 128x128 gradient kernel, gemm_big.hip)
    auto slab_of = [&](size_t M, size_t Nn) {
        const size_t big = gemm_wres_grad((int)M, (int)Nn, (int)N) ? (size_t)gemm_wres_grad_splits((int)N, 1) * M * Nn
                         : gemm_big_grad((int)M, (int)Nn, (int)N) ? (size_t)gemm_big_grad_splits((int)N) * M * Nn : 0;
        return std::max<size_t>(std::max<size_t>(512 * 64 * 64, (size_t)B * M * Nn), big) + 2 * M * Nn;
    };
    slab += slab_of(F, H) + (L + 2) * slab_of(H, H) + 2 * slab_of(H, H) + slab_of(H, 2 * H) + 3 * slab_of(C, H);
    if (e->gin) slab += L * slab_of(H, H);            // two weight gradients per GIN layer
    if (e->K > 0) slab += L * (size_t)1024 * H;      // GATConv: <= 512 partial rows of d att [2H] per layer
    if (assign) e->slab_floats = slab;
    F32(e->slabs, slab);
    {
        float* tmp = nullptr;
        F32(tmp, 2 * (size_t)e->arena_n);
        if (assign) e->arena = (double*)tmp;
        // partial rows of the cross-row sums: <= 1024 producer blocks x (2..6)H columns per set,
        // ~(3L + 16) sets alive per step
        size_t pcap = std::max<size_t>((N + 15) / 16 + 1, (std::min<int64_t>(N, 16384) + 7) / 8 + 1);
        size_t pd = pcap * (size_t)H * 2 * (3 * L + 20);
        F32(tmp, 2 * pd);
        if (assign) { e->parts = (double*)tmp; e->parts_doubles = pd; }
    }
    I32(e->rowptr_dst, N + 1); I32(e->nbr_dst, E); I32(e->eid_dst, E);
    I32(e->rowptr_src, N + 1); I32(e->nbr_src, E); I32(e->eid_src, E);
    I32(e->row32, E); I32(e->col32, E); I32(e->work, 4 * (N + 1) + 4 * E); I32(e->status, 4); I32(e->gptr, B + 1);
    I32(e->iperm, B);
    { int* tmp = nullptr; I32(tmp, 2 * B + 2); if (assign) e->perm_dev = (int64_t*)tmp; }
    I32(e->eptr, B + 1);
    F32(e->coef, 3 * E);
    F32(e->coef_src, E);
    F32(e->wslot, 2 * E);
    if (e->gin) {
        const size_t Lg = L > 0 ? L : 1;
        F32(e->gagg, Lg * N * H); F32(e->gt1, Lg * N * H); F32(e->gy, Lg * N * H); F32(e->ones, N);
    }
    if (e->K > 0) {
        const size_t K = e->K;
        F32(e->gz, (L > 0 ? L : 1) * N * H); F32(e->gsc, (L > 0 ? L : 1) * 4 * al(N * K));
        F32(e->gws, (size_t)cal_gat_bwd_ws(N, E, K, H / K));
    }
    return off * 4;
}

// Switch the backbone to GATConv(H, H/heads, heads, dropout=p) layers (model.py:340,390).  att_offs[i]: float offset of
// convs.i.att [heads, 2*H/heads] inside P / G; seeds[i]: attention-dropout seed of layer i; ctr: device counter the
// engine advances once per training step and folds into the seeds (null: the seeds are used as given -- tests).
// Call before cal_engine_workspace_bytes / cal_engine_set_workspace (the layout grows by the saved GAT activations).
CAL_EXPORT int cal_engine_set_gat(void* h, int64_t heads, float p, float slope, const int64_t* att_offs,
                                  const uint64_t* seeds, void* ctr) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(heads > 0 && e->H % heads == 0, "heads must divide hidden");
    const int64_t D = e->H / heads;
    CAL_REQUIRE(D % 4 == 0 && ((D / 4) & (D / 4 - 1)) == 0, "head dim must be 4 * a power of two");
    CAL_REQUIRE(p >= 0.f && p < 1.f, "dropout p must be in [0,1)");
    e->K = (int)heads; e->gat_p = p; e->gat_slope = slope; e->gat_ctr = (unsigned long long*)ctr;
    for (int i = 0; i < e->L; ++i) { e->o_conv_att[i] = (int)att_offs[i]; e->gat_seed[i] = seeds[i]; }
    return 0;
}

CAL_EXPORT int64_t cal_engine_workspace_bytes(void* h, int64_t N, int64_t E, int64_t B) {
    return (int64_t)engine_layout((Engine*)h, N, E, B, false) + 256;
}

CAL_EXPORT int cal_engine_set_workspace(void* h, void* ws, int64_t bytes, int64_t capN, int64_t capE, int64_t capB) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
    e->ws = (char*)ws;
    e->capN = capN; e->capE = capE; e->capB = capB;
    size_t need = engine_layout(e, capN, capE, capB, true);
    CAL_REQUIRE((int64_t)need <= bytes, "workspace too small");
    e->ws_bytes = bytes;
    e->bn0_dirty_host = 1;          // nothing is known about the new arena: the next step zeroes all of it (k_zero_f64)
    return 0;
}

// float offset (from the workspace base) of a named intermediate, for tests / debugging; -1 if unknown
CAL_EXPORT int64_t cal_engine_buffer_offset(void* h, const char* name) {
    Engine* e = (Engine*)h;
    struct { const char* n; void* p; } tab[] = {
        {"h", e->h}, {"z", e->z}, {"zco", e->zco}, {"hco", e->hco}, {"anode", e->anode}, {"pq", e->pq}, {"att", e->att},
        {"dis_unit", e->dis_unit}, {"dis_co", e->dis_co}, {"pooled", e->pooled}, {"xco", e->xco}, {"y1", e->y1},
        {"zl", e->zl}, {"logp", e->logp}, {"stats", e->stats}, {"dzl", e->dzl}, {"dyh1", e->dyh1}, {"dy1", e->dy1},
        {"dxh", e->dxh}, {"dpool", e->dpool}, {"dZco", e->dZco}, {"gn", e->gn}, {"gself", e->gself}, {"ddeg", e->ddeg},
        {"dl", e->dl}, {"dzco", e->dzco}, {"dXhco", e->dXhco}, {"dZ", e->dZ}, {"dzi", e->dzi}, {"dXh", e->dXh},
        {"arena", e->arena}, {"gptr", e->gptr}, {"status", e->status}, {"eptr", e->eptr},
        {"gt1", e->gin ? e->gt1 : nullptr}, {"gy", e->gin ? e->gy : nullptr},
        {"rowptr_dst", e->rowptr_dst}, {"nbr_dst", e->nbr_dst}, {"eid_dst", e->eid_dst},
        {"rowptr_src", e->rowptr_src}, {"nbr_src", e->nbr_src}, {"eid_src", e->eid_src}, {"perm", e->perm_dev}, {"ones", e->ones},
    };
    for (auto& t : tab)
        if (!strcmp(t.n, name)) return t.p ? ((char*)t.p - e->ws) / 4 : -1;
    return -1;
}

namespace {

struct Ctx {
    Engine* e;
    hipStream_t st;
    int N, B;
    const int64_t* batch;   // [N] graph id of every node (the step's input)
    int T;              // units of the per-graph kernels: tiles of consecutive graphs when the batch is packed, else graphs (= B)
    int64_t E;
    int training;
    int rpb_n, rpb_b;   // rows per block for node-level / graph-level row walkers
    int nfork;          // weight-gradient GEMMs forked to the side stream so far
    const int64_t* y; const int64_t* perm; float wc, wo, wco; int want_grad;
    int draw_perm;      // the step draws its own intervention permutation (first kernel) into Engine::perm_dev
    int tick_in_finish; // the step ends with the Adam update: k_finish advances the step counter
    int ro_done;        // the forward's k_ro_step already ran the readout backward
    int adam_in_finish; // k_finish applies Adam to every gradient it completes (no k_adam launch)
    size_t parts_off;   // bump allocator over Engine::parts
    FinalArgs fin;      // pending k_stats_final tasks
};

BNRef bnref(const Ctx& c, int k, int rows, int update) {
    const Engine* e = c.e;
    const BNSlot& b = e->bn[k];
    BNRef r;
    const int wp = (b.width + 3) / 4 * 4;
    r.sum = e->arena + b.arena; r.sq = e->arena + b.arena + wp;
    r.gamma = e->P + b.gamma; r.beta = e->P + b.beta;
    r.inv_n = 1.0 / (double)rows; r.eps = 1e-5f;
    r.run_mean = b.rm; r.run_var = b.rv; r.nbt = b.nbt;
    r.unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    r.update = (update && c.training) ? 1 : 0;
    r.use_running = c.training ? 0 : 1;
    r.ss = e->bn_plane;
    return r;
}
double* bn_stsum(const Ctx& c, int k) { return c.e->arena + c.e->bn[k].arena; }
double* bn_stsq(const Ctx& c, int k) { return c.e->arena + c.e->bn[k].arena + (c.e->bn[k].width + 3) / 4 * 4; }
double* bn_dsum(const Ctx& c, int k) { return c.e->arena + c.e->bn[k].arena + 2 * ((c.e->bn[k].width + 3) / 4 * 4); }
double* bn_dprod(const Ctx& c, int k) { return c.e->arena + c.e->bn[k].arena + 3 * ((c.e->bn[k].width + 3) / 4 * 4); }

// partial-row accumulator: `cols` columns x P producer blocks, finalised into `dst` by flush_finals
double* parts_alloc(Ctx& c, size_t n) {
    n = (n + 1) & ~(size_t)1;
    if (c.parts_off + n > c.e->parts_doubles) return nullptr;
    double* p = c.e->parts + c.parts_off;
    c.parts_off += n;
    return p;
}
void final_task(Ctx& c, const double* parts, int P, int stride, int n, double* dst) {
    c.fin.t[c.fin.nt++] = FinalTask{parts, P, stride, n, dst};
}
int flush_finals(Ctx& c) {
    if (c.fin.nt == 0) return 0;
    int nmax = 0;
    for (int i = 0; i < c.fin.nt; ++i) nmax = std::max(nmax, c.fin.t[i].n);
    hipLaunchKernelGGL(k_stats_final, dim3(cdiv(nmax, 8), c.fin.nt), dim3(256), 0, c.st, c.fin);
    c.fin.nt = 0;
    cal::g_last_launch = "k_stats_final";
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { set_error("k_stats_final: %s", hipGetErrorString(e_)); return 1; }
    return 0;
}
// aggregation launches: one feature row per lane group when nothing is reduced across rows
// (most waves in flight for the gather), two rows per group when the epilogue carries statistics
int spmm_rpb(int H, bool stats, int N = 1 << 30) {
    // (small batches are latency-bound: a second row per group is a second chain of three dependent gathers, 14 vs 8 us
    //  at 7.5 k nodes -- one row per group there, the partial-row buffer holds N / 8 rows up to 16 k nodes)
    // (big batches: every workgroup leaves one fp64 partial row per statistic -- at two rows per group a 160 k-node batch wrote
    //  2 x 20 000 rows of 2 KB and k_stats_final spent 65 us walking them; rows per workgroup grow with N while the launch keeps
    //  >= 4096 workgroups: 131 + 68 us -> 127 + 17 us per layer at config 5, 16 rows per group lose to the tail: 152 + 9)
    const int rows = 256 / group_for(H, 4);
    return stats && N > 16384 ? std::min(8, std::max(2, N / (rows * 4096))) * rows : rows;
}
Acc spmm_acc(Ctx& c, double* dst, int cols, int rpb) {
    const int P = cdiv(c.N, rpb);
    double* p = parts_alloc(c, (size_t)P * cols);
    if (!p) return Acc(dst);
    final_task(c, p, P, cols, cols, dst);
    return Acc(dst, p, cols);
}

// statistics destination for a node-level kernel with P = cdiv(N, rpb_n) blocks: partial rows
// (finalised right away into dst) when the launch is large, atomics otherwise
Acc node_acc(Ctx& c, double* dst, int cols) {
    const int P = cdiv(c.N, c.rpb_n);
    if ((size_t)P * cols <= 4096) return Acc(dst);
    double* p = parts_alloc(c, (size_t)P * cols);
    if (!p) return Acc(dst);
    final_task(c, p, P, cols, cols, dst);
    return Acc(dst, p, cols);
}
// activation x weight products: the K-split kernel wins on latency for few rows (readouts, 4.6 vs
// 6.9 us at 128 rows), the 64x64 tiled kernel on throughput for node-level row counts
bool use_ks(int M) { return M <= 2048; }
int fwd_gemm(Ctx& c, bool transB, const GemmArgs& a, int nbatch) {
    return use_ks(a.M) ? launch_gemm_ks(transB, a, nbatch, c.st) : launch_gemm(false, transB, a, nbatch, c.st);
}
// same for a GEMM epilogue (two sums per column, P = row tiles)
// st: the site's readers are striped readers (engine.hpp) -- k_gemm's row tiles add into the accumulator planes, no finishing launch
void gemm_stats(Ctx& c, GemmProb& pr, int M, int N, double* d0, double* d1, bool dot, int K = 0, bool st = false) {
    if (dot) { pr.dot_sum = d0; pr.dot_prod = d1; } else { pr.st_sum = d0; pr.st_sq = d1; }
    if (st && (use_ks(M) || (!gemm_wres_rows(M, N, K) && !gemm_big_rows(M, K)))) { pr.st_ss = c.e->bn_plane; return; }
    const int P = use_ks(M) ? gemm_ks_row_tiles(M) : gemm_row_tiles(M, N, K);
    if ((size_t)P * N * 2 <= 4096) return;
    double* p = parts_alloc(c, (size_t)P * 2 * N);
    if (!p) return;
    pr.parts = p;
    final_task(c, p, P, 2 * N, N, d0);
    final_task(c, p + N, P, 2 * N, N, d1);
}

GemmArgs gemm_args(int M, int N, int K, bool transA, bool transB, int relu) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K;
    a.lda = transA ? M : K; a.ldb = transB ? K : N; a.ldc = N;
    a.relu = relu;
    gemm_set_split(a, 1);
    return a;
}

// pick G for H (VEC = 4): lanes per row
template <typename F>
int with_g(int H, F f) {
    int g = group_for(H, 4);
    if (g <= 8) return f(std::integral_constant<int, 8>());
    if (g == 16) return f(std::integral_constant<int, 16>());
    if (g == 32) return f(std::integral_constant<int, 32>());
    return f(std::integral_constant<int, 64>());
}

int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// k_espmm launch: lanes per row from the width, per-edge weights / output statistics from the branch (both branches alike)
int launch_espmm(hipStream_t st, const CSR& csr, const SpmmBranch2& bb, int nbranch, int relu, float loop_w, int N, int H, int rpb);

#define RC0(x) do { int rc0_ = (x); if (rc0_) return rc0_; } while (0)
// weight-gradient GEMM (TN) with split-K slabs when the reduction axis is long
int grad_gemm(Ctx& c, GemmArgs& a, int nbatch, float** dst, FinishArgs& fa, size_t& slab_off) {
    int S = splitk_for(a.M, a.N, a.K, nbatch);
    gemm_set_split(a, S);
    if (a.nsplit > 1) {
        for (int b = 0; b < nbatch; ++b) {
            size_t need = (size_t)a.nsplit * a.M * a.N;
            if (slab_off + need > c.e->slab_floats || fa.nst >= MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
            a.p[b].C = c.e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{c.e->slabs + slab_off, dst[b], a.M * a.N, a.nsplit};
            slab_off += need;
        }
    } else {
        for (int b = 0; b < nbatch; ++b) a.p[b].C = dst[b];
    }
    // fork: everything enqueued on the main stream so far (the producers of A and B) precedes it
    hipStream_t s = c.st;
    // Measured on MI355X / ROCm 7.2: forking the 9 dW GEMMs costs more than it hides (graph replay
    // 514 -> 648 us per step, eager 523 -> 554 us: cross-stream edges serialise through heavier
    // barrier packets), so the fork is compiled in but disabled.
    constexpr bool kForkWeightGrads = false;
    if (kForkWeightGrads && c.nfork < 24) {
        hipEventRecord(c.e->ev_fork[c.nfork], c.st);
        hipStreamWaitEvent(c.e->side, c.e->ev_fork[c.nfork], 0);
        s = c.e->side;
    }
    int rc = launch_gemm(true, false, a, nbatch, s);
    if (s != c.st) { hipEventRecord(c.e->ev_join[c.nfork], s); c.nfork++; }
    return rc;
}
// the main stream waits for every forked GEMM (before the final commit, and on any early exit)
void join_side(Ctx& c) {
    for (int i = 0; i < c.nfork; ++i) hipStreamWaitEvent(c.st, c.e->ev_join[i], 0);
    c.nfork = 0;
}

// dX (NT, `ax`) and dW (TN, `aw`, split-K slabs like grad_gemm) of one layer in a single launch when both
// go to the tiled kernel; otherwise the two launches of before.
int dual_gemm(Ctx& c, GemmArgs& ax, int nbx, GemmArgs& aw, int nbw, float** dst, FinishArgs& fa, size_t& slab_off) {
    if (use_ks(ax.M)) {
        RC0(grad_gemm(c, aw, nbw, dst, fa, slab_off));
        return fwd_gemm(c, true, ax, nbx);
    }
    int S = splitk_for(aw.M, aw.N, aw.K, nbw);
    gemm_set_split(aw, S);
    if (aw.nsplit > 1) {
        for (int b = 0; b < nbw; ++b) {
            size_t need = (size_t)aw.nsplit * aw.M * aw.N;
            if (slab_off + need > c.e->slab_floats || fa.nst >= MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
            aw.p[b].C = c.e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{c.e->slabs + slab_off, dst[b], aw.M * aw.N, aw.nsplit};
            slab_off += need;
        }
    } else {
        for (int b = 0; b < nbw; ++b) aw.p[b].C = dst[b];
    }
    return launch_gemm_dual(ax, nbx, aw, nbw, c.st);
}

// live kernel timing for bench.py's roofline block: when enabled, HIP events are attached to the backbone's kernels:
// unfused forward GEMM ([N,H]x[H,H], class 0), aggregation (k_espmm forward / transposed backward, class 1), per-graph
// fused convolution forward (class 2, flops), the backward's dX + dW dual GEMM (class 3, flops), the per-graph fused
// backward (class 4, flops), the GATConv classes 5-8.  Single-kernel scopes (`single`) launch their kernel with
// hipExtLaunchKernelGGL, which stamps the two events with the START and END of that dispatch -- the kernel's own
// duration, what rocprofv3 reports; event records on the stream around the launch also measured the two marker packets
// and the gap between them (20.1 against 16.1 us for k_gconv_bwd).  Multi-kernel scopes (the GEMM helpers of gemm*.hip,
// the GAT backward) keep the records around the launches.  Events are created here (never in a normal step).
struct ProfRec { hipEvent_t e0, e1; int cls; double work; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
struct ProfScope;
static ProfScope* g_prof_cur = nullptr;
struct ProfScope {
    hipStream_t st; bool on, single, used; ProfRec r;
    ProfScope(hipStream_t s, int cls, double work, bool single_kernel = false) : st(s), on(g_prof_on), single(single_kernel), used(false) {
        if (!on) return;
        r.cls = cls; r.work = work;
        hipEventCreate(&r.e0); hipEventCreate(&r.e1);
        if (single) g_prof_cur = this; else hipEventRecord(r.e0, st);
    }
    ~ProfScope() {
        if (!on) return;
        if (single) { g_prof_cur = nullptr; if (used) g_prof.push_back(r); }
        else { hipEventRecord(r.e1, st); g_prof.push_back(r); }
    }
};
// launch inside a single-kernel ProfScope: the dispatch carries the scope's events when profiling is on
#define PROF_LAUNCH(kernel, grid, block, shmem, stream, ...) do { \
        if (g_prof_cur) { hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, g_prof_cur->r.e0, g_prof_cur->r.e1, 0, __VA_ARGS__); g_prof_cur->used = true; } \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); } while (0)

int launch_espmm(hipStream_t st, const CSR& csr, const SpmmBranch2& bb, int nbranch, int relu, float loop_w, int N, int H, int rpb) {
    const bool wt = bb.b[0].w != nullptr, stt = bb.b[0].st_sum.on(), sd = bb.b[0].sd_z != nullptr;
    if (wt != (bb.b[nbranch - 1].w != nullptr) || stt != bb.b[nbranch - 1].st_sum.on() || sd != (bb.b[nbranch - 1].sd_z != nullptr) || rpb % 4 != 0 || H > 256) {
        set_error("k_espmm: branches differ in kind"); return 2;
    }
    const dim3 grid(cdiv(N, rpb), nbranch);
    if (sd) {       // transposed weighted aggregation + SDDMM
        if (!wt || stt) { set_error("k_espmm: the fused SDDMM is built for the weighted branches without statistics"); return 2; }
        const bool pb = bb.b[0].pb_g != nullptr;
        if (pb != (bb.b[nbranch - 1].pb_g != nullptr)) { set_error("k_espmm: branches differ in kind"); return 2; }
        return with_g(H, [&](auto g) {
            constexpr int G = decltype(g)::value;
            if (pb) PROF_LAUNCH((k_espmm<4, G, true, false, true, true>), grid, dim3(256), 0, st, csr, bb, relu, loop_w, N, H, rpb);
            else PROF_LAUNCH((k_espmm<4, G, true, false, true>), grid, dim3(256), 0, st, csr, bb, relu, loop_w, N, H, rpb);
            return 0;
        });
    }
    return with_g(H, [&](auto g) {
        constexpr int G = decltype(g)::value;
        if (wt && stt) PROF_LAUNCH((k_espmm<4, G, true, true>), grid, dim3(256), 0, st, csr, bb, relu, loop_w, N, H, rpb);
        else if (wt) PROF_LAUNCH((k_espmm<4, G, true, false>), grid, dim3(256), 0, st, csr, bb, relu, loop_w, N, H, rpb);
        else if (stt) PROF_LAUNCH((k_espmm<4, G, false, true>), grid, dim3(256), 0, st, csr, bb, relu, loop_w, N, H, rpb);
        else PROF_LAUNCH((k_espmm<4, G, false, false>), grid, dim3(256), 0, st, csr, bb, relu, loop_w, N, H, rpb);
        return 0;
    });
}

// per-graph fused convolution (engine_gconv.hpp): needs the batch's per-graph bounds from the host
bool use_gc(const Ctx& c) {
    const Engine* e = c.e;
    return e->max_nodes > 0 && e->max_nodes <= GC_T && e->max_edges <= GC_E && e->H % GC_N == 0 && e->H <= GC_K && c.B > 0;
}
bool gc_small(const Ctx& c) { return c.e->max_nodes <= 64 && c.e->max_edges <= gc_edge_cap(64); }
// wide per-graph convolutions, both ways (engine_gwide.hpp): graphs of 129 .. 256 nodes (the reference's default SPMotif shape)
bool use_gw(const Ctx& c) {
    const Engine* e = c.e;
    return e->gwide && e->max_nodes > GC_T && e->max_nodes <= GW_T && e->max_edges <= GW_E && e->H % GC_N == 0 && e->H <= GW_K && c.B > 0 &&
           e->ntiles == 0 && e->K == 0 && !e->gin;
}
// the attention block between the backbone and the causal convolutions per graph for the wide shapes (engine_attwide.hpp): needs the
// per-graph plan's facts (a graph's edges are one run of edge_index columns, no self loops: slot ranges = edge ranges)
bool use_aw(const Ctx& c) {
    const Engine* e = c.e;
    return use_gw(c) && e->attwide && e->node_ptr && e->edge_ptr && e->max_nodes <= AW_T && e->max_edges <= AW_E && e->H <= 256 && c.T <= 512;
}
// per-graph fused backward (engine_gconv_bwd.hpp): 64-node graphs only
bool use_gcb(const Ctx& c) { return use_gc(c) && gc_small(c) && c.T <= 128 * 4; }
// Which BatchNorm sites of a step go through the accumulator planes (engine.hpp: stripe_sum): those whose producers are per-graph
// kernels and whose EVERY reader is a striped reader (k_gconv_fwd, k_gconv_bwd, k_att_bwd_graph, k_feat_bwd*, the final commit).
//   co: bnc / bno -- statistics from k_att_fwd_graph, backward sums from the two-branch k_gconv_bwd; read by the two-branch
//       k_gconv_fwd / k_gconv_bwd and by k_att_bwd_graph (the node-level k_att_bwd of > 256 units is a plain reader)
//   bb: the GCNConv backbone -- statistics of layers 2..L from k_gconv_fwd, backward sums of layers L..1 from k_gconv_bwd
//       (a GAT / GIN backbone has its own per-graph kernels: plain readers)
bool att_graph_fwd(const Ctx& c) {
    const Engine* e = c.e;
    return use_gc(c) && e->max_nodes <= 4 * (512 / group_for(e->H, 4)) && e->max_edges <= GP_E;
}
bool striped_co(const Ctx& c) { return c.e->striped && c.training && att_graph_fwd(c) && use_gcb(c) && c.T <= 256; }
bool striped_bb(const Ctx& c) { return c.e->striped && c.training && use_gc(c) && use_gcb(c) && c.e->K == 0 && !c.e->gin; }
//   gat: the same for a GATConv backbone whose layers run per graph both ways (k_ggat_fwd / k_ggat_bwd)
bool striped_gat(const Ctx& c) {
    const Engine* e = c.e;
    if (!(e->striped && c.training && e->K > 0 && use_gc(c) && gc_small(c) && use_gcb(c))) return false;
    const int D = e->H / e->K;
    return (D == 32 || D == 64) && e->max_edges <= GG_E && e->max_edges <= GGB_E;
}
//   gin: a GINConv backbone whose layers run per graph both ways (k_ggin_fwd<1|2> / k_ggin_bwd<2|1>): the BatchNorm inside the layer
bool striped_gin(const Ctx& c) {
    const Engine* e = c.e;
    return e->striped && c.training && e->gin && use_gc(c) && gc_small(c) && use_gcb(c) && e->max_edges <= GB_E && e->max_edges <= gc_edge_cap(64);
}
//   node: the node-level GCNConv chain of a SMALL batch (graphs beyond the per-graph kernels: configs[0], 230-250 nodes each) -- the
//       sums that leave a GEMM epilogue (feature-layer statistics, every backward sum) go into the planes; their readers are the
//       small-batch GEMMs' prologues (gemm.hip / gemm_ks.hip: striped readers), k_bn_bwd<.., ST> and k_att_bwd<.., ST>.  The
//       aggregation / attention kernels' statistics keep their partial rows (thousands of short workgroups).
bool striped_node(const Ctx& c) {
    const Engine* e = c.e;
    return e->striped && c.training && !use_gc(c) && e->K == 0 && !e->gin && c.N < 16384 && c.N > 0;
}
// the feature layer's output statistics (BatchNorm_1): read by the first backbone layer both ways and by k_feat_bwd*
bool striped_feat(const Ctx& c) { return ((striped_bb(c) || striped_gat(c)) && c.e->F <= FM_F && c.e->H <= FB_H) || striped_node(c); }
// partial-row statistics of a per-graph kernel: one row per graph; st: into the workgroup's accumulator plane, no finishing launch
Acc graph_acc(Ctx& c, double* dst, int cols, bool st = false) {
    if (st) return Acc(dst, nullptr, c.e->bn_plane);
    double* p = parts_alloc(c, (size_t)c.T * cols);
    if (!p) return Acc(dst);
    final_task(c, p, c.T, cols, cols, dst);
    return Acc(dst, p, cols);
}

// Launch the per-graph fused backward for nb branches: slabs (one per graph) and the BatchNorm-backward partial
// rows are registered like those of the GEMM path.  gb[k].dot_parts / .slab are filled in here.
int gconv_bwd(Ctx& c, const CSR& gd, GconvBwdBranch* gb, int nb, float** dst, double** dsum, double** dprod,
              FinishArgs& fa, size_t& slab_off, bool rs, bool st, const CSR* gs_wide = nullptr) {
    Engine* e = c.e;
    const int H = e->H, B = c.T, nsl = H / GC_N;         // (B: units of this launch -- tiles or graphs)
    // wide kernels: the dX' and dW products of a (graph, slice) go to two or four workgroups when one each would leave CUs without one
    const int nsplit = !gs_wide ? 1 : (int64_t)B * nsl * nb * 4 <= e->num_cus ? 4 : (int64_t)B * nsl * nb * 2 <= e->num_cus ? 2 : 1;
    const int np2 = nsplit == 4 ? 2 : 1;                 // workgroups per (graph, slice) that leave BatchNorm-backward partial rows
    for (int k = 0; k < nb; ++k) {
        gb[k].batch = c.batch; gb[k].tile_gptr = e->ntiles > 0 ? e->tile_gptr : nullptr;
        const size_t need = (size_t)B * H * H;
        if (slab_off + need > e->slab_floats || fa.nst >= MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
        gb[k].slab = e->slabs + slab_off;
        fa.st[fa.nst++] = SlabTask{e->slabs + slab_off, dst[k], H * H, B};
        slab_off += need;
        if (st) {        // the BatchNorm-backward sums go into the accumulator planes of the site (every reader adds them)
            gb[k].dacc_sum = dsum[k]; gb[k].dacc_prod = dprod[k]; gb[k].dacc_ss = e->bn_plane;
            continue;
        }
        double* p = parts_alloc(c, (size_t)B * nsl * np2 * 2 * H);
        if (!p) { set_error("engine: partial-row workspace exhausted"); return 2; }
        gb[k].dot_parts = p;
        final_task(c, p, B * nsl * np2, 2 * H, H, dsum[k]);
        final_task(c, p + H, B * nsl * np2, 2 * H, H, dprod[k]);
    }
    if (gs_wide) {
        // graphs of up to 256 nodes (engine_gwide.hpp): CSR by SOURCE
        const dim3 gridw(B, nsl * nsplit, nb);
        const GconvBwdBranch2 b2{{gb[0], gb[nb - 1]}};
        if (rs) PROF_LAUNCH((k_gw_bwd<true, 2>), gridw, dim3(GW_NT), 0, c.st, *gs_wide, e->gptr, e->eptr, b2, e->loop_w, c.N, H, H, nsplit, e->status);
        else if (gb[0].dout) PROF_LAUNCH((k_gw_bwd<false, 0>), gridw, dim3(GW_NT), 0, c.st, *gs_wide, e->gptr, e->eptr, b2, e->loop_w, c.N, H, H, nsplit, e->status);
        else PROF_LAUNCH((k_gw_bwd<false, 1>), gridw, dim3(GW_NT), 0, c.st, *gs_wide, e->gptr, e->eptr, b2, e->loop_w, c.N, H, H, nsplit, e->status);
        CAL_CHECK_LAUNCH("k_gw_bwd");
        return 0;
    }
    const dim3 grid(B, nsl, nb);
    // the two-branch launch: 2 x B x H/64 workgroups -- at 128 graphs twice the CUs; the 80 KB instantiation runs them as ONE round
    // (its gn goes out in slot order: only when the per-graph attention backward consumes it)
    const bool lean2 = rs && (int64_t)B * nsl * nb > e->num_cus && gb[0].gn_slot && gb[nb - 1].gn_slot;
    if (lean2 && e->ntiles > 0) PROF_LAUNCH((k_gconv_bwd<true, 2, true, true>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    else if (lean2) PROF_LAUNCH((k_gconv_bwd<true, 2, false, true>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    else if (rs && e->ntiles > 0) PROF_LAUNCH((k_gconv_bwd<true, 2, true>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    else if (rs) PROF_LAUNCH((k_gconv_bwd<true, 2>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    // more workgroups than CUs (packed batches, batches of > 128 graphs): the 80 KB instantiation, two workgroups per CU
    else if (gb[0].dout && (int64_t)B * nsl * nb > e->num_cus) PROF_LAUNCH((k_gconv_bwd<false, 0, false, true>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    else if (gb[0].dout) PROF_LAUNCH((k_gconv_bwd<false, 0>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    else if ((int64_t)B * nsl * nb > e->num_cus) PROF_LAUNCH((k_gconv_bwd<false, 1, false, true>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    else PROF_LAUNCH((k_gconv_bwd<false, 1>), grid, dim3(GB_NT), 0, c.st, gd, e->gptr, e->eptr, GconvBwdBranch2{{gb[0], gb[nb - 1]}}, e->loop_w, c.N, H, H, e->status);
    CAL_CHECK_LAUNCH("k_gconv_bwd");
    return 0;
}

// the one-launch readout in row blocks (k_ro_step<true>): training steps of 129 .. 512 graphs
bool use_ro_rows(const Ctx& c) {
    const Engine* e = c.e;
    const int B = c.B, H = e->H, C = e->C;
    return e->ro_rows && e->ro_step && c.training && !e->cat && B > RS_B && B <= RBK_MAXRB * RS_B && H <= RS_K && H % RO_CW == 0 &&
           RS_B * C <= 2048 && C <= 64 && (size_t)3 * cdiv(B, RS_B) * ((size_t)H * H + (size_t)C * H) <= e->slab_floats;
}
bool use_ro(const Ctx& c) {
    const int B = c.B, H = c.e->H, C = c.e->C;
    const int B4 = (B + 15) & ~15;
    if (c.e->cat) return false;                     // the 2H-wide co head takes the GEMM path
    return (size_t)B4 * (H + 4) <= (size_t)RO_LDS && B * H <= 16384 && H % RO_CW == 0 && B4 <= 256 && H <= 256 &&
           B * C <= 2048 && C <= 64;
}
RoArgs make_ro(const Ctx& c) {
    Engine* e = c.e;
    const int L = e->L, H = e->H, C = e->C, B = c.B;
    RoArgs a;
    memset(&a, 0, sizeof(a));
    for (int hd = 0; hd < 3; ++hd) {
        RoHead& h = a.h[hd];
        const int k1 = L + 3 + 2 * hd, k2 = L + 4 + 2 * hd;
        h.W1 = e->P + e->o_fc1_w[hd]; h.b1 = e->P + e->o_fc1_b[hd];
        h.W2 = e->P + e->o_fc2_w[hd]; h.b2 = e->P + e->o_fc2_b[hd];
        h.bn1 = bnref(c, k1, B, 1); h.bn2 = bnref(c, k2, B, 1);
        h.st2_sum = bn_stsum(c, k2); h.st2_sq = bn_stsq(c, k2);
        h.d1_sum = bn_dsum(c, k1); h.d1_prod = bn_dprod(c, k1);
        h.d2_sum = bn_dsum(c, k2); h.d2_prod = bn_dprod(c, k2);
        h.db1 = e->arena + e->a_db1 + hd * H; h.db2 = e->arena + e->a_db2 + hd * C;
        h.gW1 = e->G + e->o_fc1_w[hd]; h.gW2 = e->G + e->o_fc2_w[hd];
    }
    a.pooled = e->pooled; a.perm = c.perm; a.iperm = e->iperm; a.xco = e->xco; a.y1 = e->y1;
    a.zl = e->zl; a.logp = e->logp; a.dzl = e->dzl; a.dy1 = e->dy1; a.dxin = e->dxh;
    a.rowloss = e->dyh1;                  // [6,B] scratch: the unfused path's dyh1 is idle here
    a.y = c.y; a.stats = e->stats; a.B = B; a.H = H; a.C = C;
    a.wc = c.wc; a.wo = c.wo; a.wco = c.wco; a.training = c.training; a.want_grad = c.want_grad;
    return a;
}

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
// profiling aid: cal_engine_debug_stop(k) makes the step return after its k-th launch site (0 = run all)
static int g_stop_after = 0;
static int g_stage = 0;
static std::vector<const char*> g_stage_names;      // launch-site names of the latest full step
#define STAGE() do { if (g_stop_after == 0) g_stage_names.push_back(cal::g_last_launch); \
                     if (g_stop_after > 0 && ++g_stage >= g_stop_after) return -12345; } while (0)

// launch helper of the row-wise GIN passes (engine_gin.hpp): mode 0 forward BN + ReLU, 1 masked BatchNorm-backward sums,
// 2 BatchNorm backward behind the ReLU, 3 ReLU mask + column sums
int gin_rows(Ctx& c, int mode, const GinRowArgs& ga) {
    const int N = c.N, H = c.e->H;
    hipStream_t st = c.st;
    return with_g(H, [&](auto g) {
        constexpr int G = decltype(g)::value;
        const dim3 grid(cdiv(N, c.rpb_n));
        if (mode == 0) hipLaunchKernelGGL((k_gin_rows<4, G, 0>), grid, dim3(256), 0, st, ga, N, H, c.rpb_n);
        else if (mode == 1) hipLaunchKernelGGL((k_gin_rows<4, G, 1>), grid, dim3(256), 0, st, ga, N, H, c.rpb_n);
        else if (mode == 2) hipLaunchKernelGGL((k_gin_rows<4, G, 2>), grid, dim3(256), 0, st, ga, N, H, c.rpb_n);
        else hipLaunchKernelGGL((k_gin_rows<4, G, 3>), grid, dim3(256), 0, st, ga, N, H, c.rpb_n);
        return 0;
    });
}

// narrow feature matrices of big batches: the feature layer as row kernels (engine_feat.hpp) instead of tiled GEMMs
bool feat_rows(const Ctx& c) {
    const Engine* e = c.e;
    return e->F <= FEAT_FMAX && c.N > 16384 && e->H % 4 == 0 && e->H <= 256 && group_for(e->H, 4) >= FEAT_FMAX;
}
template <typename Fn>
int with_g_fp(int H, int F, Fn f) {
    return with_g(H, [&](auto g) {
        constexpr int G = decltype(g)::value;
        if constexpr (G >= FEAT_FMAX) {
            if (F <= 4) return f(g, std::integral_constant<int, 4>());
            if (F <= 8) return f(g, std::integral_constant<int, 8>());
            if (F <= 12) return f(g, std::integral_constant<int, 12>());
            return f(g, std::integral_constant<int, 16>());
        } else {
            set_error("feature row kernels: hidden width too small"); return 2;
        }
    });
}

// add-pool of the two causal branches (model.py:115-116) + the per-graph positive counts the node-level backward uses
int launch_pool2(Ctx& c) {
    Engine* e = c.e;
    const int N = (int)c.N, B = (int)c.B, H = e->H;
    const size_t NH = (size_t)N * H;
    const int tc = std::min(256, pow2ceil(H / 4));
    // few large graphs: split every graph's rows over S workgroups (slices parked in dZco, a backward-only buffer that the
    // backward itself no longer writes; slices of >= 128 rows: at S = 4 the 256 workgroups of config 5 read its 328 MB at 3.3 TB/s)
    const int S = std::max(1, std::min(32, N / std::max(1, B * 128)));
    hipLaunchKernelGGL((k_pool2<4>), dim3(B, 2, S), dim3(256), 0, c.st, e->hco, e->hco + NH, e->gptr, e->pooled,
                       e->pooled + (size_t)B * H, H, tc, e->dZco, e->pcnt);
    CAL_CHECK_LAUNCH("k_pool2");
    if (S > 1) {
        hipLaunchKernelGGL(k_pool2_sum, dim3(cdiv(4 * (int64_t)B * H, 256)), dim3(256), 0, c.st, e->dZco, S, 2 * (int64_t)B * H, e->pooled, e->pcnt);
        CAL_CHECK_LAUNCH("k_pool2_sum");
    }
    return 0;
}
// the counts alone (forward pooled per graph inside k_gconv_fwd, backward node-level)
int launch_pool_cnt(Ctx& c) {
    Engine* e = c.e;
    const int N = (int)c.N, B = (int)c.B, H = e->H;
    hipLaunchKernelGGL(k_zero_f32, dim3(cdiv(2 * (int64_t)B * H, 256)), dim3(256), 0, c.st, e->pcnt, 2 * (int64_t)B * H);
    CAL_CHECK_LAUNCH("k_zero_f32");
    return with_g(H, [&](auto g) {
        constexpr int G = decltype(g)::value;
        hipLaunchKernelGGL((k_pool_cnt<4, G>), dim3(cdiv(N, c.rpb_n), 2), dim3(256), 0, c.st, e->hco, e->hco + (size_t)N * H, c.batch, e->pcnt, N, B, H, c.rpb_n);
        CAL_CHECK_LAUNCH("k_pool_cnt");
        return 0;
    });
}

int engine_forward(Ctx& c, const float* x0, const int64_t* edge_index, const int64_t* batch, const int64_t* y,
                   const int64_t* perm, float wc, float wo, float wco, int want_grad) {
    Engine* e = c.e;
    const int N = c.N, B = c.B, T = c.T, H = e->H, F = e->F, C = e->C, L = e->L;
    const int64_t E = c.E;
    hipStream_t st = c.st;
    const size_t NH = (size_t)N * H;
    const int64_t* tgp = e->ntiles > 0 ? e->tile_gptr : nullptr;
    // per-graph plan (engine_plan.hpp) when the host vouches for the batch layout
    const bool fast_plan = e->node_ptr && e->edge_ptr && B > 0 && e->max_nodes > 0 && e->max_nodes <= GP_T2 && e->max_edges <= GP_E2;
    const bool wide_plan = fast_plan && (e->max_nodes > GP_T || e->max_edges > GP_E);
    // ... and for graphs of up to 8192 nodes (config 5): one 1024-thread workgroup per graph, rows ranked from a global scratch
    const bool big_plan = !fast_plan && e->node_ptr && e->edge_ptr && B > 0 && e->ntiles == 0 && e->max_nodes > 0 && e->max_nodes <= GPB_T;
    const bool plan_stats = (fast_plan || big_plan) && c.training && F <= 64;      // bn_feat's statistics ride in the plan kernel
    // 0. zero the fp64 arena and the GraphPlan counters (one kernel, not memset nodes) -- or, for a forward + backward step on
    //    the per-graph plan, nothing: those duties ride in k_plan_graph (engine_plan.hpp: PlanFold) and bn_feat's statistics
    //    range was zeroed by the previous step's k_finish
    const bool fold = e->fold_zero && fast_plan && c.training && c.want_grad && !e->bn0_dirty_host && g_stop_after == 0;
    if (c.training) e->bn0_dirty_host = 1;          // (cleared where k_finish is enqueued)
    if (!fold) {
        const int64_t ni = (fast_plan || big_plan) ? 0 : 4 * ((int64_t)N + 1);
        hipLaunchKernelGGL(k_zero_f64, dim3(cdiv(std::max<int64_t>(e->arena_n, ni), 256) + (c.draw_perm ? cdiv(B, ZP_EPB) : 0)), dim3(256), 0, st,
                           e->arena, (int64_t)e->arena_n, e->work, ni, e->status, (e->K > 0 && c.training) ? e->gat_ctr : nullptr,
                           c.draw_perm ? e->perm_dev : nullptr, B, e->perm_seed, e->perm_ctr, c.adam_in_finish ? e->step : nullptr,
                           c.training ? e->status + 3 : nullptr);
        CAL_CHECK_LAUNCH("k_zero_f64"); STAGE();
    }
    // 1. GraphPlan
    if (fast_plan) {
        auto kern = wide_plan ? k_plan_graph<GP_T2, GP_E2, 1024> : k_plan_graph<GP_T, GP_E, 256>;
        const int nt = wide_plan ? 1024 : 256;
        PlanFold pf;
        memset(&pf, 0, sizeof(pf));
        if (fold) {
            const int wp0 = (e->bn[0].width + 3) / 4 * 4;
            pf.on = 1; pf.arena = e->arena; pf.n = e->arena_n; pf.skip_lo = e->bn[0].arena; pf.skip_hi = e->bn[0].arena + 2 * wp0;
            pf.gat_tick = e->K > 0 ? e->gat_ctr : nullptr; pf.adam_step = c.adam_in_finish ? e->step : nullptr;
            pf.perm = c.draw_perm ? e->perm_dev : nullptr; pf.permB = B; pf.seed = e->perm_seed; pf.perm_ctr = e->perm_ctr;
            pf.dirty = e->status + 3;
        }
        hipLaunchKernelGGL(kern, dim3(T + ((fold && c.draw_perm) ? cdiv(B, nt / 4) : 0)), dim3(nt), 0, st, edge_index, E, N, T, e->node_ptr, e->edge_ptr, batch, tgp, B, e->loop_w,
                           e->rowptr_dst, e->nbr_dst, e->eid_dst, e->rowptr_src, e->nbr_src, e->eid_src, e->row32, e->col32,
                           e->gptr, e->eptr, e->dis_unit, e->status, plan_stats ? x0 : nullptr, F, bn_stsum(c, 0), bn_stsq(c, 0),
                           use_gw(c) ? e->coef : nullptr, use_gw(c) ? e->coef_src : nullptr, pf);
        CAL_CHECK_LAUNCH("k_plan_graph"); STAGE();
    } else if (big_plan) {
        int* scratch = e->work + 4 * ((size_t)e->capN + 1);
        hipLaunchKernelGGL(k_plan_big, dim3(B, 2), dim3(1024), 0, st, edge_index, E, N, B, e->node_ptr, e->edge_ptr, batch, e->loop_w,
                           e->rowptr_dst, e->rowptr_src, e->row32, e->col32, e->gptr, e->eptr, e->dis_unit, e->status, scratch,
                           plan_stats ? x0 : nullptr, F, bn_stsum(c, 0), bn_stsq(c, 0));
        CAL_CHECK_LAUNCH("k_plan_big"); STAGE();
        RC(plan_rank(E, N, e->rowptr_dst, e->nbr_dst, e->eid_dst, e->rowptr_src, e->nbr_src, e->eid_src, e->row32, e->col32, scratch, st)); STAGE();
    } else {
        RC(plan_build(edge_index, E, N, e->rowptr_dst, e->nbr_dst, e->eid_dst, e->rowptr_src, e->nbr_src, e->eid_src,
                      e->row32, e->col32, e->work, e->status, true, st)); STAGE();
        hipLaunchKernelGGL(k_gptr_dis, dim3(cdiv(N + 1, 256)), dim3(256), 0, st, batch, N, B, e->gptr, e->rowptr_src, e->loop_w,
                           e->dis_unit, e->status, e->rowptr_dst, e->eptr);
        CAL_CHECK_LAUNCH("k_gptr_dis"); STAGE();
    }
    const CSR gd{e->rowptr_dst, e->nbr_dst, e->eid_dst, (int)c.E}, gs{e->rowptr_src, e->nbr_src, e->eid_src, (int)c.E};
    (void)gs;
    // 2. bn_feat statistics (model.py:90)
    if (c.training && !plan_stats) {
        // wide feature matrices (one-hot degrees: F = 139 at the NCI1-like shape): many short blocks whose column sums go to
        // partial rows (k_stats_final) -- F fp64 atomics per block on the same 2 F addresses cost more than the reads
        int tc = std::min(256, pow2ceil(F));
        const bool wide = (size_t)cdiv(N, 32) * F > 16384;
        int rpb = wide ? std::max(32, cdiv(N, 1024)) : std::max(128, cdiv(N, 256));
        const Acc a0 = wide ? spmm_acc(c, bn_stsum(c, 0), F, rpb) : Acc(bn_stsum(c, 0));
        const Acc a1 = wide ? spmm_acc(c, bn_stsq(c, 0), F, rpb) : Acc(bn_stsq(c, 0));
        hipLaunchKernelGGL(k_colstats, dim3(cdiv(N, rpb)), dim3(256), 0, st, x0, N, F, tc, rpb, a0, a1);
        CAL_CHECK_LAUNCH("k_colstats"); STAGE();
        if (wide) { RC(flush_finals(c)); STAGE(); }
    }
    // 3. h0 = relu(BN(x0) @ W_feat)   (model.py:90-91, gcn_conv.py:75-77)
    if (feat_rows(c)) {
        FeatFwdArgs fw;
        fw.x0 = x0; fw.W = e->P + e->o_feat_w; fw.out = e->h; fw.bn0 = bnref(c, 0, N, 1);
        if (c.training && L > 0 && !e->gin) { fw.st_sum = node_acc(c, bn_stsum(c, 1), H); fw.st_sq = node_acc(c, bn_stsq(c, 1), H); }
        RC(with_g_fp(H, F, [&](auto g, auto fp) {
            hipLaunchKernelGGL((k_feat_fwd_rows<decltype(g)::value, decltype(fp)::value>), dim3(cdiv(N, c.rpb_n)), dim3(256), 0, st, fw, N, H, F, c.rpb_n);
            return 0;
        }));
        CAL_CHECK_LAUNCH("k_feat_fwd_rows"); STAGE();
        RC(flush_finals(c)); STAGE();
    } else {
        GemmArgs a = gemm_args(N, H, F, false, false, 1);
        a.p[0].A = x0; a.p[0].B = e->P + e->o_feat_w; a.p[0].C = e->h;
        a.p[0].xa.has_bn = 1; a.p[0].xa.bn = bnref(c, 0, N, 1);
        // (a GIN layer starts with the aggregation, not a BatchNorm; striped: BatchNorm_1 is read by k_gconv_fwd / k_gconv_bwd / k_feat_bwd* only)
        if (c.training && L > 0 && !e->gin) gemm_stats(c, a.p[0], N, H, bn_stsum(c, 1), bn_stsq(c, 1), false, F, striped_feat(c));
        RC(fwd_gemm(c, false, a, 1)); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    // 4. backbone: h_i = relu(A_hat (BN_i(h_{i-1}) @ W_i) + b_i)   (model.py:93-95)
    const bool gc = use_gc(c);
    const bool gw = use_gw(c);
    const bool gw_st = gw && e->striped && c.training;   // the wide kernels' BatchNorm sums go through the accumulator planes (every reader is a striped reader)
    // 32-column slices (two workgroups per CU) when 64-column slices would leave half of the CUs without a workgroup
    auto gw_narrow = [&](int nb) { return e->gw_cols ? e->gw_cols == 32 : (int64_t)T * (H / GC_N) * nb * 2 <= e->num_cus; };
    const bool gat = e->K > 0;
    for (int i = 1; i <= L; ++i) {
        // A training forward must leave what the backward of the SAME shape reads: the fused layer writes gt1 only, the
        // node-level backward (T > 512 units, max_edges > GB_E) needs gagg / gy -> fused only where the backward is fused too.
        if (e->gin && gc && gc_small(c) && e->max_edges <= gc_edge_cap(64) && (!c.training || (use_gcb(c) && e->max_edges <= GB_E))) {
            // GINConv per graph (engine_ggin.hpp): (A + I)(h W1^T) + b1 with the BatchNorm statistics | BN, ReLU, Linear, ReLU
            const float* hin = e->h + (size_t)(i - 1) * NH;
            float* t1 = e->gt1 + (size_t)(i - 1) * NH;
            GginFwdArgs ga;
            memset(&ga, 0, sizeof(ga));
            ga.x = hin; ga.W = e->P + e->o_conv_w[i - 1]; ga.bias = e->P + e->o_conv_b[i - 1]; ga.out = t1;
            if (c.training) { ga.st_sum = graph_acc(c, bn_stsum(c, i), H, striped_gin(c)); ga.st_sq = graph_acc(c, bn_stsq(c, i), H, striped_gin(c)); }
            {
                ProfScope ps(st, 9, 2.0 * N * H * H + 2.0 * (double)(c.E + N) * H, true);
                PROF_LAUNCH((k_ggin_fwd<1>), dim3(T, H / GC_N), dim3(512), 0, st, gd, e->gptr, e->eptr, ga, H, H, e->status);
            }
            CAL_CHECK_LAUNCH("k_ggin_fwd<1>"); STAGE();
            RC(flush_finals(c)); STAGE();
            memset(&ga, 0, sizeof(ga));
            ga.x = t1; ga.W = e->P + e->o_gin_w2[i - 1]; ga.bias = e->P + e->o_gin_b2[i - 1]; ga.out = e->h + (size_t)i * NH;
            ga.bn = bnref(c, i, N, 1);
            hipLaunchKernelGGL((k_ggin_fwd<2>), dim3(T, H / GC_N), dim3(512), 0, st, gd, e->gptr, e->eptr, ga, H, H, e->status);
            CAL_CHECK_LAUNCH("k_ggin_fwd<2>"); STAGE();
            continue;
        }
        if (e->gin) {    // GINConv: unit-coefficient aggregation -> Linear -> BN -> ReLU -> Linear -> ReLU (model.py:188-194)
            const float* hin = e->h + (size_t)(i - 1) * NH;
            float* agg = e->gagg + (size_t)(i - 1) * NH;
            float* t1 = e->gt1 + (size_t)(i - 1) * NH;
            float* yy = e->gy + (size_t)(i - 1) * NH;
            {
                SpmmBranch br{hin, agg, nullptr, nullptr, e->ones, Acc(), Acc(), nullptr, nullptr, nullptr};
                RC(launch_espmm(st, gd, SpmmBranch2{{br, br}}, 1, 0, 1.0f, N, H, spmm_rpb(H, false)));
                CAL_CHECK_LAUNCH("k_espmm(gin)"); STAGE();
            }
            {
                GemmArgs a = gemm_args(N, H, H, false, true, 0);
                a.p[0].A = agg; a.p[0].B = e->P + e->o_conv_w[i - 1]; a.p[0].bias = e->P + e->o_conv_b[i - 1]; a.p[0].C = t1;
                if (c.training) gemm_stats(c, a.p[0], N, H, bn_stsum(c, i), bn_stsq(c, i), false, H);
                RC(fwd_gemm(c, true, a, 1)); STAGE();
                RC(flush_finals(c)); STAGE();
            }
            {
                GinRowArgs ga;
                memset(&ga, 0, sizeof(ga));
                ga.a = t1; ga.out = yy; ga.bn = bnref(c, i, N, 1);
                RC(gin_rows(c, 0, ga));
                CAL_CHECK_LAUNCH("k_gin_bn_relu"); STAGE();
            }
            {
                GemmArgs a = gemm_args(N, H, H, false, true, 1);
                a.p[0].A = yy; a.p[0].B = e->P + e->o_gin_w2[i - 1]; a.p[0].bias = e->P + e->o_gin_b2[i - 1];
                a.p[0].C = e->h + (size_t)i * NH;
                RC(fwd_gemm(c, true, a, 1)); STAGE();
            }
            continue;
        }
        if (gat) {       // z = BN_i(h) W_i; attention scores, edge softmax (+dropout), aggregation, bias, ReLU (GATConv)
            const int K = e->K, D = H / K;
            float* zi = e->gz + (size_t)(i - 1) * NH;
            float* sc = e->gsc + (size_t)(i - 1) * 4 * al((size_t)e->capN * K);
            const size_t nk = al((size_t)e->capN * K);
            if (gc && gc_small(c) && (D == 32 || D == 64) && e->max_edges <= GG_E) {      // the whole layer per graph (engine_ggat.hpp)
                GgatArgs ga;
                memset(&ga, 0, sizeof(ga));
                ga.x = e->h + (size_t)(i - 1) * NH; ga.W = e->P + e->o_conv_w[i - 1]; ga.bias = e->P + e->o_conv_b[i - 1];
                ga.att = e->P + e->o_conv_att[i - 1]; ga.bn = bnref(c, i, N, 1); ga.out = e->h + (size_t)i * NH; ga.z = zi;
                ga.adst = sc; ga.asrc = sc + nk; ga.mx = sc + 2 * nk; ga.den = sc + 3 * nk;
                if (c.training && i < L) { ga.st_sum = graph_acc(c, bn_stsum(c, i + 1), H, striped_gat(c)); ga.st_sq = graph_acc(c, bn_stsq(c, i + 1), H, striped_gat(c)); }
                ga.heads = K; ga.D = D; ga.slope = e->gat_slope; ga.p = c.training ? e->gat_p : 0.f;
                ga.seed = e->gat_seed[i - 1]; ga.ctr = (const uint64_t*)e->gat_ctr; ga.E = E;
                {
                    ProfScope ps(st, 7, 2.0 * N * H * H + 2.0 * (double)(c.E + N) * H, true);
                    PROF_LAUNCH(k_ggat_fwd, dim3(T, H / GC_N), dim3(256), 0, st, gd, e->gptr, e->eptr, ga, H, H, e->status);
                }
                CAL_CHECK_LAUNCH("k_ggat_fwd"); STAGE();
                RC(flush_finals(c)); STAGE();
                continue;
            }
            GemmArgs a = gemm_args(N, H, H, false, false, 0);
            a.p[0].A = e->h + (size_t)(i - 1) * NH; a.p[0].B = e->P + e->o_conv_w[i - 1]; a.p[0].C = zi;
            a.p[0].xa.has_bn = 1; a.p[0].xa.bn = bnref(c, i, N, 1);
            { ProfScope ps(st, 0, 2.0 * N * H * H); RC(fwd_gemm(c, false, a, 1)); } STAGE();
            float* hi = e->h + (size_t)i * NH;
            {
                ProfScope ps(st, 5, 2.0 * N * H * 4 + (double)(c.E + N) * (8 + 12.0 * K) + (N + 1) * 4.0);
                RC(gat_forward(e->rowptr_dst, e->nbr_dst, e->eid_dst, zi, e->P + e->o_conv_att[i - 1], e->P + e->o_conv_b[i - 1], 1,
                               e->gat_slope, c.training ? e->gat_p : 0.f, e->gat_seed[i - 1], (const uint64_t*)e->gat_ctr, hi,
                               sc, sc + nk, sc + 2 * nk, sc + 3 * nk, N, E, K, D, st));
            }
            STAGE();
            if (c.training && i < L) {
                int tc = std::min(256, pow2ceil(H));
                int rpb = std::max(32, cdiv(N, 1024));
                if (H == 256 && aligned16(hi))
                    hipLaunchKernelGGL(k_colstats4, dim3(cdiv(N, rpb)), dim3(256), 0, st, hi, N, rpb, Acc(bn_stsum(c, i + 1)), Acc(bn_stsq(c, i + 1)));
                else
                    hipLaunchKernelGGL(k_colstats, dim3(cdiv(N, rpb)), dim3(256), 0, st, hi, N, H, tc, rpb, Acc(bn_stsum(c, i + 1)), Acc(bn_stsq(c, i + 1)));
                CAL_CHECK_LAUNCH("k_colstats"); STAGE();
            }
            continue;
        }
        if (gw) {        // GCNConv per graph of up to 256 nodes: MFMA product + sparse aggregation from LDS + statistics (engine_gwide.hpp)
            GconvBranch gb;
            memset(&gb, 0, sizeof(gb));
            gb.x = e->h + (size_t)(i - 1) * NH; gb.W = e->P + e->o_conv_w[i - 1]; gb.bias = e->P + e->o_conv_b[i - 1];
            gb.dis = e->dis_unit; gb.bn = bnref(c, i, N, 1); gb.out = e->h + (size_t)i * NH;
            if (i == 1 && !fast_plan) gb.coef_out = e->coef; else gb.coef_in = e->coef;      // (k_plan_graph writes the unit coefficients in slot order)
            if (c.training && i < L) { gb.st_sum = graph_acc(c, bn_stsum(c, i + 1), H, gw_st); gb.st_sq = graph_acc(c, bn_stsq(c, i + 1), H, gw_st); }
            {
                ProfScope ps(st, 2, 2.0 * N * H * H + 2.0 * (double)(c.E + N) * H, true);
                if (gw_narrow(1)) PROF_LAUNCH((k_gw_fwd<false, 32>), dim3(T, H / 32, 1), dim3(GW_NT), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb, gb}}, 1, e->loop_w, H, H, e->status);
                else PROF_LAUNCH((k_gw_fwd<false, 64>), dim3(T, H / GC_N, 1), dim3(GW_NT), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb, gb}}, 1, e->loop_w, H, H, e->status);
            }
            CAL_CHECK_LAUNCH("k_gw_fwd"); STAGE();
            RC(flush_finals(c)); STAGE();
            continue;
        }
        if (gc) {        // GCNConv: GEMM + aggregation + statistics in one per-graph kernel
            GconvBranch gb;
            memset(&gb, 0, sizeof(gb));
            gb.x = e->h + (size_t)(i - 1) * NH; gb.W = e->P + e->o_conv_w[i - 1]; gb.bias = e->P + e->o_conv_b[i - 1];
            gb.dis = e->dis_unit; gb.bn = bnref(c, i, N, 1); gb.out = e->h + (size_t)i * NH;
            if (i == 1) gb.coef_out = e->coef; else gb.coef_in = e->coef;
            if (c.training && i < L) { gb.st_sum = graph_acc(c, bn_stsum(c, i + 1), H, striped_bb(c)); gb.st_sq = graph_acc(c, bn_stsq(c, i + 1), H, striped_bb(c)); }
            {
                ProfScope ps(st, 2, 2.0 * N * H * H + 2.0 * (double)(c.E + N) * H, true);
                if (gc_small(c)) PROF_LAUNCH((k_gconv_fwd<false, 64, 512>), dim3(T, H / GC_N, 1), dim3(512), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb, gb}}, 1,
                                                    e->loop_w, H, H, e->status);
                else PROF_LAUNCH((k_gconv_fwd<false, GC_T>), dim3(T, H / GC_N, 1), dim3(256), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb, gb}}, 1,
                                        e->loop_w, H, H, e->status);
            }
            CAL_CHECK_LAUNCH("k_gconv_fwd"); STAGE();
            RC(flush_finals(c)); STAGE();
            continue;
        }
        GemmArgs a = gemm_args(N, H, H, false, false, 0);
        a.p[0].A = e->h + (size_t)(i - 1) * NH; a.p[0].B = e->P + e->o_conv_w[i - 1]; a.p[0].C = e->z;
        a.p[0].xa.has_bn = 1; a.p[0].xa.bn = bnref(c, i, N, 1);
        { ProfScope ps(st, 0, 2.0 * N * H * H); RC(fwd_gemm(c, false, a, 1)); } STAGE();
        SpmmBranch br{e->z, e->h + (size_t)i * NH, e->P + e->o_conv_b[i - 1], nullptr, e->dis_unit, Acc(), Acc(), nullptr, nullptr, nullptr};
        const bool wst = c.training && i < L;
        const int rpb = spmm_rpb(H, wst, N);
        if (wst) { br.st_sum = spmm_acc(c, bn_stsum(c, i + 1), H, rpb); br.st_sq = spmm_acc(c, bn_stsq(c, i + 1), H, rpb); }
        {
            ProfScope ps(st, 1, 2.0 * N * H * 4 + (double)(c.E + N) * 8 + (N + 1) * 4.0, true);
            RC(launch_espmm(st, gd, SpmmBranch2{{br, br}}, 1, 1, e->loop_w, N, H, rpb));
        }
        CAL_CHECK_LAUNCH("k_espmm"); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    const float* x = e->h + (size_t)L * NH;
    // 5+6 per graph: node attention, edge softmax and weighted degrees in one kernel (4 rows per lane group)
    // (forward: only when the launch has enough workgroups -- 32 graphs on 256 CUs lose to the node-parallel pair, 19.3 vs 12.6 us;
    //  128 graphs win, 20.5 vs 24.0 us with their finishing launch.  The backward per graph wins at both: 22.8 vs 28.1, 30.8 vs 47.7 us)
    const bool att_wide = use_aw(c) && 3 * T >= e->num_cus;
    if (att_wide) {          // graphs of 129-256 nodes: the same kernel in four row passes, four slots per lane (engine_attwide.hpp)
        const bool sw = gw_st;
        const Acc a0 = graph_acc(c, bn_stsum(c, L + 1), H, sw), a1 = graph_acc(c, bn_stsq(c, L + 1), H, sw);
        const Acc a2 = graph_acc(c, bn_stsum(c, L + 2), H, sw), a3 = graph_acc(c, bn_stsq(c, L + 2), H, sw);
        RC(with_g(H, [&](auto g) {
            constexpr int G = decltype(g)::value;
            hipLaunchKernelGGL((k_att_fwd_wide<4, G>), dim3(T), dim3(512), 0, st, e->gptr, gs, x, e->P + e->o_natt_w, e->P + e->o_natt_b,
                               e->P + e->o_eatt_w, e->P + e->o_eatt_b, e->anode, e->pq, e->att, e->dis_co, e->dis_co + N, a0, a1, a2, a3,
                               e->loop_w, H, E, e->status, e->no_node_att ? 0.f : 1.f, e->no_edge_att ? 0.f : 1.f, e->eptr);
            return 0;
        }));
        CAL_CHECK_LAUNCH("k_att_fwd_wide"); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    const bool att_graph = att_wide || att_graph_fwd(c);
    if (att_graph && !att_wide) {
        const bool sco = striped_co(c);
        const Acc a0 = graph_acc(c, bn_stsum(c, L + 1), H, sco), a1 = graph_acc(c, bn_stsq(c, L + 1), H, sco);
        const Acc a2 = graph_acc(c, bn_stsum(c, L + 2), H, sco), a3 = graph_acc(c, bn_stsq(c, L + 2), H, sco);
        RC(with_g(H, [&](auto g) {
            constexpr int G = decltype(g)::value;
            hipLaunchKernelGGL((k_att_fwd_graph<4, G>), dim3(T), dim3(512), 0, st, e->gptr, gs, x, e->P + e->o_natt_w, e->P + e->o_natt_b,
                               e->P + e->o_eatt_w, e->P + e->o_eatt_b, e->anode, e->pq, e->att, e->dis_co, e->dis_co + N, a0, a1, a2, a3,
                               e->loop_w, H, E, e->status, e->no_node_att ? 0.f : 1.f, e->no_edge_att ? 0.f : 1.f, e->eptr);
            return 0;
        }));
        CAL_CHECK_LAUNCH("k_att_fwd_graph"); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    // 5. node attention, edge projections, bnc/bno statistics (model.py:97-111)
    if (!att_graph) {
        const Acc a0 = node_acc(c, bn_stsum(c, L + 1), H), a1 = node_acc(c, bn_stsq(c, L + 1), H);
        const Acc a2 = node_acc(c, bn_stsum(c, L + 2), H), a3 = node_acc(c, bn_stsq(c, L + 2), H);
        RC(with_g(H, [&](auto g) {
            constexpr int G = decltype(g)::value;
            hipLaunchKernelGGL((k_node_att_fwd<4, G>), dim3(cdiv(N, c.rpb_n)), dim3(256), 0, st, x, e->P + e->o_natt_w,
                               e->P + e->o_natt_b, e->P + e->o_eatt_w, e->anode, e->pq, a0, a1, a2, a3, N, H, c.rpb_n, e->no_node_att ? 0.f : 1.f);
            return 0;
        }));
        CAL_CHECK_LAUNCH("k_node_att_fwd"); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    // 6. edge softmax + weighted degrees (model.py:102-104, gcn_conv.py:63-68)
    if (!att_graph) {
        hipLaunchKernelGGL(k_edge_att_deg, dim3(cdiv(N, 32)), dim3(256), 0, st, gs, e->pq, e->P + e->o_eatt_b, e->att, e->dis_co,
                           e->dis_co + N, e->loop_w, N, E, e->no_edge_att ? 0.f : 1.f);
        CAL_CHECK_LAUNCH("k_edge_att_deg"); STAGE();
    }
    // 7-9 fused: both weighted convolutions and the add-pool in one per-graph launch
    if (gc) {
        GconvBranch gb[2];
        memset(gb, 0, sizeof(gb));
        for (int k = 0; k < 2; ++k) {
            gb[k].x = x; gb[k].W = e->P + (k ? e->o_ow : e->o_cw); gb[k].bias = e->P + (k ? e->o_ob : e->o_cb);
            gb[k].ew = e->att + (size_t)k * E; gb[k].dis = e->dis_co + (size_t)k * N;
            gb[k].rs = e->anode + k; gb[k].rs_stride = 2; gb[k].bn = bnref(c, L + 1 + k, N, 1);
            gb[k].out = e->hco + (size_t)k * NH; gb[k].z = e->zco + (size_t)k * NH; gb[k].pooled = e->pooled + (size_t)k * B * H;
            gb[k].coef_out = e->coef + (size_t)(1 + k) * E;
            gb[k].w_out = e->wslot + (size_t)k * E;
            gb[k].batch = batch; gb[k].tile_gptr = tgp;          // packed batch: the add-pool is per GRAPH inside the tile
        }
        if (tgp) hipLaunchKernelGGL((k_gconv_fwd<true, 64, 512, true>), dim3(T, H / GC_N, 2), dim3(512), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb[0], gb[1]}}, 1,
                                    e->loop_w, H, H, e->status);
        else if (gc_small(c)) hipLaunchKernelGGL((k_gconv_fwd<true, 64, 512>), dim3(T, H / GC_N, 2), dim3(512), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb[0], gb[1]}}, 1,
                                            e->loop_w, H, H, e->status);
        else hipLaunchKernelGGL((k_gconv_fwd<true, GC_T>), dim3(T, H / GC_N, 2), dim3(256), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb[0], gb[1]}}, 1,
                                e->loop_w, H, H, e->status);
        CAL_CHECK_LAUNCH("k_gconv_fwd(co)"); STAGE();
    }
    if (gw) {        // 7-9 for graphs of up to 256 nodes (engine_gwide.hpp)
        GconvBranch gb[2];
        memset(gb, 0, sizeof(gb));
        for (int k = 0; k < 2; ++k) {
            gb[k].x = x; gb[k].W = e->P + (k ? e->o_ow : e->o_cw); gb[k].bias = e->P + (k ? e->o_ob : e->o_cb);
            gb[k].ew = e->att + (size_t)k * E; gb[k].dis = e->dis_co + (size_t)k * N;
            gb[k].rs = e->anode + k; gb[k].rs_stride = 2; gb[k].bn = bnref(c, L + 1 + k, N, 1);
            gb[k].out = e->hco + (size_t)k * NH; gb[k].z = e->zco + (size_t)k * NH; gb[k].pooled = e->pooled + (size_t)k * B * H;
        }
        if (gw_narrow(2)) hipLaunchKernelGGL((k_gw_fwd<true, 32>), dim3(T, H / 32, 2), dim3(GW_NT), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb[0], gb[1]}}, 1, e->loop_w, H, H, e->status);
        else hipLaunchKernelGGL((k_gw_fwd<true, 64>), dim3(T, H / GC_N, 2), dim3(GW_NT), 0, st, gd, e->gptr, e->eptr, GconvBranch2{{gb[0], gb[1]}}, 1, e->loop_w, H, H, e->status);
        CAL_CHECK_LAUNCH("k_gw_fwd(co)"); STAGE();
    }
    // 7. z_k = BN_k(a_k * x) @ W_k for k in (context, objects)   (model.py:112-113)
    if (!gc && !gw) {
        GemmArgs a = gemm_args(N, H, H, false, false, 0);
        for (int k = 0; k < 2; ++k) {
            a.p[k].A = x; a.p[k].B = e->P + (k ? e->o_ow : e->o_cw); a.p[k].C = e->zco + (size_t)k * NH;
            a.p[k].xa.rs = e->anode + k; a.p[k].xa.rs_stride = 2;
            a.p[k].xa.has_bn = 1; a.p[k].xa.bn = bnref(c, L + 1 + k, N, 1);
        }
        RC(fwd_gemm(c, false, a, 2)); STAGE();
    }
    // 8. h_k = relu(A_hat_k z_k + b_k)
    if (!gc && !gw) {
        SpmmBranch b0{e->zco, e->hco, e->P + e->o_cb, e->att, e->dis_co, Acc(), Acc(), nullptr, nullptr, nullptr};
        SpmmBranch b1{e->zco + NH, e->hco + NH, e->P + e->o_ob, e->att + E, e->dis_co + N, Acc(), Acc(), nullptr, nullptr, nullptr};
        RC(launch_espmm(st, gd, SpmmBranch2{{b0, b1}}, 2, 1, e->loop_w, N, H, spmm_rpb(H, false)));
        CAL_CHECK_LAUNCH("k_espmm(co)"); STAGE();
    }
    // 9. add-pool (model.py:115-116)
    if (!gc && !gw) {
        RC(launch_pool2(c));
        STAGE();
    }
    // 10. readouts (model.py:125-164)
    if (use_ro(c) && want_grad && c.training && B <= RS_B && H <= RS_K && e->ro_step) {
        // the whole readout, forward and backward, in one launch (engine_ro_step.hpp)
        RoStepArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.a = make_ro(c); sa.zpart = e->zpart; sa.sync = reinterpret_cast<int*>(e->arena + e->a_sync); sa.status = e->status;
        hipLaunchKernelGGL(k_ro_step<false>, dim3(3, H / RO_CW), dim3(256), 0, st, sa);
        CAL_CHECK_LAUNCH("k_ro_step"); STAGE();
        c.ro_done = 1;
        return 0;
    }
    if (use_ro_rows(c) && want_grad) {
        // 128 < B <= 512: the same launch in row blocks of 128 graphs (k_ro_step<true>): the sums over all rows are exchanged
        // between the row blocks inside the kernel, the weight gradients leave as one slab per row block (front of the slab
        // workspace; engine_backward registers them with k_finish)
        const int nrb = cdiv(B, RS_B), nch = H / RO_CW;
        RoStepArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.a = make_ro(c); sa.zpart = e->zpart; sa.sync = reinterpret_cast<int*>(e->arena + e->a_sync); sa.status = e->status;
        sa.nrb = nrb;
        sa.xch = parts_alloc(c, (size_t)3 * nch * RBK_XCH);
        if (!sa.xch) { set_error("engine: partial-row workspace exhausted"); return 2; }
        sa.gw1_slab = e->slabs; sa.gw2_slab = e->slabs + (size_t)3 * nrb * H * H;
        hipLaunchKernelGGL(k_ro_step<true>, dim3(3, nch, nrb), dim3(256), 0, st, sa);
        CAL_CHECK_LAUNCH("k_ro_step"); STAGE();
        c.ro_done = 2;
        return 0;
    }
    if (use_ro(c)) {
        const RoArgs ra = make_ro(c);
        hipLaunchKernelGGL(k_ro_fwd_a, dim3(3, H / RO_CW), dim3(256), 0, st, ra);
        CAL_CHECK_LAUNCH("k_ro_fwd_a"); STAGE();
        hipLaunchKernelGGL(k_ro_fwd_b, dim3(3, cdiv(B, RO_RB)), dim3(256), 0, st, ra);
        CAL_CHECK_LAUNCH("k_ro_fwd_b"); STAGE();
        if (!want_grad) {                  // no backward to finish the loss sums
            hipLaunchKernelGGL(k_ro_loss, dim3(3), dim3(256), 0, st, ra);
            CAL_CHECK_LAUNCH("k_ro_loss");
        }
        return 0;
    }
    const int bn_fc1 = L + 3, bn_fc2 = L + 4;   // + 2*head
    {
        int tc = std::min(256, pow2ceil(H));
        hipLaunchKernelGGL(k_readout_prep, dim3(cdiv(B, c.rpb_b)), dim3(256), 0, st, e->pooled, perm, e->iperm, e->xco, B, H, tc,
                           c.rpb_b, bn_stsum(c, bn_fc1), bn_stsq(c, bn_fc1), bn_stsum(c, bn_fc1 + 2), bn_stsq(c, bn_fc1 + 2),
                           bn_stsum(c, bn_fc1 + 4), bn_stsq(c, bn_fc1 + 4), e->cat);
        CAL_CHECK_LAUNCH("k_readout_prep"); STAGE();
    }
    const float* xin[3] = {e->pooled, e->pooled + (size_t)B * H, e->xco};
    const int Wco = e->cat ? 2 * H : H;                 // input width of the co head
    {
        // fc1 of the three heads: one batched launch, or (cat: the co head reduces over 2H) heads c / o batched + co alone
        auto fc1 = [&](int hd0, int nh, int Kin) -> int {
            GemmArgs a = gemm_args(B, H, Kin, false, true, 1);
            for (int q = 0; q < nh; ++q) {
                const int hd = hd0 + q;
                a.p[q].A = xin[hd]; a.p[q].B = e->P + e->o_fc1_w[hd]; a.p[q].bias = e->P + e->o_fc1_b[hd];
                a.p[q].C = e->y1 + (size_t)hd * B * H;
                a.p[q].xa.has_bn = 1; a.p[q].xa.bn = bnref(c, bn_fc1 + 2 * hd, B, 1);
                if (c.training) gemm_stats(c, a.p[q], B, H, bn_stsum(c, bn_fc2 + 2 * hd), bn_stsq(c, bn_fc2 + 2 * hd), false);
            }
            return fwd_gemm(c, true, a, nh);
        };
        if (e->cat) { RC(fc1(0, 2, H)); STAGE(); RC(fc1(2, 1, Wco)); STAGE(); }
        else { RC(fc1(0, 3, H)); STAGE(); }
        RC(flush_finals(c)); STAGE();
    }
    {
        GemmArgs a = gemm_args(B, C, H, false, true, 0);
        for (int hd = 0; hd < 3; ++hd) {
            a.p[hd].A = e->y1 + (size_t)hd * B * H; a.p[hd].B = e->P + e->o_fc2_w[hd]; a.p[hd].bias = e->P + e->o_fc2_b[hd];
            a.p[hd].C = e->zl + (size_t)hd * B * C;
            a.p[hd].xa.has_bn = 1; a.p[hd].xa.bn = bnref(c, bn_fc2 + 2 * hd, B, 1);
        }
        RC(fwd_gemm(c, true, a, 3)); STAGE();
    }
    if (B > 256) hipLaunchKernelGGL(k_loss<1024>, dim3(1), dim3(1024), 0, st, e->zl, y, e->logp, e->dzl, e->stats, e->arena + e->a_db2, B, C, wc, wo,
                                    wco, want_grad);
    else hipLaunchKernelGGL(k_loss<256>, dim3(1), dim3(256), 0, st, e->zl, y, e->logp, e->dzl, e->stats, e->arena + e->a_db2, B, C, wc, wo,
                            wco, want_grad);
    CAL_CHECK_LAUNCH("k_loss"); STAGE();
    return 0;
}

int engine_backward(Ctx& c, const float* x0, const int64_t* batch) {
    Engine* e = c.e;
    const int N = c.N, B = c.B, T = c.T, H = e->H, F = e->F, C = e->C, L = e->L;
    const int64_t E = c.E;
    hipStream_t st = c.st;
    const size_t NH = (size_t)N * H, BH = (size_t)B * H;
    const CSR gd{e->rowptr_dst, e->nbr_dst, e->eid_dst, (int)c.E}, gs{e->rowptr_src, e->nbr_src, e->eid_src, (int)c.E};
    const int bn_fc1 = L + 3, bn_fc2 = L + 4;
    const float* xin[3] = {e->pooled, e->pooled + BH, e->xco};
    FinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.tick = (c.tick_in_finish && !c.adam_in_finish) ? e->step : nullptr;      // (adam_in_finish: k_zero_f64 did it)
    fa.ticked = c.adam_in_finish ? e->step : nullptr;
    fa.perm_ctr = c.draw_perm ? e->perm_ctr : nullptr;
    fa.status = e->status;
    fa.host_status = e->host_status;
    { const int wp0 = (e->bn[0].width + 3) / 4 * 4; fa.bn0z = e->arena + e->bn[0].arena; fa.bn0z_n = 2 * wp0; fa.dirty = e->status + 3; }
    size_t slab_off = 0;
    auto commit_p = [&](const double* src, int P, int stride, int dst, int n, float scale) {
        if (fa.nct < MAX_COMMITS) fa.ct[fa.nct] = CommitTask{src, P, stride, dst, n, scale};
        fa.nct++;
    };
    auto commit = [&](int src, int dst, int n, float scale) { commit_p(e->arena + src, 1, 0, dst, n, scale); };
    // deferred column sums: partial rows kept until the final commit (no finalise launch needed)
    const int PN = cdiv(N, c.rpb_n);
    struct Deferred { double* p; int P; int stride; };
    auto deferred = [&](int cols, Deferred& d) -> Acc {
        d.p = parts_alloc(c, (size_t)PN * cols); d.P = PN; d.stride = cols;
        return d.p ? Acc(nullptr, d.p, cols) : Acc();
    };
    const bool awb = use_aw(c);                                    // ... of k_att_bwd_wide (129-256-node graphs): two while 2 T workgroups fit the chip
    const int ag_split = awb ? std::max(1, std::min(8, e->num_cus / std::max(T, 1))) : 2;       // workgroups per graph of k_att_bwd_graph
    auto deferred_g = [&](int cols, Deferred& d) -> Acc {          // one partial row per workgroup of k_att_bwd_graph / k_att_bwd_wide
        d.p = parts_alloc(c, (size_t)ag_split * T * cols); d.P = ag_split * T; d.stride = cols;
        return d.p ? Acc(nullptr, d.p, cols) : Acc();
    };
    Deferred d_convb[MAX_LAYERS], d_cb, d_ob, d_dwn, d_dwe, d_bn0;
    memset(d_convb, 0, sizeof(d_convb));
    d_bn0.p = nullptr;

    const bool ro = use_ro(c) || c.ro_done == 2;
    if (c.ro_done == 2) {                               // the row-blocked readout left one weight-gradient slab per row block
        const int nrb = cdiv(B, RS_B);
        for (int hd = 0; hd < 3; ++hd) {
            fa.st[fa.nst++] = SlabTask{e->slabs + (size_t)hd * nrb * H * H, e->G + e->o_fc1_w[hd], H * H, nrb};
            fa.st[fa.nst++] = SlabTask{e->slabs + (size_t)3 * nrb * H * H + (size_t)hd * nrb * C * H, e->G + e->o_fc2_w[hd], C * H, nrb};
        }
        slab_off = (size_t)3 * nrb * ((size_t)H * H + (size_t)C * H);
        slab_off = (slab_off + 63) & ~(size_t)63;
    }
    if (ro) {
        if (!c.ro_done) {
            const RoArgs ra = make_ro(c);
            hipLaunchKernelGGL(k_ro_bwd_a, dim3(3, H / RO_CW), dim3(256), 0, st, ra);
            CAL_CHECK_LAUNCH("k_ro_bwd_a"); STAGE();
            hipLaunchKernelGGL(k_ro_bwd_b, dim3(3, H / RO_CW), dim3(256), 0, st, ra);
            CAL_CHECK_LAUNCH("k_ro_bwd_b"); STAGE();
        }
        fa.stats = e->stats; fa.wc = c.wc; fa.wo = c.wo; fa.wco = c.wco;
    }
    // R1. dW2_h = dz_h^T @ BN2(y1_h)
    if (!ro) {
        GemmArgs a = gemm_args(C, H, B, true, false, 0);
        float* dst[3];
        for (int hd = 0; hd < 3; ++hd) {
            a.p[hd].A = e->dzl + (size_t)hd * B * C; a.p[hd].B = e->y1 + hd * BH;
            a.p[hd].xb.has_bn = 1; a.p[hd].xb.bn = bnref(c, bn_fc2 + 2 * hd, B, 0);
            dst[hd] = e->G + e->o_fc2_w[hd];
        }
        RC(grad_gemm(c, a, 3, dst, fa, slab_off)); STAGE();
    }
    // R2. d(BN2 out)_h = dz_h @ W2_h, with the BN2-backward sums
    if (!ro) {
        GemmArgs a = gemm_args(B, H, C, false, false, 0);
        for (int hd = 0; hd < 3; ++hd) {
            a.p[hd].A = e->dzl + (size_t)hd * B * C; a.p[hd].B = e->P + e->o_fc2_w[hd]; a.p[hd].C = e->dyh1 + hd * BH;
            a.p[hd].aux = e->y1 + hd * BH; a.p[hd].has_aux = 1; a.p[hd].aux_bn = bnref(c, bn_fc2 + 2 * hd, B, 0);
            gemm_stats(c, a.p[hd], B, H, bn_dsum(c, bn_fc2 + 2 * hd), bn_dprod(c, bn_fc2 + 2 * hd), true);
        }
        RC(fwd_gemm(c, false, a, 3)); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    // R3. BN2 backward + ReLU mask + fc1 bias gradients
    if (!ro) {
        BnBwdProb p[3];
        for (int hd = 0; hd < 3; ++hd)
            p[hd] = BnBwdProb{e->dyh1 + hd * BH, e->y1 + hd * BH, e->dy1 + hd * BH, bnref(c, bn_fc2 + 2 * hd, B, 0),
                              bn_dsum(c, bn_fc2 + 2 * hd), bn_dprod(c, bn_fc2 + 2 * hd), e->arena + e->a_db1 + hd * H};
        RC(with_g(H, [&](auto g) {
            constexpr int G = decltype(g)::value;
            hipLaunchKernelGGL((k_bn_bwd<4, G>), dim3(cdiv(B, c.rpb_b), 3), dim3(256), 0, st, BnBwdProb3{{p[0], p[1], p[2]}}, 1, B, H, c.rpb_b);
            return 0;
        }));
        CAL_CHECK_LAUNCH("k_bn_bwd(readout)"); STAGE();
    }
    const int Wco = e->cat ? 2 * H : H;                 // input width of the co head (cat: [xc[perm] | xo])
    float* const dxh_hd[3] = {e->dxh, e->dxh + BH, e->dxh + 2 * BH};
    // R4. dW1_h = dy1_h^T @ BN1(xin_h)
    if (!ro) {
        auto r4 = [&](int hd0, int nh, int Kin) -> int {
            GemmArgs a = gemm_args(H, Kin, B, true, false, 0);
            float* dst[3];
            for (int q = 0; q < nh; ++q) {
                const int hd = hd0 + q;
                a.p[q].A = e->dy1 + hd * BH; a.p[q].B = xin[hd];
                a.p[q].xb.has_bn = 1; a.p[q].xb.bn = bnref(c, bn_fc1 + 2 * hd, B, 0);
                dst[q] = e->G + e->o_fc1_w[hd];
            }
            return grad_gemm(c, a, nh, dst, fa, slab_off);
        };
        if (e->cat) { RC(r4(0, 2, H)); STAGE(); RC(r4(2, 1, Wco)); STAGE(); }
        else { RC(r4(0, 3, H)); STAGE(); }
    }
    // R5. d(BN1 out)_h = dy1_h @ W1_h with the BN1-backward sums
    if (!ro) {
        auto r5 = [&](int hd0, int nh, int Kin) -> int {
            GemmArgs a = gemm_args(B, Kin, H, false, false, 0);
            for (int q = 0; q < nh; ++q) {
                const int hd = hd0 + q;
                a.p[q].A = e->dy1 + hd * BH; a.p[q].B = e->P + e->o_fc1_w[hd]; a.p[q].C = dxh_hd[hd];
                a.p[q].aux = xin[hd]; a.p[q].has_aux = 1; a.p[q].aux_bn = bnref(c, bn_fc1 + 2 * hd, B, 0);
                gemm_stats(c, a.p[q], B, Kin, bn_dsum(c, bn_fc1 + 2 * hd), bn_dprod(c, bn_fc1 + 2 * hd), true);
            }
            return fwd_gemm(c, false, a, nh);
        };
        if (e->cat) { RC(r5(0, 2, H)); STAGE(); RC(r5(2, 1, Wco)); STAGE(); }
        else { RC(r5(0, 3, H)); STAGE(); }
        RC(flush_finals(c)); STAGE();
    }
    // R6. BN1 backward + un-permute the random intervention -> d pooled
    if (!ro) {
        BnIn in[3];
        for (int hd = 0; hd < 3; ++hd)
            in[hd] = BnIn{dxh_hd[hd], xin[hd], bnref(c, bn_fc1 + 2 * hd, B, 0), bn_dsum(c, bn_fc1 + 2 * hd), bn_dprod(c, bn_fc1 + 2 * hd)};
        hipLaunchKernelGGL(k_readout_bwd_tail, dim3(cdiv((int64_t)BH, 256)), dim3(256), 0, st, in[0], in[1], in[2], e->iperm, e->dpool, B, H, e->cat);
        CAL_CHECK_LAUNCH("k_readout_bwd_tail"); STAGE();
    }
    const bool gwb = use_gw(c);                 // wide per-graph backward (engine_gwide.hpp); its BatchNorm sums through the planes when the
    const bool gw_st = gwb && e->striped && c.training;  // node-level readers of this batch (k_att_bwd, k_bn_bwd) are then the striped instantiations
    const bool gcb = use_gcb(c) || gwb;
    const bool agb = gcb && !gwb && T <= 256;           // per-graph attention backward (two workgroups per graph: one wave of the chip)
    // P1. add-pool backward + ReLU of the causal/trivial convs: d(conv output)[v] = relu'(h_k[v]) * (gradient of the pooled row of v's
    // graph) is never stored -- the transposed aggregation below (P5) builds it per gathered row from the activation, and the bias
    // gradients are count x pooled-row gradient per graph (k_pool2 counted the positive rows).  Rounds 1-3 wrote and re-read the
    // two [N, H] matrices (656 MB and 155 us per step at config 5).
    // (per-graph fused backward: both are built while k_gconv_bwd stages dOut, and it emits gn / gself as well)
    if (!gcb) {
        if (use_gc(c)) { RC(launch_pool_cnt(c)); STAGE(); }             // the forward pooled inside k_gconv_fwd: no counts yet
        d_cb.p = parts_alloc(c, (size_t)B * H); d_cb.P = B; d_cb.stride = H;
        d_ob.p = parts_alloc(c, (size_t)B * H); d_ob.P = B; d_ob.stride = H;
        if (!d_cb.p || !d_ob.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
        // (fused readout: d pooled is still two addends per row -- combined here into dpool, which that path leaves unused)
        hipLaunchKernelGGL(k_pool_bias_grad, dim3(B, 2), dim3(256), 0, st, e->pcnt, ro ? e->dxh : e->dpool, ro ? e->dxh + 2 * (size_t)B * H : nullptr,
                           e->iperm, e->dpool, d_cb.p, d_ob.p, B, H);
        CAL_CHECK_LAUNCH("k_pool_bias_grad"); STAGE();
    }
    // P2-P4. gradient w.r.t. the edge weights through propagate and through the normalisation
    auto norm_bwd = [&](const float* gn2, const float* gself2) -> int {
        hipLaunchKernelGGL(k_normbwd_node2, dim3(cdiv(N, 32), 2), dim3(256), 0, st, gs, gd, e->att, e->dis_co, e->gn, e->gself, e->ddeg,
                           e->loop_w, N, E, gn2, gself2);
        CAL_CHECK_LAUNCH("k_normbwd_node2"); STAGE();
        if (E > 0) {
            hipLaunchKernelGGL(k_normbwd_edge, dim3(cdiv(E, 256)), dim3(256), 0, st, e->row32, e->col32, e->att, e->dis_co, e->gn, e->ddeg,
                               e->dl, N, E, gn2, e->no_edge_att ? 0.f : 1.f);
            CAL_CHECK_LAUNCH("k_normbwd_edge"); STAGE();
        }
        return 0;
    };
    // P5. dz_k = A_hat_k^T dZ_k with the SDDMM riding on it, then P2-P4 on its output.  The transposed aggregation gathers
    // dZ_k[dst] for every out-edge of a source row anyway, so gn_e = <dZ_k[dst_e], z_k[src_e]> costs it one more row (z_k[src]) and a
    // lane-group reduction per slot instead of a second pass over both matrices (the separate SDDMM kernel of rounds 1-3: 265 us per
    // step at config 5).  Input self-loop edges have no slot: their gn entry is never read (k_normbwd_* skip them like the plan does).
    if (!gcb) {
        SpmmBranch b0{e->hco, e->dzco, nullptr, e->att, e->dis_co, Acc(), Acc(), nullptr, nullptr, nullptr};
        SpmmBranch b1{e->hco + NH, e->dzco + NH, nullptr, e->att + E, e->dis_co + N, Acc(), Acc(), nullptr, nullptr, nullptr};
        b0.sd_z = e->zco; b0.sd_gn = e->gn; b0.sd_gself = e->gself;
        b1.sd_z = e->zco + NH; b1.sd_gn = e->gn + E; b1.sd_gself = e->gself + N;
        b0.pb_g = e->dpool; b0.pb_batch = batch;
        b1.pb_g = e->dpool + (size_t)B * H; b1.pb_batch = batch;
        RC(launch_espmm(st, gs, SpmmBranch2{{b0, b1}}, 2, 0, e->loop_w, N, H, spmm_rpb(H, false)));
        CAL_CHECK_LAUNCH("k_espmm(co,T)"); STAGE();
        RC(norm_bwd(nullptr, nullptr));
    }
    const float* x = e->h + (size_t)L * NH;
    // P5-P7 fused per graph: dz_k stays in LDS; dX'_k arrives as one partial per output-column slice
    if (gcb) {
        GconvBwdBranch gb[2];
        memset(gb, 0, sizeof(gb));
        float* dst[2]; double* dsum[2]; double* dprod[2];
        for (int k = 0; k < 2; ++k) {
            gb[k].x = x; gb[k].W = e->P + (k ? e->o_ow : e->o_cw);
            // dOut = relu'(h_k) * (gradient of the graph's pooled row); gn / gself partials per 64-column slice
            gb[k].y = e->hco + (size_t)k * NH; gb[k].z = e->zco + (size_t)k * NH;
            if (ro) { gb[k].gp0 = e->dxh + (size_t)k * B * H; gb[k].gp1 = e->dxh + (size_t)2 * B * H; gb[k].iperm = k == 0 ? e->iperm : nullptr; }
            else gb[k].gp0 = e->dpool + (size_t)k * B * H;
            gb[k].gn = e->gn + (size_t)k * E; gb[k].gself = e->gself + (size_t)k * N;
            gb[k].gn_stride = 2 * (size_t)E; gb[k].gself_stride = 2 * (size_t)N;
            Deferred& db = k ? d_ob : d_cb;
            db.p = parts_alloc(c, (size_t)T * H); db.P = T; db.stride = H;
            if (!db.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
            gb[k].bias_parts = db.p;
            gb[k].ew = e->att + (size_t)k * E; gb[k].dis = e->dis_co + (size_t)k * N;
            gb[k].rs = e->anode + k; gb[k].rs_stride = 2; gb[k].bn = bnref(c, L + 1 + k, N, 0);
            gb[k].dxp0 = e->dXhco + (size_t)k * NH; gb[k].dxp1 = e->dzco + (size_t)k * NH;
            gb[k].coef_in = gwb ? nullptr : e->coef + (size_t)(1 + k) * E;      // (the wide kernels walk the CSR by source: other slot order)
            gb[k].gn_slot = agb ? 1 : 0;        // consumed by k_att_bwd_graph in slot order (else by k_normbwd_* in edge-id order)
            dst[k] = e->G + (k ? e->o_ow : e->o_cw); dsum[k] = bn_dsum(c, L + 1 + k); dprod[k] = bn_dprod(c, L + 1 + k);
        }
        RC(gconv_bwd(c, gd, gb, 2, dst, dsum, dprod, fa, slab_off, true, gwb ? gw_st : striped_co(c), gwb ? &gs : nullptr)); STAGE();
        RC(flush_finals(c)); STAGE();
        // the edge-weight gradients through the normalisation are part of the per-graph attention backward below; big
        // batches (a per-graph launch would be several waves of one-per-CU workgroups) keep the node- / edge-parallel kernels
        if (!agb && !awb) {
            const bool two = H > GC_N;
            RC(norm_bwd(two ? e->gn + 2 * (size_t)E : nullptr, two ? e->gself + 2 * (size_t)N : nullptr));
        }
    }
    // P6. dW_k = BN_k(a_k x)^T @ dz_k
    if (!gcb) {
        GemmArgs a = gemm_args(H, H, N, true, false, 0);
        float* dst[2];
        for (int k = 0; k < 2; ++k) {
            a.p[k].A = x; a.p[k].B = e->dzco + (size_t)k * NH;
            a.p[k].xa.rs = e->anode + k; a.p[k].xa.rs_stride = 2;
            a.p[k].xa.has_bn = 1; a.p[k].xa.bn = bnref(c, L + 1 + k, N, 0);
            dst[k] = e->G + (k ? e->o_ow : e->o_cw);
        }
    // P7. d(BN_k out) = dz_k @ W_k^T with the BN_k-backward sums (same launch as P6)
        GemmArgs aw = a;
        a = gemm_args(N, H, H, false, true, 0);
        for (int k = 0; k < 2; ++k) {
            a.p[k].A = e->dzco + (size_t)k * NH; a.p[k].B = e->P + (k ? e->o_ow : e->o_cw); a.p[k].C = e->dXhco + (size_t)k * NH;
            a.p[k].aux = x; a.p[k].aux_rs = e->anode + k; a.p[k].aux_rs_stride = 2; a.p[k].has_aux = 1;
            a.p[k].aux_bn = bnref(c, L + 1 + k, N, 0);
            gemm_stats(c, a.p[k], N, H, bn_dsum(c, L + 1 + k), bn_dprod(c, L + 1 + k), true, H, striped_node(c));
        }
        RC(dual_gemm(c, a, 2, aw, 2, dst, fa, slab_off)); STAGE();
        RC(flush_finals(c)); STAGE();
    }
    // P8. everything between the last backbone conv and the two causal convs
    {
        AttBwdArgs aa;
        aa.x = x; aa.anode = e->anode; aa.dxhc = e->dXhco; aa.dxho = e->dXhco + NH;
        const bool two = gcb && H > GC_N;               // second output-column slice of the fused backward
        aa.dxhc2 = two ? e->dzco : nullptr; aa.dxho2 = two ? e->dzco + NH : nullptr;
        aa.bnc = bnref(c, L + 1, N, 0); aa.bno = bnref(c, L + 2, N, 0);
        aa.dsc = bn_dsum(c, L + 1); aa.dpc = bn_dprod(c, L + 1); aa.dso = bn_dsum(c, L + 2); aa.dpo = bn_dprod(c, L + 2); aa.dss = e->bn_plane;
        aa.Wn = e->P + e->o_natt_w; aa.We = e->P + e->o_eatt_w; aa.dl = e->dl;
        aa.fnode = e->no_node_att ? 0.f : 1.f; aa.fedge = e->no_edge_att ? 0.f : 1.f;
        aa.gs = gs; aa.gd = gd; aa.dZ = e->dZ;
        if (awb) {       // per graph of up to 256 nodes: the same with a sparse edge phase (engine_attwide.hpp); inputs in edge-id order
            aa.dbias = L > 0 ? deferred_g(H, d_convb[L - 1]) : Acc();
            aa.dWn = deferred_g(H + 4, d_dwn); aa.dWe = deferred_g(2 * H + 4, d_dwe);
            if (!aa.dWn.on() || !aa.dWe.on()) { set_error("engine: partial-row workspace exhausted"); return 2; }
            AttBwdWideArgs ag;
            ag.a = aa; ag.gptr = e->gptr; ag.eptr = e->eptr; ag.row32 = e->row32; ag.col32 = e->col32; ag.att = e->att; ag.dis = e->dis_co;
            ag.gn = e->gn; ag.gn2 = two ? e->gn + 2 * (size_t)E : nullptr;
            ag.gself = e->gself; ag.gself2 = two ? e->gself + 2 * (size_t)N : nullptr;
            ag.loop_w = e->loop_w; ag.E = E; ag.N = N; ag.status = e->status;
            RC(with_g(H, [&](auto g) {
                constexpr int G = decltype(g)::value;
                hipLaunchKernelGGL((k_att_bwd_wide<4, G>), dim3(ag_split * T), dim3(512), 0, st, ag, 1, H, ag_split);
                return 0;
            }));
            CAL_CHECK_LAUNCH("k_att_bwd_wide"); STAGE();
        } else if (agb) {       // per graph: d deg, d edge logits and the row pass in one kernel (engine_attbwd.hpp)
            aa.dbias = L > 0 ? deferred_g(H, d_convb[L - 1]) : Acc();
            aa.dWn = deferred_g(H + 4, d_dwn); aa.dWe = deferred_g(2 * H + 4, d_dwe);
            if (!aa.dWn.on() || !aa.dWe.on()) { set_error("engine: partial-row workspace exhausted"); return 2; }
            AttBwdGraphArgs ag;
            ag.a = aa; ag.gptr = e->gptr; ag.eptr = e->eptr; ag.att = e->wslot; ag.dis = e->dis_co;
            ag.gn = e->gn; ag.gn2 = two ? e->gn + 2 * (size_t)E : nullptr;
            ag.gself = e->gself; ag.gself2 = two ? e->gself + 2 * (size_t)N : nullptr;
            ag.loop_w = e->loop_w; ag.E = E; ag.N = N; ag.status = e->status;
            RC(with_g(H, [&](auto g) {
                constexpr int G = decltype(g)::value;
                hipLaunchKernelGGL((k_att_bwd_graph<4, G>), dim3(ag_split * T), dim3(512), 0, st, ag, 1, H, ag_split);
                return 0;
            }));
            CAL_CHECK_LAUNCH("k_att_bwd_graph"); STAGE();
        } else {
        aa.dbias = L > 0 ? deferred(H, d_convb[L - 1]) : Acc();
        aa.dWn = deferred(H + 4, d_dwn); aa.dWe = deferred(2 * H + 4, d_dwe);
        if (!aa.dWn.on() || !aa.dWe.on()) { set_error("engine: partial-row workspace exhausted"); return 2; }
        RC(with_g(H, [&](auto g) {
            constexpr int G = decltype(g)::value;
            if (striped_node(c) || gw_st) hipLaunchKernelGGL((k_att_bwd<4, G, true>), dim3(cdiv(N, c.rpb_n)), dim3(256), 0, st, aa, 1, N, H, c.rpb_n);
            else hipLaunchKernelGGL((k_att_bwd<4, G>), dim3(cdiv(N, c.rpb_n)), dim3(256), 0, st, aa, 1, N, H, c.rpb_n);
            return 0;
        }));
        CAL_CHECK_LAUNCH("k_att_bwd"); STAGE();
        }
    }
    bool feat_done = false;     // the per-graph feature-layer backward (k_feat_bwd_mma) has run
    // feature layer h0 = relu(BN0(x0) W_feat) per unit, fed from the first backbone layer's partial input gradients
    // (engine_gconv_bwd.hpp: one MFMA product per unit, any F <= 160); nobn: no BatchNorm between h0 and that layer (GIN)
    auto feat_bwd = [&](const float* p0, const float* p1, bool nobn) -> int {
        FeatBwdArgs fb;
        memset(&fb, 0, sizeof(fb));
        fb.dy0 = p0; fb.dy1 = p1; fb.y = e->h;
        if (!nobn) { fb.ubn = bnref(c, 1, N, 0); fb.udot_sum = bn_dsum(c, 1); fb.udot_prod = bn_dprod(c, 1); }
        fb.x0 = x0; fb.W = e->P + e->o_feat_w; fb.bn0 = bnref(c, 0, N, 0);
        // units: the graphs (tiles) of the per-graph kernels -- or, behind the wide convolutions (129-256-node graphs), uniform chunks
        // of FB_T rows: the layer is row-wise, the chunks need not be graphs (k_feat_bwd with a null unit table)
        const bool chunks = gwb;
        const int U = chunks ? cdiv(N, FB_T) : T;
        const size_t need = (size_t)U * F * H;
        if (slab_off + need > e->slab_floats || fa.nst >= MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
        fb.slab = e->slabs + slab_off;
        fa.st[fa.nst++] = SlabTask{e->slabs + slab_off, e->G + e->o_feat_w, F * H, U};
        slab_off += need;
        d_bn0.p = parts_alloc(c, (size_t)U * 2 * F); d_bn0.P = U; d_bn0.stride = 2 * F;
        if (!d_bn0.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
        fb.parts = d_bn0.p;
        const dim3 grid(U), blk(GB_NT);
        if (F <= FB_F && !nobn) {
            // few features (SPMotif: F = 10): the FMA-loop kernel is 1.7 us shorter than one MFMA tile behind three barriers
            if (F <= 16) hipLaunchKernelGGL(k_feat_bwd<16>, grid, blk, 0, st, chunks ? (const int*)nullptr : (const int*)e->gptr, fb, H, F, e->status, N);
            else hipLaunchKernelGGL(k_feat_bwd<FB_F>, grid, blk, 0, st, chunks ? (const int*)nullptr : (const int*)e->gptr, fb, H, F, e->status, N);
        } else if (F <= 64) {
            hipLaunchKernelGGL((k_feat_bwd_mma<8, true>), grid, blk, 0, st, e->gptr, fb, H, F, e->status);
        } else {
            if (nobn) hipLaunchKernelGGL((k_feat_bwd_mma<20, true>), grid, blk, 0, st, e->gptr, fb, H, F, e->status);
            else hipLaunchKernelGGL((k_feat_bwd_mma<20, false>), grid, blk, 0, st, e->gptr, fb, H, F, e->status);
        }
        CAL_CHECK_LAUNCH("k_feat_bwd");
        feat_done = true;
        return 0;
    };
    // Q. backbone layers, last to first
    Deferred d_gin_b1[MAX_LAYERS];
    memset(d_gin_b1, 0, sizeof(d_gin_b1));
    for (int i = L; i >= 1; --i) {
        float* dzi = e->dzi + (size_t)(i - 1) * NH;     // per layer: the side-stream dW GEMM reads it later
        const bool gat = e->K > 0;
        if (e->gin && gcb && e->max_edges <= GB_E) {
            // GINConv backward per graph (engine_ggin.hpp): second Linear (+ the BatchNorm-backward sums behind the ReLU) |
            // BatchNorm backward, first Linear, transposed aggregation.  Partial input gradients (one per output-column
            // slice): PART 2 -> (dXh, gy[i-1]), PART 1 -> (z, gagg[i-1]); the layer below (or the mask pass in front of the
            // feature layer) adds the two and masks by h_{i-1} > 0.
            const int nsl = H / GC_N;
            const bool two = nsl > 1;
            float* c0 = e->dXh; float* c1 = e->gy + (size_t)(i - 1) * NH;
            float* d0 = e->z; float* d1 = e->gagg + (size_t)(i - 1) * NH;
            const float* t1 = e->gt1 + (size_t)(i - 1) * NH;
            if (slab_off + 2 * (size_t)T * H * H > e->slab_floats || fa.nst + 2 > MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
            GginBwdArgs ga;
            memset(&ga, 0, sizeof(ga));
            if (i == L) ga.dout = e->dZ;
            else {
                ga.dy0 = e->z; ga.dy1 = two ? e->gagg + (size_t)i * NH : nullptr; ga.hmask = e->h + (size_t)i * NH;
                Deferred& db = d_convb[i - 1];           // d b2 of this layer: column sums of the masked d h_i, one partial row per unit
                db.p = parts_alloc(c, (size_t)T * H); db.P = T; db.stride = H;
                if (!db.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
                ga.bias_parts = db.p;
            }
            ga.x = t1; ga.W = e->P + e->o_gin_w2[i - 1]; ga.bn = bnref(c, i, N, 0);
            ga.dxp0 = c0; ga.dxp1 = c1;
            ga.slab = e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{ga.slab, e->G + e->o_gin_w2[i - 1], H * H, T};
            slab_off += (size_t)T * H * H;
            if (striped_gin(c)) { ga.dacc_sum = bn_dsum(c, i); ga.dacc_prod = bn_dprod(c, i); ga.dacc_ss = e->bn_plane; }
            else {
                double* pp = parts_alloc(c, (size_t)T * nsl * 2 * H);
                if (!pp) { set_error("engine: partial-row workspace exhausted"); return 2; }
                ga.dot_parts = pp;
                final_task(c, pp, T * nsl, 2 * H, H, bn_dsum(c, i));
                final_task(c, pp + H, T * nsl, 2 * H, H, bn_dprod(c, i));
            }
            hipLaunchKernelGGL((k_ggin_bwd<2>), dim3(T, nsl), dim3(GB_NT), 0, st, gd, e->gptr, e->eptr, ga, N, H, H, e->status);
            CAL_CHECK_LAUNCH("k_ggin_bwd<2>"); STAGE();
            RC(flush_finals(c)); STAGE();
            memset(&ga, 0, sizeof(ga));
            ga.dy0 = c0; ga.dy1 = two ? c1 : nullptr; ga.t1 = t1;
            ga.dot_sum = bn_dsum(c, i); ga.dot_prod = bn_dprod(c, i);
            {
                Deferred& db = d_gin_b1[i - 1];          // d b1: column sums of dt1
                db.p = parts_alloc(c, (size_t)T * H); db.P = T; db.stride = H;
                if (!db.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
                ga.bias_parts = db.p;
            }
            ga.x = e->h + (size_t)(i - 1) * NH; ga.W = e->P + e->o_conv_w[i - 1]; ga.bn = bnref(c, i, N, 0);
            ga.dxp0 = d0; ga.dxp1 = d1;
            ga.slab = e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{ga.slab, e->G + e->o_conv_w[i - 1], H * H, T};
            slab_off += (size_t)T * H * H;
            {
                ProfScope ps(st, 10, 4.0 * N * H * H + 2.0 * (double)(c.E + N) * H, true);
                PROF_LAUNCH((k_ggin_bwd<1>), dim3(T, nsl), dim3(GB_NT), 0, st, gd, e->gptr, e->eptr, ga, N, H, H, e->status);
            }
            CAL_CHECK_LAUNCH("k_ggin_bwd<1>"); STAGE();
            if (i == 1 && F <= FM_F && H <= FB_H) {
                RC(feat_bwd(d0, two ? d1 : nullptr, true)); STAGE();
            } else if (i == 1) {                         // the feature layer below takes d h0 masked by h0 > 0 in e->dZ
                GinRowArgs gr;
                memset(&gr, 0, sizeof(gr));
                gr.a = d0; gr.a2 = two ? d1 : nullptr; gr.y = e->h; gr.out = e->dZ;
                RC(gin_rows(c, 3, gr));
                CAL_CHECK_LAUNCH("k_gin_mask"); STAGE();
            }
            continue;
        }
        if (e->gin) {
            // GINConv backward (model.py:188-194 differentiated).  e->dZ holds dz2 = d h_i masked by h_i > 0 (its column sums,
            // d b2, are already deferred): dW2 = dz2^T y, dy = dz2 W2, BatchNorm backward behind the ReLU -> dt1 (+ d b1),
            // dW1 = dt1^T agg, dagg = dt1 W1, d h_{i-1} = dagg + A^T dagg, masked by h_{i-1} > 0 for the layer below
            const float* agg = e->gagg + (size_t)(i - 1) * NH;
            const float* t1 = e->gt1 + (size_t)(i - 1) * NH;
            const float* yy = e->gy + (size_t)(i - 1) * NH;
            {
                GemmArgs a = gemm_args(H, H, N, true, false, 0);
                a.p[0].A = e->dZ; a.p[0].B = yy;
                float* dst[1] = {e->G + e->o_gin_w2[i - 1]};
                RC(grad_gemm(c, a, 1, dst, fa, slab_off)); STAGE();
            }
            {
                GemmArgs a = gemm_args(N, H, H, false, false, 0);
                a.p[0].A = e->dZ; a.p[0].B = e->P + e->o_gin_w2[i - 1]; a.p[0].C = e->dXh;      // dy
                RC(fwd_gemm(c, false, a, 1)); STAGE();
            }
            {
                GinRowArgs ga;
                memset(&ga, 0, sizeof(ga));
                ga.a = e->dXh; ga.y = yy; ga.t1 = t1; ga.bn = bnref(c, i, N, 0);
                ga.acc0 = node_acc(c, bn_dsum(c, i), H); ga.acc1 = node_acc(c, bn_dprod(c, i), H);
                RC(gin_rows(c, 1, ga));
                CAL_CHECK_LAUNCH("k_gin_dots"); STAGE();
                RC(flush_finals(c)); STAGE();
            }
            {
                GinRowArgs ga;
                memset(&ga, 0, sizeof(ga));
                ga.a = e->dXh; ga.y = yy; ga.t1 = t1; ga.out = dzi; ga.bn = bnref(c, i, N, 0);       // dt1
                ga.dot_sum = bn_dsum(c, i); ga.dot_prod = bn_dprod(c, i);
                ga.acc0 = deferred(H, d_gin_b1[i - 1]);
                if (!ga.acc0.on()) { set_error("engine: partial-row workspace exhausted"); return 2; }
                RC(gin_rows(c, 2, ga));
                CAL_CHECK_LAUNCH("k_gin_bn_bwd"); STAGE();
            }
            {
                GemmArgs a = gemm_args(H, H, N, true, false, 0);
                a.p[0].A = dzi; a.p[0].B = agg;
                float* dst[1] = {e->G + e->o_conv_w[i - 1]};
                RC(grad_gemm(c, a, 1, dst, fa, slab_off)); STAGE();
            }
            {
                GemmArgs a = gemm_args(N, H, H, false, false, 0);
                a.p[0].A = dzi; a.p[0].B = e->P + e->o_conv_w[i - 1]; a.p[0].C = e->dXh;            // dagg
                RC(fwd_gemm(c, false, a, 1)); STAGE();
            }
            {
                SpmmBranch br{e->dXh, e->z, nullptr, nullptr, e->ones, Acc(), Acc(), nullptr, nullptr, nullptr};                // d h_{i-1}
                RC(launch_espmm(st, gs, SpmmBranch2{{br, br}}, 1, 0, 1.0f, N, H, spmm_rpb(H, false)));
                CAL_CHECK_LAUNCH("k_espmm(gin,T)"); STAGE();
            }
            {
                GinRowArgs ga;
                memset(&ga, 0, sizeof(ga));
                ga.a = e->z; ga.y = e->h + (size_t)(i - 1) * NH; ga.out = e->dZ;                     // masked by h_{i-1} > 0
                ga.acc0 = i >= 2 ? deferred(H, d_convb[i - 2]) : Acc();                                  // d b2 of the layer below
                RC(gin_rows(c, 3, ga));
                CAL_CHECK_LAUNCH("k_gin_mask"); STAGE();
            }
            continue;
        }
        if (gat && gcb && (H / e->K == 32 || H / e->K == 64) && e->max_edges <= GGB_E) {
            // GATConv layer backward per graph (engine_ggat.hpp): attention backward + dX' (two slice partials) + dW / d att
            // slabs.  As in the GCNConv path below, layer i < L builds its dOut from layer i+1's partials while staging
            // (BatchNorm_{i+1}-backward + ReLU mask + bias sums: no k_bn_bwd launch, no dZ round trip); slice-0 partials
            // ping-pong between dXh and z (idle in the fused forward), slice-1 partials are per layer.
            const int K = e->K, D = H / K, nsl = H / GC_N;
            const size_t nk = al((size_t)e->capN * K);
            const float* sc = e->gsc + (size_t)(i - 1) * 4 * nk;
            const float* hin = e->h + (size_t)(i - 1) * NH;
            float* p0 = ((L - i) & 1) ? e->z : e->dXh;
            GgatBwdArgs ga;
            memset(&ga, 0, sizeof(ga));
            ga.x = hin; ga.W = e->P + e->o_conv_w[i - 1]; ga.att = e->P + e->o_conv_att[i - 1];
            ga.z = e->gz + (size_t)(i - 1) * NH; ga.adst = sc; ga.asrc = sc + nk; ga.mx = sc + 2 * nk; ga.den = sc + 3 * nk;
            ga.bn = bnref(c, i, N, 0); ga.dxp0 = p0; ga.dxp1 = dzi;
            ga.heads = K; ga.D = D; ga.slope = e->gat_slope; ga.p = c.training ? e->gat_p : 0.f;
            ga.seed = e->gat_seed[i - 1]; ga.ctr = (const uint64_t*)e->gat_ctr; ga.E = c.E;
            if (i == L) ga.dout = e->dZ;
            else {
                ga.dy0 = ((L - i - 1) & 1) ? e->z : e->dXh;
                ga.dy1 = H > GC_N ? e->dzi + (size_t)i * NH : nullptr;
                ga.y = e->h + (size_t)i * NH;
                ga.ubn = bnref(c, i + 1, N, 0); ga.udot_sum = bn_dsum(c, i + 1); ga.udot_prod = bn_dprod(c, i + 1);
                Deferred& db = d_convb[i - 1];
                db.p = parts_alloc(c, (size_t)T * H); db.P = T; db.stride = H;
                if (!db.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
                ga.bias_parts = db.p;
            }
            const size_t need_w = (size_t)T * H * H, need_a = (size_t)T * 2 * H;
            if (slab_off + need_w + need_a > e->slab_floats || fa.nst + 2 > MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
            ga.slab = e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{ga.slab, e->G + e->o_conv_w[i - 1], H * H, T};
            slab_off += need_w;
            ga.att_slab = e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{ga.att_slab, e->G + e->o_conv_att[i - 1], 2 * H, T};
            slab_off += need_a;
            if (striped_gat(c) && (i > 1 || (F <= FM_F && H <= FB_H))) {      // accumulator planes: the readers below are striped readers
                ga.dacc_sum = bn_dsum(c, i); ga.dacc_prod = bn_dprod(c, i); ga.dacc_ss = e->bn_plane;
            } else {
                double* pp = parts_alloc(c, (size_t)T * nsl * 2 * H);
                if (!pp) { set_error("engine: partial-row workspace exhausted"); return 2; }
                ga.dot_parts = pp;
                final_task(c, pp, T * nsl, 2 * H, H, bn_dsum(c, i));
                final_task(c, pp + H, T * nsl, 2 * H, H, bn_dprod(c, i));
            }
            {
                ProfScope ps(st, 8, 4.0 * N * H * H + 4.0 * (double)(c.E + N) * H, true);
                if (i == L) PROF_LAUNCH((k_ggat_bwd<false>), dim3(T, nsl), dim3(GB_NT), 0, st, gd, e->gptr, e->eptr, ga, N, H, H, e->status);
                else PROF_LAUNCH((k_ggat_bwd<true>), dim3(T, nsl), dim3(GB_NT), 0, st, gd, e->gptr, e->eptr, ga, N, H, H, e->status);
            }
            CAL_CHECK_LAUNCH("k_ggat_bwd"); STAGE();
            RC(flush_finals(c)); STAGE();
            if (i == 1 && F <= FM_F && H <= FB_H) {
                // the feature layer's backward per graph, fed from this layer's partial dX' (as in the GCNConv path)
                RC(feat_bwd(p0, H > GC_N ? dzi : nullptr, false)); STAGE();
            } else if (i == 1) {       // the feature layer below is a plain GEMM: materialise dZ for it
                BnBwdProb p{p0, hin, e->dZ, bnref(c, i, N, 0), bn_dsum(c, i), bn_dprod(c, i), Acc(), H > GC_N ? dzi : nullptr};
                RC(with_g(H, [&](auto g) {
                    constexpr int G = decltype(g)::value;
                    hipLaunchKernelGGL((k_bn_bwd<4, G>), dim3(cdiv(N, c.rpb_n), 1), dim3(256), 0, st, BnBwdProb3{{p, p, p}}, 1, N, H, c.rpb_n);
                    return 0;
                }));
                CAL_CHECK_LAUNCH("k_bn_bwd"); STAGE();
            }
            continue;
        }
        if (gcb && !gat) {
            // Layer i reads dOut = e->dZ (i == L, written by k_att_bwd) or builds it while staging from layer i+1's
            // partial dX' (BatchNorm_{i+1}-backward + ReLU mask fused in: no k_bn_bwd launch, no dZ round trip);
            // slice-0 partials ping-pong between dXh and z (idle in the fused forward), slice-1 partials are per layer.
            const float* hin = e->h + (size_t)(i - 1) * NH;
            float* p0 = ((L - i) & 1) ? e->z : e->dXh;
            GconvBwdBranch gb;
            memset(&gb, 0, sizeof(gb));
            gb.x = hin; gb.W = e->P + e->o_conv_w[i - 1]; gb.dis = e->dis_unit; gb.bn = bnref(c, i, N, 0);
            gb.dxp0 = p0; gb.dxp1 = dzi;
            gb.coef_in = e->coef;               // written by the forward's first fused layer
            if (gwb) {                          // CSR by source: the unit coefficients k_plan_graph left in that slot order, if it ran
                const bool fastp = e->node_ptr && e->edge_ptr && e->max_nodes > 0 && e->max_nodes <= GP_T2 && e->max_edges <= GP_E2;
                gb.coef_in = fastp ? e->coef_src : nullptr;
            }
            if (i == L) gb.dout = e->dZ;
            else {
                gb.dy0 = ((L - i - 1) & 1) ? e->z : e->dXh;
                gb.dy1 = H > GC_N ? e->dzi + (size_t)i * NH : nullptr;
                gb.y = e->h + (size_t)i * NH;
                gb.ubn = bnref(c, i + 1, N, 0); gb.udot_sum = bn_dsum(c, i + 1); gb.udot_prod = bn_dprod(c, i + 1);
                Deferred& db = d_convb[i - 1];  // bias of conv i: column sums of dOut, one partial row per graph
                db.p = parts_alloc(c, (size_t)T * H); db.P = T; db.stride = H;
                if (!db.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
                gb.bias_parts = db.p;
            }
            float* dst[1] = {e->G + e->o_conv_w[i - 1]};
            double* dsum[1] = {bn_dsum(c, i)}; double* dprod[1] = {bn_dprod(c, i)};
            { ProfScope ps(st, 4, 4.0 * N * H * H + 2.0 * (double)(c.E + N) * H, true); RC(gconv_bwd(c, gd, &gb, 1, dst, dsum, dprod, fa, slab_off, false, gwb ? gw_st : striped_bb(c) && (i > 1 || (F <= FM_F && H <= FB_H)), gwb ? &gs : nullptr)); } STAGE();
            RC(flush_finals(c)); STAGE();
            if (i == 1 && (gwb ? F <= FB_F : F <= FM_F) && H <= FB_H) {
                // the feature layer's backward per graph, fed from this layer's partial dX' (no k_bn_bwd, no dZ round trip)
                // (wide graphs: the same FMA kernel over uniform 64-row chunks, F <= 64 -- round 6; it replaces k_bn_bwd + the dual GEMM + a finishing launch)
                RC(feat_bwd(p0, H > GC_N ? dzi : nullptr, false)); STAGE();
            } else if (i == 1) {       // the feature layer below is a plain GEMM: materialise dZ for it
                BnBwdProb p{p0, hin, e->dZ, bnref(c, i, N, 0), bn_dsum(c, i), bn_dprod(c, i), Acc(), H > GC_N ? dzi : nullptr};
                RC(with_g(H, [&](auto g) {
                    constexpr int G = decltype(g)::value;
                    if (gw_st) hipLaunchKernelGGL((k_bn_bwd<4, G, true>), dim3(cdiv(N, c.rpb_n), 1), dim3(256), 0, st, BnBwdProb3{{p, p, p}}, 1, N, H, c.rpb_n);
                    else hipLaunchKernelGGL((k_bn_bwd<4, G>), dim3(cdiv(N, c.rpb_n), 1), dim3(256), 0, st, BnBwdProb3{{p, p, p}}, 1, N, H, c.rpb_n);
                    return 0;
                }));
                CAL_CHECK_LAUNCH("k_bn_bwd"); STAGE();
            }
            continue;
        }
        if (gat) {       // dz_i and d att_i from dOut_i (GATConv backward; alpha recomputed from the saved max / denominator)
            const int K = e->K, D = H / K;
            const size_t nk = al((size_t)e->capN * K);
            const float* sc = e->gsc + (size_t)(i - 1) * 4 * nk;
            ProfScope ps(st, 6, 4.0 * N * H * 4 + (double)(c.E + N) * (16 + 24.0 * K));
            // d att: per-block partial rows parked in the slab area, summed by the final k_finish (no finishing launch)
            const size_t need = (size_t)gat_datt_parts(N) * 2 * H;
            if (slab_off + need > e->slab_floats || fa.nst >= MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
            float* part = e->slabs + slab_off;
            int nparts = 0;
            RC(gat_backward(e->rowptr_dst, e->nbr_dst, e->eid_dst, e->rowptr_src, e->nbr_src, e->eid_src,
                            e->gz + (size_t)(i - 1) * NH, e->P + e->o_conv_att[i - 1], sc, sc + nk, sc + 2 * nk, sc + 3 * nk, e->dZ,
                            e->gat_slope, c.training ? e->gat_p : 0.f, e->gat_seed[i - 1], (const uint64_t*)e->gat_ctr, dzi,
                            e->G + e->o_conv_att[i - 1], e->gws, N, c.E, K, D, st, part, &nparts));
            fa.st[fa.nst++] = SlabTask{part, e->G + e->o_conv_att[i - 1], 2 * H, nparts};
            slab_off += need;
        } else {
            SpmmBranch br{e->dZ, dzi, nullptr, nullptr, e->dis_unit, Acc(), Acc(), nullptr, nullptr, nullptr};
            ProfScope ps(st, 1, 2.0 * N * H * 4 + (double)(c.E + N) * 8 + (N + 1) * 4.0, true);
            RC(launch_espmm(st, gs, SpmmBranch2{{br, br}}, 1, 0, e->loop_w, N, H, spmm_rpb(H, false)));
            CAL_CHECK_LAUNCH("k_espmm(T)");
        }
        STAGE();
        const float* hin = e->h + (size_t)(i - 1) * NH;
        {
            GemmArgs a = gemm_args(H, H, N, true, false, 0);
            a.p[0].A = hin; a.p[0].B = dzi;
            a.p[0].xa.has_bn = 1; a.p[0].xa.bn = bnref(c, i, N, 0);
            float* dst[1] = {e->G + e->o_conv_w[i - 1]};
            GemmArgs aw = a;
            a = gemm_args(N, H, H, false, true, 0);
            a.p[0].A = dzi; a.p[0].B = e->P + e->o_conv_w[i - 1]; a.p[0].C = e->dXh;
            a.p[0].aux = hin; a.p[0].has_aux = 1; a.p[0].aux_bn = bnref(c, i, N, 0);
            gemm_stats(c, a.p[0], N, H, bn_dsum(c, i), bn_dprod(c, i), true, H, striped_node(c));
            { ProfScope ps(st, 3, 4.0 * N * H * H); RC(dual_gemm(c, a, 1, aw, 1, dst, fa, slab_off)); } STAGE();
            RC(flush_finals(c)); STAGE();
        }
        if (i == 1 && !feat_done && feat_rows(c)) {
            // the last BatchNorm backward feeds only the feature layer: dZ stays in registers (engine_feat.hpp)
            const int P = cdiv(N, c.rpb_n), cols = (F + 1) * H;
            FeatBwdRowsArgs fb{e->dXh, hin, bnref(c, 1, N, 0), bn_dsum(c, 1), bn_dprod(c, 1), x0, bnref(c, 0, N, 0), parts_alloc(c, (size_t)P * cols)};
            double* sums = parts_alloc(c, cols);
            d_bn0.p = parts_alloc(c, 2 * (size_t)F); d_bn0.P = 1; d_bn0.stride = 2 * F;
            if (!fb.parts || !sums || !d_bn0.p) { set_error("engine: partial-row workspace exhausted"); return 2; }
            if (slab_off + (size_t)F * H > e->slab_floats || fa.nst >= MAX_SLABS) { set_error("engine: slab workspace exhausted"); return 2; }
            float* dw = e->slabs + slab_off;
            fa.st[fa.nst++] = SlabTask{dw, e->G + e->o_feat_w, F * H, 1};
            slab_off += (size_t)F * H;
            RC(with_g_fp(H, F, [&](auto g, auto fp) {
                hipLaunchKernelGGL((k_bn_bwd_feat<decltype(g)::value, decltype(fp)::value>), dim3(P), dim3(256), 0, st, fb, N, H, F, c.rpb_n);
                return 0;
            }));
            CAL_CHECK_LAUNCH("k_bn_bwd_feat"); STAGE();
            final_task(c, fb.parts, P, cols, cols, sums);
            RC(flush_finals(c)); STAGE();
            hipLaunchKernelGGL(k_feat_bwd_final, dim3(F), dim3(256), 0, st, sums, e->P + e->o_feat_w, bnref(c, 0, N, 0), dw, d_bn0.p, H, F);
            CAL_CHECK_LAUNCH("k_feat_bwd_final"); STAGE();
            feat_done = true;
        } else {
            BnBwdProb p{e->dXh, hin, e->dZ, bnref(c, i, N, 0), bn_dsum(c, i), bn_dprod(c, i),
                        i >= 2 ? deferred(H, d_convb[i - 2]) : Acc()};
            RC(with_g(H, [&](auto g) {
                constexpr int G = decltype(g)::value;
                if (striped_node(c)) hipLaunchKernelGGL((k_bn_bwd<4, G, true>), dim3(cdiv(N, c.rpb_n), 1), dim3(256), 0, st, BnBwdProb3{{p, p, p}}, 1, N, H, c.rpb_n);
                else hipLaunchKernelGGL((k_bn_bwd<4, G>), dim3(cdiv(N, c.rpb_n), 1), dim3(256), 0, st, BnBwdProb3{{p, p, p}}, 1, N, H, c.rpb_n);
                return 0;
            }));
            CAL_CHECK_LAUNCH("k_bn_bwd"); STAGE();
        }
    }
    // S. conv_feat weight and bn_feat affine gradients
    if (!feat_done) {
        GemmArgs a = gemm_args(F, H, N, true, false, 0);
        a.p[0].A = x0; a.p[0].B = e->dZ;
        a.p[0].xa.has_bn = 1; a.p[0].xa.bn = bnref(c, 0, N, 0);
        float* dst[1] = {e->G + e->o_feat_w};
        GemmArgs aw = a;
        a = gemm_args(N, F, H, false, true, 0);
        a.p[0].A = e->dZ; a.p[0].B = e->P + e->o_feat_w; a.p[0].C = nullptr;
        a.p[0].aux = x0; a.p[0].has_aux = 1; a.p[0].aux_bn = bnref(c, 0, N, 0);
        a.p[0].dot_sum = bn_dsum(c, 0); a.p[0].dot_prod = bn_dprod(c, 0);
        {   // bn_feat's sums are only needed by the commit: keep the partial rows, no finalise launch
            const int P0 = use_ks(N) ? gemm_ks_row_tiles(N) : gemm_row_tiles(N, F, H, false);
            if ((size_t)P0 * F * 2 > 4096) {
                d_bn0.p = parts_alloc(c, (size_t)P0 * 2 * F); d_bn0.P = P0; d_bn0.stride = 2 * F;
                a.p[0].parts = d_bn0.p;
            }
        }
        RC(dual_gemm(c, a, 1, aw, 1, dst, fa, slab_off)); STAGE();
    }
    // commits: fp64 arena -> fp32 gradients
    for (int k = 0; k < e->nbn; ++k) {
        const BNSlot& b = e->bn[k];
        const int wp = (b.width + 3) / 4 * 4;
        if (k == 0 && d_bn0.p) {
            commit_p(d_bn0.p + F, d_bn0.P, d_bn0.stride, b.gamma, F, 1.f);
            commit_p(d_bn0.p, d_bn0.P, d_bn0.stride, b.beta, F, 1.f);
            continue;
        }
        // (the NSTRIPE accumulator planes of the site: engine.hpp, stripe_sum)
        commit_p(e->arena + b.arena + 3 * wp, NSTRIPE, e->bn_plane, b.gamma, b.width, 1.f);   // d gamma = sum dyh * x_n
        commit_p(e->arena + b.arena + 2 * wp, NSTRIPE, e->bn_plane, b.beta, b.width, 1.f);    // d beta  = sum dyh
    }
    for (int i = 0; i < L; ++i) {
        if (!d_convb[i].p) { set_error("engine: missing bias-gradient partials"); return 2; }
        // GCNConv / GATConv: the layer's bias; GINConv: the bias of its second Linear (the first one's has its own sums)
        commit_p(d_convb[i].p, d_convb[i].P, d_convb[i].stride, e->gin ? e->o_gin_b2[i] : e->o_conv_b[i], H, 1.f);
        if (e->gin) {
            if (!d_gin_b1[i].p) { set_error("engine: missing bias-gradient partials"); return 2; }
            commit_p(d_gin_b1[i].p, d_gin_b1[i].P, d_gin_b1[i].stride, e->o_conv_b[i], H, 1.f);
        }
    }
    commit_p(d_cb.p, d_cb.P, d_cb.stride, e->o_cb, H, 1.f);
    commit_p(d_ob.p, d_ob.P, d_ob.stride, e->o_ob, H, 1.f);
    commit_p(d_dwn.p, d_dwn.P, d_dwn.stride, e->o_natt_w, H, 1.f);
    commit_p(d_dwn.p, d_dwn.P, d_dwn.stride, e->o_natt_w + H, H, -1.f);
    commit_p(d_dwn.p + H, d_dwn.P, d_dwn.stride, e->o_natt_b, 1, 1.f);
    commit_p(d_dwn.p + H, d_dwn.P, d_dwn.stride, e->o_natt_b + 1, 1, -1.f);
    commit_p(d_dwe.p, d_dwe.P, d_dwe.stride, e->o_eatt_w, 2 * H, 1.f);
    commit_p(d_dwe.p, d_dwe.P, d_dwe.stride, e->o_eatt_w + 2 * H, 2 * H, -1.f);
    commit_p(d_dwe.p + 2 * H, d_dwe.P, d_dwe.stride, e->o_eatt_b, 1, 1.f);
    commit_p(d_dwe.p + 2 * H, d_dwe.P, d_dwe.stride, e->o_eatt_b + 1, 1, -1.f);
    for (int hd = 0; hd < 3; ++hd) {
        commit(e->a_db1 + hd * H, e->o_fc1_b[hd], H, 1.f);
        commit(e->a_db2 + hd * C, e->o_fc2_b[hd], C, 1.f);
    }
    if (fa.nct > MAX_COMMITS) { set_error("engine: too many commit tasks"); return 2; }
    join_side(c);
    {
        if (c.adam_in_finish) {
            // Adam rides in this kernel: parameter ranges no task writes (gradients stored directly by the readout /
            // single-slab GEMMs) become update-only tasks
            fa.adam = AdamArgs{e->P, e->M1, e->M2, e->step, e->lr, e->beta1, e->beta2, e->eps, e->wd, e->grad_scale, 1};
            std::vector<std::pair<int64_t, int64_t>> iv;
            for (int i = 0; i < fa.nst; ++i) iv.push_back({fa.st[i].dst - e->G, (fa.st[i].dst - e->G) + fa.st[i].n});
            for (int i = 0; i < fa.nct; ++i) iv.push_back({fa.ct[i].dst, (int64_t)fa.ct[i].dst + fa.ct[i].n});
            std::sort(iv.begin(), iv.end());
            int64_t pos = 0;
            bool ok = true;
            auto gap = [&](int64_t a, int64_t b) {
                if (b <= a) return;
                if (fa.nar >= MAX_ADAM_RANGES) { ok = false; return; }
                fa.ar[fa.nar++] = AdamRange{a, b};
            };
            for (auto& r : iv) {
                if (r.first < pos) { ok = false; break; }          // two tasks finish the same element: not expected
                gap(pos, r.first);
                pos = r.second;
            }
            gap(pos, e->nparam);
            if (!ok || pos > e->nparam) {                          // keep the separate k_adam launch
                fa.adam.on = 0; fa.nar = 0; c.adam_in_finish = 2;     // 2: k_adam follows, counter already advanced
            }
        }
        int nblk = 0;
        for (int i = 0; i < fa.nst; ++i) { fa.blk0[i] = nblk; nblk += std::max(1, cdiv(fa.st[i].n, 64)); }
        for (int i = 0; i < fa.nct; ++i) { fa.blk0[fa.nst + i] = nblk; nblk += std::max(1, cdiv(fa.ct[i].n, 16)); }
        for (int i = 0; i < fa.nar; ++i) { fa.blk0[fa.nst + fa.nct + i] = nblk; nblk += (int)cdiv(fa.ar[i].end - fa.ar[i].begin, 256); }
        fa.blk0[fa.nst + fa.nct + fa.nar] = nblk;
        hipLaunchKernelGGL(k_finish, dim3(nblk), dim3(256), 0, st, fa, e->G);
    }
    CAL_CHECK_LAUNCH("k_finish"); STAGE();
    e->bn0_dirty_host = 0;          // bn_feat's statistics range is clean again behind this launch (PlanFold)
    return 0;
}

}  // namespace

// One training step (or, with mode bits cleared, parts of it) for a device-resident batch.
//   x0 [N,F] fp32, edge_index [2,E] int64, batch [N] int64 (sorted), y [B] int64, perm [B] int64
//   (the random-intervention index, model.py:147-152; identity = arange).
//   mode: bit0 = forward in training mode (batch statistics, running-stat updates);
//         bit1 = loss gradient + backward (fills the flat gradient buffer);
//         bit2 = Adam update.  mode = 0 is an eval-mode forward.
//   Outputs live in the workspace: "logp" [3,B,C] log-probabilities (c, o, co), "stats" [5].
CAL_EXPORT int cal_engine_step(void* h, const float* x0, const int64_t* edge_index, const int64_t* batch, const int64_t* y,
                               const int64_t* perm, int64_t N, int64_t E, int64_t B, float wc, float wo, float wco, int mode,
                               void* stream_) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e && e->ws, "engine has no workspace");
    CAL_REQUIRE(N <= e->capN && E <= e->capE && B <= e->capB, "batch exceeds the workspace capacity");
    CAL_REQUIRE(N > 0 && B > 0, "empty batch");
    Ctx c;
    c.e = e; c.st = (hipStream_t)stream_; c.N = (int)N; c.B = (int)B; c.E = E;
    CAL_REQUIRE(e->ntiles <= B, "more tiles than graphs (cal_engine_set_tiles belongs to another batch)");
    c.T = e->ntiles > 0 ? e->ntiles : (int)B;
    c.batch = batch;
    CAL_REQUIRE(e->ntiles == 0 || (e->H % GC_N == 0 && e->H <= GC_K && e->node_ptr && e->edge_ptr && e->max_nodes > 0 && e->max_nodes <= 64 &&
                                   e->max_edges <= gc_edge_cap(64)),
                "cal_engine_set_tiles needs the per-graph kernels: hidden in {64, 128}, the tiles' offsets (cal_engine_set_graph_ptrs) and bounds <= 64 nodes / 1024 edges");
    c.training = (mode & 1) ? 1 : 0;
    // (16 k - 64 k rows: 512 workgroups -- a workgroup's prologue / column-sum epilogue over 32 rows was a third of k_att_bwd at 30 k rows)
    c.rpb_n = std::max(32, cdiv(N, e->rpb_div ? e->rpb_div : (N >= 16384 && N <= 65536 ? 512 : 1024)));
    c.rpb_b = std::max(32, cdiv(B, 64));
    c.parts_off = 0;
    c.fin.nt = 0;
    c.nfork = 0;
    c.ro_done = 0;
    c.adam_in_finish = ((mode & 4) && e->adam_fused) ? 1 : 0;
    const int want_grad = (mode & 2) ? 1 : 0;
    c.draw_perm = (mode & 16) ? 1 : 0;
    CAL_REQUIRE(!c.draw_perm || (e->perm_ctr && B <= ZP_CAP && want_grad), "mode bit 16 needs cal_engine_set_perm_rng, at most 1024 graphs per batch and the backward pass (its last kernel advances the permutation counter)");
    CAL_REQUIRE(c.draw_perm || perm, "perm is null and the step does not draw its own (mode bit 16)");
    if (c.draw_perm) perm = e->perm_dev;
    c.y = y; c.perm = perm; c.wc = wc; c.wo = wo; c.wco = wco; c.want_grad = want_grad;
    c.tick_in_finish = (mode & (4 | 8)) ? 1 : 0;
    CAL_REQUIRE(!(mode & 8) || want_grad, "mode bit 8 (Adam follows a gradient exchange) needs the backward pass");
    CAL_REQUIRE(!want_grad || c.training, "backward needs a training-mode forward");
    CAL_REQUIRE(!(mode & 4) || want_grad, "the Adam update needs the backward pass in the same step");
    g_stage = 0;
    g_stage_names.clear();
    {
        int rc = engine_forward(c, x0, edge_index, batch, y, perm, wc, wo, wco, want_grad);
        if (rc == -12345) return 0;
        if (rc) return rc;
        if (want_grad) {
            rc = engine_backward(c, x0, batch);
            if (rc) join_side(c);
            if (rc == -12345) return 0;
            if (rc) return rc;
        }
    }
    if ((mode & 4) && c.adam_in_finish != 1) {          // k_finish has already advanced the step counter (c.tick_in_finish)
        hipLaunchKernelGGL(k_adam, dim3(cdiv(e->nparam, 256)), dim3(256), 0, c.st, e->P, e->G, e->M1, e->M2, e->step, e->lr, e->beta1,
                           e->beta2, e->eps, e->wd, e->nparam, 1, e->grad_scale, e->status);
        CAL_CHECK_LAUNCH("k_adam");
    }
    return 0;
}

// Backward from an external gradient (autograd surface): `dlogp` [3,B,C] = d loss / d log-probs of
// the LAST training-mode forward of the same batch (cal_engine_step with mode = 1); fills the flat
// gradient buffer exactly like mode bit 2.  The forward's activations, plan and statistics are
// taken from the workspace, so no other step may run in between.
CAL_EXPORT int cal_engine_backward_from(void* h, const float* x0, const int64_t* batch, const float* dlogp, int64_t N,
                                        int64_t E, int64_t B, void* stream_) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e && e->ws, "engine has no workspace");
    CAL_REQUIRE(N <= e->capN && E <= e->capE && B <= e->capB && N > 0 && B > 0, "bad batch sizes");
    Ctx c;
    c.e = e; c.st = (hipStream_t)stream_; c.N = (int)N; c.B = (int)B; c.E = E;
    CAL_REQUIRE(e->ntiles <= B, "more tiles than graphs (cal_engine_set_tiles belongs to another batch)");
    c.T = e->ntiles > 0 ? e->ntiles : (int)B;
    c.batch = batch;
    c.training = 1;
    // (16 k - 64 k rows: 512 workgroups -- a workgroup's prologue / column-sum epilogue over 32 rows was a third of k_att_bwd at 30 k rows)
    c.rpb_n = std::max(32, cdiv(N, e->rpb_div ? e->rpb_div : (N >= 16384 && N <= 65536 ? 512 : 1024)));
    c.rpb_b = std::max(32, cdiv(B, 64));
    c.parts_off = 0; c.fin.nt = 0; c.nfork = 0;
    c.y = nullptr; c.perm = nullptr; c.wc = c.wo = c.wco = 0.f; c.want_grad = 1;
    c.tick_in_finish = 0; c.draw_perm = 0; c.ro_done = 0; c.adam_in_finish = 0;
    hipLaunchKernelGGL(k_logsoftmax_bwd, dim3(1), dim3(256), 0, c.st, e->logp, dlogp, e->dzl, e->arena + e->a_db2, (int)B, e->C);
    CAL_CHECK_LAUNCH("k_logsoftmax_bwd");
    g_stage = 0;
    int rc = engine_backward(c, x0, batch);
    if (rc) join_side(c);
    return rc == -12345 ? 0 : rc;
}

// The 3-term loss of train_causal.py:176-183 on log-probabilities logp [3, B, C] (heads c, o, co) with labels y [B]: out [4] =
// {loss, c_loss, o_loss, co_loss}, and (dlogp non-null) d loss / d logp [3, B, C] -- what cal_engine_backward_from takes.
// flag (may be null): bit 1 is set when a label lies outside [0, C).
CAL_EXPORT int cal_causal_loss(const float* logp, const int64_t* y, int64_t B, int64_t C, float wc, float wo, float wco, float* out,
                               float* dlogp, int32_t* flag, void* stream_) {
    CAL_REQUIRE(logp && y && out && B > 0 && C > 0 && B * C < (1ll << 30), "bad arguments");
    hipLaunchKernelGGL(k_causal_loss, dim3(1), dim3(256), 0, (hipStream_t)stream_, logp, y, (int)B, (int)C, wc, wo, wco, out, dlogp, (int*)flag);
    CAL_CHECK_LAUNCH("k_causal_loss");
    return 0;
}

// Per-graph bounds of the batches that follow (largest node count / stored-edge count of any single graph;
// 0 = unknown).  With bounds that fit (engine_gconv.hpp) the step runs the per-graph fused convolutions; a
// bound that turns out too small sets bit 3 of the status word and leaves that graph's outputs unwritten.
CAL_EXPORT int cal_engine_set_graph_bounds(void* h, int64_t max_nodes, int64_t max_edges) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr && max_nodes >= 0 && max_edges >= 0, "bad arguments");
    e->max_nodes = (int)std::min<int64_t>(max_nodes, 1 << 30);
    e->max_edges = (int)std::min<int64_t>(max_edges, 1 << 30);
    return 0;
}
// Layout of the coming batch, as every collate knows it: graph b owns nodes [node_ptr[b], node_ptr[b+1]) and the
// contiguous edge_index columns [edge_ptr[b], edge_ptr[b+1]), and no edge is a self loop (both [B+1] int64 DEVICE
// arrays, alive until the step has run; null = unknown).  Together with the per-graph bounds this selects the
// one-kernel per-graph GraphPlan; what the kernel finds violated is flagged in the status word.
CAL_EXPORT int cal_engine_set_graph_ptrs(void* h, const int64_t* node_ptr, const int64_t* edge_ptr) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr, "bad arguments");
    e->node_ptr = node_ptr; e->edge_ptr = edge_ptr;
    return 0;
}
// Small-graph packing.  The per-graph kernels give every workgroup a 64-row MFMA tile; a batch of ~30-node graphs (NCI1,
// MUTAG: BASELINE.json configs[2..3]) leaves half of it empty and launches twice the workgroups.  A batch is a disjoint union,
// so any run of CONSECUTIVE graphs is itself a block-diagonal graph: the caller groups consecutive graphs into tiles
// (<= 64 nodes, <= 1024 edges, <= 8 graphs each), passes the TILES' node / edge offsets and bounds through
// cal_engine_set_graph_ptrs / cal_engine_set_graph_bounds, and here the first graph of every tile: tile_gptr [ntiles + 1]
// (device int64, tile_gptr[ntiles] = B; alive until the step has run).  Only the add-pool and its backward look at graphs
// inside a tile (model.py:115-116).  ntiles = 0 / null: one graph per workgroup.
CAL_EXPORT int cal_engine_set_tiles(void* h, const int64_t* tile_gptr, int64_t ntiles) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr && ntiles >= 0 && (ntiles == 0 || tile_gptr != nullptr), "bad arguments");
    e->tile_gptr = ntiles > 0 ? tile_gptr : nullptr; e->ntiles = (int)ntiles;
    return 0;
}
// Deterministic mode (atomic-free, bit-reproducible reductions): on = 1 turns the striped fp64 accumulators
// (engine.hpp: stripe_sum -- order-dependent in the last bits) off for this engine: every BatchNorm sum is then a partial row per
// workgroup, added in a fixed order by k_stats_final (what CAL_AMD_STRIPED=0 selects process-wide).  Same results to ~1e-16
// relative, bit-identical from run to run; a few finishing launches per step slower.  Call before the first step is captured.
CAL_EXPORT int cal_engine_set_deterministic(void* h, int on) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr, "bad arguments");
    e->striped = on ? 0 : 1;
    return 0;
}
CAL_EXPORT int cal_engine_debug_stop(int k) { g_stop_after = k; return 0; }
// name of launch site k (1-based, as counted by cal_engine_debug_stop) in the latest untruncated step; "" past the end
CAL_EXPORT const char* cal_engine_stage_name(int k) {
    return k >= 1 && k <= (int)g_stage_names.size() ? g_stage_names[k - 1] : "";
}
#ifdef CAL_BLK_CLOCKS
CAL_EXPORT int cal_debug_blk_clocks(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cal::g_blk_clk), sizeof(long long) * 8192);
}
#endif
#ifdef CAL_RO_CLOCKS
CAL_EXPORT int cal_debug_ro_clocks(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cal::g_ro_clk), sizeof(long long) * 64);
}
#endif

// Live timing (bench.py roofline): enable, run steps eagerly, synchronise, read.
CAL_EXPORT int cal_engine_profile(int on) {
    g_prof_on = on != 0;
    if (on) {
        for (auto& r : g_prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
        g_prof.clear();
    }
    return 0;
}
// Fills out[3*i + {0,1,2}] = (class, milliseconds, algorithmic work: flops for class 0, bytes for
// class 1) for up to cap records; returns the number of records.  Call after a stream sync.
CAL_EXPORT int64_t cal_engine_profile_read(double* out, int64_t cap) {
    int64_t n = 0;
    for (auto& r : g_prof) {
        if (n >= cap) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
        out[3 * n] = r.cls; out[3 * n + 1] = ms; out[3 * n + 2] = r.work;
        ++n;
    }
    return n;
}

// Adam alone (after an external gradient all-reduce)
CAL_EXPORT int cal_engine_adam(void* h, void* stream_) {
    Engine* e = (Engine*)h;
    hipStream_t st = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_adam, dim3(cdiv(e->nparam, 256)), dim3(256), 0, st, e->P, e->G, e->M1, e->M2, e->step, e->lr, e->beta1,
                       e->beta2, e->eps, e->wd, e->nparam, 0, e->grad_scale, e->status);
    CAL_CHECK_LAUNCH("k_adam");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, e->step);
    CAL_CHECK_LAUNCH("k_adam_tick");
    return 0;
}
// Adam after a gradient exchange when the step ran with mode bit 8 (its k_finish already advanced the step counter):
// ONE launch -- [graph: forward + backward] -> all-reduce(sum) -> this, with the 1/world mean folded into grad_scale
CAL_EXPORT int cal_engine_adam_ticked(void* h, void* stream_) {
    Engine* e = (Engine*)h;
    hipStream_t st = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_adam, dim3(cdiv(e->nparam, 256)), dim3(256), 0, st, e->P, e->G, e->M1, e->M2, e->step, e->lr, e->beta1,
                       e->beta2, e->eps, e->wd, e->nparam, 1, e->grad_scale, e->status);
    CAL_CHECK_LAUNCH("k_adam");
    return 0;
}
// ---- one-shot gradient exchange over peer-mapped memory (k_p2p_adam, engine_kernels.hpp) ------------------------------
// The regions must be FINE-GRAINED device memory: a kernel that polls a flag a peer GPU writes and then reads the peer's bucket
// while both kernels run needs stores that become visible across devices without a kernel boundary; coarse-grained hipMalloc
// memory (what torch allocates) promises that only at kernel boundaries (RCCL allocates its buffers the same way).
CAL_EXPORT int cal_p2p_alloc(int64_t bytes, void** out) {
    CAL_REQUIRE(bytes > 0 && out, "bad arguments");
    void* p = nullptr;
    hipError_t rc = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    if (rc != hipSuccess || !p) { (void)hipGetLastError(); set_error("cal_p2p_alloc: hipExtMallocWithFlags(fine-grained, %lld bytes): %s", (long long)bytes, hipGetErrorString(rc)); return 3; }
    rc = hipMemset(p, 0, (size_t)bytes);
    if (rc == hipSuccess) rc = hipDeviceSynchronize();
    if (rc != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); set_error("cal_p2p_alloc: zeroing failed: %s", hipGetErrorString(rc)); return 3; }
    *out = p;
    return 0;
}
CAL_EXPORT int cal_p2p_free(void* p) {
    if (p && hipFree(p) != hipSuccess) { (void)hipGetLastError(); set_error("cal_p2p_free failed"); return 3; }
    return 0;
}
// 64-byte IPC handle of a cal_p2p_alloc region (hipIpcGetMemHandle) / its mapping in another process (hipIpcOpenMemHandle)
CAL_EXPORT int cal_p2p_export(void* p, void* handle64) {
    CAL_REQUIRE(p && handle64, "bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t hd;
    hipError_t rc = hipIpcGetMemHandle(&hd, p);
    if (rc != hipSuccess) { (void)hipGetLastError(); set_error("cal_p2p_export: hipIpcGetMemHandle: %s", hipGetErrorString(rc)); return 3; }
    memcpy(handle64, &hd, 64);
    return 0;
}
CAL_EXPORT int cal_p2p_open(const void* handle64, void** out) {
    CAL_REQUIRE(handle64 && out, "bad arguments");
    hipIpcMemHandle_t hd;
    memcpy(&hd, handle64, 64);
    void* p = nullptr;
    hipError_t rc = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
    if (rc != hipSuccess || !p) { (void)hipGetLastError(); set_error("cal_p2p_open: hipIpcOpenMemHandle: %s", hipGetErrorString(rc)); return 3; }
    *out = p;
    return 0;
}
CAL_EXPORT int cal_p2p_close(void* p) {
    if (p && hipIpcCloseMemHandle(p) != hipSuccess) { (void)hipGetLastError(); set_error("cal_p2p_close failed"); return 3; }
    return 0;
}
// Bytes of the region every rank allocates (cal_p2p_alloc: zero-initialised, its own allocation) and shares with the others.
CAL_EXPORT int64_t cal_engine_p2p_region_bytes(void* h) {
    Engine* e = (Engine*)h;
    const int64_t np = (e->nparam + 63) / 64 * 64;
    return (2 * np + 64) * 4;
}
// peer_bases[r]: device address, in THIS process, of rank r's region (its own at index `rank`); peer_devices[r]: the HIP device
// that owns it (peer access is enabled here when it is another device), or null when all regions live on the current device
CAL_EXPORT int cal_engine_p2p_bind(void* h, void* const* peer_bases, const int64_t* peer_devices, int64_t world, int64_t rank) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e && e->P && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world && peer_bases, "bad arguments (bind the parameters first; at most 8 ranks)");
    int dev = 0;
    hipGetDevice(&dev);
    memset(&e->p2p, 0, sizeof(e->p2p));
    for (int r = 0; r < world; ++r) {
        CAL_REQUIRE(peer_bases[r] != nullptr, "null peer region");
        e->p2p.region[r] = (float*)peer_bases[r];
        if (peer_devices && (int)peer_devices[r] != dev) {
            hipError_t rc = hipDeviceEnablePeerAccess((int)peer_devices[r], 0);
            if (rc != hipSuccess && rc != hipErrorPeerAccessAlreadyEnabled) { set_error("cal_engine_p2p_bind: no peer access to device %d", (int)peer_devices[r]); return 3; }
            (void)hipGetLastError();
        }
    }
    if (!e->p2p_host_status) {
        hipError_t rc = hipHostMalloc((void**)&e->p2p_host_status, 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (rc != hipSuccess) { (void)hipGetLastError(); e->p2p_host_status = nullptr; set_error("cal_engine_p2p_bind: hipHostMalloc failed"); return 3; }
    }
    *e->p2p_host_status = 0;
    e->p2p.host_status = e->p2p_host_status; e->p2p.max_polls = e->p2p_max_polls;
    e->p2p.rank = (int)rank; e->p2p.world = (int)world; e->p2p.np = (e->nparam + 63) / 64 * 64;
    e->p2p_on = 1;
    return 0;
}
// Bound of the peer-flag wait in polls (~0.5-2 us each; default 2^22).  A launch ENQUEUED (or captured) after this call uses it.
CAL_EXPORT int cal_engine_p2p_set_timeout(void* h, int64_t max_polls) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e && max_polls >= 1 && max_polls <= (1ll << 30), "bad arguments");
    e->p2p_max_polls = (int)max_polls; e->p2p.max_polls = (int)max_polls;
    return 0;
}
// The status words (this step's | the sticky earlier ones) as the latest COMPLETED training step left them: a host-mapped word the
// step's last kernel refreshes, so the loops can look at it before every step without a synchronisation -- a flagged batch
// (whose step updated nothing, and neither does any later one) surfaces a step or two later instead of at the end of the epoch.
CAL_EXPORT int64_t cal_engine_peek_status(void* h) {
    Engine* e = (Engine*)h;
    if (!e || !e->host_status) return 0;
    return (int64_t)__atomic_load_n(e->host_status, __ATOMIC_ACQUIRE);
}
// 0, or 64 once an exchange has timed out (the word is host-mapped: no device synchronisation, cheap enough for every step).
// After a timeout k_p2p_adam updates nothing any more; allocate fresh regions and bind again to resume.
CAL_EXPORT int64_t cal_engine_p2p_status(void* h) {
    Engine* e = (Engine*)h;
    if (!e || !e->p2p_host_status) return 0;
    return (int64_t)__atomic_load_n(e->p2p_host_status, __ATOMIC_ACQUIRE);
}
// The exchange + update of a step that ran with mode bit 8 (its last kernel advanced the Adam step counter): ONE launch.
// The gradient factor is cal_engine_set_grad_scale's (1 / world for the replicas' mean).
CAL_EXPORT int cal_engine_p2p_adam(void* h, void* stream_) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e && e->p2p_on && e->ws, "cal_engine_p2p_bind first");
    const AdamArgs A{e->P, e->M1, e->M2, e->step, e->lr, e->beta1, e->beta2, e->eps, e->wd, e->grad_scale, 1};
    const int nb = (int)std::min<int64_t>(P2P_BLOCKS, cdiv(e->nparam, 256));
    hipLaunchKernelGGL(k_p2p_adam, dim3(nb), dim3(256), 0, (hipStream_t)stream_, e->p2p, e->G, A, e->nparam, e->status);
    CAL_CHECK_LAUNCH("k_p2p_adam");
    return 0;
}
// Model variants (opts.py:96-103 / model.py:24-31): cat = the random-intervention readout concatenates xc[perm] and xo
// (fc1_bn_co and fc1_co are 2H wide) instead of adding them; no_node_att / no_edge_att = the ablation flags (constant 0.5
// node / edge masks, no gradient into the corresponding attention MLP).  Call before cal_engine_bind.
CAL_EXPORT int cal_engine_set_options(void* h, int cat, int no_node_att, int no_edge_att) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr, "bad arguments");
    e->cat = cat ? 1 : 0; e->no_node_att = no_node_att ? 1 : 0; e->no_edge_att = no_edge_att ? 1 : 0;
    return 0;
}
// CausalGIN (model.py:166-264): the backbone layers are GINConv(Sequential(Linear, BatchNorm1d, ReLU, Linear, ReLU)) (model.py:188-194,
// eps = 0): per layer the slots {nn.1.weight, nn.1.bias, nn.0.weight, nn.0.bias, nn.3.weight, nn.3.bias} replace the four of a
// GCNConv layer in `offs`, and BatchNorm i of `bn_ptrs` is convs.(i-1).nn.1.  Call before cal_engine_bind.  The workspace buffer
// "ones" ([N] floats) must hold 1.0 (unit aggregation coefficients).
CAL_EXPORT int cal_engine_set_gin(void* h, int on) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr && (!on || e->K == 0), "bad arguments");
    e->gin = on ? 1 : 0;
    return 0;
}
// Random-intervention permutation drawn by the step itself (mode bit 16; model.py:147-152): keyed by (seed, *counter), the
// device counter advancing once per drawing step.  The drawn permutation is the workspace buffer "perm" (int64 [B]).
CAL_EXPORT int cal_engine_set_perm_rng(void* h, uint64_t seed, uint64_t* counter) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr, "bad arguments");
    e->perm_seed = seed; e->perm_ctr = (unsigned long long*)counter;
    return 0;
}
// Adam hyper-parameters after cal_engine_bind (an optimizer object's param group may change them between steps:
// torch.optim.Adam(params, lr, betas, eps, weight_decay), train_causal.py:21,76); the learning rate is the bound device float
CAL_EXPORT int cal_engine_set_adam(void* h, float beta1, float beta2, float eps, float weight_decay) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f, "bad arguments");
    e->beta1 = beta1; e->beta2 = beta2; e->eps = eps; e->wd = weight_decay;
    return 0;
}
// Factor applied to the bound gradient buffer inside the Adam update (1/world_size turns the all-reduced SUM of the
// replicas' gradients into the mean, train_causal.py:187-192 semantics per replica); default 1
CAL_EXPORT int cal_engine_set_grad_scale(void* h, float scale) {
    Engine* e = (Engine*)h;
    CAL_REQUIRE(e != nullptr && scale > 0.f, "bad arguments");
    e->grad_scale = scale;
    return 0;
}
