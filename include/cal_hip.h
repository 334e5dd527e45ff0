/* libcalhip -- C ABI of the MI355X (gfx950) kernels behind cal_amd.
 *
 * The reference (yongduosui/CAL) has no FFI: its hot path is Python calling
 * PyTorch-Geometric / torch_scatter operators.  Each entry point below replaces
 * one such operator call site (cited file:line, paths relative to the reference
 * root) and is what a binding of that path would bind (INTEGRATION.md shows the
 * ctypes stub).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; cal_last_error()
 *    returns the (thread-local) message of the last failure;
 *  - all pointers are DEVICE pointers owned by the caller (torch allocations),
 *    contiguous row-major; float buffers 16-byte aligned for the vector paths
 *    (unaligned / H % 4 != 0 inputs take a scalar path);
 *  - features fp32, reference indices int64 (edge_index, batch), plan indices
 *    int32;
 *  - `stream` is the hipStream_t to launch on (torch's current stream);
 *    functions only enqueue work: no allocation, no synchronisation, safe to
 *    capture in a hipGraph;
 *  - E = number of edges of the input edge_index (self loops included),
 *    N = nodes, H = feature width, B = graphs, K = heads, D = head width.
 */
#ifndef CAL_HIP_H
#define CAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAL_API __attribute__((visibility("default")))

CAL_API const char* cal_last_error(void);
CAL_API int cal_version(void);

/* ---- GraphPlan ---------------------------------------------------------------
 * COO -> CSR-by-destination and CSR-by-source with edge ids, explicit self loops
 * dropped (remove_self_loops, gcn_conv.py:56); the N loops add_self_loops
 * appends (gcn_conv.py:57-63) stay implicit.  work: 4*(N+1) + 4*E int32.  status: 1
 * int32, bit0 = edge index out of range, bit1 = batch not sorted/out of range. */
CAL_API int cal_plan_build(const int64_t* edge_index, int64_t E, int64_t N,
                           int32_t* rowptr_dst, int32_t* nbr_dst, int32_t* eid_dst,
                           int32_t* rowptr_src, int32_t* nbr_src, int32_t* eid_src,
                           int32_t* row32, int32_t* col32, int32_t* work, int32_t* status,
                           void* stream);
/* node offsets of each graph from the sorted `batch` vector (train_causal.py:174 batch object) */
CAL_API int cal_graph_ptr(const int64_t* batch, int64_t N, int64_t B, int32_t* gptr,
                          int32_t* status, void* stream);

/* ---- GCNConv (gcn_conv.py:44-104) ------------------------------------------ */
/* GCNConv.norm, gcn_conv.py:44-70 */
CAL_API int cal_gcn_norm_fwd(const int32_t* rowptr_src, const int32_t* eid_src,
                             const int32_t* row32, const int32_t* col32, const float* w,
                             float loop_w, int64_t N, int64_t E, float* dis, float* norm_e,
                             void* stream);
/* propagate + message + update (+ fused ReLU of model.py:95), gcn_conv.py:92-104 */
CAL_API int cal_spmm_fwd(const int32_t* rowptr, const int32_t* nbr, const int32_t* eid,
                         const float* norm_e, const float* dis, float loop_w, const float* h,
                         const float* bias, int relu, float* out, int64_t N, int64_t H,
                         void* stream);
CAL_API int64_t cal_colsum_parts(int64_t N);
/* ReLU backward + bias gradient (gcn_conv.py:103, model.py:95) */
CAL_API int cal_relu_bwd_colsum(const float* dout, const float* y, float* dz, float* dbias,
                                float* part, int64_t N, int64_t H, void* stream);
/* autograd of gcn_conv.py:63-70,97 w.r.t. edge_weight */
CAL_API int cal_gcn_norm_bwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                             const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src,
                             const int32_t* row32, const int32_t* col32, const float* w,
                             const float* dis, float loop_w, const float* h, const float* dz,
                             float* gn_e, float* gself, float* ddeg, float* dw, int64_t N,
                             int64_t E, int64_t H, void* stream);

/* ---- dense linear layers on the matrix cores (fp32 MFMA) ----------------------
 * x @ W (gcn_conv.py:75), torch.nn.Linear (model.py:46-75) and their gradients.
 * C[M,N] = op(A) op(B) (+bias[N]) (ReLU); transA: A stored [K,M]; transB: B stored [N,K]. */
CAL_API int64_t cal_gemm_ws(int64_t M, int64_t N, int64_t K);
CAL_API int cal_gemm(int transA, int transB, const float* A, const float* B, float* C,
                     const float* bias, int relu, float* ws, int64_t M, int64_t N, int64_t K,
                     void* stream);

/* K-split (latency-oriented) variant, A not transposed, K <= 512 recommended */
CAL_API int cal_gemm_ks(int transB, const float* A, const float* B, float* C, const float* bias,
                        int relu, int64_t M, int64_t N, int64_t K, void* stream);

/* ---- causal / trivial soft masks (model.py:97-111) ------------------------- */
CAL_API int cal_edge_att_fwd(const float* x, const float* W, const float* b, const int32_t* row32,
                             const int32_t* col32, float* pq, float* att, int64_t N, int64_t E,
                             int64_t H, void* stream);
CAL_API int64_t cal_edge_att_bwd_ws(int64_t N, int64_t E, int64_t H);
CAL_API int cal_edge_att_bwd(const float* x, const float* W, const float* att, const float* datt,
                             const int32_t* rowptr_src, const int32_t* eid_src,
                             const int32_t* rowptr_dst, const int32_t* eid_dst, float* dx,
                             int accumulate, float* dW, float* db, float* ws, int64_t N, int64_t E,
                             int64_t H, void* stream);
CAL_API int cal_node_att_split_fwd(const float* x, const float* Wn, const float* bn, float* att_n,
                                   float* xc, float* xo, int64_t N, int64_t H, void* stream);
CAL_API int64_t cal_node_att_bwd_ws(int64_t N, int64_t H);
CAL_API int cal_node_att_split_bwd(const float* x, const float* Wn, const float* att_n,
                                   const float* dxc, const float* dxo, float* dx, float* dWn,
                                   float* dbn, float* ws, int64_t N, int64_t H, void* stream);

/* ---- global_add_pool (model.py:115-116, 403-404) --------------------------- */
CAL_API int cal_add_pool_fwd(const float* x, const int32_t* gptr, float* out, float* part,
                             int64_t B, int64_t H, int64_t S, void* stream);
CAL_API int cal_add_pool_bwd(const float* dout, const int64_t* batch, float* dx, int64_t N,
                             int64_t H, void* stream);

/* ---- GATConv (PyG, call sites model.py:340,390) ----------------------------- */
CAL_API int cal_gat_fwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                        const float* z, const float* att, const float* bias, int relu, float slope,
                        float p, uint64_t seed, float* out, float* adst, float* asrc, float* mx,
                        float* den, int64_t N, int64_t E, int64_t K, int64_t D, void* stream);
CAL_API int64_t cal_gat_bwd_ws(int64_t N, int64_t E, int64_t K, int64_t D);
CAL_API int cal_gat_bwd(const int32_t* rowptr_dst, const int32_t* nbr_dst, const int32_t* eid_dst,
                        const int32_t* rowptr_src, const int32_t* nbr_src, const int32_t* eid_src,
                        const float* z, const float* att, const float* adst, const float* asrc,
                        const float* mx, const float* den, const float* gout, float slope, float p,
                        uint64_t seed, float* dz, float* datt, float* ws, int64_t N, int64_t E,
                        int64_t K, int64_t D, void* stream);
CAL_API int cal_gat_dropout_mask(uint64_t seed, int64_t E, int64_t N, int64_t K, float p,
                                 float* mask, void* stream);

/* ---- on-device mini-batch assembly (PyG DataLoader / Batch collate, train_causal.py:13-15,171-174)
 * over a device-resident concatenated dataset; offsets of the selected graphs computed by the host */
CAL_API int cal_collate(const float* X, const int64_t* EI, int64_t sumE, int64_t F,
                        const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* Y,
                        const int64_t* sel, const int64_t* out_node_off, const int64_t* out_edge_off,
                        float* xo, int64_t* eio, int64_t Eout, int64_t* batcho, int64_t* yo, int64_t B,
                        void* stream);

/* host-side batch assembly of a HOST-resident dataset (DataLoader / Batch collate, train_causal.py:13-15,171-174): X [Ntot, F],
 * EI [2, Etot] (edge ids local to their graph), node_ptr / edge_ptr [G + 1], Y [G]; graphs idx [B] -> xo [N, F], eio [2, Eout]
 * (rebased by the batch's node offsets), batcho [N], yo [B] in caller-owned staging memory.  Host pointers, no device work. */
CAL_API int cal_collate_host(const float* X, const int64_t* EI, int64_t Etot, int64_t F, const int64_t* node_ptr,
                             const int64_t* edge_ptr, const int64_t* Y, const int64_t* idx, int64_t B, float* xo,
                             int64_t* eio, int64_t Eout, int64_t* batcho, int64_t* yo);
/* host-side helper of the small-graph packing (cal_engine_set_tiles): first-fit-decreasing order of the B graphs of a
 * mini-batch under the tile bounds; order_out [B] positions, first_out [T + 1] first emitted graph of every tile; returns T
 * (-1: a graph exceeds a bound).  Host pointers, no device work. */
CAL_API int64_t cal_pack_order(const int64_t* node_sizes, const int64_t* edge_sizes, int64_t B, int64_t max_nodes,
                               int64_t max_edges, int64_t max_graphs, int64_t* order_out, int64_t* first_out);

/* ---- random-intervention permutation (model.py:147-152, `random.shuffle(range(num))`) drawn on the
 * device: perm[0..B) <- uniformly random permutation keyed by (seed, *counter); the kernel advances
 * *counter (a device uint64), so a replayed hipGraph draws a fresh permutation every step.  B <= 4096 */
CAL_API int cal_randperm(int64_t* perm, int64_t B, uint64_t seed, uint64_t* counter, void* stream);

/* CausalGAT (model.py:315-409): switch the engine's backbone layers to GATConv(H, H/heads, heads, dropout=p)
 * (PyG GATConv at model.py:340,390).  att_offs[i] = float offset of convs.i.att [heads, 2H/heads] in the bound
 * parameter buffer; seeds[i] = attention-dropout seed of layer i; ctr = device uint64 the engine advances once
 * per training step and folds into the seeds so a replayed hipGraph still draws fresh masks (NULL: seeds as given).
 * Call before cal_engine_workspace_bytes / cal_engine_set_workspace. */
CAL_API int cal_engine_set_gat(void* engine, int64_t heads, float p, float slope, const int64_t* att_offs,
                               const uint64_t* seeds, void* ctr);

/* ---- native CausalGCN step engine ------------------------------------------------
 * The whole train step of train_causal.py:173-192 on model.py:85-164 (forward, 3-term loss,
 * backward, Adam) as one call enqueuing a few dozen fused kernels (24 at BASELINE config 2); see cal_amd/csrc/engine.hip for the
 * slot order of `offs` / `bn_ptrs`.  mode bits: 1 = training-mode forward, 2 = loss gradient +
 * backward into the flat gradient buffer, 4 = Adam (applied inside the step's last kernel; the step counter is advanced by
 * its first one), 8 = Adam follows separately (cal_engine_adam_ticked),
 * 16 = draw the random-intervention permutation on the device inside the step's first kernel (`perm` ignored; cal_engine_set_perm_rng).  Outputs ("logp" [3,B,C], "stats" [7] =
 * loss, c_loss, o_loss, co_loss, correct_o, correct_c, correct_co: the hit counts eval_acc_causal needs, train_causal.py:214-218)
 * live in the caller-owned workspace at
 * cal_engine_buffer_offset(name) floats from its base. */
CAL_API void* cal_engine_create(int64_t F, int64_t H, int64_t C, int64_t L);
CAL_API void cal_engine_destroy(void* engine);
/* model variants (model.py:24-31,65-69,99-107): cat = cat_or_add "cat" (fc1_bn_co / fc1_co are 2H wide), no_node_att / no_edge_att =
 * without_node_attention / without_edge_attention (constant 0.5 masks).  Call before cal_engine_bind. */
CAL_API int cal_engine_set_options(void* engine, int cat, int no_node_att, int no_edge_att);
/* CausalGIN (model.py:166-264): GINConv(Sequential(Linear, BatchNorm1d, ReLU, Linear, ReLU)) backbone layers (model.py:188-194); per
 * layer the `offs` slots are {nn.1.weight, nn.1.bias, nn.0.weight, nn.0.bias, nn.3.weight, nn.3.bias}, BatchNorm i of `bn_ptrs` is
 * convs.(i-1).nn.1; the workspace buffer "ones" ([N] floats) must hold 1.0.  Call before cal_engine_bind. */
CAL_API int cal_engine_set_gin(void* engine, int on);
CAL_API int64_t cal_engine_num_param_slots(void* engine);
CAL_API int64_t cal_engine_num_bn(void* engine);
/* P / G / M1 / M2: flat parameter, gradient and Adam-moment buffers of `nparam` floats; offs[slot] = offset of a parameter in
 * them.  Keep every offset a multiple of 4 floats (16 bytes; cal_amd.trainer.flat_offsets pads with zeros, which Adam leaves
 * alone): batches of >= 16 384 nodes read the weight matrices with 16-byte loads and cal_engine_step fails with "operands of a
 * statistics GEMM must be 16-byte aligned" otherwise. */
CAL_API int cal_engine_bind(void* engine, float* P, float* G, float* M1, float* M2, float* step,
                            float* lr, int64_t nparam, const int64_t* offs, const int64_t* bn_ptrs,
                            float beta1, float beta2, float eps, float weight_decay);
CAL_API int64_t cal_engine_workspace_bytes(void* engine, int64_t N, int64_t E, int64_t B);
CAL_API int cal_engine_set_workspace(void* engine, void* ws, int64_t bytes, int64_t capN,
                                     int64_t capE, int64_t capB);
CAL_API int64_t cal_engine_buffer_offset(void* engine, const char* name);
CAL_API int cal_engine_step(void* engine, const float* x0, const int64_t* edge_index,
                            const int64_t* batch, const int64_t* y, const int64_t* perm, int64_t N,
                            int64_t E, int64_t B, float wc, float wo, float wco, int mode,
                            void* stream);
CAL_API int cal_engine_adam(void* engine, void* stream);
/* small-graph packing: the per-graph kernels run one workgroup per TILE of consecutive graphs (<= 64 nodes, <= 1024 edges,
 * <= 8 graphs); the tiles' node / edge offsets and bounds go through cal_engine_set_graph_ptrs / _set_graph_bounds, and
 * tile_gptr [ntiles + 1] (device int64) names the first graph of every tile.  Only global_add_pool (model.py:115-116) and its
 * backward distinguish the graphs of a tile.  ntiles = 0: one graph per workgroup. */
CAL_API int cal_engine_set_tiles(void* engine, const int64_t* tile_gptr, int64_t ntiles);
/* torch.optim.Adam's betas / eps / weight_decay (train_causal.py:21,76) after the bind, e.g. when an optimizer object that
 * owns them is attached to an engine-backed model (cal_amd/optim.py); the learning rate is the bound device float `lr` */
CAL_API int cal_engine_set_adam(void* engine, float beta1, float beta2, float eps, float weight_decay);
/* model.py:147-152 on the device, inside the step (mode bit 16): permutation keyed by (seed, *counter), as cal_randperm draws it;
 * the device counter advances once per drawing step, so a replayed hipGraph draws a fresh permutation.  B <= 1024. */
CAL_API int cal_engine_set_perm_rng(void* engine, uint64_t seed, uint64_t* counter);
/* data parallel (train_causal.py:187-192 per replica): cal_engine_step(mode = 1|2|8) runs forward + backward
 * and advances the Adam step counter, the caller all-reduces (sum) the bound gradient buffer, cal_engine_adam_ticked applies
 * the update in ONE launch with the gradient multiplied by cal_engine_set_grad_scale's factor (1 / world_size = mean) */
CAL_API int cal_engine_adam_ticked(void* engine, void* stream);
CAL_API int cal_engine_set_grad_scale(void* engine, float scale);
/* the same exchange WITHOUT a collective library (the ~0.5 MB bucket is latency-bound): every rank shares one
 * region (cal_engine_p2p_region_bytes, from cal_p2p_alloc: FINE-GRAINED device memory, zero-initialised -- coarse-grained
 * hipMalloc memory is not visible across devices inside a running kernel) with the others through IPC / xGMI peer mappings
 * (cal_p2p_export -> 64-byte handle -> cal_p2p_open in the peer process); cal_engine_p2p_adam is ONE launch that publishes the
 * bucket, waits for every rank's flag, sums in rank order and applies Adam.  peer_bases[r] / peer_devices[r]: rank r's region as
 * mapped in this process and the device that owns it.  If a peer's flag does not arrive within the timeout
 * (cal_engine_p2p_set_timeout, polls) NO parameter is updated in that or any later launch, status bit 64 is set and
 * cal_engine_p2p_status (a host-mapped word, no synchronisation) returns 64. */
CAL_API int cal_p2p_alloc(int64_t bytes, void** out);
CAL_API int cal_p2p_free(void* region);
CAL_API int cal_p2p_export(void* region, void* handle64);
CAL_API int cal_p2p_open(const void* handle64, void** out);
CAL_API int cal_p2p_close(void* mapped);
CAL_API int64_t cal_engine_p2p_region_bytes(void* engine);
CAL_API int cal_engine_p2p_bind(void* engine, void* const* peer_bases, const int64_t* peer_devices, int64_t world, int64_t rank);
CAL_API int cal_engine_p2p_set_timeout(void* engine, int64_t max_polls);
CAL_API int64_t cal_engine_p2p_status(void* engine);
/* status words (latest step | sticky) as of the latest completed training step: host-mapped mirror, no synchronisation
 * (train_causal.py:171-192 has no counterpart: the reference validates nothing; here a stale batch attribute must not train on
 * garbage silently) */
CAL_API int64_t cal_engine_peek_status(void* engine);
CAL_API int cal_engine_p2p_adam(void* engine, void* stream);
/* the 3-term loss of train_causal.py:176-183 on log-probabilities logp [3,B,C] (heads c, o, co) and labels y [B]:
 * out [4] = {loss, c_loss, o_loss, co_loss}; dlogp [3,B,C] (or null) = d loss / d logp, the input of cal_engine_backward_from.
 * flag (or null): bit 1 set when a label is outside [0, C).  One launch. */
CAL_API int cal_causal_loss(const float* logp, const int64_t* y, int64_t B, int64_t C, float wc, float wo, float wco, float* out,
                            float* dlogp, int32_t* flag, void* stream);
/* backward from an external d loss / d log-probs [3,B,C] of the last training-mode forward (autograd surface) */
CAL_API int cal_engine_backward_from(void* engine, const float* x0, const int64_t* batch,
                                     const float* dlogp, int64_t N, int64_t E, int64_t B, void* stream);
/* per-graph bounds of the coming batches (largest node count / edge count of a single graph; 0 = unknown):
 * when they fit, the step runs its per-graph fused convolution kernels (GEMM + aggregation + add-pool in
 * LDS); a violated bound sets bit 3 of the engine's status word */
CAL_API int cal_engine_set_graph_bounds(void* engine, int64_t max_nodes, int64_t max_edges);
/* layout of the coming batch as the collate knows it: node / edge ranges per graph ([B+1] int64 device arrays, graph
 * b owns the contiguous edge_index columns [edge_ptr[b], edge_ptr[b+1])) and no self loops; null = unknown.  With the
 * per-graph bounds this selects the one-kernel per-graph CSR build; violations are flagged in the status word */
CAL_API int cal_engine_set_graph_ptrs(void* engine, const int64_t* node_ptr, const int64_t* edge_ptr);
/* profiling aid: make cal_engine_step return after its k-th launch site (0 = run everything) */
/* Deterministic mode of one engine: every BatchNorm sum as fixed-order partial rows (no fp64 atomics): bit-reproducible steps.
 * No reference counterpart (torch.use_deterministic_algorithms is the closest notion); SURVEY.md section 7 "determinism". */
CAL_API int cal_engine_set_deterministic(void* engine, int on);
CAL_API int cal_engine_debug_stop(int k);
/* name of launch site k (1-based) of the latest untruncated cal_engine_step; "" past the end */
CAL_API const char* cal_engine_stage_name(int k);
/* live HIP-event timing of the node-level GEMMs (class 0, work = flops) and aggregations (class 1,
 * work = algorithmic bytes) inside the step, for bench.py's roofline block; `out` is a HOST array */
CAL_API int cal_engine_profile(int on);
CAL_API int64_t cal_engine_profile_read(double* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* CAL_HIP_H */
