// Internal types of the native CausalGCN step engine (not part of the C ABI).
#pragma once
#include "common.hpp"

namespace cal {

// Reference to one BatchNorm1d instance whose batch statistics live in the fp64 accumulator arena
// (sum, sum of squares per column, filled by the producer kernel's epilogue / a stats kernel).
// Consumers derive scale = gamma*rstd, shift = beta - mean*scale (torch BatchNorm1d, eps 1e-5,
// biased variance for normalisation) on the fly.
struct BNRef {
    const double* sum;     // [W]
    const double* sq;      // [W]
    const float* gamma;    // [W] (null -> 1)
    const float* beta;     // [W] (null -> 0)
    double inv_n;          // 1 / rows, in DOUBLE: as a float, 1 / 3 is off by 3e-8 and var = q / n - (s / n)^2 inherits 3e-8 * mean^2 --
                           // 1e-4 of the variance of a pooled column with mean / sigma = 60 in a batch of 3 graphs (round 6: the
                           // one report of the random-shape sweep the ReLU masks did not explain, tests/tools/fuzz_engine.py)
    float eps;
    // running statistics (updated by exactly one consumer block when update != 0; momentum 0.1,
    // unbiased variance, num_batches_tracked += 1)
    float* run_mean;
    float* run_var;
    int64_t* nbt;
    float unbias;          // n / (n - 1)
    int update;
    int use_running;       // eval mode: normalise with running statistics instead of batch statistics
    int ss;                // distance in doubles of the site's NSTRIPE accumulator planes (sum / sq point at plane 0)
};

// Striped fp64 accumulators (round 5).  A BatchNorm's column sums were one partial row per producer workgroup plus a finishing
// launch (k_stats_final: 8 x 4.7 us of the 233 us headline step).  Where producers AND consumers are the per-graph kernels
// (128-512 workgroups ending together), or the GEMM epilogues / row kernels of a small node-level batch, the producers now add
// their sums atomically into ONE OF NSTRIPE ROWS (row = workgroup %
// NSTRIPE; the rows -- "planes" of the whole BatchNorm-sum region -- are `ss` doubles apart in the arena and zeroed with it), and
// the consumers add the NSTRIPE rows in one fixed order: a chain of 32-128 same-address atomics per column instead of 256-512
// (scripts/micro/atomics64.hip, profiles/r5/micro_atomics64.txt: +1.5 / +2.1 us per producer / consumer pair at eight / four
// rows against +5.0 us for partial rows + finishing kernel and +6.1 us for one row).  Producers that finish into a single row
// (k_stats_final, plain atomics, plain stores) write plane 0 and leave the others zero, so a STRIPED reader (the *_st helpers
// below) is right for either kind; the plain readers are right only for single-row sites -- the engine stripes a site only
// when every kernel that reads it is a striped reader (engine.hip: striped_co / striped_bb / striped_gat / striped_gin / striped_feat / striped_node).  ss == 0 (no planes): the same value read
// NSTRIPE times and scaled back, exactly -- no branch in a kernel prologue (see bn_raw_load).
constexpr int NSTRIPE = 4;
__device__ __forceinline__ double stripe_sum(const double* __restrict__ p, int c, int ss) {
    double v[NSTRIPE];
#pragma unroll
    for (int r = 0; r < NSTRIPE; ++r) v[r] = p[(size_t)r * ss + c];
    double t = v[0];
#pragma unroll
    for (int r = 1; r < NSTRIPE; ++r) t += v[r];
    return ss ? t : t * (1.0 / NSTRIPE);
}
// The same in three steps for kernel prologues: request (no arithmetic on the values: an add placed right behind the loads is
// an s_waitcnt in front of every LATER load of the prologue -- a whole memory round trip per kernel, 6.7 us per headline step
// when stripe_sum was used there), pin with the prologue's other loads, add afterwards.
struct StripeVal { double v[NSTRIPE]; };
__device__ __forceinline__ StripeVal stripe_load(const double* __restrict__ p, int c, int ss) {
    StripeVal s;
#pragma unroll
    for (int r = 0; r < NSTRIPE; ++r) s.v[r] = p[(size_t)r * ss + c];
    return s;
}
__device__ __forceinline__ void stripe_pin(StripeVal& s) {
#pragma unroll
    for (int r = 0; r < NSTRIPE; ++r) asm volatile("" : "+v"(s.v[r]));
}
__device__ __forceinline__ double stripe_total(const StripeVal& s, int ss) {
    double t = s.v[0];
#pragma unroll
    for (int r = 1; r < NSTRIPE; ++r) t += s.v[r];
    return ss ? t : t * (1.0 / NSTRIPE);
}
__device__ __forceinline__ int stripe_of_block() { return (int)((blockIdx.x + blockIdx.y + blockIdx.z) % NSTRIPE); }

// (ST: striped reader -- the site's NSTRIPE accumulator planes are added here: the small-batch GEMMs, k_bn_bwd<.., ST>)
template <bool ST = false>
__device__ __forceinline__ void bn_scale_shift(const BNRef& bn, int c, float& sc, float& sh) {
    float mean, var;
    if (bn.use_running) {
        mean = bn.run_mean[c];
        var = bn.run_var[c];
    } else {
        double m = (ST ? stripe_sum(bn.sum, c, bn.ss) : bn.sum[c]) * (double)bn.inv_n;
        double v = (ST ? stripe_sum(bn.sq, c, bn.ss) : bn.sq[c]) * (double)bn.inv_n - m * m;
        mean = (float)m;
        var = (float)(v > 0.0 ? v : 0.0);
    }
    float rstd = 1.0f / sqrtf(var + bn.eps);
    float g = bn.gamma ? bn.gamma[c] : 1.f;
    float b = bn.beta ? bn.beta[c] : 0.f;
    sc = g * rstd;
    sh = b - mean * sc;
}

// mean / rstd only (for the normalised value x_n = (x - mean) * rstd used by BN backward)
template <bool ST = false>
__device__ __forceinline__ void bn_mean_rstd(const BNRef& bn, int c, float& mean, float& rstd) {
    float var;
    if (bn.use_running) {
        mean = bn.run_mean[c];
        var = bn.run_var[c];
    } else {
        double m = (ST ? stripe_sum(bn.sum, c, bn.ss) : bn.sum[c]) * (double)bn.inv_n;
        double v = (ST ? stripe_sum(bn.sq, c, bn.ss) : bn.sq[c]) * (double)bn.inv_n - m * m;
        mean = (float)m;
        var = (float)(v > 0.0 ? v : 0.0);
    }
    rstd = 1.0f / sqrtf(var + bn.eps);
}

// The same for the VEC consecutive columns c .. c+VEC-1 with every load issued before the first use:
// called per column, each column's loads sit in their own conditional block and hipcc waits for one
// column before it requests the next -- VEC dependent round trips in a kernel prologue.
template <int VEC, bool ST = false>
__device__ __forceinline__ void bn_mean_rstd_v(const BNRef& bn, int c, float (&mean)[VEC], float (&rstd)[VEC]) {
    if (bn.use_running) {
        float rv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { mean[j] = bn.run_mean[c + j]; rv[j] = bn.run_var[c + j]; }
#pragma unroll
        for (int j = 0; j < VEC; ++j) rstd[j] = 1.0f / sqrtf(rv[j] + bn.eps);
    } else {
        double s[VEC], q[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s[j] = ST ? stripe_sum(bn.sum, c + j, bn.ss) : bn.sum[c + j]; q[j] = ST ? stripe_sum(bn.sq, c + j, bn.ss) : bn.sq[c + j]; }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double m = s[j] * (double)bn.inv_n, v = q[j] * (double)bn.inv_n - m * m;
            mean[j] = (float)m;
            rstd[j] = 1.0f / sqrtf((float)(v > 0.0 ? v : 0.0) + bn.eps);
        }
    }
}

template <bool ST = false>
__device__ __forceinline__ void bn_update_running(const BNRef& bn, int c) {
    double m = (ST ? stripe_sum(bn.sum, c, bn.ss) : bn.sum[c]) * (double)bn.inv_n;
    double v = (ST ? stripe_sum(bn.sq, c, bn.ss) : bn.sq[c]) * (double)bn.inv_n - m * m;
    if (v < 0.0) v = 0.0;
    bn.run_mean[c] = 0.9f * bn.run_mean[c] + 0.1f * (float)m;
    bn.run_var[c] = 0.9f * bn.run_var[c] + 0.1f * (float)(v * (double)bn.unbias);
    if (c == 0 && bn.nbt) *bn.nbt += 1;
}

// Branch-free access to a BatchNorm's constants for kernel prologues.  bn_scale_shift / bn_mean_rstd select between
// running and batch statistics (and test gamma / beta for null) with BRANCHES; hipcc then waits for every load in
// flight at each join and requests the next operand only afterwards -- a prologue with two BatchNorms became 4-10
// serial round trips behind the tile loads (1.5-2 us each when the statistics were just written by another XCD).
// bn_raw_load issues all six loads unconditionally (every pointer of a BNRef built by the engine is valid; `c` must be
// a valid column), bn_raw_pin keeps them from sinking, bn_raw_* turn the values into the constants with selects.
struct BNRaw { double s, q; float rm, rv, g, b; };
__device__ __forceinline__ BNRaw bn_raw_load(const BNRef& bn, int c) {
    BNRaw r;
    r.s = bn.sum[c]; r.q = bn.sq[c]; r.rm = bn.run_mean[c]; r.rv = bn.run_var[c]; r.g = bn.gamma[c]; r.b = bn.beta[c];
    return r;
}
struct BNRawS { StripeVal s, q; float rm, rv, g, b; };       // striped reader of a kernel prologue: load, pin, then bn_raws_sum
__device__ __forceinline__ BNRawS bn_raws_load(const BNRef& bn, int c) {
    BNRawS r;
    r.s = stripe_load(bn.sum, c, bn.ss); r.q = stripe_load(bn.sq, c, bn.ss);
    r.rm = bn.run_mean[c]; r.rv = bn.run_var[c]; r.g = bn.gamma[c]; r.b = bn.beta[c];
    return r;
}
// one register set for two BatchNorms read by disjoint lanes of a workgroup (lane-wise choice of the pointers, not a branch)
__device__ __forceinline__ BNRawS bn_raws_load2(const BNRef& a, int ca, const BNRef& b, int cb, bool use_b) {
    const double* ps = use_b ? b.sum : a.sum; const double* pq = use_b ? b.sq : a.sq;
    const float* prm = use_b ? b.run_mean : a.run_mean; const float* prv = use_b ? b.run_var : a.run_var;
    const float* pg = use_b ? b.gamma : a.gamma; const float* pb = use_b ? b.beta : a.beta;
    const int c = use_b ? cb : ca, ss = use_b ? b.ss : a.ss;
    BNRawS r;
    r.s = stripe_load(ps, c, ss); r.q = stripe_load(pq, c, ss);
    r.rm = prm[c]; r.rv = prv[c]; r.g = pg[c]; r.b = pb[c];
    return r;
}
__device__ __forceinline__ void bn_raws_pin(BNRawS& r) {
    stripe_pin(r.s); stripe_pin(r.q);
    asm volatile("" : "+v"(r.rm), "+v"(r.rv), "+v"(r.g), "+v"(r.b));
}
__device__ __forceinline__ BNRaw bn_raws_sum(const BNRef& bn, const BNRawS& r) {
    BNRaw o;
    o.s = stripe_total(r.s, bn.ss); o.q = stripe_total(r.q, bn.ss); o.rm = r.rm; o.rv = r.rv; o.g = r.g; o.b = r.b;
    return o;
}
__device__ __forceinline__ BNRaw bn_raw_load_st(const BNRef& bn, int c) {      // striped reader (outside a prologue)
    BNRaw r;
    r.s = stripe_sum(bn.sum, c, bn.ss); r.q = stripe_sum(bn.sq, c, bn.ss);
    r.rm = bn.run_mean[c]; r.rv = bn.run_var[c]; r.g = bn.gamma[c]; r.b = bn.beta[c];
    return r;
}
__device__ __forceinline__ void bn_raw_pin(BNRaw& r) {
    asm volatile("" : "+v"(r.s), "+v"(r.q), "+v"(r.rm), "+v"(r.rv), "+v"(r.g), "+v"(r.b));
}
__device__ __forceinline__ void bn_raw_mean_rstd(const BNRef& bn, const BNRaw& r, float& mean, float& rstd) {
    const double m = r.s * (double)bn.inv_n, v = r.q * (double)bn.inv_n - m * m;
    mean = bn.use_running ? r.rm : (float)m;
    const float var = bn.use_running ? r.rv : (float)(v > 0.0 ? v : 0.0);
    rstd = 1.0f / sqrtf(var + bn.eps);
}
__device__ __forceinline__ void bn_raw_scale_shift(const BNRef& bn, const BNRaw& r, float& sc, float& sh) {
    float mean, rstd;
    bn_raw_mean_rstd(bn, r, mean, rstd);
    sc = r.g * rstd;
    sh = r.b - mean * sc;
}
// running-statistics update from the raw values (stores only; call under the one-block-per-BatchNorm condition)
__device__ __forceinline__ void bn_raw_update_running(const BNRef& bn, const BNRaw& r, int c) {
    const double m = r.s * (double)bn.inv_n;
    double v = r.q * (double)bn.inv_n - m * m;
    if (v < 0.0) v = 0.0;
    bn.run_mean[c] = 0.9f * r.rm + 0.1f * (float)m;
    bn.run_var[c] = 0.9f * r.rv + 0.1f * (float)(v * (double)bn.unbias);
    if (c == 0 && bn.nbt) *bn.nbt += 1;
}

// Optional transform applied to a GEMM operand while it is staged, in STORAGE coordinates
// (storage row = node / sample index, storage column = feature index):
//     v' = (rs[row] * v) * sc[col] + sh[col]        (sc, sh from `bn`; rs may be null)
struct Xform {
    const float* rs;   // per storage row scale (node attention a_k[v]), stride rs_stride
    int rs_stride;
    int has_bn;
    BNRef bn;
};

// One problem of a (batched) GEMM launch.
struct GemmProb {
    const float* A;
    const float* B;
    float* C;            // may be null (statistics only)
    const float* bias;   // [N] or null
    Xform xa, xb;
    // epilogue statistics over the rows of C (after bias / ReLU): sum and sum of squares per column
    double* st_sum;
    double* st_sq;
    // epilogue "dot statistics": dot_sum[col] += C, dot_prod[col] += C * aux_n where
    // aux_n = ((aux_rs[row] * aux[row, col]) - mean[col]) * rstd[col]   (BN backward sums)
    const float* aux;
    const float* aux_rs;
    int aux_rs_stride;
    int has_aux;
    BNRef aux_bn;
    double* dot_sum;
    double* dot_prod;
    // when non-null the two epilogue sums are written as one partial row per row tile,
    // parts[(row_tile * 2 + {0,1}) * N + col], instead of being added atomically (k_stats_final
    // sums them): hot-address fp64 atomics cost ~0.65 ns each chip-wide on MI355X
    double* parts;
    // ... or (st_ss != 0, atomic mode) into the row tile's plane of the site's NSTRIPE accumulator planes, st_ss doubles apart
    // (stripe_sum above: every reader of the site adds the planes; k_gemm and k_gemm_ks)
    int st_ss;
};

struct GemmArgs {
    GemmProb p[3];
    int M, N, K;
    int lda, ldb, ldc;
    int relu;
    int kchunk;          // split-K slice length (multiple of BK); slices write C + z * M * ldc
    int nsplit;
};

int launch_gemm(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream);
int launch_gemm_dual(const GemmArgs& ax, int nbx, const GemmArgs& aw, int nbw, hipStream_t stream);
int splitk_for(int64_t M, int64_t N, int64_t K, int nbatch);
void gemm_set_split(GemmArgs& a, int S);
int gemm_row_tiles(int M, int N, int K, bool hasC = true);
// throughput variant (gemm_big.hip): 128x128 tiles for node-level products with >= 16k rows; 1 launched, 0 n/a, < 0 error
bool gemm_big_rows(int M, int K);
bool gemm_big_grad(int M, int N, int K);
int gemm_big_grad_splits(int K);
int launch_gemm_big(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream);
int launch_gemm_big_dual(const GemmArgs& ax, int nbx, const GemmArgs& aw, int nbw, hipStream_t stream);
// weight-resident / direct-operand variants (gemm_wres.hip): hidden width 128 / 256, >= 16k rows; same return convention
bool gemm_wres_rows(int M, int N, int K);
bool gemm_wres_grad(int M, int N, int K);
int gemm_wres_grad_splits(int K, int nbatch);
int gemm_wres_parts();
int launch_gemm_wres(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream);
// latency-oriented variant (gemm_ks.hip): 32x32 tiles, K split across the four waves
int launch_gemm_ks(bool transB, const GemmArgs& a, int nbatch, hipStream_t stream);
int gemm_ks_row_tiles(int M);


}  // namespace cal
