// The whole readout of a training step -- the four kernels of engine_readout.hpp (model.py:125-164, the three-term
// loss of train_causal.py:176-183 and their backward) -- as ONE launch, for B <= 128 graphs and H <= 128.
//
// The four kernels are 24 workgroups each and exchange [B,H]-sized tiles through HBM; what they cost is three kernel
// boundaries and, in every one of them, a first round of loads from buffers another XCD just wrote (45 us of a 287 us
// step for 25 MFLOP).  Here the 24 workgroups stay resident and meet at two barriers per head instead:
//   grid (3 heads, H/16 chunks), 256 threads, one workgroup per CU (133 KB of LDS) -- all co-resident by
//   construction (24 <= 256 CUs and nothing else runs on the stream), which is what makes a spin barrier legal.
//   A  (k_ro_fwd_a)  x_co = xc[perm] + xo, BN1 of all columns, y1[:, chunk] = relu(BN1(x) W1[chunk]^T + b1), BN2 of the
//                    chunk (column-local), and the chunk's PARTIAL logits zp = BN2(y1)[:, chunk] W2[:, chunk]^T
//   -- barrier 1 of the head: the H/16 partial logit tiles ([B,C] each) are the only exchange --
//   B  (k_ro_fwd_b + k_ro_bwd_a)  every workgroup sums the partials in chunk order (all of them get the same bits),
//                    log_softmax / loss / dz for all B graphs (redundantly: B*C values), then fc2 + BN2 + ReLU backward
//                    of its chunk: dy1[:, chunk], d W2[:, chunk], d b1, BN2's d gamma / d beta sums
//   -- barrier 2 of the head: dy1 chunks --
//   C  (k_ro_bwd_b)  over INPUT columns: d(BN1 out)[:, chunk] = dy1 W1[:, chunk], BN1 backward -> dxin chunk, d W1[:, chunk]
// Barriers: one counter per (head, barrier) in the step's zeroed fp64 arena; arrive = workgroup barrier + agent-scope
// release increment by lane 0, wait = acquire spin by lane 0 + workgroup barrier + agent-scope acquire fence (the
// partner may sit on another XCD, whose L2 is not coherent with ours: the release writes the stores back, the acquire
// drops our stale lines).
#pragma once
#include "engine_readout.hpp"

namespace cal {

constexpr int RS_B = 128;                 // graphs
constexpr int RS_K = 128;                 // hidden width
constexpr int RS_LD = RS_K + 4;           // row stride of the [B][K] tile and of the weight chunks

struct RoStepArgs {
    RoArgs a;
    float* zpart;                         // [3][H/16][B*C] partial logits
    int* sync;                            // [3][2] barrier counters (zero at the start of the step); row-blocked: RBK_SYNC_INTS of them
    int* status;                          // the engine's status word (bit 256: a barrier timed out)
    // row-blocked variant (128 < B <= 512, k_ro_step<true>): grid (3, H/16, nrb); workgroup (hd, ch, rb) owns rows rb*128 ..
    double* xch;                          // [3][H/16][RBK_XCH] exchange of the row blocks' partial sums
    float* gw1_slab; float* gw2_slab;     // [3][nrb][H*H], [3][nrb][C*H]: per-row-block weight gradients (k_finish sums them)
    int nrb;
};
// Exchange layout per (head, chunk): four sites (BN1 statistics of all K columns | BN2 statistics | BN2-backward sums + the
// head's loss / hit partials | BN1-backward sums + d b1 + d b2), each [RBK_MAXRB][values per row block]
constexpr int RBK_MAXRB = 4;
constexpr int RBK_N0 = 2 * RS_K, RBK_N1 = 2 * RO_CW, RBK_N2 = 2 * (RO_CW + 1), RBK_N3 = 4 * 64;
constexpr int RBK_O1 = RBK_MAXRB * RBK_N0, RBK_O2 = RBK_O1 + RBK_MAXRB * RBK_N1, RBK_O3 = RBK_O2 + RBK_MAXRB * RBK_N2;
constexpr int RBK_XCH = RBK_O3 + RBK_MAXRB * RBK_N3;
constexpr int RBK_SYNC_INTS = 3 * RBK_MAXRB * 2 + 3 * (RS_K / RO_CW) * 4;      // chunk barriers [3][4][2], then group barriers [3][H/16][4]

// Fences by ONE lane, around the workgroup barriers: the L2 write-back of a release and the invalidate of an acquire are
// cache-wide operations, so one wave's covers the workgroup once __syncthreads has drained every wave's stores
// (s_waitcnt vmcnt(0) precedes the s_barrier) -- issued by all waves they queue up behind each other
// (scripts/micro/gridbar.hip: 65 vs 20 us per barrier at 256 workgroups x 8 waves); the poll is a relaxed load.
__device__ __forceinline__ void ro_step_arrive(int* ctr) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// The poll gives up after ~2^21 rounds (a second or so) and flags status bit 256 instead of hanging the stream: the 24
// workgroups are co-resident on an idle GPU (the engine checks the CU count), but a co-tenant holding CUs could keep a
// partner from being dispatched; the step's results are garbage then and check_status says so.
__device__ __forceinline__ void ro_step_wait(int* ctr, int n, int* status) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
            if (++spins > (1 << 21)) { if (status) atomicOr(status, 256); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ __forceinline__ void ro_step_barrier(int* ctr, int n, int* status) {
    ro_step_arrive(ctr);
    ro_step_wait(ctr, n, status);
}
// 16 x 16 output tiles (wave w owns tiles w, w + 4, ..) of a product whose operands are both ROW-MAJOR in k in LDS:
// A[row][k] = X[min(row, nrow - 1) * ld + k] (optionally x * sc[k] + sh[k]), B[k][col] = Wr[col * ld + k].  Lane
// (lr = l & 15, lk = l >> 4) takes the four consecutive k  k0 + 4 lk .. + 3  of every sixteen with one 16 B read per
// operand and feeds element j to MFMA step j -- the four lane groups of a step then hold four distinct k, every k once per
// sixteen: a valid reduction order.  One read per four MFMAs and operand instead of one (three with the BatchNorm) per MFMA.
template <int NT, bool BN>
__device__ __forceinline__ void ro_mfma_rowk_nt(int kred, const float* X, int ld, int nrow, const float* Wr, const float* sc,
                                                const float* sh, ro_f32x4 (&acc)[4]) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
    const float* ar[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ar[t] = X + min((w + 4 * t) * 16 + lr, nrow - 1) * ld + 4 * lk;
    const float* br = Wr + lr * ld + 4 * lk;
    // operands of the next sixteen k are requested before the current ones are multiplied (one LDS latency per product)
    float4 a4[NT], b4, s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int k0, float4 (&a)[NT], float4& b, float4& s, float4& h) {
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const float4*>(ar[t] + k0);
        b = *reinterpret_cast<const float4*>(br + k0);
        if (BN) { s = *reinterpret_cast<const float4*>(sc + k0 + 4 * lk); h = *reinterpret_cast<const float4*>(sh + k0 + 4 * lk); }
    };
    fetch(0, a4, b4, s4, h4);
    for (int k0 = 0; k0 < kred; k0 += 16) {
        float4 an[NT], bn, sn = s4, hn = h4;
        fetch(min(k0 + 16, kred - 16), an, bn, sn, hn);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float aa[4] = {a4[t].x, a4[t].y, a4[t].z, a4[t].w};
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(BN ? fmaf(aa[j], ss[j], hh[j]) : aa[j], bb[j], acc[t], 0, 0, 0);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) a4[t] = an[t];
        b4 = bn; s4 = sn; h4 = hn;
    }
}
template <bool BN>
__device__ __forceinline__ void ro_mfma_rowk(int ntiles, int kred, const float* X, int ld, int nrow, const float* Wr, const float* sc,
                                             const float* sh, ro_f32x4 (&acc)[4]) {
    switch ((ntiles + 3) / 4) {
        case 1: ro_mfma_rowk_nt<1, BN>(kred, X, ld, nrow, Wr, sc, sh, acc); break;
        case 2: ro_mfma_rowk_nt<2, BN>(kred, X, ld, nrow, Wr, sc, sh, acc); break;
        case 3: ro_mfma_rowk_nt<3, BN>(kred, X, ld, nrow, Wr, sc, sh, acc); break;
        default: ro_mfma_rowk_nt<4, BN>(kred, X, ld, nrow, Wr, sc, sh, acc); break;
    }
}

// Row-blocked variant: the nrb workgroups (hd, ch, *) hold partial sums over their 128-row blocks and every one of them needs
// the totals.  Thread t < n writes its NV partials to xs[rb][v * n + t]; group barrier (release / acquire as the chunk
// barriers); every workgroup adds the partials in block order, so all of them get the same bits.  Called by the whole workgroup.
template <int NV>
__device__ __forceinline__ void rbk_total(double (&v)[NV], int t, int n, double* xs, int nrb, int rb, int* ctr, int* status) {
    if (t < n) {
#pragma unroll
        for (int q = 0; q < NV; ++q) xs[(size_t)rb * NV * n + q * n + t] = v[q];
    }
    ro_step_barrier(ctr, nrb, status);
    if (t < n) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double tot = 0.0;
            for (int r = 0; r < nrb; ++r) tot += xs[(size_t)r * NV * n + q * n + t];
            v[q] = tot;
        }
    }
}

// RBK = false: B <= 128, grid (3, H/16).  RBK = true: 128 < B <= 512 in row blocks of 128, grid (3, H/16, cdiv(B, 128)): the
// same phases per row block, the five sums over ALL rows (the two BatchNorms' statistics, their backward sums, the loss)
// exchanged between the row blocks of a (head, chunk) at four more barriers, the weight gradients as one slab per row block.
template <bool RBK>
__global__ void __launch_bounds__(256) k_ro_step(const RoStepArgs sa) {
    warm_kernargs<(sizeof(RoStepArgs) < 1024 ? sizeof(RoStepArgs) : 1024)>();
    const RoArgs& a = sa.a;
    __shared__ __attribute__((aligned(16))) float Xs[RS_B * RS_LD];         // raw input rows; phase C: dy1 rows
    __shared__ __attribute__((aligned(16))) float Ws[RO_CW * RS_LD];        // W1[j0 + j][:]
    __shared__ __attribute__((aligned(16))) float W1t[RO_CW * RS_LD];       // W1[:, j0 + i] stored [i][j]
    __shared__ __attribute__((aligned(16))) float scratch[2 * RS_B * RO_CW];// phase A: 8 x 256 doubles; phase B: dyh | yn
    __shared__ __attribute__((aligned(16))) float xn[RS_B * RO_CW];         // input chunk [B][16] (raw -> BN1 output)
    __shared__ __attribute__((aligned(16))) float y1c[RS_B * RO_CW];        // relu(fc1) chunk [B][16] (raw)
    __shared__ __attribute__((aligned(16))) float zs[2048];                 // logits -> dz, [B][C]
    __shared__ __attribute__((aligned(16))) float W2c[64 * RO_CW];          // W2[c][j0 + j]
    __shared__ __attribute__((aligned(16))) float sc_s[RS_K], sh_s[RS_K], mean1_s[RS_K], rstd1_s[RS_K];     // (sc / sh are read 16 B at a time)
    __shared__ int perm_s[256];
    __shared__ double red[2][256];
    __shared__ float c2[6][RO_CW];             // BN2 of the chunk: mean, rstd, gamma, beta; m1, m2 of a BatchNorm backward
    __shared__ double lossp_s[2];              // RBK: this row block's loss / hit sums of the head (duty_loss workgroups)
    const int hd = blockIdx.x, ch = blockIdx.y, j0 = ch * RO_CW, nch = gridDim.y;
    const int Bt = a.B, rb = RBK ? (int)blockIdx.z : 0, r0 = rb * RS_B, nrb = RBK ? sa.nrb : 1;      // rows r0 .. r0 + B of the batch's Bt
    const int B = RBK ? min(RS_B, Bt - r0) : Bt, K = a.H, C = a.C, ld = RS_LD, K4 = K / 4, BC = B * C, BCt = Bt * C;
    int* const cbar = RBK ? sa.sync + (hd * RBK_MAXRB + rb) * 2 : sa.sync + hd * 2;                 // the head's (row block's) two chunk barriers
    int* const gbar = sa.sync + 3 * RBK_MAXRB * 2 + (hd * nch + ch) * 4;                           // RBK: the (head, chunk)'s four exchanges
    double* const xch = RBK ? sa.xch + (size_t)(hd * nch + ch) * RBK_XCH : nullptr;
    const RoHead& h = a.h[hd];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, j = l & 15, lk = l >> 4;     // j == threadIdx.x % 16
    const int rl = threadIdx.x / RO_CW;
    float* dyh = scratch;
    float* yn = scratch + RS_B * RO_CW;
    double (*red8)[256] = reinterpret_cast<double (*)[256]>(scratch);
    const int B16 = (B + 15) & ~15, ntiles = B16 / 16;
    // the per-head extras are spread over the chunks (the barriers wait for the slowest workgroup of the head)
    const bool duty_out = ch == 1 % nch, duty_loss = ch == 2 % nch, duty_db2 = ch == 3 % nch,
               duty_xco = ch == 4 % nch, duty_run1 = ch == 5 % nch;

    RO_CLK(40);
    // =============================== A: fc1 forward of the chunk ===============================================
    const bool colfix = 256 % K4 == 0;
    // a lane's <= 16 terms per column are summed in fp32 (fp64 conversions and adds issue at a fraction of the fp32 rate:
    // 192 of them per lane were most of this phase), everything across lanes in fp64
    // The fp32 part is SHIFTED by the lane's first value of the column (pivot): sum (v - p), sum (v - p)^2, un-shifted in fp64
    // when the lane hands its partial over.  Plain fp32 sums of v^2 lose (mean / sigma)^2 ulps of the variance -- nothing at
    // B = 128, but a batch of 2-3 graphs (the last batch of an epoch) has sigma << mean in most columns and its gradients
    // came out 10-100 % off (tests/tools/fuzz_engine.py against the fp64 oracle).
    float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f}, pvt[4] = {0.f, 0.f, 0.f, 0.f};
    int nst = 0;
    auto add_stats = [&](const float4 v) {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            pvt[c] = nst == 0 ? vv[c] : pvt[c];
            const float d = vv[c] - pvt[c];
            sf[c] += d; qf[c] = fmaf(d, d, qf[c]);
        }
        ++nst;
    };
    // Loads of the kernel's first round.  The co head gathers xc[perm]: its perm entry is requested FIRST (loads return in
    // order: asked for behind the tiles it arrived after all of them, and the gather was a second full round, 6.9 us to
    // 41), the weight chunks go out behind it, and the input rows and the gathered rows follow once it is here.
    int pv = hd == 2 ? (int)a.perm[r0 + min((int)threadIdx.x, B - 1)] : 0;
    float g2 = h.bn2.gamma[j0 + j], b2n = h.bn2.beta[j0 + j], rm2 = h.bn2.run_mean[j0 + j], rv2 = h.bn2.run_var[j0 + j];
    float bias1 = h.b1[j0 + j];
    float rm1 = h.bn1.run_mean[min((int)threadIdx.x, K - 1)], rv1 = h.bn1.run_var[min((int)threadIdx.x, K - 1)];
    int ylab = (int)a.y[r0 + min((int)threadIdx.x, B - 1)];
    {
        RoBatch<float4, 2> bw;
        RoBatch<float, 8> bt;
        RoBatch<float, 4> bw2;
        ro_issue(bw, RO_CW, K4, [&](int jj, int c) { return *reinterpret_cast<const float4*>(h.W1 + (size_t)(j0 + jj) * K + c * 4); });
        ro_issue(bt, K, RO_CW, [&](int jj, int ii) { return h.W1[(size_t)jj * K + j0 + ii]; });
        ro_issue(bw2, C, RO_CW, [&](int c, int jj) { return h.W2[(size_t)c * K + j0 + jj]; });
        auto commit_weights = [&]() {
            ro_commit(bw, RO_CW, K4, [&](int jj, int c, const float4 v) { *reinterpret_cast<float4*>(Ws + jj * ld + c * 4) = v; });
            ro_commit(bt, K, RO_CW, [&](int jj, int ii, float v) { W1t[ii * ld + jj] = v; });
            ro_commit(bw2, C, RO_CW, [&](int c, int jj, float v) { W2c[c * RO_CW + jj] = v; });
            asm volatile("" : "+v"(g2), "+v"(b2n), "+v"(rm2), "+v"(rv2), "+v"(bias1), "+v"(ylab), "+v"(rm1), "+v"(rv1));
        };
        RoBatch<float4, 16> bx;
        if (hd == 2) {
            asm volatile("" : "+v"(pv));
            perm_s[threadIdx.x] = pv;
            __syncthreads();
            RoBatch<float4, 16> bp;
            ro_issue(bx, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(a.pooled + (size_t)(Bt + r0 + b) * K + c * 4); });
            ro_issue(bp, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(a.pooled + (size_t)perm_s[b] * K + c * 4); });
            commit_weights();
            ro_commit2(bx, bp, B, K4, [&](int b, int c, const float4 o, const float4 p) {
                const float4 v = make_float4(p.x + o.x, p.y + o.y, p.z + o.z, p.w + o.w);
                *reinterpret_cast<float4*>(Xs + b * ld + c * 4) = v;
                if (colfix) add_stats(v);
                if (duty_xco) *reinterpret_cast<float4*>(a.xco + (size_t)(r0 + b) * K + c * 4) = v;
            });
            if (duty_xco && (int)threadIdx.x < B) a.iperm[pv] = r0 + threadIdx.x;
        } else {
            const float* src = a.pooled + (hd == 0 ? (size_t)0 : (size_t)Bt * K) + (size_t)r0 * K;
            ro_issue(bx, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(src + (size_t)b * K + c * 4); });
            commit_weights();
            ro_commit(bx, B, K4, [&](int b, int c, const float4 v) {
                *reinterpret_cast<float4*>(Xs + b * ld + c * 4) = v;
                if (colfix) add_stats(v);
            });
        }
        __syncthreads();
    }
    RO_CLK(41);
    // BN1 statistics of all K columns: lane = (4-column group, row part), then one lane per column
    {
        const int np = 256 / K4, cg = threadIdx.x % K4, part = threadIdx.x / K4;
        if (!colfix && part < np) {
#pragma unroll 4
            for (int b = part; b < B; b += np) add_stats(*reinterpret_cast<const float4*>(Xs + b * ld + 4 * cg));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double P = (double)pvt[c], S0 = (double)sf[c], n0 = (double)nst;
            red8[c][threadIdx.x] = S0 + n0 * P;
            red8[4 + c][threadIdx.x] = (double)qf[c] + 2.0 * P * S0 + n0 * P * P;
        }
        __syncthreads();
        double SQ[2] = {0.0, 0.0};
        if ((int)threadIdx.x < K) {
            const int g = threadIdx.x >> 2, c = threadIdx.x & 3;
            for (int p = 0; p < np; ++p) { SQ[0] += red8[c][p * K4 + g]; SQ[1] += red8[4 + c][p * K4 + g]; }
        }
        if (RBK) rbk_total<2>(SQ, threadIdx.x, K, xch, nrb, rb, gbar + 0, sa.status);
        if ((int)threadIdx.x < K) {
            const int k = threadIdx.x;
            const double S = SQ[0], Q = SQ[1];
            float sc, sh, mean, rstd;
            ro_bn_from_sums(h.bn1, k, S, Q, sc, sh, mean, rstd);
            sc_s[k] = sc; sh_s[k] = sh; mean1_s[k] = mean; rstd1_s[k] = rstd;
            if (duty_run1 && rb == 0) {
                const double m = S * (double)h.bn1.inv_n;
                double v = Q * (double)h.bn1.inv_n - m * m;
                if (v < 0.0) v = 0.0;
                h.bn1.run_mean[k] = 0.9f * rm1 + 0.1f * (float)m;
                h.bn1.run_var[k] = 0.9f * rv1 + 0.1f * (float)(v * (double)h.bn1.unbias);
                if (k == 0 && h.bn1.nbt) *h.bn1.nbt += 1;
            }
        }
        __syncthreads();
    }
    RO_CLK(42);
    // this workgroup's INPUT chunk for phase C (the big tile is overwritten by dy1 there); rows B .. B16 zero
    for (int idx = threadIdx.x; idx < B16 * RO_CW; idx += 256) {
        const int b = idx / RO_CW, ii = idx % RO_CW;
        xn[idx] = b < B ? Xs[b * ld + j0 + ii] : 0.f;
    }
    // y1[:, chunk] = relu(BN1(x) W1[chunk]^T + b1) on MFMA
    ro_f32x4 acc[4] = {};
    ro_mfma_rowk<true>(ntiles, K, Xs, ld, B, Ws, sc_s, sh_s, acc);
    float s1 = 0.f, s2 = 0.f, pv2 = 0.f;       // (16 terms per lane in fp32, shifted by the lane's first value: see add_stats)
    int n2 = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (w + 4 * t) * 16 + lk * 4 + r;
            const float v = b < B ? fmaxf(acc[t][r] + bias1, 0.f) : 0.f;
            if (b < B) y1c[b * RO_CW + j] = v;
            pv2 = (b < B && n2 == 0) ? v : pv2;
            const float d = b < B ? v - pv2 : 0.f;
            n2 += b < B ? 1 : 0;
            s1 += d; s2 = fmaf(d, d, s2);
        }
    RO_CLK(43);
    {
        const double P = (double)pv2, S0 = (double)s1, n0 = (double)n2;
        red[0][threadIdx.x] = S0 + n0 * P; red[1][threadIdx.x] = (double)s2 + 2.0 * P * S0 + n0 * P * P;
    }
    __syncthreads();
    double SQ2[2] = {0.0, 0.0};
    if (threadIdx.x < RO_CW)
        for (int p = 0; p < 16; ++p) { SQ2[0] += red[0][p * RO_CW + threadIdx.x]; SQ2[1] += red[1][p * RO_CW + threadIdx.x]; }
    if (RBK) rbk_total<2>(SQ2, threadIdx.x, RO_CW, xch + RBK_O1, nrb, rb, gbar + 1, sa.status);
    if (threadIdx.x < RO_CW) {                 // BN2 of the chunk (column-local: final values)
        const double S = SQ2[0], Q = SQ2[1];
        if (rb == 0) { h.st2_sum[j0 + threadIdx.x] = S; h.st2_sq[j0 + threadIdx.x] = Q; }
        const double m = S * (double)h.bn2.inv_n;
        double v = Q * (double)h.bn2.inv_n - m * m;
        if (v < 0.0) v = 0.0;
        c2[0][threadIdx.x] = (float)m; c2[1][threadIdx.x] = 1.0f / sqrtf((float)v + h.bn2.eps);
        c2[2][threadIdx.x] = g2; c2[3][threadIdx.x] = b2n;
        if (rb == 0) {
            h.bn2.run_mean[j0 + threadIdx.x] = 0.9f * rm2 + 0.1f * (float)m;
            h.bn2.run_var[j0 + threadIdx.x] = 0.9f * rv2 + 0.1f * (float)(v * (double)h.bn2.unbias);
            if (ch == 0 && threadIdx.x == 0 && h.bn2.nbt) *h.bn2.nbt += 1;
        }
    }
    __syncthreads();
    const float mean2 = c2[0][j], rstd2 = c2[1][j], gam2 = c2[2][j], bet2 = c2[3][j];
    {   // BN2 output of the chunk (fc2's input; d W2 needs it again), then the partial logits
        const float sc = gam2 * rstd2, sh = bet2 - mean2 * sc;
        for (int b = rl; b < B; b += 16) yn[b * RO_CW + j] = fmaf(y1c[b * RO_CW + j], sc, sh);
    }
    __syncthreads();
    {
        float* zp = sa.zpart + ((size_t)hd * nch + ch) * BCt + (size_t)r0 * C;
        for (int o = threadIdx.x; o < BC; o += 256) {
            const int b = o / C, c = o % C;
            const float4* yr = reinterpret_cast<const float4*>(yn + b * RO_CW);
            const float4* wr = reinterpret_cast<const float4*>(W2c + c * RO_CW);
            zp[o] = dot4(yr[3], wr[3], dot4(yr[2], wr[2], dot4(yr[1], wr[1], dot4(yr[0], wr[0], 0.f))));
        }
    }
    RO_CLK(44);
    ro_step_barrier(cbar, nch, sa.status);
    RO_CLK(45);

    // =============================== B: logits, loss, dz; fc2 + BN2 + ReLU backward of the chunk ===============
    for (int o0 = 0; o0 < BC; o0 += 512) {     // two outputs per lane and round: B*C <= 512 is one round of loads
        float pz[2][8], bz[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int o = min(o0 + v * 256 + (int)threadIdx.x, BC - 1);
            bz[v] = h.b2[o % C];
#pragma unroll
            for (int p = 0; p < 8; ++p) pz[v][p] = sa.zpart[((size_t)hd * nch + min(p, nch - 1)) * BCt + (size_t)r0 * C + o];
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            ro_pin(bz[v]);
#pragma unroll
            for (int p = 0; p < 8; ++p) ro_pin(pz[v][p]);
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int o = o0 + v * 256 + (int)threadIdx.x;
            float z = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) z += p < nch ? pz[v][p] : 0.f;
            z += bz[v];
            if (o < BC) {
                zs[o] = z;
                if (duty_out) a.zl[(size_t)hd * BCt + (size_t)r0 * C + o] = z;
            }
        }
    }
    __syncthreads();
    float lv = 0.f, cv = 0.f;
    if ((int)threadIdx.x < B) {                // log_softmax / per-graph loss / dz (in place): one lane per graph
        const int b = threadIdx.x;
        const float u = 1.0f / (float)C, invB = 1.0f / (float)Bt, logu = logf(u);
        const float wgt = hd == 0 ? a.wc : (hd == 1 ? a.wo : a.wco);
        float* zr = zs + b * C;
        const int yy = ylab;
        int arg = 0; float best = -INFINITY;
        double lrow = 0.0;
        auto finish = [&](int k, float zk, float lse) {       // class k of this graph: log-prob, loss term, dz
            const float lp = zk - lse;
            if (lp > best) { best = lp; arg = k; }
            if (hd == 0) lrow += (double)(u * (logu - lp));
            else if (k == yy) lrow = (double)(-lp);
            const float p = expf(lp);
            const float dz = wgt * invB * (hd == 0 ? (p - u) : (p - (k == yy ? 1.f : 0.f)));
            zr[k] = dz;
            if (duty_out) { a.logp[(size_t)hd * BCt + (size_t)(r0 + b) * C + k] = lp; a.dzl[(size_t)hd * BCt + (size_t)(r0 + b) * C + k] = dz; }
        };
        if (C <= 8) {                              // the row in registers: one batch of LDS reads instead of three passes
            float zc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) zc[k] = zr[min(k, C - 1)];
            float m = zc[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) m = fmaxf(m, zc[k]);
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) se += k < C ? expf(zc[k] - m) : 0.f;
            const float lse = m + logf(se);
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k < C) finish(k, zc[k], lse);
        } else {
            float m = -INFINITY;
            for (int k = 0; k < C; ++k) m = fmaxf(m, zr[k]);
            float se = 0.f;
            for (int k = 0; k < C; ++k) se += expf(zr[k] - m);
            const float lse = m + logf(se);
            for (int k = 0; k < C; ++k) finish(k, zr[k], lse);
        }
        lv = (float)lrow; cv = arg == yy ? 1.f : 0.f;
    }
    if (duty_loss) {                           // per-head loss = mean of the per-graph losses, correct_o = number of hits
        double ls = (double)lv, cs = (double)cv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ls += __shfl_xor(ls, o, 64); cs += __shfl_xor(cs, o, 64); }
        if (l == 0) { red[0][w] = ls; red[1][w] = cs; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double lsum = red[0][0] + red[0][1] + red[0][2] + red[0][3], csum = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            if (RBK) { lossp_s[0] = lsum; lossp_s[1] = csum; }       // (summed over the row blocks with the BN2-backward sums below)
            else {
                a.stats[1 + hd] = (float)(lsum / (double)Bt);
                a.stats[hd == 1 ? 4 : (hd == 0 ? 5 : 6)] = (float)csum;
            }
        }
    }
    __syncthreads();
    RO_CLK(46);
    const float* dzs = zs;
    double db1_keep = 0.0, db2_keep = 0.0;
    {
        // d(BN2 out)[b, j] = sum_c dz[b, c] W2[c, j] for this lane's 8 rows: the class loop outside, so that a step is
        // nine independent LDS reads (inside the row guard it was one read pair + wait per (row, class))
        float t1f = 0.f, t2f = 0.f;
        float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const float wc2 = W2c[c * RO_CW + j];
            float dzv[8];
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) dzv[qq] = dzs[min(rl + 16 * qq, B - 1) * C + c];
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) dq[qq] = fmaf(dzv[qq], wc2, dq[qq]);
        }
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            const int b = rl + 16 * qq;
            const float yv = y1c[min(b, B - 1) * RO_CW + j];
            if (b < B) {
                const float n = (yv - mean2) * rstd2;
                dyh[b * RO_CW + j] = dq[qq];
                t1f += dq[qq]; t2f = fmaf(dq[qq], n, t2f);
            }
        }
        double t1 = (double)t1f, t2 = (double)t2f;
        red[0][threadIdx.x] = t1; red[1][threadIdx.x] = t2;
        __syncthreads();
        if (rl == 0) for (int p = 1; p < 16; ++p) { t1 += red[0][p * RO_CW + j]; t2 += red[1][p * RO_CW + j]; }
        if (RBK) {             // threads 0-15: the chunk's sums; thread 16: the head's loss / hits (duty_loss workgroups, else zeros)
            double tt[2] = {rl == 0 ? t1 : 0.0, rl == 0 ? t2 : 0.0};
            if (threadIdx.x == RO_CW) { tt[0] = duty_loss ? lossp_s[0] : 0.0; tt[1] = duty_loss ? lossp_s[1] : 0.0; }
            rbk_total<2>(tt, threadIdx.x, RO_CW + 1, xch + RBK_O2, nrb, rb, gbar + 2, sa.status);
            t1 = tt[0]; t2 = tt[1];
            if (threadIdx.x == RO_CW && duty_loss && rb == 0) {
                a.stats[1 + hd] = (float)(tt[0] / (double)Bt);
                a.stats[hd == 1 ? 4 : (hd == 0 ? 5 : 6)] = (float)tt[1];
            }
        }
        if (rl == 0) {
            if (rb == 0) { h.d2_sum[j0 + j] = t1; h.d2_prod[j0 + j] = t2; }
            c2[4][j] = (float)(t1 * (double)h.bn2.inv_n); c2[5][j] = (float)(t2 * (double)h.bn2.inv_n);
        }
        __syncthreads();
        const float m1 = c2[4][j], m2 = c2[5][j], gs = gam2 * rstd2;
        float sbf = 0.f;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            const int b = rl + 16 * qq;
            if (b < B) {
                const float yv = y1c[b * RO_CW + j];
                const float n = (yv - mean2) * rstd2;
                const float dy = yv > 0.f ? gs * (dyh[b * RO_CW + j] - m1 - n * m2) : 0.f;     // ReLU mask
                a.dy1[((size_t)hd * Bt + r0 + b) * K + j0 + j] = dy;
                sbf += dy;
            }
        }
        double sb = (double)sbf;
        red[0][threadIdx.x] = sb;
        __syncthreads();
        if (rl == 0) {
            for (int p = 1; p < 16; ++p) sb += red[0][p * RO_CW + j];
            if (RBK) db1_keep = sb; else h.db1[j0 + j] = sb;       // (RBK: summed over the row blocks with the BN1-backward sums)
        }
    }
    // barrier 2 of the head, then phase C's dy1 tile is requested at once; d b2 and d W2 (LDS only; gradients nobody in
    // this kernel reads) run while it is in flight
    RO_CLK(47);
    ro_step_arrive(cbar + 1);
    RO_CLK(48);
    ro_step_wait(cbar + 1, nch, sa.status);
    RO_CLK(49);
    // =============================== C: fc1 + BN1 backward over the INPUT chunk =================================
    float* Ds = Xs;
    RoBatch<float4, 16> bd;
    ro_issue(bd, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(a.dy1 + ((size_t)hd * Bt + r0 + b) * K + c * 4); });
    if (duty_db2) {                            // d b2[c] = sum_b dz[b, c]
        const int np = 256 / C, c = threadIdx.x % C, part = threadIdx.x / C;
        double sdz = 0.0;
        if (part < np) for (int b = part; b < B; b += np) sdz += (double)dzs[b * C + c];
        red[0][threadIdx.x] = sdz;
        __syncthreads();
        if (part == 0) {
            for (int p2 = 1; p2 < np; ++p2) sdz += red[0][p2 * C + c];
            if (RBK) db2_keep = sdz; else h.db2[c] = sdz;
        }
        __syncthreads();
    }
    {   // d W2[c, chunk] = sum_b dz[b,c] * BN2out[b, chunk]
        const int nout = C * RO_CW;
        float* fred = dyh;                     // dead (all reads precede the barrier above)
        for (int o0 = 0; o0 < nout; o0 += 256) {
            const int no = min(256, nout - o0), np = 256 / no;
            const int o = o0 + threadIdx.x % no, part = threadIdx.x / no;
            const int c = o / RO_CW, jj = o % RO_CW;
            float accw = 0.f;
            if (part < np) {
#pragma unroll 4
                for (int b = part; b < B; b += np) accw = fmaf(dzs[b * C + c], yn[b * RO_CW + jj], accw);
            }
            __syncthreads();
            fred[threadIdx.x] = accw;
            __syncthreads();
            if (part == 0) {
                for (int p = 1; p < np; ++p) accw += fred[p * no + threadIdx.x];
                (RBK ? sa.gw2_slab + ((size_t)hd * nrb + rb) * C * K : h.gW2)[(size_t)c * K + j0 + jj] = accw;
            }
        }
    }
    ro_commit(bd, B, K4, [&](int b, int c, const float4 v) { *reinterpret_cast<float4*>(Ds + b * ld + c * 4) = v; });
    for (int idx = threadIdx.x; idx < (B16 - B) * ld; idx += 256) Ds[B * ld + idx] = 0.f;
    __syncthreads();
    RO_CLK(50);
    const int i = j, i0 = j0;
    const float mean1 = mean1_s[i0 + i], rstd1 = rstd1_s[i0 + i], gs1 = sc_s[i0 + i], sh1 = sh_s[i0 + i];
    ro_f32x4 dacc[4] = {};
    ro_mfma_rowk<false>(ntiles, K, Ds, ld, B, W1t, nullptr, nullptr, dacc);
    float u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (w + 4 * t) * 16 + lk * 4 + r;
            const float n = (xn[min(b, B16 - 1) * RO_CW + i] - mean1) * rstd1;      // (rows past B: dacc is zero there)
            const float dv = b < B ? dacc[t][r] : 0.f;
            u1 += dv; u2 = fmaf(dv, n, u2);
        }
    red[0][threadIdx.x] = (double)u1; red[1][threadIdx.x] = (double)u2;
    __syncthreads();
    {
        double uu[4] = {0.0, 0.0, 0.0, 0.0};       // BN1-backward sums; RBK: + this row block's d b1 (threads 0-15) and d b2 (threads 0 .. C-1 of the duty workgroup)
        if (threadIdx.x < RO_CW)
            for (int p = 0; p < 16; ++p) { uu[0] += red[0][p * RO_CW + threadIdx.x]; uu[1] += red[1][p * RO_CW + threadIdx.x]; }
        if (RBK) {
            uu[2] = threadIdx.x < RO_CW ? db1_keep : 0.0;
            uu[3] = (duty_db2 && (int)threadIdx.x < C) ? db2_keep : 0.0;
            rbk_total<4>(uu, threadIdx.x, 64, xch + RBK_O3, nrb, rb, gbar + 3, sa.status);
            if (rb == 0) {
                if (threadIdx.x < RO_CW) h.db1[j0 + threadIdx.x] = uu[2];
                if (duty_db2 && (int)threadIdx.x < C) h.db2[threadIdx.x] = uu[3];
            }
        }
        if (threadIdx.x < RO_CW) {
            const double S = uu[0], Q = uu[1];
            if (rb == 0) { h.d1_sum[i0 + threadIdx.x] = S; h.d1_prod[i0 + threadIdx.x] = Q; }
            c2[4][threadIdx.x] = (float)(S * (double)h.bn1.inv_n); c2[5][threadIdx.x] = (float)(Q * (double)h.bn1.inv_n);
        }
    }
    __syncthreads();
    {
        const float m1 = c2[4][i], m2 = c2[5][i];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = (w + 4 * t) * 16 + lk * 4 + r;
                if (b < B) {
                    const float xr = xn[b * RO_CW + i];
                    const float n = (xr - mean1) * rstd1;
                    a.dxin[((size_t)hd * Bt + r0 + b) * K + i0 + i] = gs1 * (dacc[t][r] - m1 - n * m2);
                    xn[b * RO_CW + i] = fmaf(xr, gs1, sh1);      // BN1 output (fc1 input) for d W1
                }
            }
    }
    __syncthreads();
    RO_CLK(51);
    // d W1[j, chunk] = sum_b dy1[b, j] * BN1out[b, chunk] on MFMA (rows = j, cols = chunk, reduction over graphs)
    ro_f32x4 wacc[4] = {};
    ro_mfma_tiles(K / 16, B16, [&](int row, int k) { return Ds[k * ld + min(row, K - 1)]; },
                  [&](int k, int col) { return xn[k * RO_CW + col]; }, wacc);
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (w + 4 * t < K / 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                (RBK ? sa.gw1_slab + ((size_t)hd * nrb + rb) * K * K : h.gW1)[(size_t)((w + 4 * t) * 16 + lk * 4 + r) * K + i0 + i] = wacc[t][r];
        }
    RO_CLK(52);
}

}  // namespace cal
