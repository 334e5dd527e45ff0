// Latency-oriented fp32 MFMA GEMM for the "activation x small weight" products of the hot path
// (C[M,N] = A[M,K] op(B), K = hidden <= 512): the shapes here have K = 128 -- four K tiles -- so a
// classic K loop is four dependent load -> LDS -> MFMA rounds per workgroup and the kernel is pure
// latency (measured 8.8 us for [7294,128]x[128,128], of which < 2 us is matrix-pipe time).
//
// Here a workgroup owns a 32x32 output tile and its four waves split K: wave w stages and
// multiplies k in [32w, 32w+32) (+128 per extra round), so for K = 128 every operand load of the
// tile is in flight at once and there is ONE round; the four partial accumulators are summed through
// LDS and the epilogue (bias, ReLU, BN statistics / BN-backward sums, 16 B coalesced stores) runs
// on the reduced tile.  4x more workgroups (916 for N = 7294) also means ~3.6 resident per CU.
// Operand transforms and epilogues are those of gemm.hip (engine.hpp).
#include "engine.hpp"

namespace cal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 32, TN = 32, TK = 32;      // per-wave K slice
constexpr int LA = 33;                         // LDS stride of transposed (k-major) slices
constexpr int LB = 36;                         // LDS stride of directly stored B slices (16 B aligned)
constexpr int KS_XMAX = 512;

// XA: 0 plain, 1 BN on A's feature (k) axis, 2 per-row scale then BN.  B_KC: B stored [N,K].
template <bool B_KC, int XA>
__global__ void __launch_bounds__(256) k_gemm_ks(const GemmArgs a) {
    // operand slices; the reduction buffer Rs[4][32*32] aliases the A slices once the MFMAs are done
    __shared__ __attribute__((aligned(16))) float smem[4 * TK * LA + 4 * TK * LB];
    static_assert(4 * TM * TN <= 4 * TK * LA, "reduction buffer must fit in the A slices");
    warm_kernargs<sizeof(GemmArgs)>();
    __shared__ float xsc[XA > 0 ? KS_XMAX : 4], xsh[XA > 0 ? KS_XMAX : 4];
    __shared__ double cred[4][2][TN];

    const GemmProb& pr = a.p[blockIdx.z];
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lk = lane >> 5;
    const bool vecA = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(pr.A) & 15) == 0);
    const bool vecB = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(pr.B) & 15) == 0);
    const bool vecC = (a.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(pr.C) & 15) == 0);

    if (XA > 0) {
        for (int t = threadIdx.x; t < K; t += 256) {
            bn_scale_shift<true>(pr.xa.bn, t, xsc[t], xsh[t]);
            if (pr.xa.bn.update && blockIdx.x == 0 && blockIdx.y == 0) bn_update_running<true>(pr.xa.bn, t);
        }
        __syncthreads();
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float* as = smem + wave * TK * LA;
    float* bs = smem + 4 * TK * LA + wave * TK * LB;
    float* Rs = smem;
    // lane -> (row/col r8 + 8q, k quad kq) for k-contiguous slices; (k r8 + 8q, col quad) for B[K,N]
    const int r8 = lane >> 3, q4 = (lane & 7) * 4;
    for (int kr = 0; kr < K; kr += 4 * TK) {
        const int k0 = kr + wave * TK;
        if (k0 < K) {                                          // wave-uniform
            // ---- A slice [32 rows][32 k], transposed into as[k][i]
            float4 ra[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = min(m0 + r8 + 8 * q, M - 1);
                const float* src = pr.A + (size_t)row * a.lda;
                const int kk = k0 + q4;
                if (vecA && kk + 3 < K) ra[q] = *reinterpret_cast<const float4*>(src + kk);
                else ra[q] = make_float4(src[min(kk, K - 1)], src[min(kk + 1, K - 1)], src[min(kk + 2, K - 1)], src[min(kk + 3, K - 1)]);
            }
            // ---- B slice
            float4 rb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (B_KC) {                                    // B stored [N,K]: like A
                    const int col = min(n0 + r8 + 8 * q, N - 1);
                    const float* src = pr.B + (size_t)col * a.ldb;
                    const int kk = k0 + q4;
                    if (vecB && kk + 3 < K) rb[q] = *reinterpret_cast<const float4*>(src + kk);
                    else rb[q] = make_float4(src[min(kk, K - 1)], src[min(kk + 1, K - 1)], src[min(kk + 2, K - 1)], src[min(kk + 3, K - 1)]);
                } else {                                       // B stored [K,N]
                    const int kk = min(k0 + r8 + 8 * q, K - 1);
                    const float* src = pr.B + (size_t)kk * a.ldb;
                    const int col = n0 + q4;
                    if (vecB && col + 3 < N) rb[q] = *reinterpret_cast<const float4*>(src + col);
                    else rb[q] = make_float4(src[min(col, N - 1)], src[min(col + 1, N - 1)], src[min(col + 2, N - 1)], src[min(col + 3, N - 1)]);
                }
            }
            // ---- transform + store A (zero outside the matrix)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = r8 + 8 * q;
                const bool rok = m0 + i < M;
                float v[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w};
                float rs = 1.f;
                if (XA == 2) rs = pr.xa.rs[(size_t)min(m0 + i, M - 1) * pr.xa.rs_stride];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = k0 + q4 + j;
                    const int kc = min(kk, K - 1);
                    if (XA > 0) v[j] = fmaf(XA == 2 ? rs * v[j] : v[j], xsc[kc], xsh[kc]);
                    as[(q4 + j) * LA + i] = (rok && kk < K) ? v[j] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (B_KC) {
                    const int jn = r8 + 8 * q;
                    const bool cok = n0 + jn < N;
                    const float v[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) bs[(q4 + j) * LB + jn] = (cok && k0 + q4 + j < K) ? v[j] : 0.f;
                } else {
                    const int kk = r8 + 8 * q;
                    const bool kok = k0 + kk < K;
                    float4 v = rb[q];
                    v.x = (kok && n0 + q4 + 0 < N) ? v.x : 0.f; v.y = (kok && n0 + q4 + 1 < N) ? v.y : 0.f;
                    v.z = (kok && n0 + q4 + 2 < N) ? v.z : 0.f; v.w = (kok && n0 + q4 + 3 < N) ? v.w : 0.f;
                    *reinterpret_cast<float4*>(bs + kk * LB + q4) = v;
                }
            }
        }
        __syncthreads();       // (only the wave's own slice is read, but this also orders the next round)
        if (k0 < K) {
            float av[TK / 2], bv[TK / 2];
#pragma unroll
            for (int i = 0; i < TK / 2; ++i) {
                av[i] = as[(2 * i + lk) * LA + li];
                bv[i] = bs[(2 * i + lk) * LB + li];
            }
#pragma unroll
            for (int i = 0; i < TK / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        }
        if (kr + 4 * TK < K) __syncthreads();
    }
    // ---- reduce the four K-slice accumulators: Rs[w][row][col]
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) Rs[wave * TM * TN + ((r & 3) + 8 * (r >> 2) + 4 * lk) * TN + li] = acc[r];
    __syncthreads();
    const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;       // 32 rows x 8 column quads
    float4 v = *reinterpret_cast<const float4*>(&Rs[row * TN + c4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(&Rs[w * TM * TN + row * TN + c4]);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int grow = m0 + row, gcol = n0 + c4;
    const bool rok = grow < M;
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (pr.bias && gcol + j < N) o[j] += pr.bias[gcol + j];
        if (a.relu) o[j] = fmaxf(o[j], 0.f);
    }
    if (pr.C && rok) {
        float* dst = pr.C + (size_t)grow * a.ldc + gcol;
        if (gcol + 3 < N && vecC) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (gcol + j < N) dst[j] = o[j];
    }
    const bool want_st = pr.st_sum != nullptr, want_dot = pr.dot_sum != nullptr;
    if (want_st || want_dot) {
        double s1[4], s2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] = 0.0; s2[j] = 0.0;
            if (rok && gcol + j < N) {
                const double val = (double)o[j];
                s1[j] = val;
                if (want_st) s2[j] = val * val;
                else {
                    float x = pr.aux[(size_t)grow * N + gcol + j];
                    if (pr.aux_rs) x *= pr.aux_rs[(size_t)grow * pr.aux_rs_stride];
                    float mean, rstd;
                    bn_mean_rstd<true>(pr.aux_bn, gcol + j, mean, rstd);
                    s2[j] = val * (double)((x - mean) * rstd);
                }
            }
            // rows live on lane bits 3..5 (8 rows per wave): butterfly over them
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { s1[j] += __shfl_xor(s1[j], off, 64); s2[j] += __shfl_xor(s2[j], off, 64); }
        }
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { cred[wave][0][lane * 4 + j] = s1[j]; cred[wave][1][lane * 4 + j] = s2[j]; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * TN) {
            const int which = threadIdx.x / TN, c = threadIdx.x % TN;
            if (n0 + c < N) {
                const double t = cred[0][which][c] + cred[1][which][c] + cred[2][which][c] + cred[3][which][c];
                if (pr.parts) pr.parts[((size_t)blockIdx.x * 2 + which) * N + n0 + c] = t;
                else atomicAdd((which ? (want_st ? pr.st_sq : pr.dot_prod) : (want_st ? pr.st_sum : pr.dot_sum)) + (size_t)(blockIdx.x % NSTRIPE) * pr.st_ss + n0 + c, t);
            }
        }
    }
}

int gemm_ks_row_tiles(int M) { return cdiv(M, TM); }

// A must be [M,K] row-major (no transA); transB selects B stored [N,K].  No split-K.
int launch_gemm_ks(bool transB, const GemmArgs& a, int nbatch, hipStream_t stream) {
    if (a.M == 0 || a.N == 0 || nbatch == 0) return 0;
    int xa = -1;
    for (int b = 0; b < nbatch; ++b) {
        const int ma = a.p[b].xa.has_bn ? (a.p[b].xa.rs ? 2 : 1) : 0;
        if (a.p[b].xb.has_bn || a.p[b].xb.rs || (a.p[b].xa.rs && !a.p[b].xa.has_bn)) { set_error("launch_gemm_ks: unsupported operand transform"); return 2; }
        if (xa >= 0 && xa != ma) { set_error("launch_gemm_ks: mixed operand transforms in one batch"); return 2; }
        xa = ma;
    }
    if (xa > 0 && a.K > KS_XMAX) { set_error("launch_gemm_ks: K > %d with a BN-transformed operand", KS_XMAX); return 2; }
    dim3 grid(cdiv(a.M, TM), cdiv(a.N, TN), nbatch);
    if (transB) {
        if (xa == 0) hipLaunchKernelGGL((k_gemm_ks<true, 0>), grid, dim3(256), 0, stream, a);
        else if (xa == 1) hipLaunchKernelGGL((k_gemm_ks<true, 1>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_gemm_ks<true, 2>), grid, dim3(256), 0, stream, a);
    } else {
        if (xa == 0) hipLaunchKernelGGL((k_gemm_ks<false, 0>), grid, dim3(256), 0, stream, a);
        else if (xa == 1) hipLaunchKernelGGL((k_gemm_ks<false, 1>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_gemm_ks<false, 2>), grid, dim3(256), 0, stream, a);
    }
    CAL_CHECK_LAUNCH("k_gemm_ks");
    return 0;
}

}  // namespace cal
