// Microbenchmark: the P2 || P3 phase of k_gconv_bwd in isolation -- 8 waves per workgroup, one workgroup per CU,
// v_mfma_f32_32x32x2_f32 fed from k-major LDS tiles with the kernel's strides -- to see what the operand path costs:
//   variant 0: waves 0-3 run P2 (2 row tiles x 1 column tile, kred 64), waves 4-7 run P3 (1 x 2, kred 64)   [the kernel]
//   variant 1: the same MFMA count with operands held in registers (no LDS reads inside the loop)
//   variant 2: variant 0 with only waves 0-3 active;  variant 3: only waves 4-7
//   variant 4: variant 0, operands read as ds_read_b128 from row-major-in-k tiles (4 k-steps per read)
// Prints us per phase (hipEvents over REP back-to-back phases inside one launch) and the MFMA-pipe utilisation.
// hipcc --offload-arch=gfx950 -O3 -w mma_lds.hip -o mma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDJ = 65, LDW = 129, LDX = 132, LDD = 68;

template <int NA, int NB, int LDA, int LDB>
__device__ __forceinline__ void mma(const float* a0, const float* a1, const float* b0, const float* b1, int kred, int lk, f32x16 (&acc)[2]) {
    float av[2][2][16], bv[2][2][16];
    auto read_ops = [&](int kb, int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kb * 32 + 2 * i + lk;
            av[s][0][i] = a0[k * LDA];
            if (NA == 2) av[s][1][i] = a1[k * LDA];
            bv[s][0][i] = b0[k * LDB];
            if (NB == 2) bv[s][1][i] = b1[k * LDB];
        }
    };
    auto mul = [&](int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][i], bv[s][0][i], acc[0], 0, 0, 0);
            if (NA == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][i], bv[s][0][i], acc[1], 0, 0, 0);
            if (NB == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][i], bv[s][1][i], acc[1], 0, 0, 0);
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
// row-major-in-k operands: A[row * LDA + k], B[col * LDB + k]; lane (li, lk) takes 4 consecutive k per read (any
// bijection of k onto (step, lk) is a valid reduction order as long as A and B use the same one)
template <int NA, int NB, int LDA, int LDB>
__device__ __forceinline__ void mma128(const float* a0, const float* a1, const float* b0, const float* b1, int kred, int lk, f32x16 (&acc)[2]) {
    for (int k0 = 0; k0 < kred; k0 += 32) {
        float4 av[2][4], bv[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 8 * i + 4 * lk;
            av[0][i] = *reinterpret_cast<const float4*>(a0 + k);
            if (NA == 2) av[1][i] = *reinterpret_cast<const float4*>(a1 + k);
            bv[0][i] = *reinterpret_cast<const float4*>(b0 + k);
            if (NB == 2) bv[1][i] = *reinterpret_cast<const float4*>(b1 + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a[4] = {av[0][i].x, av[0][i].y, av[0][i].z, av[0][i].w};
            const float a2[4] = {av[1][i].x, av[1][i].y, av[1][i].z, av[1][i].w};
            const float bb[4] = {bv[0][i].x, bv[0][i].y, bv[0][i].z, bv[0][i].w};
            const float b2[4] = {bv[1][i].x, bv[1][i].y, bv[1][i].z, bv[1][i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc[0], 0, 0, 0);
                if (NA == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], bb[j], acc[1], 0, 0, 0);
                if (NB == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b2[j], acc[1], 0, 0, 0);
            }
        }
    }
}

template <int V>
__global__ void __launch_bounds__(512) k_phase(int reps, float* sink, long long* clk) {
    __shared__ __attribute__((aligned(16))) float Zt[64 * LDJ], Wt[64 * LDW], Xs[64 * LDX], Ds[64 * LDD];
    __shared__ __attribute__((aligned(16))) float Ar[128 * LDD], Br[128 * LDD];       // row-major-in-k tiles (variant 4): [row][64 k]
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, lk = lane >> 5, w = t >> 6;
    if (V != 4) {
        for (int i = t; i < 64 * LDJ; i += 512) Zt[i] = 0.001f * (i % 97);
        for (int i = t; i < 64 * LDW; i += 512) Wt[i] = 0.002f * (i % 89);
        for (int i = t; i < 64 * LDX; i += 512) Xs[i] = 0.003f * (i % 83);
        for (int i = t; i < 64 * LDD; i += 512) Ds[i] = 0.004f * (i % 79);
    } else {
        for (int i = t; i < 128 * LDD; i += 512) { Ar[i] = 0.001f * (i % 71); Br[i] = 0.002f * (i % 67); }
    }
    __syncthreads();
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    const long long c0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        if (V == 0 || V == 2 || V == 3) {
            if (w < 4 && V != 3) mma<2, 1, LDJ, LDW>(Zt + li, Zt + 32 + li, Wt + w * 32 + li, nullptr, 64, lk, acc);
            if (w >= 4 && V != 2) mma<1, 2, LDX, LDD>(Xs + (w - 4) * 32 + li, nullptr, Ds + li, Ds + 32 + li, 64, lk, acc);
        } else if (V == 1) {
            float a = Zt[t], b = Wt[t];
#pragma unroll 8
            for (int i = 0; i < 32; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            }
        } else if (V == 4) {
            if (w < 4) mma128<2, 1, LDD, LDD>(Ar + li * LDD, Ar + (32 + li) * LDD, Br + (w * 32 + li) * LDD, nullptr, 64, lk, acc);
            else mma128<1, 2, LDD, LDD>(Ar + ((w - 4) * 32 + li) * LDD, nullptr, Br + li * LDD, Br + (32 + li) * LDD, 64, lk, acc);
        }
        __syncthreads();
    }
    const long long c1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    if (s == 12345.f) sink[0] = s;
    if (t == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
template <int V> void run(const char* name, float* sink, long long* clk, int active_waves) {
    const int reps = 200;
    k_phase<V><<<256, 512>>>(reps, sink, clk);
    hipDeviceSynchronize();
    k_phase<V><<<256, 512>>>(reps, sink, clk);
    long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double us = h * 0.01 / reps;                       // 100 MHz ticks
    const double mfma_per_simd = 64.0 * active_waves / 4.0;    // 64 MFMAs per active wave and phase
    printf("%-46s %6.2f us / phase   (%.0f MFMAs per SIMD: %.2f us at 26.9 ns -> %.0f %% of the pipe)\n", name, us, mfma_per_simd,
           mfma_per_simd * 0.0269, 100.0 * mfma_per_simd * 0.0269 / us);
}
int main() {
    float* sink; long long* clk;
    hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    run<0>("P2 (waves 0-3) || P3 (waves 4-7), ds_read_b32", sink, clk, 8);
    run<1>("same MFMA count, register operands", sink, clk, 8);
    run<2>("P2 only (waves 0-3)", sink, clk, 4);
    run<3>("P3 only (waves 4-7)", sink, clk, 4);
    run<4>("P2 || P3, ds_read_b128 (row-major-in-k tiles)", sink, clk, 8);
    return 0;
}
