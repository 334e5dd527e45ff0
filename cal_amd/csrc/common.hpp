// Shared device/host helpers for libcalhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAL_EXPORT extern "C" __attribute__((visibility("default")))

namespace cal {

void set_error(const char* fmt, ...);

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // default workgroup: 4 waves, one per SIMD

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// lanes cooperating on one feature row: smallest power of two covering H/vec, <= 64
inline int group_for(int H, int vec) {
    int need = (H + vec - 1) / vec;
    int g = 1;
    while (g < need && g < 64) g <<= 1;
    return g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Reductions over aligned groups of G lanes (G a power of two <= 64; every lane of the group gets the same bits).  Inside a DPP row
// of 16 lanes: four FUSED v_add_f32_dpp / v_max_f32_dpp steps -- quad_perm xor 1, quad_perm xor 2, row_half_mirror (l <-> 7 - l: the
// other quad of the eight), row_mirror (l <-> 15 - l: the other eight) -- ~10 cycles each; across rows: the four row totals by
// v_readlane and two or three adds.  The __shfl_xor form this replaces is one ds_bpermute_b32 per step (address arithmetic + an LDS
// crossbar pass of ~100 cycles of dependent latency, and LDS-pipe time: k_node_att_fwd issued 36 of them per row).  A DPP read
// needs two wait states behind the VALU write of its source, which the compiler cannot see into an asm for: s_nop 1 in front.
#define CAL_DPP_STEP(NAME, OP, CTRL)                                                                                      \
    __device__ __forceinline__ float NAME(float v) {                                                                      \
        float r;                                                                                                          \
        asm("s_nop 1\n\t" OP " %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));                       \
        return r;                                                                                                         \
    }
CAL_DPP_STEP(dpp_add_x1, "v_add_f32_dpp", "quad_perm:[1,0,3,2]")
CAL_DPP_STEP(dpp_add_x2, "v_add_f32_dpp", "quad_perm:[2,3,0,1]")
CAL_DPP_STEP(dpp_add_hm, "v_add_f32_dpp", "row_half_mirror")
CAL_DPP_STEP(dpp_add_rm, "v_add_f32_dpp", "row_mirror")
CAL_DPP_STEP(dpp_max_x1, "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
CAL_DPP_STEP(dpp_max_x2, "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
CAL_DPP_STEP(dpp_max_hm, "v_max_f32_dpp", "row_half_mirror")
CAL_DPP_STEP(dpp_max_rm, "v_max_f32_dpp", "row_mirror")
#undef CAL_DPP_STEP
__device__ __forceinline__ float rdlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "group of 1..64 lanes, a power of two");
    if constexpr (G >= 2) v = dpp_add_x1(v);
    if constexpr (G >= 4) v = dpp_add_x2(v);
    if constexpr (G >= 8) v = dpp_add_hm(v);
    if constexpr (G >= 16) v = dpp_add_rm(v);
    if constexpr (G == 32) {
        const float lo = rdlane_f(v, 0) + rdlane_f(v, 16), hi = rdlane_f(v, 32) + rdlane_f(v, 48);
        v = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 32u) ? hi : lo;     // (lane id)
    }
    if constexpr (G == 64) v = (rdlane_f(v, 0) + rdlane_f(v, 16)) + (rdlane_f(v, 32) + rdlane_f(v, 48));
    return v;
}

template <int G>
__device__ __forceinline__ double group_sum_d(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int G>
__device__ __forceinline__ float group_max(float v) {
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "group of 1..64 lanes, a power of two");
    if constexpr (G >= 2) v = dpp_max_x1(v);
    if constexpr (G >= 4) v = dpp_max_x2(v);
    if constexpr (G >= 8) v = dpp_max_hm(v);
    if constexpr (G >= 16) v = dpp_max_rm(v);
    if constexpr (G == 32) {
        const float lo = fmaxf(rdlane_f(v, 0), rdlane_f(v, 16)), hi = fmaxf(rdlane_f(v, 32), rdlane_f(v, 48));
        v = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 32u) ? hi : lo;     // (lane id)
    }
    if constexpr (G == 64) v = fmaxf(fmaxf(rdlane_f(v, 0), rdlane_f(v, 16)), fmaxf(rdlane_f(v, 32), rdlane_f(v, 48)));
    return v;
}

// Column-wise sum of `nparts` partial rows: out-of-line device helper used by every *_finish kernel.
// Block = 256 threads = 16 columns x 16 part-lanes; lanes of a column reduce through LDS.
// Returns the total for column c (valid on part-lane 0).
__device__ __forceinline__ float finish_colsum(const float* __restrict__ part, int nparts, int stride, int c,
                                               bool valid, float* red /* [256] */) {
    const int pl = threadIdx.x >> 4;           // part lane 0..15
    float s0 = 0.f, s1 = 0.f;
    if (valid) {
        int p = pl;
        for (; p + 16 < nparts; p += 32) {
            s0 += part[(size_t)p * stride + c];
            s1 += part[(size_t)(p + 16) * stride + c];
        }
        if (p < nparts) s0 += part[(size_t)p * stride + c];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    float t = 0.f;
    if (pl == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k * 16 + (threadIdx.x & 15)];
    }
    __syncthreads();
    return t;
}

// vector-of-VEC float load/store helpers (VEC = 1 or 4)
template <int VEC> struct Vec;
template <> struct Vec<4> {
    float4 v;
    __device__ __forceinline__ static Vec ld(const float* p) { Vec r; r.v = *reinterpret_cast<const float4*>(p); return r; }
    __device__ __forceinline__ void st(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ __forceinline__ static Vec zero() { Vec r; r.v = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
    __device__ __forceinline__ void fma(float a, const Vec& x) {
        v.x = fmaf(a, x.v.x, v.x); v.y = fmaf(a, x.v.y, v.y); v.z = fmaf(a, x.v.z, v.z); v.w = fmaf(a, x.v.w, v.w);
    }
    __device__ __forceinline__ void add(const Vec& x) { v.x += x.v.x; v.y += x.v.y; v.z += x.v.z; v.w += x.v.w; }
    __device__ __forceinline__ void scale(float a) { v.x *= a; v.y *= a; v.z *= a; v.w *= a; }
    __device__ __forceinline__ void relu() { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    __device__ __forceinline__ float dot(const Vec& x) const {
        return fmaf(v.w, x.v.w, fmaf(v.z, x.v.z, fmaf(v.y, x.v.y, v.x * x.v.x)));
    }
    __device__ __forceinline__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
    // keep a batch of loads issued together: hipcc may not sink this value's load past the pin
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
};
template <> struct Vec<1> {
    float v;
    __device__ __forceinline__ static Vec ld(const float* p) { Vec r; r.v = *p; return r; }
    __device__ __forceinline__ void st(float* p) const { *p = v; }
    __device__ __forceinline__ static Vec zero() { Vec r; r.v = 0.f; return r; }
    __device__ __forceinline__ void fma(float a, const Vec& x) { v = fmaf(a, x.v, v); }
    __device__ __forceinline__ void add(const Vec& x) { v += x.v; }
    __device__ __forceinline__ void scale(float a) { v *= a; }
    __device__ __forceinline__ void relu() { v = fmaxf(v, 0.f); }
    __device__ __forceinline__ float dot(const Vec& x) const { return v * x.v; }
    __device__ __forceinline__ float get(int) const { return v; }
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v)); }
};

// Random-intervention permutation (model.py:147-152, `random.shuffle(range(num))`) drawn on the device: graph b gets
// the key splitmix64(seed, *counter, b), the permutation is the argsort of the keys (rank sort in LDS, one
// workgroup of NT threads, B <= CAP); *counter is advanced, so a replayed hipGraph draws a fresh permutation every
// step.  Shared by cal_randperm (collate.hip, stand-alone) and the step engine's first kernel.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <int NT>
__device__ __forceinline__ void randperm_block(int64_t* __restrict__ perm, int B, unsigned long long seed,
                                               unsigned long long* __restrict__ counter, unsigned long long* key, int* idx) {
    // perm = argsort of B hashed 63-bit keys, ties (probability ~B^2 / 2^64) by index so the result is always a permutation.
    // Rank sort: element i goes to position #{j : key_j < key_i or (key_j == key_i and j < i)} -- B broadcast LDS reads and
    // compares per element and ONE barrier, against log2(B)^2 / 2 barrier-separated exchange stages of a bitonic network
    // (28 stages at B = 128, 45 at 512: the step's first kernel waited 5-19 us for this one workgroup).
    const unsigned long long cnt = *counter;
    for (int i = threadIdx.x; i < B; i += NT)
        key[i] = splitmix64(splitmix64(seed ^ (cnt * 0xD1342543DE82EF95ull)) + (unsigned long long)i) >> 1;
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += NT) {
        const unsigned long long k = key[i];
        int r = 0;
        int j = 0;
        for (; j + 8 <= B; j += 8) {
            unsigned long long kj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kj[u] = key[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += (kj[u] < k || (kj[u] == k && j + u < i)) ? 1 : 0;
        }
        for (; j < B; ++j) { const unsigned long long kk = key[j]; r += (kk < k || (kk == k && j < i)) ? 1 : 0; }
        idx[r] = i;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += NT) perm[i] = idx[i];
    if (threadIdx.x == 0) *counter = cnt + 1;
}

// The same permutation by SEVERAL workgroups (the step's first kernel: B = 512 took 30 us in one workgroup -- B^2 64-bit
// compares at 4 cycles per wave instruction).  Workgroup `part` of `nparts` hashes all B keys into LDS and ranks the elements
// [part * NT / 4, (part + 1) * NT / 4) with four lanes per element (a quarter of the keys each); positions are scattered to
// global memory, so no workgroup waits for another.  The counter is only READ: the caller advances it in a later kernel
// of the same step (k_finish), after every workgroup here has used it.
template <int NT>
__device__ __forceinline__ void randperm_slice(int64_t* __restrict__ perm, int B, unsigned long long seed,
                                               const unsigned long long* __restrict__ counter, unsigned long long* key, int part) {
    const unsigned long long cnt = *counter;
    for (int i = threadIdx.x; i < B; i += NT)
        key[i] = splitmix64(splitmix64(seed ^ (cnt * 0xD1342543DE82EF95ull)) + (unsigned long long)i) >> 1;
    __syncthreads();
    const int i = part * (NT / 4) + ((int)threadIdx.x >> 2), q = threadIdx.x & 3;
    const int ic = min(i, B - 1);
    const unsigned long long k = key[ic];
    const int per = (B + 3) >> 2, lo = q * per, hi = min(B, lo + per);
    int r = 0;
    for (int j = lo; j < hi; j += 8) {
        unsigned long long kj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kj[u] = key[min(j + u, B - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) r += (j + u < hi && (kj[u] < k || (kj[u] == k && j + u < ic))) ? 1 : 0;
    }
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    if (q == 0 && i < B) perm[r] = i;
}

// Touch every 64 B line of the kernel-argument segment in ONE round of scalar loads at the top of a kernel: hipcc sinks
// argument loads into the blocks that use them, and each first touch of another line is a scalar-cache miss (a few
// hundred ns) in its own dependent round; after this every later argument load is a hit.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned* ka = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NL = (BYTES + 63) / 64 < 32 ? (BYTES + 63) / 64 : 32;      // at most 2 KB
#pragma unroll
    for (int g = 0; g < NL; g += 16) {
        unsigned v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = ka[16 * (g + i < NL ? g + i : 0)];
        asm volatile("" :: "s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]), "s"(v[8]), "s"(v[9]),
                     "s"(v[10]), "s"(v[11]), "s"(v[12]), "s"(v[13]), "s"(v[14]), "s"(v[15]));
    }
#endif
}

}  // namespace cal

namespace cal { inline const char* g_last_launch = ""; }     // name of the latest launch site (profiling aid)

#define CAL_CHECK_LAUNCH(name)                                               \
    do {                                                                     \
        cal::g_last_launch = name;                                           \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            cal::set_error("%s: %s", name, hipGetErrorString(e_));           \
            return 1;                                                        \
        }                                                                    \
    } while (0)

#define CAL_REQUIRE(cond, msg)                                               \
    do {                                                                     \
        if (!(cond)) {                                                       \
            cal::set_error("%s: %s", __func__, msg);                         \
            return 2;                                                        \
        }                                                                    \
    } while (0)

// Dispatch on (vectorisable?, group width) -> KERNEL<VEC, G>
#define CAL_DISPATCH_VG(H, vec_ok, BODY)                                     \
    do {                                                                     \
        if (vec_ok) {                                                        \
            constexpr int VEC = 4;                                           \
            int g_ = cal::group_for((H), 4);                                 \
            if (g_ <= 8) { constexpr int G = 8; BODY; }                      \
            else if (g_ == 16) { constexpr int G = 16; BODY; }               \
            else if (g_ == 32) { constexpr int G = 32; BODY; }               \
            else { constexpr int G = 64; BODY; }                             \
        } else {                                                             \
            constexpr int VEC = 1;                                           \
            int g_ = cal::group_for((H), 1);                                 \
            if (g_ <= 8) { constexpr int G = 8; BODY; }                      \
            else if (g_ == 16) { constexpr int G = 16; BODY; }               \
            else if (g_ == 32) { constexpr int G = 32; BODY; }               \
            else { constexpr int G = 64; BODY; }                             \
        }                                                                    \
    } while (0)
