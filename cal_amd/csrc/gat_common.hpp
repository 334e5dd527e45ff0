// Device helpers shared by the GATConv kernels (gat.hip) and the per-graph fused GAT layer of the step engine
// (engine_ggat.hpp): the counter-based attention-dropout mask must be bit-identical in both.
#pragma once
#include "common.hpp"

namespace cal {

// 32-bit finaliser (two multiply-xorshift rounds).  Round 4: the 64-bit splitmix used before cost three 64-bit multiplies per
// decision -- ~12 quarter-rate v_mul_lo/hi_u32, ~250 cycles per wave -- and every row needs (slots + 1) x heads of them: the
// training-mode GAT kernels spent more VALU time hashing than on the softmax.  Keys are (slot id * K + head) < 2^32.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
// Counter-based keep decision for (edge slot id, head): reproducible in the backward.
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t id, int k, int K, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    const uint32_t key = (uint32_t)(id * K + k);
    const uint32_t r = mix32(mix32(key + (uint32_t)seed) ^ (uint32_t)(seed >> 32));
    return ((float)r * (1.0f / 4294967296.0f)) >= p ? inv_keep : 0.f;
}

// Inside a replayed hipGraph the seed argument is frozen; the engine passes the address of its per-step device
// counter so every training step still draws a fresh mask (null: the seed is used as given).
__device__ __forceinline__ uint64_t step_seed(uint64_t seed, const uint64_t* ctr) {
    return ctr ? seed + *ctr * 0x9E3779B97F4A7C15ull : seed;
}

}  // namespace cal
