// Device helpers shared by the GATConv kernels (gat.hip) and the per-graph fused GAT layer of the step engine
// (engine_ggat.hpp): the counter-based attention-dropout mask must be bit-identical in both.
#pragma once
#include "common.hpp"

namespace cal {

__device__ __forceinline__ uint32_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}
// Counter-based keep decision for (edge slot id, head): reproducible in the backward.
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t id, int k, int K, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    uint32_t r = mix64(seed ^ (uint64_t)(id * K + k) * 0xD6E8FEB86659FD93ull);
    return ((float)r * (1.0f / 4294967296.0f)) >= p ? inv_keep : 0.f;
}

// Inside a replayed hipGraph the seed argument is frozen; the engine passes the address of its per-step device
// counter so every training step still draws a fresh mask (null: the seed is used as given).
__device__ __forceinline__ uint64_t step_seed(uint64_t seed, const uint64_t* ctr) {
    return ctr ? seed + *ctr * 0x9E3779B97F4A7C15ull : seed;
}

}  // namespace cal
