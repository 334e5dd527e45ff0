// Where do the four waves of a 256-thread workgroup land?  (HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13] ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned* out, int spin) {
    __shared__ float big[18000];
    big[threadIdx.x] = threadIdx.x;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (big[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int G = 512;
    unsigned* d; (void)hipMalloc(&d, G * 4 * 2 * 4);
    k<<<G, 256>>>(d, 2000);
    std::vector<unsigned> h(G * 8);
    (void)hipMemcpy(h.data(), d, G * 8 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<int, int>>> cu;   // key (xcc, se, sh, cu) -> (wg, simd)
    for (int b = 0; b < G; ++b)
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            unsigned key = (xcc << 16) | (hw & 0xff00);
            cu[key].push_back({b, (int)((hw >> 4) & 3)});
        }
    printf("distinct CUs seen: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 4) break;
        printf("cu key %06x:", kv.first);
        for (auto& p : kv.second) printf(" wg%d@simd%d", p.first, p.second);
        printf("\n");
    }
    // histogram: for each (cu, wg) how many distinct SIMDs do its 4 waves use
    int hist[5] = {0};
    for (auto& kv : cu) {
        std::map<int, unsigned> m;
        for (auto& p : kv.second) m[p.first] |= 1u << p.second;
        for (auto& q : m) hist[__builtin_popcount(q.second)]++;
    }
    printf("workgroups whose 4 waves span 1/2/3/4 SIMDs: %d %d %d %d\n", hist[1], hist[2], hist[3], hist[4]);
    return 0;
}
