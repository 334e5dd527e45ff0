// Microbenchmark (round 4, review item 4a): what does a fork / join inside a captured step cost and buy?
// Geometry of the headline step's kernels: 256 workgroups x 512 threads, one per CU, each a latency chain (dependent loads) of
// ~T us that uses little of the CU -- so two such kernels COULD share the chip.  Chains compared, as replayed hipGraphs:
//   serial     A -> B -> C -> D                       (one stream)
//   forked     A -> { B || C } -> D                   (C on a side stream: event fork after A, event join before D)
//   forked x3  the same pattern three times per graph (three fork / join pairs, as the review's three candidate sites)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/forkjoin.hip -o scripts/micro/forkjoin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LDS_KB>
__global__ void __launch_bounds__(512) k_chain(const int* __restrict__ next, int hops, int* __restrict__ sink) {
    __shared__ int pad[LDS_KB * 256];
    int p = (blockIdx.x * 97 + threadIdx.x) & 65535;
    for (int h = 0; h < hops; ++h) p = next[p];            // dependent L2-resident loads: ~0.5 us per hop under load
    pad[threadIdx.x] = p;
    __syncthreads();
    if (threadIdx.x == 0 && pad[1] == -1) sink[0] = p;
}

int main() {
    std::vector<int> h(65536);
    for (int i = 0; i < 65536; ++i) h[i] = (i * 40503 + 12345) & 65535;
    int *d_next, *d_sink;
    CK(hipMalloc(&d_next, h.size() * 4)); CK(hipMalloc(&d_sink, 64));
    CK(hipMemcpy(d_next, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t ef[3], ej[3], t0, t1;
    for (int i = 0; i < 3; ++i) { CK(hipEventCreateWithFlags(&ef[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming)); }
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int hops : {8, 16, 32}) {
        auto K = [&](hipStream_t s) { hipLaunchKernelGGL((k_chain<60>), dim3(256), dim3(512), 0, s, d_next, hops, d_sink); };
        float res[4] = {0, 0, 0, 0};
        for (int variant = 0; variant < 4; ++variant) {       // 0 one kernel, 1 serial x3 groups, 2 forked x1 + serial x2, 3 forked x3
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            if (variant == 0) { for (int r = 0; r < 12; ++r) K(s0); }
            else {
                for (int grp = 0; grp < 3; ++grp) {
                    const bool fork = variant == 3 || (variant == 2 && grp == 0);
                    K(s0);                                     // A
                    if (fork) {
                        CK(hipEventRecord(ef[grp], s0)); CK(hipStreamWaitEvent(s1, ef[grp], 0));
                        K(s0); K(s1);                          // B || C
                        CK(hipEventRecord(ej[grp], s1)); CK(hipStreamWaitEvent(s0, ej[grp], 0));
                    } else { K(s0); K(s0); }
                    K(s0);                                     // D
                }
            }
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s0));
            CK(hipStreamSynchronize(s0));
            CK(hipEventRecord(t0, s0));
            for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(t1, s0));
            CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            res[variant] = ms * 1e3f / 50;
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        printf("hops %2d: 12 kernels serial %.1f us (%.2f us per kernel) | 3 x [A B C D] serial %.1f us | one group forked %.1f us (%+.1f) | all three forked %.1f us (%+.1f; ideal %+.1f)\n",
               hops, res[0], res[0] / 12, res[1], res[2], res[2] - res[1], res[3], res[3] - res[1], -3 * res[0] / 12);
    }
    return 0;
}
