// Scattered global atomicAdd throughput against the size of the table they fall into (the
// count-delta block of k_fused_batch grows from 1 MB to 131 MB during a training run).  2 active lanes per wave
// instruction (sparse, as in the fused pass) and 64.
// Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_footprint.hip -o build/atomic_footprint
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z >> 32);
}

__global__ void k_atom(uint32_t *tab, uint32_t mask, uint32_t per_lane, uint32_t active) {
    const uint32_t lane = threadIdx.x & 63;
    if (lane >= active) return;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) atomicAdd(&tab[mix(g * 0x9E3779B97F4A7C15ull + k) & mask], 1u);
}

int main() {
    uint32_t *tab;
    CHK(hipMalloc(&tab, (size_t)512 << 20));
    CHK(hipMemset(tab, 0, (size_t)512 << 20));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const uint64_t n_atom = 67ull << 20;
    const uint32_t waves = 6144;
    for (uint32_t active : {2u, 64u}) {
        for (int lg = 16; lg <= 27; ++lg) {           // 256 KB .. 512 MB
            const uint32_t mask = (1u << lg) - 1u;
            const uint32_t per = (uint32_t)(n_atom / ((uint64_t)waves * active));
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_atom, dim3(waves / 4), dim3(256), 0, 0, tab, mask, per, active);
                CHK(hipEventRecord(e1, 0));
                CHK(hipDeviceSynchronize());
                float t;
                CHK(hipEventElapsedTime(&t, e0, e1));
                best = t < best ? t : best;
            }
            printf("active lanes %2u  table %7.2f MB  %.3f ms  %.1f G atomics/s\n", active, (double)(4ull << lg) / 1048576.0, best,
                   (double)waves * active * per / best / 1e6);
        }
    }
    return 0;
}
