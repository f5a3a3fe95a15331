// Scattered global atomics on differently allocated tables and in different forms: does any
// of them get past the 24 G/s of the default path?
// Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_modes.hip -o build/atomic_modes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z >> 32);
}

template <int MODE>
__global__ void k_atom(uint32_t *tab, uint32_t mask, uint32_t per_lane, uint32_t active) {
    const uint32_t lane = threadIdx.x & 63;
    if (lane >= active) return;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t i = mix(g * 0x9E3779B97F4A7C15ull + k) & mask;
        if (MODE == 0) atomicAdd(&tab[i], 1u);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long *>(tab) + (i >> 1), 1ull << (32 * (i & 1)));   // 64-bit
        else if (MODE == 2) { uint32_t old = atomicAdd(&tab[i], 1u); asm volatile("" :: "v"(old)); }                    // returning
        else if (MODE == 3) tab[i] += 1u;                                                                                  // plain RMW (inexact): what loads + stores cost
        else if (MODE == 4) __hip_atomic_fetch_add(&tab[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == 5) atomicAdd(reinterpret_cast<float *>(tab) + i, 1.0f);
    }
}

template <int MODE>
int run(const char *what, uint32_t *tab, uint32_t mask, uint32_t active) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const uint64_t n_atom = 67ull << 20;
    const uint32_t waves = 6144, per = (uint32_t)(n_atom / ((uint64_t)waves * active));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_atom<MODE>, dim3(waves / 4), dim3(256), 0, 0, tab, mask, per, active);
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float t;
        CHK(hipEventElapsedTime(&t, e0, e1));
        best = t < best ? t : best;
    }
    printf("  %-34s lanes %2u  %.3f ms  %.1f G/s\n", what, active, best, (double)waves * active * per / best / 1e6);
    return 0;
}

int main() {
    const size_t bytes = (size_t)64 << 20;
    const uint32_t mask = (uint32_t)(bytes / 4) - 1u;
    struct { const char *name; unsigned flags; int kind; } allocs[] = {
        {"hipMalloc", 0, 0}, {"hipExtMallocWithFlags(Finegrained)", hipDeviceMallocFinegrained, 1},
        {"hipExtMallocWithFlags(Uncached)", hipDeviceMallocUncached, 1}};
    // (hipMallocManaged: 6.9 s per launch of 7e7 atomics, i.e. 0.01 G/s -- not worth a run)
    for (auto &a : allocs) {
        uint32_t *tab = nullptr;
        hipError_t e = a.kind == 0 ? hipMalloc(&tab, bytes) : hipExtMallocWithFlags((void **)&tab, bytes, a.flags);
        if (e != hipSuccess) { printf("%s: %s\n", a.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CHK(hipMemset(tab, 0, bytes));
        printf("%s, 64 MB table\n", a.name);
        for (uint32_t active : {2u, 64u}) {
            if (run<0>("atomicAdd u32 (no return)", tab, mask, active)) return 1;
            if (run<1>("atomicAdd u64", tab, mask, active)) return 1;
            if (run<2>("atomicAdd u32 returning", tab, mask, active)) return 1;
            if (run<4>("system-scope atomic", tab, mask, active)) return 1;
            if (run<5>("atomicAdd f32", tab, mask, active)) return 1;
            if (run<3>("plain load + store (inexact)", tab, mask, active)) return 1;
        }
        CHK(hipFree(tab));
    }
    return 0;
}
