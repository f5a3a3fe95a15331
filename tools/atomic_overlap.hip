// Do scattered global atomics cost time of their own next to an HBM stream?  (the delta
// atomics of k_fused_batch.)  A 17 GB copy, N scattered atomicAdd into a 1 MB table (dense: 64 active lanes per
// instruction; sparse: 2), each alone and both at once on two streams.
// Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_overlap.hip -o build/atomic_overlap
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}

__device__ __forceinline__ uint32_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z >> 32);
}

// per_lane atomics per active lane; `active` lanes of every wave take part
__global__ void k_atom(uint32_t *tab, uint32_t mask, uint32_t per_lane, uint32_t active) {
    const uint32_t lane = threadIdx.x & 63;
    if (lane >= active) return;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) atomicAdd(&tab[mix(g * 0x9E3779B97F4A7C15ull + k) & mask], 1u);
}

// the same with an atomic of narrower scope (is it executed in the XCD's L2 instead of at the memory side?), and
// with one table per XCD (XCC_ID), which is what such atomics would need to stay exact
template <int SCOPE, bool PER_XCD>
__global__ void k_atom_scope(uint32_t *tab, uint32_t mask, uint32_t per_lane, uint32_t active) {
    const uint32_t lane = threadIdx.x & 63;
    if (lane >= active) return;
    uint32_t xcc = 0;
    if (PER_XCD) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
    }
    uint32_t *t = tab + (size_t)xcc * (mask + 1u);
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k)
        __hip_atomic_fetch_add(&t[mix(g * 0x9E3779B97F4A7C15ull + k) & mask], 1u, __ATOMIC_RELAXED, SCOPE);
}

__global__ void k_sum(const uint32_t *tab, uint64_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += tab[i];
    atomicAdd(out, s);
}

template <int SCOPE, bool PER_XCD>
int run_scope(const char *name, uint32_t *tab8, uint32_t mask, uint64_t n_atom, unsigned long long *d_sum) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const uint32_t waves = 6144, per = (uint32_t)(n_atom / ((uint64_t)waves * 2));
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipMemset(tab8, 0, (size_t)8 * (mask + 1) * 4));
        CHK(hipMemset(d_sum, 0, 8));
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_atom_scope<SCOPE, PER_XCD>), dim3(waves / 4), dim3(256), 0, 0, tab8, mask, per, 2u);
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float t;
        CHK(hipEventElapsedTime(&t, e0, e1));
        hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, 0, tab8, (uint64_t)8 * (mask + 1), d_sum);
        unsigned long long h = 0;
        CHK(hipMemcpy(&h, d_sum, 8, hipMemcpyDeviceToHost));
        const unsigned long long want = (unsigned long long)waves * 2 * per;
        printf("%-44s %.3f ms  %.1f G/s  sum %llu of %llu %s\n", name, t, want / t / 1e6, h, want, h == want ? "exact" : "LOST UPDATES");
    }
    return 0;
}

int main() {
    const uint64_t n_vec = (8600ull << 20) / 16;        // 8.6 GB each way
    uint4 *a, *b;
    uint32_t *tab;
    CHK(hipMalloc(&a, n_vec * 16));
    CHK(hipMalloc(&b, n_vec * 16));
    CHK(hipMalloc(&tab, 1 << 20));
    CHK(hipMemset(a, 1, n_vec * 16));
    CHK(hipMemset(tab, 0, 1 << 20));
    hipStream_t s1, s2;
    CHK(hipStreamCreate(&s1));
    CHK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, f0, f1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&f0)); CHK(hipEventCreate(&f1));
    const uint32_t mask = (1u << 18) - 1;               // 262,144 cells = 1 MB
    const uint64_t n_atom = 67ull << 20;
    for (int rep = 0; rep < 3; ++rep) {
        float tc, ta_d, ta_s, tb;
        CHK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s1, a, b, n_vec);
        CHK(hipEventRecord(e1, s1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&tc, e0, e1));
        // dense: 6144 waves x 64 lanes x per_lane
        const uint32_t waves = 6144;
        uint32_t per_dense = (uint32_t)(n_atom / ((uint64_t)waves * 64));
        CHK(hipEventRecord(e0, s2));
        hipLaunchKernelGGL(k_atom, dim3(waves / 4), dim3(256), 0, s2, tab, mask, per_dense, 64u);
        CHK(hipEventRecord(e1, s2));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ta_d, e0, e1));
        uint32_t per_sparse = (uint32_t)(n_atom / ((uint64_t)waves * 2));
        CHK(hipEventRecord(e0, s2));
        hipLaunchKernelGGL(k_atom, dim3(waves / 4), dim3(256), 0, s2, tab, mask, per_sparse, 2u);
        CHK(hipEventRecord(e1, s2));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ta_s, e0, e1));
        // both at once (sparse atomics, fewer workgroups so that the copy keeps its CUs)
        CHK(hipEventRecord(e0, s1));
        CHK(hipEventRecord(f0, s2));
        hipLaunchKernelGGL(k_atom, dim3(waves / 4), dim3(256), 0, s2, tab, mask, per_sparse, 2u);
        hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s1, a, b, n_vec);
        CHK(hipEventRecord(e1, s1));
        CHK(hipEventRecord(f1, s2));
        CHK(hipDeviceSynchronize());
        float tcopy2, tatom2;
        CHK(hipEventElapsedTime(&tcopy2, e0, e1));
        CHK(hipEventElapsedTime(&tatom2, f0, f1));
        tb = tcopy2 > tatom2 ? tcopy2 : tatom2;
        printf("copy 17.2 GB alone %.3f ms | %llu atomics dense alone %.3f ms (%.1f G/s) | sparse alone %.3f ms (%.1f G/s) | "
               "copy + sparse atomics together: copy %.3f, atomics %.3f, both done after %.3f ms\n",
               tc, (unsigned long long)n_atom, ta_d, n_atom / ta_d / 1e6, ta_s, n_atom / ta_s / 1e6, tcopy2, tatom2, tb);
    }
    uint32_t *tab8;
    unsigned long long *d_sum;
    CHK(hipMalloc(&tab8, (size_t)8 << 20));
    CHK(hipMalloc(&d_sum, 8));
    run_scope<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, one table", tab8, mask, n_atom, d_sum);
    run_scope<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope, one table", tab8, mask, n_atom, d_sum);
    run_scope<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, one table per XCD", tab8, mask, n_atom, d_sum);
    run_scope<__HIP_MEMORY_SCOPE_WAVEFRONT, true>("wavefront scope, one table per XCD", tab8, mask, n_atom, d_sum);
    run_scope<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, one table per XCD", tab8, mask, n_atom, d_sum);
    return 0;
}
