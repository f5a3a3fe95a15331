// What the LDS alone allows the pair-count scan: 4.29e9 fire-and-forget LDS atomics into a 32,768-word histogram from one
// 1024-thread workgroup per CU -- the scan's shape without its loads, byte extraction, checksums and flush.
//   random   : word index from a per-lane hash (what random byte pairs give: ~3.5-way conflicts per 32-lane group)
//   spread   : bank = lane by construction, random row (no two lanes of a 32-lane group on one bank)
//   u64      : ds_add_u64 on random 8-byte words (does the 64-bit form see more banks?)
//   lean     : the scan's own instruction mix per pair on register data (no loads, no checksums, no flush)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_atomic_floor.hip -o build/lds_atomic_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 1024;
constexpr int kWords = 32768;

// one hash per EIGHT atomics: the eight addresses are idx + constant (the constant rides in the instruction's offset
// field), so the vector ALUs issue ~1 instruction per atomic and the LDS is what is being measured
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_floor(uint32_t iters, uint32_t *out) {
    __shared__ uint32_t hist[kWords];
    for (uint32_t w = threadIdx.x; w < (uint32_t)kWords; w += kThreads) hist[w] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * kThreads + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            x = mix(x + 0x9E3779B9u);
            uint32_t idx = x >> 18;                          // 14 bits; + offsets below 16,384 stays inside the histogram
            if (MODE == 1) idx = (idx & ~31u) | (lane & 31u); // bank = lane (+ the same constant for every lane)
            const uint32_t inc = 1u + ((x >> 3) & 1u) * 0xFFFFu;
            if (MODE == 2) {
                unsigned long long *h64 = reinterpret_cast<unsigned long long *>(hist);
                idx >>= 1;
#pragma unroll
                for (int k = 0; k < 8; ++k) atomicAdd(&h64[idx + k * 1021], (unsigned long long)inc);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) atomicAdd(&hist[idx + k * 2039], inc);
            }
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t w = threadIdx.x; w < (uint32_t)kWords; w += kThreads) s += (hist[w] & 0xFFFFu) + (hist[w] >> 16);
    if (s == 0xFFFFFFFFu) out[0] = s;      // (keeps the histogram alive)
}

// the scan's own inner loop on register data: per dword of "text" 4 pairs at 4.75 vector instructions each
// (v_alignbit / v_perm / v_bfe / v_mad as in kernels.hip pc_lean_pairs), no loads, no checksums
__global__ __launch_bounds__(kThreads) void k_floor_lean(uint32_t iters, uint32_t *out) {
    __shared__ uint32_t hist[kWords];
    for (uint32_t w = threadIdx.x; w < (uint32_t)kWords; w += kThreads) hist[w] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * kThreads + threadIdx.x) * 2654435761u + 12345u;
    uint32_t k0xffff;
    asm volatile("s_mov_b32 %0, 0xffff" : "=s"(k0xffff));
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t w[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { x = mix(x + 0x9E3779B9u); w[i] = x; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t z = __builtin_amdgcn_alignbit(w[i + 1], w[i], 8);
            const uint32_t y = w[i] ^ z;
            const uint32_t z7 = z & 0x7F7F7F7Fu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t word = __builtin_amdgcn_perm(z7, y, 0x0C0C0000u | ((4u + k) << 8) | (uint32_t)k);
                uint32_t top, inc;
                asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(top) : "v"(z), "n"(8 * k + 7));
                asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(top), "s"(k0xffff));
                atomicAdd(&hist[word], inc);
            }
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t w = threadIdx.x; w < (uint32_t)kWords; w += kThreads) s += (hist[w] & 0xFFFFu) + (hist[w] >> 16);
    if (s == 0xFFFFFFFFu) out[0] = s;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double total = 4294967296.0;
    const uint32_t iters = (uint32_t)(total / cus / kThreads / 32);
    uint32_t *out;
    CHK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    struct V { const char *name; void (*fn)(uint32_t, uint32_t *); } vs[] = {
        {"random ds_add_u32, one hash per 8 atomics", k_floor<0>},
        {"bank = lane ds_add_u32 (conflict-free by construction)", k_floor<1>},
        {"random ds_add_u64, one hash per 8 atomics", k_floor<2>},
        {"the scan's inner loop on register data (4.75 VALU per pair)", k_floor_lean},
    };
    printf("%d CUs, %u iterations x 32 atomics x %d threads per CU = %.3e increments\n", cus, iters, kThreads,
           (double)iters * 32 * kThreads * cus);
    for (auto &v : vs) {
        std::vector<float> ms;
        for (int rep = 0; rep < 8; ++rep) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(v.fn, dim3(cus), dim3(kThreads), 0, 0, iters, out);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float t;
            CHK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 3) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const float med = ms[ms.size() / 2];
        printf("%-58s %8.3f ms  = %.3f of the HBM peak if this were the 4 GiB scan\n", v.name, med, 4294967296.0 / (med * 1e-3) / 8e12);
    }
    return 0;
}
