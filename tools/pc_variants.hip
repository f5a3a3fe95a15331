// A/B harness for the pair-count scan (k_pair_count_u8): variants of the LDS-histogram kernel,
// timed with HIP events on SplitMix64 bytes and checked against a plain global-atomic count.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pc_variants.hip -o build/pc_variants
// Run  :  build/pc_variants [bytes]          (default 4 GiB)
// The winning variant lives in minbpe-cc_amd/csrc/kernels.hip; this file keeps the A/B table
// reproducible (profiles/r02_pair_count_ab.md).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWave = 64;
constexpr int kPcWords = 32768;
constexpr uint32_t kPcHotBits = 0xE000u;

__global__ void k_gen(uint64_t *out, uint64_t n_words, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_words; i += stride) {
        uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        out[i] = z ^ (z >> 31);
    }
}

// skewed text-like data: byte = small alphabet with a very frequent pair
__global__ void k_gen_skew(uint8_t *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (i / 2 + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        const uint32_t r = (uint32_t)(z >> 40) & 0xFF;
        // 75 % of the positions: the pair "e " ; else a letter
        out[i] = r < 192 ? ((i & 1) ? ' ' : 'e') : (uint8_t)('a' + (r & 15));
    }
}

__global__ void k_ref(const uint8_t *text, uint64_t n, uint32_t *bp) {
    __shared__ uint32_t h[1];
    (void)h;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i + 1 < n; i += stride) atomicAdd(&bp[((uint32_t)text[i] << 8) | text[i + 1]], 1u);
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ uint32_t pc_table_index(uint32_t hbin) {
    const uint32_t bin = hbin ^ (hbin >> 8);
    return ((bin & 0xFFu) << 8) | (bin >> 8);
}

template <int THREADS>
__device__ __forceinline__ void pc_sweep(uint32_t *hist, uint32_t *bp) {
#pragma unroll
    for (int k = 0; k < kPcWords / 4 / THREADS; ++k) {
        const uint32_t g = k * THREADS + threadIdx.x;
        uint4 v = reinterpret_cast<uint4 *>(hist)[g];
        const uint32_t any = (v.x | v.y | v.z | v.w) & (kPcHotBits | (kPcHotBits << 16));
        if (any) {
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t word = g * 4 + c;
                const uint32_t lo = w[c] & 0xFFFFu, hi = w[c] >> 16;
                if (lo & kPcHotBits) { atomicAdd(&bp[pc_table_index(word)], lo); w[c] &= 0xFFFF0000u; }
                if (hi & kPcHotBits) { atomicAdd(&bp[pc_table_index(word | 0x8000u)], hi); w[c] &= 0x0000FFFFu; }
            }
            reinterpret_cast<uint4 *>(hist)[g] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

template <int THREADS>
__device__ __forceinline__ void pc_flush(const uint32_t *hist, uint32_t *bp) {
    for (uint32_t o = threadIdx.x; o < 65536u; o += THREADS) {
        uint32_t bin = ((o & 0xFFu) << 8) | (o >> 8);
        bin ^= bin >> 8;
        const uint32_t c = (hist[bin & 0x7FFFu] >> ((bin >> 15) * 16)) & 0xFFFFu;
        if (c) atomicAdd(&bp[o], c);
    }
}

// sum of all decoded 16-bit counters of the workgroup's histogram (exact overflow detector: a
// wrapped counter changes the decoded sum by -65535 or -65536, never by 0)
template <int THREADS>
__device__ __forceinline__ unsigned long long pc_checksum(const uint32_t *hist, unsigned long long *red) {
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < kPcWords / 4 / THREADS; ++k) {
        const uint4 v = reinterpret_cast<const uint4 *>(hist)[k * THREADS + threadIdx.x];
        s += (v.x & 0xFFFFu) + (v.x >> 16) + (v.y & 0xFFFFu) + (v.y >> 16) + (v.z & 0xFFFFu) + (v.z >> 16) +
             (v.w & 0xFFFFu) + (v.w >> 16);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += ((unsigned long long)__shfl_xor((uint32_t)(s >> 32), d, kWave) << 32) | __shfl_xor((uint32_t)s, d, kWave);
    }
    if (lane_id() == 0) red[threadIdx.x / kWave] = s;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < THREADS / kWave; ++w) t += red[w];
    __syncthreads();
    return t;
}

// SWEEP 1: epoch sweeps every EPOCH iterations (exact for any data; the shipped r01 kernel)
// SWEEP 0: none (exact only while no counter wraps)
// SWEEP 2: none inside a segment of SEG iterations; at the segment end a checksum decides between
//          flushing the segment and recounting it with sweeps (exact for any data)
// FLUSH 0: global atomics per bin from every workgroup; 1: none (timing only); 2: dump the packed
//          16-bit histogram to scratch[blockIdx] with plain stores, k_reduce sums the dumps
// LEAN inner loop: 4.75 vector instructions per pair instead of 8.  Per dword w (4 pairs): z = the stream shifted by one
// byte (v_alignbit), y = w ^ z (byte k of y = first ^ second of pair k), z7 = z & 0x7F7F7F7F; per pair one v_perm_b32
// builds the 15-bit word index ((second & 0x7F) << 8 | first ^ second: the same bin -> (word, half) map as bin ^= bin >> 8),
// one shift makes it a byte address, v_bfe + v_mad give the increment (1, or 0x10000 when the second byte's top bit is set).
// Inline asm so that the compiler does not turn the increment into compare + select (vcc hazards).
__device__ __forceinline__ void lean_pairs(uint32_t *hist, uint32_t w, uint32_t wn, uint32_t k0xffff) {
    const uint32_t z = __builtin_amdgcn_alignbit(wn, w, 8);
    const uint32_t y = w ^ z;
    const uint32_t z7 = z & 0x7F7F7F7Fu;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t h15 = __builtin_amdgcn_perm(z7, y, 0x0C0C0000u | ((4u + k) << 8) | (uint32_t)k);
        uint32_t b, inc;
        asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(b) : "v"(z), "n"(8 * k + 7));
        asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(b), "s"(k0xffff));
        atomicAdd(&hist[h15], inc);
    }
}

template <int THREADS, int SWEEP, int VPL, int FLUSH = 0, int LEAN = 0>
__global__ __launch_bounds__(THREADS) void k_pc(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ bp,
                                                uint32_t *__restrict__ n_redo, uint32_t *__restrict__ scratch = nullptr) {
    __shared__ uint32_t hist[kPcWords];
    __shared__ unsigned long long red[THREADS / kWave];
    for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += THREADS) hist[w] = 0;
    __syncthreads();
    constexpr int EPOCH = 49152 / (THREADS * 16 * VPL) > 0 ? 49152 / (THREADS * 16 * VPL) : 1;   // (VPL > 2: timing only)
    static_assert(EPOCH >= 1, "epoch");
    constexpr int ITER_VECS = THREADS * VPL;

    const uint64_t n_full = n / 16;
    uint64_t per = (n_full + gridDim.x - 1) / gridDim.x;
    per = (per + ITER_VECS - 1) / ITER_VECS * ITER_VECS;
    const uint64_t v_begin = per * blockIdx.x;
    uint64_t v_end = v_begin + per;
    if (v_end > n_full) v_end = n_full;
    const uint32_t lane = lane_id();
    const uint64_t last_vec = n_full ? n_full - 1 : 0;

    uint4 q[VPL];
    uint32_t xb[VPL];
    auto issue = [&](uint64_t base) {
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            uint64_t vec = base + u * THREADS + threadIdx.x;
            vec = vec < last_vec ? vec : last_vec;
            const uint64_t byte0 = vec * 16;
            q[u] = *reinterpret_cast<const uint4 *>(text + byte0);
            const uint64_t nx = byte0 + 16 < n ? byte0 + 16 : n - 1;
            xb[u] = text[lane == kWave - 1 ? nx : byte0];
        }
    };
    if (v_begin < v_end) issue(v_begin);

    auto count_iter = [&](uint64_t base, const uint4 *cq, const uint32_t *cxb) {
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const uint64_t vec = base + u * THREADS + threadIdx.x;
            uint32_t nb = __shfl_down(cq[u].x, 1, kWave) & 0xFFu;
            if (lane == kWave - 1) nb = cxb[u];
            const uint32_t valid = vec < v_end ? (vec + 1 < n_full ? 0xFFFFu : 0x7FFFu) : 0u;
            const uint32_t w[5] = {cq[u].x, cq[u].y, cq[u].z, cq[u].w, nb};
            if (LEAN && __ballot(valid != 0xFFFFu) == 0ull) {
                uint32_t c;
                asm volatile("s_mov_b32 %0, 0xffff" : "=s"(c));
#pragma unroll
                for (int i = 0; i < 4; ++i) lean_pairs(hist, w[i], w[i + 1], c);
            } else if (__ballot(valid != 0xFFFFu) == 0ull) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int wi = i >> 2, sh = 8 * (i & 3);
                    uint32_t bin;
                    if (sh <= 16) bin = (w[wi] >> sh) & 0xFFFFu;
                    else bin = ((w[wi] >> 24) | (w[wi + 1] << 8)) & 0xFFFFu;
                    bin ^= bin >> 8;
                    atomicAdd(&hist[bin & 0x7FFFu], 1u + (bin >> 15) * 0xFFFFu);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int wi = i >> 2, sh = 8 * (i & 3);
                    uint32_t bin;
                    if (sh <= 16) bin = (w[wi] >> sh) & 0xFFFFu;
                    else bin = ((w[wi] >> 24) | (w[wi + 1] << 8)) & 0xFFFFu;
                    bin ^= bin >> 8;
                    const uint32_t inc = ((valid >> i) & 1u) ? 1u + (bin >> 15) * 0xFFFFu : 0u;
                    atomicAdd(&hist[bin & 0x7FFFu], inc);
                }
            }
        }
    };

    if (SWEEP != 2) {
        int epoch_iter = 0;
        for (uint64_t base = v_begin; base < v_end; base += ITER_VECS) {
            uint4 cq[VPL];
            uint32_t cxb[VPL];
#pragma unroll
            for (int u = 0; u < VPL; ++u) { cq[u] = q[u]; cxb[u] = xb[u]; }
            issue(base + ITER_VECS < v_end ? base + ITER_VECS : base);
            count_iter(base, cq, cxb);
            if (SWEEP == 1 && ++epoch_iter == EPOCH) {
                epoch_iter = 0;
                __syncthreads();
                pc_sweep<THREADS>(hist, bp);
                __syncthreads();
            }
        }
        __syncthreads();
        pc_flush<THREADS>(hist, bp);
    } else {
        // segments of SEG iterations without sweeps; checksum at the end of each
        constexpr uint64_t SEG = 4096;    // iterations: 64 Mi pairs at 1024 threads (a bin overflows at 65,536)
        for (uint64_t seg = v_begin; seg < v_end; seg += SEG * ITER_VECS) {
            const uint64_t seg_end = seg + SEG * ITER_VECS < v_end ? seg + SEG * ITER_VECS : v_end;
            if (seg != v_begin) issue(seg);
            for (uint64_t base = seg; base < seg_end; base += ITER_VECS) {
                uint4 cq[VPL];
                uint32_t cxb[VPL];
#pragma unroll
                for (int u = 0; u < VPL; ++u) { cq[u] = q[u]; cxb[u] = xb[u]; }
                issue(base + ITER_VECS < seg_end ? base + ITER_VECS : base);
                count_iter(base, cq, cxb);
            }
            __syncthreads();
            // pairs this segment counted: 16 per vector, minus the straddling pair of the very last full vector
            unsigned long long want = (seg_end - seg) * 16ull;
            if (seg_end == n_full && seg_end > seg) want -= 1;
            const unsigned long long got = pc_checksum<THREADS>(hist, red);
            if (got == want) {
                if (FLUSH == 0) pc_flush<THREADS>(hist, bp);
                if (FLUSH == 2) {
                    uint4 *dst = reinterpret_cast<uint4 *>(scratch + (size_t)blockIdx.x * kPcWords);
                    for (int k = 0; k < kPcWords / 4 / THREADS; ++k)
                        dst[k * THREADS + threadIdx.x] = reinterpret_cast<uint4 *>(hist)[k * THREADS + threadIdx.x];
                }
                __syncthreads();
                for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += THREADS) hist[w] = 0;
                __syncthreads();
            } else {
                // a counter wrapped: forget the segment and recount it with epoch sweeps
                if (threadIdx.x == 0) atomicAdd(n_redo, 1u);
                for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += THREADS) hist[w] = 0;
                __syncthreads();
                issue(seg);
                int epoch_iter = 0;
                for (uint64_t base = seg; base < seg_end; base += ITER_VECS) {
                    uint4 cq[VPL];
                    uint32_t cxb[VPL];
#pragma unroll
                    for (int u = 0; u < VPL; ++u) { cq[u] = q[u]; cxb[u] = xb[u]; }
                    issue(base + ITER_VECS < seg_end ? base + ITER_VECS : base);
                    count_iter(base, cq, cxb);
                    if (++epoch_iter == EPOCH) {
                        epoch_iter = 0;
                        __syncthreads();
                        pc_sweep<THREADS>(hist, bp);
                        __syncthreads();
                    }
                }
                __syncthreads();
                pc_flush<THREADS>(hist, bp);
                __syncthreads();
                for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += THREADS) hist[w] = 0;
                __syncthreads();
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint64_t i = n_full * 16;
        if (i > 0) --i;
        if (n_full * 16 == n) i = n;
        for (; i + 1 < n; ++i) atomicAdd(&bp[((uint32_t)text[i] << 8) | text[i + 1]], 1u);
    }
}

struct Variant {
    const char *name;
    void (*launch)(const uint8_t *, uint64_t, uint32_t *, uint32_t *, int);
};

uint32_t *g_scratch = nullptr;

// sums the per-workgroup dumps: thread = one packed word (two bins)
__global__ void k_reduce(const uint32_t *__restrict__ scratch, int n_wg, uint32_t *__restrict__ bp) {
    const uint32_t word = blockIdx.x * blockDim.x + threadIdx.x;      // < 32768
    uint32_t lo = 0, hi = 0;
    for (int g = 0; g < n_wg; ++g) {
        const uint32_t v = scratch[(size_t)g * kPcWords + word];
        lo += v & 0xFFFFu;
        hi += v >> 16;
    }
    if (lo) atomicAdd(&bp[pc_table_index(word)], lo);
    if (hi) atomicAdd(&bp[pc_table_index(word | 0x8000u)], hi);
}

template <int THREADS, int SWEEP, int VPL, int FLUSH = 0, int LEAN = 0>
void launch(const uint8_t *text, uint64_t n, uint32_t *bp, uint32_t *redo, int cus) {
    hipLaunchKernelGGL((k_pc<THREADS, SWEEP, VPL, FLUSH, LEAN>), dim3(cus), dim3(THREADS), 0, 0, text, n, bp, redo, g_scratch);
    if (FLUSH == 2) hipLaunchKernelGGL(k_reduce, dim3(kPcWords / 256), dim3(256), 0, 0, g_scratch, cus, bp);
}

int main(int argc, char **argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (4ull << 30);
    int skew = argc > 2 ? atoi(argv[2]) : 0;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, n = %llu bytes, %s data\n", prop.name, cus, (unsigned long long)n, skew ? "skewed" : "uniform");
    uint8_t *text;
    CHK(hipMalloc(&text, n + 64));
    if (skew) hipLaunchKernelGGL(k_gen_skew, dim3(4096), dim3(256), 0, 0, text, n);
    else hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, (uint64_t *)text, (n + 7) / 8, 42ull);
    uint32_t *bp, *ref, *redo;
    CHK(hipMalloc(&bp, 65536 * 4));
    CHK(hipMalloc(&ref, 65536 * 4));
    CHK(hipMalloc(&redo, 4));
    CHK(hipMemset(ref, 0, 65536 * 4));
    hipLaunchKernelGGL(k_ref, dim3(8192), dim3(256), 0, 0, text, n, ref);
    CHK(hipDeviceSynchronize());
    std::vector<uint32_t> h_ref(65536), h(65536);
    CHK(hipMemcpy(h_ref.data(), ref, 65536 * 4, hipMemcpyDeviceToHost));

    Variant vs[] = {
        {"r01: 1024 thr, epoch sweeps, 1 vec/lane", launch<1024, 1, 1>},
        {"1024 thr, NO sweeps (inexact on skew), 1 vec", launch<1024, 0, 1>},
        {"1024 thr, segment checksum, 1 vec/lane", launch<1024, 2, 1>},
        {"1024 thr, segment checksum, 2 vec/lane", launch<1024, 2, 2>},
        {"512 thr, segment checksum, 1 vec/lane", launch<512, 2, 1>},
        {"512 thr, segment checksum, 2 vec/lane", launch<512, 2, 2>},
        {"512 thr, epoch sweeps, 2 vec/lane", launch<512, 1, 2>},
        {"1024 thr, NO sweeps, 2 vec", launch<1024, 0, 2>},
        {"1024 thr, segment checksum, 3 vec/lane", launch<1024, 2, 3>},
        {"1024 thr, segment checksum, 4 vec/lane", launch<1024, 2, 4>},
        {"LEAN 4.75 VALU/pair: 1024 thr, seg checksum, 2 vec", launch<1024, 2, 2, 0, 1>},
        {"LEAN: 1024 thr, seg checksum, 1 vec", launch<1024, 2, 1, 0, 1>},
        {"LEAN: 1024 thr, seg checksum, 4 vec", launch<1024, 2, 4, 0, 1>},
        {"1024 thr, seg checksum, 2 vec, NO FLUSH (timing)", launch<1024, 2, 2, 1>},
        {"1024 thr, seg checksum, 2 vec, dump + reduce", launch<1024, 2, 2, 2>},
        {"1024 thr, seg checksum, 4 vec, dump + reduce", launch<1024, 2, 4, 2>},
    };
    CHK(hipMalloc(&g_scratch, (size_t)cus * kPcWords * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (auto &v : vs) {
        std::vector<float> ms;
        uint32_t h_redo = 0;
        bool ok = true;
        for (int rep = 0; rep < 6; ++rep) {
            CHK(hipMemset(bp, 0, 65536 * 4));
            CHK(hipMemset(redo, 0, 4));
            CHK(hipEventRecord(e0, 0));
            v.launch(text, n, bp, redo, cus);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float t;
            CHK(hipEventElapsedTime(&t, e0, e1));
            if (rep) ms.push_back(t);
            if (rep == 1) {
                CHK(hipMemcpy(h.data(), bp, 65536 * 4, hipMemcpyDeviceToHost));
                CHK(hipMemcpy(&h_redo, redo, 4, hipMemcpyDeviceToHost));
                ok = h == h_ref;
            }
        }
        std::sort(ms.begin(), ms.end());
        const float med = ms[ms.size() / 2];
        printf("%-48s  %8.3f ms  %7.1f GB/s  frac %.3f  %s  redo %u\n", v.name, med, n / (med * 1e-3) / 1e9,
               n / (med * 1e-3) / 1e9 / 8000.0, ok ? "exact" : "WRONG", h_redo);
    }
    return 0;
}
