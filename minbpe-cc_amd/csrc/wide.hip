// Kernels of the 32-bit continuation of a training: see wide.h.  gfx950, wave64.
#include "wide.h"

#include <hip/hip_runtime.h>

namespace mbpe {
namespace {

constexpr int kWave = 64;
constexpr int kSpanIters = kWideSpan / kWave;
constexpr int kThreads = 256;                   // 4 waves = 4 spans per workgroup
constexpr int kScanThreads = 1024;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

__host__ __device__ inline uint32_t wide_hash(unsigned long long key, uint32_t shift) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift);
}

// create_or_modify_pair, PairCount.h:249-260: find the pair and add `delta`, or insert it with `delta`.  Entries are
// never removed.  Concurrent inserts of one key meet at the same free slot and the CAS lets exactly one of them in.
__device__ void wide_add(const WideTable &t, WideCtl *ctl, unsigned long long key, int32_t delta) {
    uint32_t h = wide_hash(key, t.shift);
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        unsigned long long k = t.keys[h];
        if (k == kWideEmpty) {
            k = atomicCAS(&t.keys[h], kWideEmpty, key);
            if (k == kWideEmpty) {
                atomicAdd(&ctl->n_entries, 1u);
                k = key;
            }
        }
        if (k == key) {
            atomicAdd(&t.cnts[h], delta);
            return;
        }
        h = (h + 1) & t.mask;
    }
    atomicOr(&ctl->err, 1u);
}

__global__ void k_wide_table_clear(WideTable t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i <= t.mask; i += stride) { t.keys[i] = kWideEmpty; t.cnts[i] = 0; }
}

__global__ void k_wide_table_from16(const uint32_t *__restrict__ ekey, const int32_t *__restrict__ ecnt, uint32_t n,
                                    WideTable t, WideCtl *ctl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t k16 = ekey[i];
        wide_add(t, ctl, ((unsigned long long)(k16 >> 16) << 32) | (k16 & 0xFFFFu), ecnt[i]);
    }
}

__global__ void k_wide_rehash(WideTable from, WideTable to, WideCtl *ctl) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i <= from.mask; i += stride) {
        const unsigned long long k = from.keys[i];
        if (k != kWideEmpty) wide_add(to, ctl, k, from.cnts[i]);
    }
}

// ---- argmax: (count desc, first asc, second asc) = the larger of (count, ~key) -------------------------------
struct Cand128 { long long count; unsigned long long nkey; };      // count = -1: nothing
__device__ __forceinline__ bool better(const Cand128 &a, const Cand128 &b) {
    return a.count > b.count || (a.count == b.count && a.nkey > b.nkey);
}
__device__ __forceinline__ Cand128 wave_best(Cand128 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Cand128 o;
        o.count = ((long long)__shfl_xor((int)(v.count >> 32), d, kWave) << 32) | (uint32_t)__shfl_xor((int)v.count, d, kWave);
        o.nkey = ((unsigned long long)(uint32_t)__shfl_xor((int)(v.nkey >> 32), d, kWave) << 32) |
                 (uint32_t)__shfl_xor((int)v.nkey, d, kWave);
        if (better(o, v)) v = o;
    }
    return v;
}
__device__ Cand128 block_best(Cand128 v, Cand128 *sh) {
    v = wave_best(v);
    __syncthreads();
    if (lane_id() == 0) sh[threadIdx.x / kWave] = v;
    __syncthreads();
    Cand128 r = sh[0];
    for (uint32_t w = 1; w < blockDim.x / kWave; ++w)
        if (better(sh[w], r)) r = sh[w];
    return r;
}

constexpr int kArgBlocks = 1024;
__global__ __launch_bounds__(256) void k_wide_argmax_partial(WideTable t, const WideCtl *ctl,
                                                             unsigned long long *__restrict__ part) {
    __shared__ Cand128 sh[4];
    Cand128 v = {-1, 0};
    if (ctl->k < ctl->k_limit) {
        uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (; i <= t.mask; i += stride) {
            const unsigned long long k = t.keys[i];
            if (k == kWideEmpty) continue;
            Cand128 c = {(long long)t.cnts[i], ~k};
            if (better(c, v)) v = c;
        }
    }
    v = block_best(v, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = (unsigned long long)v.count; part[2 * blockIdx.x + 1] = v.nkey; }
}

__global__ __launch_bounds__(kArgBlocks) void k_wide_argmax_final(const unsigned long long *__restrict__ part,
                                                                  WideCtl *ctl, WideBest *best) {
    __shared__ Cand128 sh[kArgBlocks / kWave];
    if (threadIdx.x == 0) ctl->ran = 0;        // (whatever follows: no scatter without a merge of its own)
    if (ctl->k >= ctl->k_limit) return;
    Cand128 v = {(long long)part[2 * threadIdx.x], part[2 * threadIdx.x + 1]};
    v = block_best(v, sh);
    if (threadIdx.x == 0) {
        const bool any = v.count >= 0;         // (a count can never be negative: every decrement undoes an increment)
        const unsigned long long key = ~v.nkey;
        ctl->live = any ? 1u : 0u;
        ctl->a = (uint32_t)(key >> 32);
        ctl->b = (uint32_t)key;
        ctl->count = any ? (int32_t)v.count : 0;
        ctl->matches = 0;
        if (any) {
            WideBest wb = {(int32_t)v.count, (uint32_t)(key >> 32), (uint32_t)key, 0u};
            best[ctl->k] = wb;
        }
    }
}

// ---- scans over the spans (one workgroup, two sweeps; as in encode.hip) ---------------------------------------
__global__ __launch_bounds__(kScanThreads) void k_wide_scan_parity(const uint32_t *__restrict__ span_sum,
                                                                   const WideCtl *ctl, uint32_t *__restrict__ in_par) {
    __shared__ uint32_t sh[kScanThreads];
    const uint64_t n_spans = (ctl->n + kWideSpan - 1) / kWideSpan;
    const uint64_t per = (n_spans + kScanThreads - 1) / kScanThreads;
    const uint64_t lo = per * threadIdx.x, hi = lo + per < n_spans ? lo + per : n_spans;
    uint32_t all = 1, par = 0;
    for (uint64_t s = lo; s < hi; ++s) {
        const uint32_t v = span_sum[s];
        if (v & 1u) par ^= v >> 1; else { all = 0; par = v >> 1; }
    }
    sh[threadIdx.x] = all | (par << 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t p = 0;
        for (int t = 0; t < kScanThreads; ++t) {
            const uint32_t v = sh[t];
            sh[t] = p;
            if (v & 1u) p ^= v >> 1; else p = v >> 1;
        }
    }
    __syncthreads();
    par = sh[threadIdx.x];
    for (uint64_t s = lo; s < hi; ++s) {
        in_par[s] = par;
        const uint32_t v = span_sum[s];
        if (v & 1u) par ^= v >> 1; else par = v >> 1;
    }
}

// exclusive sums of the spans' kept-token counts over the spans of n_in tokens; the total becomes ctl->n (and, when
// `advance`, the merge counter moves on)
__global__ __launch_bounds__(kScanThreads) void k_wide_scan_sum(const uint32_t *__restrict__ cnt, uint64_t n_in_fixed,
                                                                unsigned long long *__restrict__ off, WideCtl *ctl,
                                                                int advance) {
    __shared__ unsigned long long sh[kScanThreads];
    if (advance && (ctl->k >= ctl->k_limit || !ctl->live)) return;
    const uint64_t n_in = advance ? ctl->n : n_in_fixed;
    const uint64_t n_spans = (n_in + kWideSpan - 1) / kWideSpan;
    const uint64_t per = (n_spans + kScanThreads - 1) / kScanThreads;
    const uint64_t lo = per * threadIdx.x, hi = lo + per < n_spans ? lo + per : n_spans;
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += cnt[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int t = 0; t < kScanThreads; ++t) { const unsigned long long v = sh[t]; sh[t] = acc; acc += v; }
        sh[0] = 0;
        ctl->n_prev = n_in;                    // (the scatter walks the stream that was read)
        ctl->n = acc;
        ctl->ran = 1;
        if (advance) ctl->k += 1;
    }
    __syncthreads();
    s = threadIdx.x == 0 ? 0 : sh[threadIdx.x];
    for (uint64_t i = lo; i < hi; ++i) { off[i] = s; s += cnt[i]; }
}

__global__ __launch_bounds__(kThreads) void k_wide_scatter(const uint32_t *__restrict__ val,
                                                           const unsigned long long *__restrict__ span_off,
                                                           uint32_t *__restrict__ out, const WideCtl *ctl) {
    // (a merge that did not run -- limit reached, empty table -- must not scatter either)
    if (!ctl->ran) return;
    const uint64_t n_upper = ctl->n_prev;
    const uint64_t span = (uint64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kWideSpan;
    if (base >= n_upper) return;
    const uint32_t lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long o = span_off[span];
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        const uint32_t v = i < n_upper ? val[i] : kWideNone;
        const unsigned long long K = __ballot(v != kWideNone);
        if (v != kWideNone) out[o + (uint32_t)__popcll(K & lt)] = v;
        o += (uint32_t)__popcll(K);
    }
}

// ---- conversion of the 16-bit slot stream ---------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_wide_from_slots(const uint16_t *__restrict__ slots, uint64_t n_live,
                                                              uint32_t barrier, uint32_t endbit, uint32_t *__restrict__ val,
                                                              uint32_t *__restrict__ span_keep) {
    const uint64_t span = (uint64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kWideSpan;
    if (base >= n_live) return;
    const uint32_t lane = lane_id();
    uint32_t kept = 0;
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        uint32_t v = kWideNone;
        if (i < n_live) {
            const uint32_t s = slots[i];
            if (s != barrier) {
                const bool last = i + 1 >= n_live;
                const uint32_t nx = last ? barrier : slots[i + 1];
                // flag bit: the slot says so; barrier layout: a barrier follows; one chunk: only the very last token
                const bool end = endbit ? (s & endbit) != 0u : barrier == 0xFFFFFFFFu ? last : nx == barrier;
                v = (s & ~endbit) | (end ? kWideEnd : 0u);
            }
            val[i] = v;
        }
        kept += (uint32_t)__popcll(__ballot(v != kWideNone));
    }
    if (lane == 0) span_keep[span] = kept;
}

// ---- one merge ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_cand(uint32_t t, uint32_t nx, uint32_t a, uint32_t b) {
    return !(t & kWideEnd) && t == a && (nx & kWideIdMask) == b;      // (t carries no end flag here, so t == a compares ids)
}

__global__ __launch_bounds__(kThreads) void k_wide_cand(const uint32_t *__restrict__ tok, uint64_t n_upper,
                                                        const WideCtl *ctl, uint32_t *__restrict__ span_sum) {
    if (ctl->k >= ctl->k_limit || !ctl->live) return;
    const uint64_t n = ctl->n;
    const uint64_t span = (uint64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kWideSpan;
    if (base >= n) return;
    const uint32_t a = ctl->a, b = ctl->b;
    const uint32_t lane = lane_id();
    bool all = true;
    uint32_t par = 0;
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        const uint32_t t = i < n ? tok[i] : kWideEnd;
        uint32_t nx = __shfl_down(t, 1, kWave);
        if (lane == kWave - 1) nx = i + 1 < n ? tok[i + 1] : kWideEnd;
        const bool c = i < n && i + 1 < n && is_cand(t, nx, a, b);
        const unsigned long long M = __ballot(c);
        if (M != ~0ull) {
            all = false;
            par = (uint32_t)__builtin_clzll(~M) & 1u;
        }
    }
    (void)n_upper;
    if (lane == 0) span_sum[span] = (all ? 1u : 0u) | (par << 1);
}

__global__ __launch_bounds__(kThreads) void k_wide_match(const uint32_t *__restrict__ tok, const uint32_t *__restrict__ in_par,
                                                         uint32_t *__restrict__ val, uint32_t *__restrict__ span_keep,
                                                         WideTable tab, WideCtl *ctl, uint32_t new_id_base) {
    if (ctl->k >= ctl->k_limit || !ctl->live) return;
    const uint64_t n = ctl->n;
    const uint64_t span = (uint64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kWideSpan;
    if (base >= n) return;
    const uint32_t a = ctl->a, b = ctl->b, X = new_id_base + ctl->k;
    const uint32_t lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t carry = in_par[span];
    uint32_t kept = 0, n_match = 0;
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        const uint32_t t = i < n ? tok[i] : kWideEnd;
        uint32_t nx = __shfl_down(t, 1, kWave);
        if (lane == kWave - 1) nx = i + 1 < n ? tok[i + 1] : kWideEnd;
        const bool c = i < n && i + 1 < n && is_cand(t, nx, a, b);
        const unsigned long long M = __ballot(c);
        // r = consecutive candidates immediately below this lane (continuing into `carry` when all of them are)
        const unsigned long long zeros_below = ~M & lt;
        uint32_t r;
        if (zeros_below == 0ull) r = lane + carry;
        else r = lane - 1u - (63u - (uint32_t)__builtin_clzll(zeros_below));
        const bool odd = r & 1u;
        uint32_t v = kWideNone;
        const bool match = c && !odd;
        if (i < n && !odd) v = match ? (X | (nx & kWideEnd)) : t;
        if (i < n) val[i] = v;
        kept += (uint32_t)__popcll(__ballot(v != kWideNone));
        n_match += (uint32_t)__popcll(__ballot(match));
        if (match) {
            // Tokenizer.h:248-260: the left neighbour as it stands AFTER the walk has passed it.  It was swallowed by a
            // match (and is X now) iff a match ends right before i: for a == b that is "(i-1, i) is a candidate too"
            // (then the run before i-1 is odd, r being even here), for a != b "(i-2, i-1) is a candidate" (such
            // candidates never touch, so every one of them is a match).
            if (i > 0) {
                const uint32_t p = tok[i - 1];
                if (!(p & kWideEnd)) {
                    bool swallowed;
                    if (a == b) swallowed = p == a;                   // (no end flag on p: (p, t) is a candidate)
                    else swallowed = i > 1 && (p & kWideIdMask) == b && tok[i - 2] == a;   // (tok[i-2] == a: no end flag, id a)
                    const uint32_t x = swallowed ? X : p;
                    wide_add(tab, ctl, ((unsigned long long)x << 32) | a, -1);
                    wide_add(tab, ctl, ((unsigned long long)x << 32) | X, 1);
                }
            }
            // :263-279: the right neighbour as it still is
            if (!(nx & kWideEnd) && i + 2 < n) {
                const uint32_t y = tok[i + 2] & kWideIdMask;
                wide_add(tab, ctl, ((unsigned long long)b << 32) | y, -1);
                wide_add(tab, ctl, ((unsigned long long)X << 32) | y, 1);
            }
        }
        if (M != ~0ull) carry = (uint32_t)__builtin_clzll(~M) & 1u;
    }
    if (lane == 0) {
        span_keep[span] = kept;
        if (n_match) {
            wide_add(tab, ctl, ((unsigned long long)a << 32) | b, -(int32_t)n_match);      // :240-246
            atomicAdd(&ctl->matches, n_match);
        }
    }
}

}  // namespace

size_t wide_scratch_words(uint64_t n_tokens) {
    const uint64_t n_spans = (n_tokens + kWideSpan - 1) / kWideSpan + 2;
    return (size_t)(n_spans * 5 + 8);          // span_sum, in_par, span_keep (u32 each), span_off (u64)
}

namespace {
struct Scratch { uint32_t *span_sum, *in_par, *span_keep; unsigned long long *span_off; };
Scratch carve(uint32_t *scratch, uint64_t n_tokens) {
    const uint64_t n_spans = (n_tokens + kWideSpan - 1) / kWideSpan + 2;
    Scratch s;
    s.span_off = reinterpret_cast<unsigned long long *>(scratch);          // (8-byte aligned: first)
    s.span_sum = scratch + 2 * n_spans;
    s.in_par = s.span_sum + n_spans;
    s.span_keep = s.in_par + n_spans;
    return s;
}
uint32_t span_grid(uint64_t n) {
    const uint64_t n_spans = (n + kWideSpan - 1) / kWideSpan;
    return (uint32_t)((n_spans + kThreads / kWave - 1) / (kThreads / kWave));
}
}  // namespace

void launch_wide_from_slots(hipStream_t s, const uint16_t *slots, uint64_t n_live, uint32_t barrier, uint32_t endbit,
                            uint32_t *val, uint32_t *span_scratch, uint32_t *tok_out, WideCtl *ctl) {
    if (!n_live) return;
    const Scratch sc = carve(span_scratch, n_live);
    hipLaunchKernelGGL(k_wide_from_slots, dim3(span_grid(n_live)), dim3(kThreads), 0, s, slots, n_live, barrier, endbit, val, sc.span_keep);
    hipLaunchKernelGGL(k_wide_scan_sum, dim3(1), dim3(kScanThreads), 0, s, sc.span_keep, n_live, sc.span_off, ctl, 0);
    hipLaunchKernelGGL(k_wide_scatter, dim3(span_grid(n_live)), dim3(kThreads), 0, s, val, sc.span_off, tok_out, ctl);
}

void launch_wide_table_clear(hipStream_t s, WideTable t) {
    hipLaunchKernelGGL(k_wide_table_clear, dim3(2048), dim3(256), 0, s, t);
}

void launch_wide_table_from16(hipStream_t s, const uint32_t *ekey, const int32_t *ecnt, uint32_t n_entries, WideTable t,
                              WideCtl *ctl) {
    if (!n_entries) return;
    const uint32_t blocks = n_entries / 256 + 1 > 4096 ? 4096 : n_entries / 256 + 1;
    hipLaunchKernelGGL(k_wide_table_from16, dim3(blocks), dim3(256), 0, s, ekey, ecnt, n_entries, t, ctl);
}

void launch_wide_rehash(hipStream_t s, WideTable from, WideTable to, WideCtl *ctl) {
    hipLaunchKernelGGL(k_wide_rehash, dim3(2048), dim3(256), 0, s, from, to, ctl);
}

void launch_wide_argmax(hipStream_t s, WideTable t, WideCtl *ctl, WideBest *best, unsigned long long *scratch) {
    hipLaunchKernelGGL(k_wide_argmax_partial, dim3(kArgBlocks), dim3(256), 0, s, t, ctl, scratch);
    hipLaunchKernelGGL(k_wide_argmax_final, dim3(1), dim3(kArgBlocks), 0, s, scratch, ctl, best);
}

void launch_wide_merge(hipStream_t s, const uint32_t *src, uint32_t *dst, uint64_t n_upper, uint32_t *val,
                       uint32_t *span_scratch, WideTable t, WideCtl *ctl, uint32_t new_id_base) {
    if (!n_upper) return;
    const Scratch sc = carve(span_scratch, n_upper);
    const dim3 grid(span_grid(n_upper)), block(kThreads);
    hipLaunchKernelGGL(k_wide_cand, grid, block, 0, s, src, n_upper, ctl, sc.span_sum);
    hipLaunchKernelGGL(k_wide_scan_parity, dim3(1), dim3(kScanThreads), 0, s, sc.span_sum, ctl, sc.in_par);
    hipLaunchKernelGGL(k_wide_match, grid, block, 0, s, src, sc.in_par, val, sc.span_keep, t, ctl, new_id_base);
    hipLaunchKernelGGL(k_wide_scan_sum, dim3(1), dim3(kScanThreads), 0, s, sc.span_keep, (uint64_t)0, sc.span_off, ctl, 1);
    hipLaunchKernelGGL(k_wide_scatter, grid, block, 0, s, val, sc.span_off, dst, ctl);
}

}  // namespace mbpe
