// Count deltas of a fused pass without scattered global atomics (gfx950, wave64).
//
// merge_incremental (reference Tokenizer.h:239-280) updates up to five pair counts per match.  A fused pass
// (k_fused_batch, kernels.hip) merges up to 1024 pairs at once and owes the pair table two count deltas per match
// -- L_j[x] += 1 for the left neighbour x of a match of pair j, R_j[y] += 1 for its right neighbour -- about 1e8 of
// them per pass over a 4 GiB corpus, scattered over 1e5..1e7 cells of the LR block.  As global atomics they execute
// at the memory side at ~24 G/s whatever their layout: more than half of the pass.  Instead the
// streaming waves LOG them: a record is the cell index (bit 31: subtract), staged per wave in LDS and written 64 at
// a time into chunks of a log in HBM.  Three small kernels then turn the log into the LR block:
//
//   k_log_plan       (one workgroup, before the pass)  the LR cells the batch can touch are cut into buckets of 32,768
//                    cells; a bucket's record count is bounded by the counts of the pairs whose rows overlap it (a match
//                    of j gives at most one L_j and two R_j records), which fixes every bucket's region in the
//                    partition buffer without a counting pass.  Decides whether the pass logs at all (LogState::on).
//   k_log_partition  multisplit: a workgroup takes slices of 8,192 records, histograms them by bucket in LDS, reserves
//                    room in every bucket's region with one atomic per bucket and slice, sorts the slice by bucket in
//                    LDS and writes it out in runs.
//   k_log_count      a bucket's records are counted in a 128-KiB LDS histogram (one 32-bit counter per cell) and the
//                    non-zero counters are added to the LR block: contiguous, so at the full rate of the atomic unit.
//                    Few buckets (early in a run: 300 ids x 800 pairs = 15 buckets) are split over several workgroups.
//
// Everything is exact integer arithmetic (a subtract record always cancels an add record of the same cell), so the LR
// block is bit-identical to what the atomics produce; a log that is full makes the waves fall back to atomics for the
// records that do not fit (LogState::spilled counts them).
#include "mbpe_dev.h"

namespace mbpe {

namespace {

constexpr int kWave = 64;
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, kWave);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

// exclusive scan of two values per thread over a 1024-thread workgroup (thread t owns entries 2t, 2t + 1);
// returns the exclusive prefix of entry 2t and leaves the grand total in *total (LDS)
__device__ __forceinline__ uint32_t block_excl_scan2(uint32_t a, uint32_t b, uint32_t *wsum /* [16] */, uint32_t *total) {
    const uint32_t lane = lane_id(), w = threadIdx.x / kWave;
    const uint32_t incl = wave_incl_scan(a + b);
    if (lane == kWave - 1) wsum[w] = incl;
    __syncthreads();
    if (threadIdx.x < 16) {
        uint32_t v = wsum[threadIdx.x], s = v;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t t = __shfl_up(s, d, 16);
            if (threadIdx.x >= (uint32_t)d) s += t;
        }
        wsum[threadIdx.x] = s - v;
        if (threadIdx.x == 15) *total = s;
    }
    __syncthreads();
    return wsum[w] + incl - (a + b);
}

constexpr int kPlanThreads = 1024;

__global__ __launch_bounds__(kPlanThreads) void k_log_plan(const DevCtl *ctl, const BatchState *bs, LogState *ls) {
    __shared__ uint32_t cap[kLogMaxBuckets];
    __shared__ uint32_t wsum[16], total;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { ls->on = 0; ls->n_alloc = 0; }
    const uint32_t n = ctl->batch_n;
    if (n < 2 || !ctl->fused || !ls->enabled) return;
    // (the frequent-pair instantiation of the pass counts in its LDS cache instead: kernels.hip, DeltaCache)
    if (dc_wanted(bs->packed[0] >> 32, ctl->n_live * ctl->n_ranks)) return;
    const uint32_t pitch = lr_pitch(256u + ctl->k_done);
    const uint64_t cells = 2ull * n * pitch;
    const uint32_t P = (uint32_t)((cells + kLogBucketCells - 1u) >> kLogBucketShift);
    if (P > kLogMaxBuckets) return;
    for (uint32_t i = tid; i < kLogMaxBuckets; i += kPlanThreads) cap[i] = 0;
    __syncthreads();
    // a row's records all fall into the buckets its cells overlap (at most three: pitch <= 65,536)
    for (uint32_t r = tid; r < 2u * n; r += kPlanThreads) {
        const unsigned long long cnt = bs->packed[r >> 1] >> 32;
        unsigned long long c = (r & 1u) ? 2ull * cnt : cnt;          // L_j: one per match; R_j: an add and a take-back
        if (c > 0x7FFFFFFFull) c = 0x7FFFFFFFull;
        const uint32_t b0 = (uint32_t)(((uint64_t)r * pitch) >> kLogBucketShift);
        const uint32_t b1 = (uint32_t)(((uint64_t)(r + 1u) * pitch - 1u) >> kLogBucketShift);
        for (uint32_t b = b0; b <= b1; ++b) atomicAdd(&cap[b], (uint32_t)c);    // (a wrap shows in the 64-bit sum below)
    }
    __syncthreads();
    // 64-bit total first (a 32-bit scan may not wrap)
    unsigned long long sum64 = 0;
    for (uint32_t r = tid; r < 2u * n; r += kPlanThreads) {
        const unsigned long long cnt = bs->packed[r >> 1] >> 32;
        const uint32_t b0 = (uint32_t)(((uint64_t)r * pitch) >> kLogBucketShift);
        const uint32_t b1 = (uint32_t)(((uint64_t)(r + 1u) * pitch - 1u) >> kLogBucketShift);
        sum64 += ((r & 1u) ? 2ull * cnt : cnt) * (b1 - b0 + 1u);
    }
    __shared__ unsigned long long s64;
    if (tid == 0) s64 = 0;
    __syncthreads();
    if (sum64) atomicAdd(&s64, sum64);
    __syncthreads();
    if (s64 > (unsigned long long)ls->part_cap) return;             // does not fit: this pass keeps its atomics
    // few matches (s64 is about three records per match): the atomics cost less than the three kernels' fixed parts
    if (ls->enabled == 1u && s64 < 3ull * kLogMinMatches) return;
    const uint32_t a = cap[2 * tid], b = cap[2 * tid + 1];
    const uint32_t ex = block_excl_scan2(a, b, wsum, &total);
    ls->base[2 * tid] = ex;
    ls->base[2 * tid + 1] = ex + a;
    ls->fill[2 * tid] = 0;
    ls->fill[2 * tid + 1] = 0;
    if (tid == 0) {
        ls->base[kLogMaxBuckets] = total;
        ls->n_buckets = P;
        ls->on = 1;
        ls->passes += 1;
    }
}

// ---- multisplit of the log by bucket -----------------------------------------------------------
constexpr int kPartThreads = 1024;
constexpr uint32_t kSlice = 8192;          // records per slice: 8 per thread
static_assert(kLogMaxBuckets == 2u * kPartThreads && kLogMaxBuckets == 2u * kPlanThreads, "a thread owns two buckets in the scans");
static_assert(kLogChunk % 8u == 0 && kSlice % kLogChunk == 0, "a thread's 8 records never straddle the end of the log");

__global__ __launch_bounds__(kPartThreads) void k_log_partition(const uint32_t *__restrict__ dlog, uint32_t *__restrict__ part,
                                                                LogState *ls, DevCtl *ctl) {
    __shared__ uint32_t hist[kLogMaxBuckets];       // records of the slice per bucket, then the running placement rank
    __shared__ uint32_t off[kLogMaxBuckets + 2];    // exclusive prefix of hist: where a bucket starts in `sorted`
    __shared__ uint32_t gb[kLogMaxBuckets];         // where the slice's records of a bucket go in the partition buffer
    __shared__ uint32_t sorted[kSlice];
    __shared__ uint32_t wsum[16], total;
    if (!ls->on) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t cap_whole = ls->cap - ls->cap % kLogChunk;
    const uint32_t n_alloc = ls->n_alloc;
    const uint32_t n_rec = n_alloc < cap_whole ? n_alloc : cap_whole;      // chunks beyond the capacity were refused
    const uint32_t n_slices = (n_rec + kSlice - 1u) / kSlice;
    if (blockIdx.x == 0 && tid == 0) ls->total_records += n_rec;
    for (uint32_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        for (uint32_t i = tid; i < kLogMaxBuckets; i += kPartThreads) hist[i] = 0;
        __syncthreads();
        // (n_rec is a multiple of kLogChunk = 1024 records, a thread's 8 records never straddle its end)
        const uint32_t r0 = sl * kSlice + tid * 8u;
        uint32_t rec[8];
        if (r0 < n_rec) {
            const uint4 q0 = reinterpret_cast<const uint4 *>(dlog + r0)[0], q1 = reinterpret_cast<const uint4 *>(dlog + r0)[1];
            rec[0] = q0.x; rec[1] = q0.y; rec[2] = q0.z; rec[3] = q0.w;
            rec[4] = q1.x; rec[5] = q1.y; rec[6] = q1.z; rec[7] = q1.w;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) rec[u] = kLogNull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rec[u] != kLogNull) atomicAdd(&hist[(rec[u] & ~kLogNeg) >> kLogBucketShift], 1u);
        __syncthreads();
        const uint32_t a = hist[2 * tid], b = hist[2 * tid + 1];
        const uint32_t ex = block_excl_scan2(a, b, wsum, &total);
        off[2 * tid] = ex;
        off[2 * tid + 1] = ex + a;
        // room in the buckets' regions: one atomic per bucket that occurs in the slice
        if (a) {
            const uint32_t at = atomicAdd(&ls->fill[2 * tid], a);
            gb[2 * tid] = ls->base[2 * tid] + at;
            if (ls->base[2 * tid] + at + a > ls->base[2 * tid + 1]) atomicOr(&ctl->err, kErrLog);
        }
        if (b) {
            const uint32_t at = atomicAdd(&ls->fill[2 * tid + 1], b);
            gb[2 * tid + 1] = ls->base[2 * tid + 1] + at;
            if (ls->base[2 * tid + 1] + at + b > ls->base[2 * tid + 2]) atomicOr(&ctl->err, kErrLog);
        }
        hist[2 * tid] = 0;
        hist[2 * tid + 1] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (rec[u] == kLogNull) continue;
            const uint32_t bk = (rec[u] & ~kLogNeg) >> kLogBucketShift;
            const uint32_t r = atomicAdd(&hist[bk], 1u);
            sorted[off[bk] + r] = rec[u];
        }
        __syncthreads();
        const uint32_t n_sorted = total;
        const bool ok = !(ctl->err & kErrLog);          // (a region overflow would write out of bounds: never expected)
        for (uint32_t t = tid; t < n_sorted; t += kPartThreads) {
            const uint32_t v = sorted[t];
            const uint32_t bk = (v & ~kLogNeg) >> kLogBucketShift;
            if (ok) part[gb[bk] + (t - off[bk])] = v;
        }
        __syncthreads();
    }
}

// ---- count a bucket in LDS, add the counters to the LR block --------------------------------------
constexpr int kCountThreads = 1024;

__global__ __launch_bounds__(kCountThreads) void k_log_count(const uint32_t *__restrict__ part, uint32_t *LR, const LogState *ls) {
    __shared__ uint32_t hist[kLogBucketCells];
    if (!ls->on) return;
    const uint32_t P = ls->n_buckets;
    const uint32_t S = P >= (uint32_t)gridDim.x ? 1u : (uint32_t)gridDim.x / P > 64u ? 64u : (uint32_t)gridDim.x / P;   // workgroups per bucket
    const uint32_t tid = threadIdx.x;
    // (a workgroup may have to take several buckets: more buckets than workgroups)
    for (uint32_t w = blockIdx.x; w < P * S; w += gridDim.x) {
        const uint32_t b = w / S, share = w - b * S;
        const uint32_t n = ls->fill[b], start = ls->base[b];
        const uint32_t lo = (uint32_t)((uint64_t)n * share / S), hi = (uint32_t)((uint64_t)n * (share + 1u) / S);
        if (lo == hi) continue;
        for (uint32_t i = tid; i < kLogBucketCells; i += kCountThreads) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = lo + tid; i < hi; i += kCountThreads) {
            const uint32_t v = part[start + i];
            atomicAdd(&hist[v & (kLogBucketCells - 1u)], (v & kLogNeg) ? 0xFFFFFFFFu : 1u);
        }
        __syncthreads();
        uint32_t *cells = LR + ((size_t)b << kLogBucketShift);
        for (uint32_t i = tid; i < kLogBucketCells; i += kCountThreads) {
            const uint32_t c = hist[i];
            if (c) atomicAdd(&cells[i], c);
        }
        __syncthreads();
    }
}

}  // namespace

size_t log_state_bytes() { return sizeof(LogState); }

void launch_log_plan(hipStream_t s, const DevCtl *ctl, const BatchState *bs, LogState *ls) {
    hipLaunchKernelGGL(k_log_plan, dim3(1), dim3(kPlanThreads), 0, s, ctl, bs, ls);
}

void launch_log_consume(hipStream_t s, const uint32_t *dlog, uint32_t *part, uint32_t *LR, LogState *ls, DevCtl *ctl,
                        int n_cus) {
    const int cus = n_cus > 0 ? n_cus : 256;
    hipLaunchKernelGGL(k_log_partition, dim3(cus * 2), dim3(kPartThreads), 0, s, dlog, part, ls, ctl);
    hipLaunchKernelGGL(k_log_count, dim3(cus), dim3(kCountThreads), 0, s, part, LR, ls);
}

}  // namespace mbpe
