// HIP kernels of the BPE training hot path for gfx950 (MI355X, wave64).
//
// Reference path replaced (code/include/ of justinhj/minbpe-cc):
//   calculate_freqs            Tokenizer.h:127-146   -> k_pair_count_u8 (+ k_table_init)
//   get_top_pair_count         PairCount.h:262-269   -> k_sel_scan + k_sel_pick / k_select_batch
//                                                       (k_argmax / k_argmax_hier for one merge at a time)
//   merge_chunks / merge_incremental
//                              Tokenizer.h:309-320, :202-306 -> k_fused_batch (large batches),
//                                                       k_scan_batch + k_rewrite_marked (small ones),
//                                                       k_merge (one pair); count updates:
//                                                       k_validate + k_apply_batch[_dense] / k_apply
//   create_lists / text_to_vector
//                              Tokenizer.h:114-124, :85-100 -> k_widen
//
// Integer / index work, HBM-bound: no MFMA anywhere.  Everything is exact
// (integer atomics commute), so repeated runs are bit-identical.
//
// The stream kernels are wave-centric: one wave owns one 512-slot tile at a
// time (8 slots = one 16-byte load per lane), keeps it in registers, finds
// neighbours with ballots / ds_bpermute instead of LDS staging, and never
// needs a workgroup barrier.  Waves walk the tiles with a grid stride and
// prefetch their next tile while working on the current one.
#include "mbpe_dev.h"
#include <hip/hip_ext.h>

#include <cstdlib>

namespace mbpe {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kSent = 0x10000u;   // "nothing in this lane" (not a 16-bit value)

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// whole-wave shifts by one lane (DPP): lane i takes lane i+1 / lane i-1, the last / first lane takes `edge`
__device__ __forceinline__ uint32_t wave_from_next(uint32_t v, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false);     // wave_shl:1
}
__device__ __forceinline__ uint32_t wave_from_prev(uint32_t v, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);     // wave_shr:1
}
// this lane's bit of a wave mask, as a condition (one v_cndmask at the use, no 64-bit lane arithmetic)
__device__ __forceinline__ bool lane_of(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

#define MBPE_GLOBAL_AS __attribute__((address_space(1)))
// The stream's tile loads and the fused pass's tile stores are non-temporal accesses: a pass reads and writes gigabytes
// once each, and what profits from the caches is the count-delta block its atomics hit at random.  Same box, whole
// training of the benchmark workload: 76,800 -> 79,100 merges/s, fused pass 5.19 -> 5.01 ms on average (0: plain, A/B).
#ifndef MBPE_NT_STREAM
#define MBPE_NT_STREAM 1
#endif
// a wave-uniform address, pinned to scalar registers
__device__ __forceinline__ uintptr_t uniform_ptr(uintptr_t p) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return ((uintptr_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t rlane(uint32_t v, uint32_t uniform_lane) {
    return __builtin_amdgcn_readlane(v, rfl(uniform_lane));
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t t = __shfl_up(v, d, kWave);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, d, kWave);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), d, kWave);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int d) {
    const uint32_t lo = __shfl_xor((uint32_t)v, d, kWave), hi = __shfl_xor((uint32_t)(v >> 32), d, kWave);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, kWave);
        v = o > v ? o : v;
    }
    return v;
}

// sum over all lanes of a per-lane count in [0, 8], via ballots (scalar popcounts)
__device__ __forceinline__ uint32_t wave_sum_le8(uint32_t v, unsigned long long lane_mask) {
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) s += __popcll(__ballot(v > k) & lane_mask);
    return s;
}

// ---- pair table device ops ------------------------------------------------

// Count-delta vectors of up to kBatchMax simultaneous merges, pair-major: member j owns the two rows 2j (L_j) and
// 2j + 1 (R_j) of `pitch` cells each, pitch = lr_pitch(ids that exist when the sequence starts), so that the cells a
// batch of n pairs can touch are the prefix [0, 2 * n * pitch) of the block -- what a multi-GPU run exchanges:
//   L_j[x] = LR[lr_idx(pitch, x, j, 0)],  R_j[y] = LR[lr_idx(pitch, y, j, 1)]
// (largest index: 2 * kBatchMax rows of at most 65,536 cells = 2^29 at the cap of 4096.  The fused pass feeds idx << 2
//  to a buffer atomic as a 32-bit byte offset -- num_records 0xFFFFFFFC -- so the whole block has to stay below 4 GiB)
static_assert(2ull * kBatchMax * 65536ull * 4ull <= 0x100000000ull,
              "the LR block must be addressable with a 32-bit byte offset (lr_idx, lr_rsrc)");
static_assert(2u * kBatchMax < (1u << 24), "lr_idx multiplies the row with v_mad_u32_u24");
__device__ __forceinline__ uint32_t lr_idx(uint32_t pitch, uint32_t x, uint32_t j, uint32_t side) {
    return __umul24(2u * j + side, pitch) + x;          // (row < 2 * kBatchMax, pitch <= 65,536: one v_mad_u32_u24)
}

// Count deltas of frequent neighbours.  In text a pair such as ("e", " ") has millions of
// occurrences whose left neighbour is one of a handful of tokens, and a global atomicAdd on
// one address runs at roughly 10 ns each whoever issues it.  Each workgroup therefore keeps a
// small direct-mapped cache of LR cells in LDS: a cell that owns its slot is counted with LDS
// atomics and written back once when the kernel ends; cells that lose the race for a slot go
// straight to memory.  `on` (uniform) is only set for batches whose top pair is frequent
// enough for this to matter; uniform-random corpora bypass the cache.
constexpr uint32_t kDcEmpty = 0xFFFFFFFFu;
template <int BITS>
struct DeltaCacheT {
    static constexpr uint32_t kSlots = 1u << BITS;
    uint32_t tag[kSlots];
    uint32_t cnt[kSlots];
};
typedef DeltaCacheT<10> DeltaCache;
typedef DeltaCacheT<9> DeltaCacheSmall;      // (k_fused_batch: its lookup table and staging buffers leave 4 KB)

template <class DC>
__device__ __forceinline__ void dc_init(DC &dc) {
    for (uint32_t i = threadIdx.x; i < DC::kSlots; i += blockDim.x) { dc.tag[i] = kDcEmpty; dc.cnt[i] = 0; }
}

template <int BITS>
__device__ __forceinline__ void dc_add(DeltaCacheT<BITS> &dc, bool on, uint32_t *LR, uint32_t idx, uint32_t delta) {
    if (on) {
        const uint32_t slot = (idx * 0x9E3779B1u) >> (32 - BITS);
        uint32_t t = dc.tag[slot];
        if (t == kDcEmpty) {
            t = atomicCAS(&dc.tag[slot], kDcEmpty, idx);
            if (t == kDcEmpty) t = idx;
        }
        if (t == idx) { atomicAdd(&dc.cnt[slot], delta); return; }
    }
    atomicAdd(&LR[idx], delta);
}

// every thread of the workgroup, once all adds are done
template <class DC>
__device__ __forceinline__ void dc_flush(DC &dc, uint32_t *LR) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < DC::kSlots; i += blockDim.x) {
        const uint32_t c = dc.cnt[i];
        if (c) atomicAdd(&LR[dc.tag[i]], c);
    }
}

// is the largest pair of the pass frequent enough for hot count cells?  (one occurrence per
// 8192 live tokens; counts are global, so n_live is this shard's live tokens times the ranks)
__device__ __forceinline__ bool dc_wanted(unsigned long long top_count, unsigned long long n_live) {
    return top_count * 8192ull >= n_live;
}

// The stream kernels come in a plain and a HOT instantiation and each pass is done by the one the data ask for
// (dc_wanted).  The host does not even launch the HOT ones while it can prove them unnecessary (hot_launched 0): should
// the proof ever be wrong, the pass would silently not happen -- so the plain instantiation says so instead.
template <bool HOT>
__device__ __forceinline__ bool hot_mismatch(unsigned long long top_count, const DevCtl *ctl, int hot_launched) {
    const bool want = dc_wanted(top_count, ctl->n_live * ctl->n_ranks);
    if (want == HOT) return false;
    if (!HOT && !hot_launched && blockIdx.x == 0 && threadIdx.x == 0)
        atomicOr(const_cast<uint32_t *>(&ctl->err), kErrHotSkipped);
    return true;
}

__device__ __forceinline__ uint32_t hash_key(uint32_t k) {
    k *= 0x9E3779B1u;
    k ^= k >> 15;
    k *= 0x85EBCA77u;
    k ^= k >> 13;
    return k;
}

// create_or_modify_pair, PairCount.h:249-260: find the pair; add delta, or
// insert it with count = delta.  Within one kernel no two threads ever insert
// the same key (see k_apply / k_table_init), so a found key always has its
// entry index published by an earlier kernel.
// Dense layout: tiles of 32 first tokens x 32 second tokens (1024 cells = 4 KB = one block of the
// argmax bounds), tiles row-major.  32 consecutive second tokens of a row are contiguous, and the
// same column of 32 consecutive rows stays inside one 4 KB tile, so that the one scattered update
// of a batch -- (x, a_j) for all x -- reuses DRAM pages instead of opening one per update.
__device__ __forceinline__ uint32_t dense_index(const PairTable &t, uint32_t key) {
    const uint32_t f = key >> 16, g = key & 0xFFFFu;
    return ((((f >> 5) << (t.vshift - 5u)) | (g >> 5)) << 10) | ((f & 31u) << 5) | (g & 31u);
}
__device__ __forceinline__ uint32_t dense_key(const PairTable &t, uint32_t e) {
    const uint32_t tile = e >> 10, w = e & 1023u;
    const uint32_t f = ((tile >> (t.vshift - 5u)) << 5) | (w >> 5);
    const uint32_t g = ((tile & ((1u << (t.vshift - 5u)) - 1u)) << 5) | (w & 31u);
    return (f << 16) | g;
}

// entries the argmax has to look at
__device__ __forceinline__ uint32_t table_size(const PairTable &t, const DevCtl *ctl) {
    if (t.cells) return t.ecap;
    return ctl->n_entries < t.ecap ? ctl->n_entries : t.ecap;
}

// packed (count, ~key) of entry e, 0 for a cell whose pair was never inserted
__device__ __forceinline__ unsigned long long entry_packed(const PairTable &t, uint32_t e) {
    if (t.cells) {
        const uint32_t v = t.cells[e];
        return (v & kPresent) ? pack_best((int32_t)(v & ~kPresent), dense_key(t, e)) : 0ull;
    }
    const int32_t c = t.ecnt[e];
    return pack_best(c < 0 ? 0 : c, t.ekey[e]);
}

__device__ void table_add(const PairTable &t, DevCtl *ctl, uint32_t key, int32_t delta, bool may_insert) {
    if (t.cells) {
        const uint32_t e = dense_index(t, key);
        if (may_insert) {
            // a pair of a NEW token: by construction absent, and written by this thread only
            t.cells[e] = kPresent | (uint32_t)delta;
            atomicAdd(&ctl->n_entries, 1u);
            const unsigned long long p = pack_best(delta, key);
            if (p > t.bmax[e >> kBlockShift]) atomicMax(&t.bmax[e >> kBlockShift], p);
            if (p > t.smax[e >> (2 * kBlockShift)]) atomicMax(&t.smax[e >> (2 * kBlockShift)], p);
        } else {
            const uint32_t old = atomicAdd(&t.cells[e], (uint32_t)delta);
            if (!(old & kPresent)) atomicOr(&ctl->err, kErrMissingPair);
            else if ((int32_t)(old & ~kPresent) + delta < 0) atomicOr(&ctl->err, kErrNegCount);
        }
        return;
    }
    uint32_t h = hash_key(key) & t.hmask;
    for (uint32_t probe = 0; probe <= t.hmask; ++probe) {
        const unsigned long long slot = t.hslot[h];
        if ((uint32_t)(slot >> 32) == key) {
            int32_t old = atomicAdd(&t.ecnt[(uint32_t)slot], delta);
            if (old + delta < 0) atomicOr(&ctl->err, kErrNegCount);
            return;
        }
        if (slot == kEmptySlot) {
            if (!may_insert) { atomicOr(&ctl->err, kErrMissingPair); return; }
            // claim the slot with the key; the index follows (nobody looks this key up in this kernel)
            const unsigned long long claim = ((unsigned long long)key << 32) | 0xFFFFFFFFull;
            const unsigned long long prev = atomicCAS(&t.hslot[h], kEmptySlot, claim);
            if (prev == kEmptySlot) {
                uint32_t idx = atomicAdd(&ctl->n_entries, 1u);
                if (idx >= t.ecap) { atomicOr(&ctl->err, kErrTableFull); return; }
                reinterpret_cast<uint32_t *>(&t.hslot[h])[0] = idx;     // low word (little endian)
                t.ekey[idx] = key;
                t.ecnt[idx] = delta;
                // upper bounds for the hierarchical argmax (counts only fall after insertion);
                // entries are appended, so a block's bound is usually already higher: read first
                const unsigned long long p = pack_best(delta, key);
                if (p > t.bmax[idx >> kBlockShift]) atomicMax(&t.bmax[idx >> kBlockShift], p);
                if (p > t.smax[idx >> (2 * kBlockShift)]) atomicMax(&t.smax[idx >> (2 * kBlockShift)], p);
                return;
            }
            if ((uint32_t)(prev >> 32) == key) {  // cannot happen by construction; keep the table sane anyway
                atomicOr(&ctl->err, kErrMissingPair);
                return;
            }
        }
        h = (h + 1) & t.hmask;
    }
    atomicOr(&ctl->err, kErrTableFull);
}

// ---- fills ----------------------------------------------------------------

__global__ void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

__global__ void k_fill_u16(uint16_t *p, uint64_t n, uint16_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ---- pair-count scan ------------------------------------------------------
// One workgroup per CU keeps a private histogram of all 65,536 byte pairs in
// LDS as packed 16-bit counters (128 KiB of the CU's 160 KiB).  Each lane
// reads 16 corpus bytes per 16-byte load (the next iteration's loads are
// issued before the current ones are consumed) and issues 16 fire-and-forget
// LDS atomics per vector: nothing in the inner loop waits for the LDS.  The kernel
// is bound by the LDS atomics themselves: 64 random bins per wave instruction
// conflict about 3.5-way on the 32 banks a 32-lane group shares.
//
// bin = first | second << 8 (the little-endian 16-bit value at the pair's
// offset), then bin ^= bin >> 8 so that the LDS bank (low bits) mixes both
// bytes -- text uses few distinct first bytes; counter word = bin & 0x7FFF,
// half = bin >> 15.
//
// Exactness of the 16-bit counters, for any input.  A counter wraps after 65,536 increments,
// and a workgroup adds 32,768 per iteration, so in the worst case (one pair repeated) the
// histogram would have to be swept every iteration -- which is what costs a third of the
// time on data that never needs it.  Instead the workgroup runs SEGMENTS of iterations with
// no sweep and no barrier, and proves afterwards that nothing wrapped:
//   * every lane counts the increments it issued; the sum over the workgroup plus the
//     decoded sum of the counters before the segment is what the decoded sum must be now;
//   * a counter that wrapped lowers the decoded sum by 65,535 (low half: its carry lands in
//     the high half) or 65,536 (high half: the carry is lost), never raises it: the sums
//     agree  <=>  no counter wrapped  <=>  every counter is exact;
//   * if they disagree the segment is void: the histogram is restored from the snapshot taken
//     at the segment's start (128 KiB of plain stores to the workgroup's scratch) and the
//     segment is recounted by the slow loop, which sweeps every 49,152 increments and
//     moves counters >= 0x2000 to the global table, so that none can wrap.
// After a good segment counters >= 0x2000 are drained as well (the slow loop relies on it).
// Segment length adapts: it starts at kPcSeg0 iterations, grows (x8, x2) while the largest counter stays
// small and shrinks to a quarter (at least 8) after a void segment.  Uniform data never takes the slow loop; a
// corpus that is one repeated byte takes it for every segment.
constexpr int kPcThreads = 1024;
constexpr int kPcWords = 32768;            // 2 counters per word
constexpr uint32_t kPcHotBits = 0xE000u;   // counter >= 0x2000
constexpr int kPcEpochIters = 3;           // slow loop: iterations of one vector per lane between sweeps
constexpr int kPcVpl = 2;                  // fast loop: vectors per lane and iteration
#ifndef MBPE_PC_SEG0
#define MBPE_PC_SEG0 64
#endif
constexpr uint32_t kPcSeg0 = MBPE_PC_SEG0;  // first segment, in fast iterations of 32 Ki pairs
constexpr uint32_t kPcSegMax = 4096;
// one-chunk corpora take k_pair_count_u8_fast (0: k_pair_count_u8 for every corpus, A/B)
#ifndef MBPE_PC_FAST
#define MBPE_PC_FAST 1
#endif
// the final flush adds two bins per 64-bit atomic (0: one 32-bit atomic per bin, A/B)
#ifndef MBPE_PC_FLUSH64
#define MBPE_PC_FLUSH64 1
#endif

__device__ __forceinline__ uint32_t pc_table_index(uint32_t hbin) {
#ifndef MBPE_PC_NOHASH
    const uint32_t bin = hbin ^ (hbin >> 8);         // undo the bank hash
#else
    const uint32_t bin = hbin;
#endif
    return ((bin & 0xFFu) << 8) | (bin >> 8);        // -> (first << 8) | second
}

// sweep: thread t owns the 16-byte groups t, t+1024, ... of the histogram; counters >= 0x2000 move
// to the global table
__device__ __forceinline__ void pc_sweep(uint32_t *hist, uint32_t *bp) {
#pragma unroll
    for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k) {
        const uint32_t g = k * kPcThreads + threadIdx.x;
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

// final flush: lanes walk consecutive OUTPUT indices so the global atomics of
// a wave are contiguous (the LDS reads are bank-conflicted, but this runs once).
// Two bins per lane and ONE 64-bit atomic for both: the flush is 65,536 x 256 atomic lanes on one 256-KiB table, which the
// memory side takes at a fixed rate of wave instructions (0.115 ms of a 4 GiB scan's 0.98 ms did not scale with the
// corpus: most of it this) -- half as many instructions.  (No carry can run from the low bin into the high one: a pair
// count stays below 2^31, k_table_init checks.)
__device__ __forceinline__ void pc_flush(const uint32_t *hist, uint32_t *bp) {
#if MBPE_PC_FLUSH64
    unsigned long long *bp64 = reinterpret_cast<unsigned long long *>(bp);
    for (uint32_t o2 = threadIdx.x; o2 < 32768u; o2 += kPcThreads) {
        unsigned long long v = 0;
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
            const uint32_t o = 2u * o2 + h;
            uint32_t bin = ((o & 0xFFu) << 8) | (o >> 8);
#ifndef MBPE_PC_NOHASH
            bin ^= bin >> 8;
#endif
            const uint32_t c = (hist[bin & 0x7FFFu] >> ((bin >> 15) * 16)) & 0xFFFFu;
            v |= (unsigned long long)c << (32u * h);
        }
        if (v) atomicAdd(&bp64[o2], v);
    }
#else
    for (uint32_t o = threadIdx.x; o < 65536u; o += kPcThreads) {
        uint32_t bin = ((o & 0xFFu) << 8) | (o >> 8);
#ifndef MBPE_PC_NOHASH
        bin ^= bin >> 8;
#endif
        const uint32_t c = (hist[bin & 0x7FFFu] >> ((bin >> 15) * 16)) & 0xFFFFu;
        if (c) atomicAdd(&bp[o], c);
    }
#endif
}

// workgroup-wide: decoded sum of all counters + `issued` summed over the threads, and the largest counter
struct PcCheck { unsigned long long sum, issued; uint32_t max; };
__device__ __forceinline__ PcCheck pc_check(const uint32_t *hist, unsigned long long issued,
                                            unsigned long long *red /* [3 * 16] */) {
    unsigned long long s = 0;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k) {
        const uint4 v = reinterpret_cast<const uint4 *>(hist)[k * kPcThreads + threadIdx.x];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t lo = w[c] & 0xFFFFu, hi = w[c] >> 16;
            s += lo + hi;
            m = lo > m ? lo : m;
            m = hi > m ? hi : m;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += shfl_xor_u64(s, d);
        issued += shfl_xor_u64(issued, d);
        const uint32_t om = __shfl_xor(m, d, kWave);
        m = om > m ? om : m;
    }
    const uint32_t w = threadIdx.x / kWave;
    __syncthreads();                       // red may still be read from the previous call
    if (lane_id() == 0) { red[w] = s; red[16 + w] = issued; red[32 + w] = m; }
    __syncthreads();
    PcCheck r = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < kPcThreads / kWave; ++i) {
        r.sum += red[i];
        r.issued += red[16 + i];
        r.max = (uint32_t)red[32 + i] > r.max ? (uint32_t)red[32 + i] : r.max;
    }
    return r;
}

// The 4 pairs that start in dword w (wn = the following dword of the byte stream), 4.75 vector instructions per pair
// instead of 8: z = the stream shifted by one byte (v_alignbit), y = w ^ z (byte k of y = first ^ second of pair k),
// z7 = z & 0x7F7F7F7F; then per pair ONE v_perm_b32 builds the 15-bit word index ((second & 0x7F) << 8 | first ^
// second: exactly bin ^= bin >> 8; word = bin & 0x7FFF), the LDS address is that index scaled by the atomic's
// pointer arithmetic, and v_bfe + v_mad give the increment: 1, or 0x10000 when the second byte's top bit is set (=
// bin >> 15).  The two inline-asm instructions keep the compiler from turning the increment into compare + select
// (vcc hazards, s_nop: round 2's attempt).  A/B: tools/pc_variants.hip, profiles/r03_pair_count_ab.md.
__device__ __forceinline__ void pc_lean_pairs(uint32_t *hist, uint32_t w, uint32_t wn, uint32_t k0xffff) {
    const uint32_t z = __builtin_amdgcn_alignbit(wn, w, 8);
    const uint32_t y = w ^ z;
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

// Counts the pairs that start in the full 16-byte vectors [begin, end) of this workgroup
// (begin, end: multiples of the iteration size from the workgroup's first vector).  VPL vectors per
// lane and iteration; SWEEPS: sweep every kPcEpochIters iterations (VPL must be 1).  Returns the number
// of increments this lane issued.
template <bool MASKED, int VPL, bool SWEEPS>
__device__ __forceinline__ uint32_t pc_count_range(uint32_t *hist, uint32_t *__restrict__ bp,
                                                   const uint8_t *__restrict__ text,
                                                   const uint8_t *__restrict__ endmask, uint64_t n, uint64_t n_full,
                                                   uint64_t wg_end, uint64_t begin, uint64_t end) {
    static_assert(!SWEEPS || VPL == 1, "the slow loop sweeps every 3 x 16,384 increments");
    constexpr uint64_t kIterVecs = (uint64_t)kPcThreads * VPL;
    const uint32_t lane = lane_id();
    const uint64_t last_vec = n_full ? n_full - 1 : 0;
    // software pipeline: q/e/xb hold the vectors of the NEXT iteration.  All
    // loads are unconditional (clamped addresses): a branch around a load or
    // an LDS atomic makes hipcc serialise them with full waits.
    uint4 q[VPL];
    uint32_t e[VPL], xb[VPL];
    auto issue = [&](uint64_t base) {
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            uint64_t vec = base + (uint64_t)u * kPcThreads + threadIdx.x;
            vec = vec < last_vec ? vec : last_vec;
            const uint64_t byte0 = vec * 16;
            q[u] = *reinterpret_cast<const uint4 *>(text + byte0);
            e[u] = MASKED ? reinterpret_cast<const uint16_t *>(endmask)[vec] : 0u;
            const uint64_t nx = byte0 + 16 < n ? byte0 + 16 : n - 1;
            xb[u] = text[lane == kWave - 1 ? nx : byte0];     // only lane 63 uses it
        }
    };
    uint32_t issued = 0;
    if (begin < end) issue(begin);
    int epoch_iter = 0;
    for (uint64_t base = begin; base < end; base += kIterVecs) {
        uint4 cq[VPL];
        uint32_t ce[VPL], cxb[VPL];
#pragma unroll
        for (int u = 0; u < VPL; ++u) { cq[u] = q[u]; ce[u] = e[u]; cxb[u] = xb[u]; }
        issue(base + kIterVecs < end ? base + kIterVecs : base);
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const uint64_t vec = base + (uint64_t)u * kPcThreads + threadIdx.x;
            // first byte of the next lane's vector = second byte of my last pair
            uint32_t nb = __shfl_down(cq[u].x, 1, kWave) & 0xFFu;
            if (lane == kWave - 1) nb = cxb[u];
            // pairs that start in this vector: 16, or 15 for the last full vector (its
            // straddling pair belongs to the tail thread of the kernel)
            uint32_t valid = vec < wg_end ? (vec + 1 < n_full ? 0xFFFFu : 0x7FFFu) : 0u;
            if (MASKED) valid &= ~ce[u];
            issued += __popc(valid);
            const uint32_t w[5] = {cq[u].x, cq[u].y, cq[u].z, cq[u].w, nb};
            if (__ballot(valid != 0xFFFFu) == 0ull) {
                // every pair of every lane counts: no per-pair predicate
#if !defined(MBPE_PC_NOHASH) && !defined(MBPE_PC_PLAIN)
                uint32_t k0xffff;
                asm volatile("s_mov_b32 %0, 0xffff" : "=s"(k0xffff));
#pragma unroll
                for (int i = 0; i < 4; ++i) pc_lean_pairs(hist, w[i], w[i + 1], k0xffff);
#else
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int wi = i >> 2, sh = 8 * (i & 3);
                    uint32_t bin;
                    if (sh <= 16) bin = (w[wi] >> sh) & 0xFFFFu;
                    else bin = ((w[wi] >> 24) | (w[wi + 1] << 8)) & 0xFFFFu;
#ifndef MBPE_PC_NOHASH
                    bin ^= bin >> 8;
#endif
                    atomicAdd(&hist[bin & 0x7FFFu], 1u + (bin >> 15) * 0xFFFFu);
                }
#endif
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int wi = i >> 2, sh = 8 * (i & 3);
                    uint32_t bin;
                    if (sh <= 16) bin = (w[wi] >> sh) & 0xFFFFu;
                    else bin = ((w[wi] >> 24) | (w[wi + 1] << 8)) & 0xFFFFu;
#ifndef MBPE_PC_NOHASH
                    bin ^= bin >> 8;
#endif
                    const uint32_t inc = ((valid >> i) & 1u) ? 1u + (bin >> 15) * 0xFFFFu : 0u;   // invalid: add 0
                    atomicAdd(&hist[bin & 0x7FFFu], inc);
                }
            }
        }
        if (SWEEPS && ++epoch_iter == kPcEpochIters) {
            epoch_iter = 0;
            __syncthreads();
            pc_sweep(hist, bp);
            __syncthreads();
        }
    }
    return issued;
}

template <bool MASKED>
__global__ __launch_bounds__(kPcThreads) void k_pair_count_u8(const uint8_t *__restrict__ text, uint64_t n,
                                                              const uint8_t *__restrict__ endmask,
                                                              uint32_t *__restrict__ bp,
                                                              uint32_t *__restrict__ snap /* [gridDim.x][kPcWords] */) {
    __shared__ uint32_t hist[kPcWords];
    __shared__ unsigned long long red[48];
    for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += kPcThreads) hist[w] = 0;
    __syncthreads();

    // The loops handle the FULL 16-byte vectors; the ragged tail (< 16
    // bytes, plus the pair that straddles into it) is left to one thread.
    constexpr uint64_t kIterVecs = (uint64_t)kPcThreads * kPcVpl;
    const uint64_t n_full = n / 16;
    uint64_t per = (n_full + gridDim.x - 1) / gridDim.x;
    per = (per + kIterVecs - 1) / kIterVecs * kIterVecs;
    const uint64_t v_begin = per * blockIdx.x;
    uint64_t v_end = v_begin + per;
    if (v_end > n_full) v_end = n_full;
    uint4 *my_snap = reinterpret_cast<uint4 *>(snap + (size_t)blockIdx.x * kPcWords);

    unsigned long long resid = 0;          // decoded sum of the counters (uniform)
    uint32_t seg_iters = kPcSeg0;
    for (uint64_t seg = v_begin; seg < v_end;) {
        // (segment ends stay on iteration boundaries; the last one is cut at v_end)
        const uint64_t seg_end = seg + seg_iters * kIterVecs < v_end ? seg + seg_iters * kIterVecs : v_end;
        if (resid) {                       // something to lose: snapshot
#pragma unroll
            for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k)
                my_snap[k * kPcThreads + threadIdx.x] = reinterpret_cast<const uint4 *>(hist)[k * kPcThreads + threadIdx.x];
        }
        const uint32_t issued = pc_count_range<MASKED, kPcVpl, false>(hist, bp, text, endmask, n, n_full, v_end, seg, seg_end);
        __syncthreads();
        PcCheck ck = pc_check(hist, issued, red);
        if (ck.sum == resid + ck.issued) {
            // exact.  Keep every counter below 0x2000 between segments
            if (ck.max >= 0x2000u) {
                pc_sweep(hist, bp);
                __syncthreads();
                ck = pc_check(hist, 0, red);
            }
            resid = ck.sum;
            // (the next segment may be as long as keeps the largest counter, growing at this rate, below half)
            if (ck.max * 8u < 0x8000u && seg_iters * 8 <= kPcSegMax) seg_iters *= 8;
            else if (ck.max * 2u < 0x8000u && seg_iters * 2 <= kPcSegMax) seg_iters *= 2;
            else if (ck.max >= 0x8000u && seg_iters > 4) seg_iters /= 2;
        } else {
            // a counter wrapped: the segment is void.  Restore and recount it with sweeps
#pragma unroll
            for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k)
                reinterpret_cast<uint4 *>(hist)[k * kPcThreads + threadIdx.x] =
                    resid ? my_snap[k * kPcThreads + threadIdx.x] : make_uint4(0, 0, 0, 0);
            __syncthreads();
            pc_count_range<MASKED, 1, true>(hist, bp, text, endmask, n, n_full, v_end, seg, seg_end);
            __syncthreads();
            pc_sweep(hist, bp);
            __syncthreads();
            resid = pc_check(hist, 0, red).sum;
            seg_iters = seg_iters >= 32 ? seg_iters / 4 : 8;      // shorter segments while the data stay this skewed
        }
        seg = seg_end;
    }
    __syncthreads();
    pc_flush(hist, bp);
    // ragged tail: pairs starting at byte 16*n_full - 1 .. n-2
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint64_t i = n_full * 16;
        if (i > 0) --i;                      // the pair straddling into the tail
        if (n_full * 16 == n) i = n;         // no tail at all
        for (; i + 1 < n; ++i) {
            if (MASKED && ((endmask[i >> 3] >> (i & 7)) & 1u)) continue;
            atomicAdd(&bp[((uint32_t)text[i] << 8) | text[i + 1]], 1u);
        }
    }
}

// ---- pair-count scan, one-chunk corpora: the same histogram, a leaner loop around it (round 4) ----------------------
// The LDS alone needs 0.81 ms for 2^32 random ds_add_u32 from 1024-thread workgroups (tools/lds_atomic_floor.hip: 7.1
// LDS cycles per wave instruction against 4.2 without bank conflicts), and the 4.75 vector instructions per pair of
// pc_lean_pairs on register data 0.89 ms.  k_pair_count_u8 ran at 1.05 ms because of what surrounds them: per vector a
// 64-bit clamped address, a byte load for lane 63's straddling pair, a ds_bpermute for the other lanes' (an LDS
// instruction itself) and the bookkeeping of which pairs count -- 6.7 vector instructions per pair all told.  Here
//   * the workgroup walks its range in iterations of 16 consecutive 2-KiB blocks, one per WAVE, lane l holding bytes
//     [32 l, 32 l + 32) of its wave's block (two 16-byte loads): 31 of its 32 pairs lie inside the lane, the 32nd needs the
//     next lane's first byte -- one DPP move, no LDS -- and lane 63's comes from a load of the dword behind the block;
//   * addresses are a wave-uniform base in scalar registers plus a constant lane offset;
//   * the loop only ever sees whole blocks that lie strictly inside the corpus, so every pair counts and no lane keeps a
//     tally: a segment of i iterations issues i x 32,768 increments per workgroup.  The blocks that are left over at the
//     end of the workgroup's range (less than one iteration of the workgroup, plus the corpus tail) go through
//     pc_count_range as before.
// Segments, checksum, snapshot / rollback and the slow recount are those of k_pair_count_u8.
constexpr uint32_t kPcBlockBytes = kWave * 32u;                  // one wave, one iteration
constexpr uint32_t kPcBlockVecs = kPcBlockBytes / 16u;
constexpr int kPcWaves = kPcThreads / kWave;

struct PcBlock { u32x4 a, b; uint32_t edge; };
#ifndef MBPE_PC_DEPTH
#define MBPE_PC_DEPTH 1
#endif
constexpr int kPcDepth = MBPE_PC_DEPTH;

__device__ __forceinline__ PcBlock pc_block_issue(const MBPE_GLOBAL_AS char *stripe /* wave-uniform */, uint32_t it, uint32_t voff) {
    PcBlock r;
    // (iteration `it` of the workgroup is 16 consecutive blocks, one per wave: the workgroup walks its range front to
    //  back.  A contiguous stripe per wave -- 4,096 streams a mebibyte apart chip-wide -- started every launch with
    //  ~40 us of page walks: its first 64 iterations took 2.25 us each instead of 1.7)
    const MBPE_GLOBAL_AS char *p = stripe + (uint64_t)it * (kPcBlockBytes * (uint32_t)kPcWaves);
    r.a = *(const MBPE_GLOBAL_AS u32x4 *)(p + voff);
    r.b = *(const MBPE_GLOBAL_AS u32x4 *)(p + voff + 16u);
    // (uniform address, constant address space: a scalar load)
    r.edge = *(const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(p + kPcBlockBytes);
    return r;
}

// iterations [it0, it1) of the workgroup, this wave's block of each; every pair that starts in them counts
__device__ __attribute__((noinline)) void pc_fast_range(uint32_t *hist, const uint8_t *text, uint64_t stripe_byte0, uint32_t it0, uint32_t it1) {
    if (it0 >= it1) return;
    const MBPE_GLOBAL_AS char *stripe = (const MBPE_GLOBAL_AS char *)uniform_ptr(reinterpret_cast<uintptr_t>(text) + stripe_byte0);
    const uint32_t voff = lane_id() * 32u;
    uint32_t k0xffff;
    asm volatile("s_mov_b32 %0, 0xffff" : "=s"(k0xffff));
    // kPcDepth blocks are in flight behind the one being counted (9 registers each)
    const uint32_t last = it1 - 1u;
    PcBlock ring[kPcDepth];
#pragma unroll
    for (int d = 0; d < kPcDepth; ++d) ring[d] = pc_block_issue(stripe, it0 + d < last ? it0 + d : last, voff);
    for (uint32_t it = it0; it < it1; ++it) {
        const PcBlock cur = ring[0];
#pragma unroll
        for (int d = 0; d + 1 < kPcDepth; ++d) ring[d] = ring[d + 1];
        ring[kPcDepth - 1] = pc_block_issue(stripe, it + kPcDepth < last ? it + kPcDepth : last, voff);   // (unconditional: see pc_count_range)
        const uint32_t w[9] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cur.b.z, cur.b.w,
                               wave_from_next(cur.a.x, cur.edge)};
#pragma unroll
        for (int i = 0; i < 8; ++i) pc_lean_pairs(hist, w[i], w[i + 1], k0xffff);
    }
}

// (out of line: inlined, the generic loops' registers spill the fast loop's)
__device__ __attribute__((noinline)) void pc_slow_range(uint32_t *hist, uint32_t *bp, const uint8_t *text, uint64_t n,
                                                        uint64_t n_full, uint64_t begin, uint64_t end) {
    pc_count_range<false, 1, true>(hist, bp, text, nullptr, n, n_full, end, begin, end);
    __syncthreads();
    pc_sweep(hist, bp);
    __syncthreads();
}
__device__ __attribute__((noinline)) uint32_t pc_generic_range(uint32_t *hist, uint32_t *bp, const uint8_t *text, uint64_t n,
                                                               uint64_t n_full, uint64_t begin, uint64_t end) {
    return pc_count_range<false, kPcVpl, false>(hist, bp, text, nullptr, n, n_full, end, begin, end);
}

__global__ __launch_bounds__(kPcThreads) void k_pair_count_u8_fast(const uint8_t *__restrict__ text, uint64_t n,
                                                                   uint32_t *__restrict__ bp,
                                                                   uint32_t *__restrict__ snap /* [gridDim.x][kPcWords] */) {
    __shared__ uint32_t hist[kPcWords];
    __shared__ unsigned long long red[48];
#ifdef MBPE_PC_STAMPS
    // diagnostic build: wall-clock stamps (100 MHz) of this workgroup's phases behind the snapshots: 16 x u64 per workgroup
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(snap + (size_t)gridDim.x * kPcWords) + 16 * blockIdx.x;
    uint32_t n_stamp = 0;
#define PC_STAMP() do { if (threadIdx.x == 0 && n_stamp < 16) stamps[n_stamp++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PC_STAMP() do {} while (0)
#endif
    PC_STAMP();
    for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += kPcThreads) hist[w] = 0;
    __syncthreads();

    // the workgroup's range of full 16-byte vectors, as in k_pair_count_u8
    constexpr uint64_t kIterVecs = (uint64_t)kPcThreads * kPcVpl;
    static_assert(kIterVecs == (uint64_t)kPcWaves * kPcBlockVecs, "one fast iteration of the workgroup = one generic iteration");
    const uint64_t n_full = n / 16;
    uint64_t per = (n_full + gridDim.x - 1) / gridDim.x;
    per = (per + kIterVecs - 1) / kIterVecs * kIterVecs;
    const uint64_t v_begin = per * blockIdx.x;
    uint64_t v_end = v_begin + per;
    if (v_end > n_full) v_end = n_full;
    uint4 *my_snap = reinterpret_cast<uint4 *>(snap + (size_t)blockIdx.x * kPcWords);

    // fast iterations: whole iterations of the workgroup whose blocks -- and the 4 bytes behind each -- lie inside the
    // corpus, so that every one of their pairs has a second byte
    uint64_t fast = v_end > v_begin ? (v_end - v_begin) / kIterVecs : 0;
    while (fast > 0 && (v_begin + fast * kIterVecs) * 16 + 4 > n) --fast;
    const uint32_t F = (uint32_t)fast;
    const uint32_t wave = threadIdx.x / kWave;
    const uint64_t stripe_vec0 = v_begin + (uint64_t)wave * kPcBlockVecs;     // this wave's block of iteration 0

    unsigned long long resid = 0;          // decoded sum of the counters (uniform)
    uint32_t seg_iters = kPcSeg0;
    // The pair that straddles from the last full vector into the ragged tail, (16 n_full - 1, 16 n_full): a fast iteration
    // counts every pair that starts in its blocks -- this one too, when the workgroup's fast iterations end at n_full and
    // the segment that holds the last of them stood; the generic loops (the remainder below, the recount of a void
    // segment) count the pairs INSIDE the full vectors and leave it to the tail code at the end of the kernel.
    bool straddle_counted = false;         // uniform
    for (uint32_t seg = 0; seg < F;) {
        const uint32_t seg_end = seg + seg_iters < F ? seg + seg_iters : F;
        if (resid) {                       // something to lose: snapshot
#pragma unroll
            for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k)
                my_snap[k * kPcThreads + threadIdx.x] = reinterpret_cast<const uint4 *>(hist)[k * kPcThreads + threadIdx.x];
        }
        PC_STAMP();
        pc_fast_range(hist, text, stripe_vec0 * 16, seg, seg_end);
        __syncthreads();
        PC_STAMP();
        PcCheck ck = pc_check(hist, (unsigned long long)(seg_end - seg) * 32ull, red);
        if (ck.sum == resid + ck.issued) {
            if (ck.max >= 0x2000u) {
                pc_sweep(hist, bp);
                __syncthreads();
                ck = pc_check(hist, 0, red);
            }
            resid = ck.sum;
            if (ck.max * 8u < 0x8000u && seg_iters * 8 <= kPcSegMax) seg_iters *= 8;
            else if (ck.max * 2u < 0x8000u && seg_iters * 2 <= kPcSegMax) seg_iters *= 2;
            else if (ck.max >= 0x8000u && seg_iters > 4) seg_iters /= 2;
            if (seg_end == F && v_begin + (uint64_t)F * kIterVecs == n_full) straddle_counted = true;
        } else {
            // a counter wrapped: the segment is void.  Restore, and recount it with sweeps
#pragma unroll
            for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k)
                reinterpret_cast<uint4 *>(hist)[k * kPcThreads + threadIdx.x] =
                    resid ? my_snap[k * kPcThreads + threadIdx.x] : make_uint4(0, 0, 0, 0);
            __syncthreads();
            pc_slow_range(hist, bp, text, n, n_full, v_begin + (uint64_t)seg * kIterVecs, v_begin + (uint64_t)seg_end * kIterVecs);
            resid = pc_check(hist, 0, red).sum;
            seg_iters = seg_iters >= 32 ? seg_iters / 4 : 8;
        }
        seg = seg_end;
    }
    // what the fast iterations left of the workgroup's range: the generic loop, as one more segment
    {
        const uint64_t r_begin = v_begin + (uint64_t)F * kIterVecs;
        if (r_begin < v_end) {
            if (resid) {
#pragma unroll
                for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k)
                    my_snap[k * kPcThreads + threadIdx.x] = reinterpret_cast<const uint4 *>(hist)[k * kPcThreads + threadIdx.x];
            }
            const uint32_t issued = pc_generic_range(hist, bp, text, n, n_full, r_begin, v_end);
            __syncthreads();
            const PcCheck ck = pc_check(hist, issued, red);
            if (ck.sum != resid + ck.issued) {
#pragma unroll
                for (int k = 0; k < kPcWords / 4 / kPcThreads; ++k)
                    reinterpret_cast<uint4 *>(hist)[k * kPcThreads + threadIdx.x] =
                        resid ? my_snap[k * kPcThreads + threadIdx.x] : make_uint4(0, 0, 0, 0);
                __syncthreads();
                pc_slow_range(hist, bp, text, n, n_full, r_begin, v_end);
            }
        }
    }
    __syncthreads();
    PC_STAMP();
    pc_flush(hist, bp);
    // ragged tail: pairs starting at byte 16*n_full - 1 .. n-2, by the workgroup whose range holds the last full vector
    // (it knows whether its fast iterations counted the straddling pair; no full vector at all: workgroup 0)
    const bool ends_here = n_full == 0 ? blockIdx.x == 0 : (v_begin < n_full && v_end == n_full);
    if (ends_here && threadIdx.x == 0) {
        uint64_t i = n_full * 16;
        if (i > 0 && !straddle_counted) --i;
        if (n_full * 16 == n) i = n;         // no tail at all
        for (; i + 1 < n; ++i) atomicAdd(&bp[((uint32_t)text[i] << 8) | text[i + 1]], 1u);
    }
#ifdef MBPE_PC_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    PC_STAMP();
    if (threadIdx.x == 0) for (; n_stamp < 16; ++n_stamp) stamps[n_stamp] = 0;
#endif
}

// ---- widen: byte corpus -> 16-bit slot stream ------------------------------

template <bool MASKED>
__global__ void k_widen(const uint8_t *__restrict__ text, uint64_t n, const uint8_t *__restrict__ endmask,
                        uint16_t *__restrict__ tok, uint64_t n_padded) {
    uint64_t vec = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_vec = n_padded / 16;
    for (; vec < n_vec; vec += stride) {
        const uint64_t byte0 = vec * 16;
        uint32_t w[4] = {0, 0, 0, 0};
        uint32_t ends = 0;
        if (byte0 + 16 <= n) {
            uint4 q = *reinterpret_cast<const uint4 *>(text + byte0);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        } else {
            for (uint64_t i = byte0; i < n; ++i)
                w[(i - byte0) >> 2] |= (uint32_t)text[i] << (8 * ((i - byte0) & 3));
        }
        if (MASKED && byte0 < n) ends = reinterpret_cast<const uint16_t *>(endmask)[vec];
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            uint32_t t0 = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            uint32_t t1 = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xFFu;
            if (MASKED) {
                t0 |= ((ends >> i) & 1u) << 15;
                t1 |= ((ends >> (i + 1)) & 1u) << 15;
            }
            if (byte0 + i >= n) t0 = kHole;
            if (byte0 + i + 1 >= n) t1 = kHole;
            o[i >> 1] = t0 | (t1 << 16);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(tok + byte0);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// Barrier layout (chunked corpora whose ids need all 16 bits): byte i goes to slot 2i, slot 2i+1 holds the
// barrier when byte i ends its chunk and a hole otherwise -- 256 bytes per tile, whatever the chunk lengths.
// The host squeezes the holes out right afterwards (the ordinary compaction).
__global__ void k_widen_barrier(const uint8_t *__restrict__ text, uint64_t n, const uint8_t *__restrict__ endmask,
                                uint16_t *__restrict__ tok, uint64_t n_slots) {
    uint64_t vec = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;          // 8 bytes -> 16 slots
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_vec = n_slots / 16;
    for (; vec < n_vec; vec += stride) {
        const uint64_t byte0 = vec * 8;
        uint32_t w[2] = {0, 0};
        uint32_t ends = 0;
        if (byte0 + 8 <= n) {
            const uint2 q = *reinterpret_cast<const uint2 *>(text + byte0);
            w[0] = q.x; w[1] = q.y;
        } else {
            for (uint64_t i = byte0; i < n; ++i) w[(i - byte0) >> 2] |= (uint32_t)text[i] << (8 * ((i - byte0) & 3));
        }
        if (byte0 < n) ends = endmask[vec];
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            uint32_t e = ((ends >> i) & 1u) ? kBarrier : kHole;
            if (byte0 + i >= n) { t = kHole; e = kHole; }
            o[i] = t | (e << 16);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(tok + vec * 16);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// ---- wave-level tile helpers ---------------------------------------------------

// How the stream kernels know where a chunk ends (template parameter MODE):
//   0  one chunk: every two neighbouring live tokens form a pair
//   1  bit 15 of the chunk's last token (ids below 0x7FFF)
//   2  a BARRIER slot after the chunk's last token (kBarrier, see mbpe_dev.h): a live token that belongs to no
//      pair, so ids use the full 16 bits.  It is never a member of a batch pair, so no match can contain or
//      span it; the only places that have to know are the ones that COUNT a neighbour pair.
// left_open(p1): a pair (p1, self) exists for the live token `self`; right_open(self, n1): a pair (self, n1) exists.
template <int MODE>
__device__ __forceinline__ bool left_open(uint32_t p1) {
    return p1 != kHole && (MODE == 1 ? !(p1 & kEndBit) : MODE == 2 ? p1 != kBarrier : true);
}
template <int MODE>
__device__ __forceinline__ bool right_open(uint32_t self, uint32_t n1) {
    return n1 != kHole && (MODE == 1 ? !(self & kEndBit) : MODE == 2 ? n1 != kBarrier : true);
}

__device__ __forceinline__ void unpack8(const uint4 &q, uint32_t s[8]) {
    s[0] = q.x & 0xFFFFu; s[1] = q.x >> 16;
    s[2] = q.y & 0xFFFFu; s[3] = q.y >> 16;
    s[4] = q.z & 0xFFFFu; s[5] = q.z >> 16;
    s[6] = q.w & 0xFFFFu; s[7] = q.w >> 16;
}

__device__ __forceinline__ uint4 pack8(const uint32_t s[8]) {
    return make_uint4(s[0] | (s[1] << 16), s[2] | (s[3] << 16), s[4] | (s[5] << 16), s[6] | (s[7] << 16));
}

// Every kernel that rewrites a tile leaves the tile's live tokens in its FIRST slots, in order, and the holes behind them
// ("prefix form": k_widen and the compaction produce it, k_merge / k_rewrite_marked / k_fused_batch keep it), so the
// token after a live slot is simply the next slot -- no chains over holes in the fused pass (fused_tile_pf).  (A
// stream with barrier slots is compacted before its first pass: mbpe_train_begin.)
// tile_compact: v[8] of all 64 lanes -> the lane's 8 slots of the compacted tile, packed; stage = this wave's 512
// 16-bit slots of LDS (a wave's LDS operations execute in order: no barrier between its writes and its reads);
// n_live (uniform) = live tokens of the tile.
constexpr uint32_t kTileSlots = kWave * 8;
__device__ __forceinline__ uint4 tile_compact(const uint32_t v[8], uint16_t *stage, uint32_t &n_live) {
    const uint32_t lane = lane_id();
    uint32_t keep = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) keep |= (v[j] != kHole ? 1u : 0u) << j;
    const uint32_t cnt = (uint32_t)__popc(keep);         // 0 .. 8: the prefix sum over the lanes bit by bit, with ballots
    uint32_t pos = 0, total = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const unsigned long long m = __ballot(((cnt >> b) & 1u) != 0u);
        pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
        total += (uint32_t)__popcll(m) << b;
    }
    n_live = total;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if ((keep >> j) & 1u) stage[pos] = (uint16_t)v[j];
        pos += (keep >> j) & 1u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4 q = *reinterpret_cast<const uint4 *>(stage + lane * 8u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int rem = (int)total - (int)(lane * 8u);       // live slots of this lane from here on (<= 0: none, >= 8: all)
    q.x = rem >= 2 ? q.x : rem == 1 ? (q.x | 0xFFFF0000u) : 0xFFFFFFFFu;
    q.y = rem >= 4 ? q.y : rem == 3 ? (q.y | 0xFFFF0000u) : 0xFFFFFFFFu;
    q.z = rem >= 6 ? q.z : rem == 5 ? (q.z | 0xFFFF0000u) : 0xFFFFFFFFu;
    q.w = rem >= 8 ? q.w : rem == 7 ? (q.w | 0xFFFF0000u) : 0xFFFFFFFFu;
    return q;
}

__device__ __forceinline__ uint4 sum_to_u4(uint32_t head0, uint32_t head1, uint32_t tail1, uint32_t tail0,
                                           uint32_t n_live, uint32_t tail_run) {
    return make_uint4(head0 | (head1 << 16), tail1 | (tail0 << 16), n_live | (tail_run << 16), 0u);
}

// The summary of one tile from its (new) slot values, computed by the whole
// wave; the result is uniform.  v[j] == kHole marks a dead slot.
__device__ __forceinline__ uint4 wave_summary(const uint32_t v[8]) {
    uint32_t cnt = 0, f1 = kSent, f2 = kSent, l1 = kSent, l2 = kSent;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (v[j] != kHole) {
            if (cnt == 0) f1 = v[j]; else if (cnt == 1) f2 = v[j];
            l2 = l1; l1 = v[j];
            ++cnt;
        }
    }
    const unsigned long long mh = __ballot(cnt > 0);
    if (!mh) return sum_to_u4(kHole, kHole, kHole, kHole, 0, 0);
    const uint32_t lowlane = __builtin_ctzll(mh), highlane = 63 - __builtin_clzll(mh);
    uint32_t head0 = rlane(f1, lowlane), head1 = rlane(f2, lowlane);
    if (head1 == kSent) {
        const unsigned long long m2 = mh & ~(1ull << lowlane);
        head1 = m2 ? rlane(f1, __builtin_ctzll(m2)) : kHole;
    }
    uint32_t tail0 = rlane(l1, highlane), tail1 = rlane(l2, highlane);
    if (tail1 == kSent) {
        const unsigned long long m2 = mh & ~(1ull << highlane);
        tail1 = m2 ? rlane(l1, 63 - __builtin_clzll(m2)) : kHole;
    }
    const uint32_t n_live = wave_sum_le8(cnt, ~0ull);
    // trailing run of tokens equal (raw) to tail0
    uint32_t trail = 0;
    bool stop = false;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        if (v[j] != kHole) {
            if (!stop && v[j] == tail0) ++trail; else stop = true;
        }
    }
    const unsigned long long brk = __ballot(cnt > 0 && stop);
    uint32_t run;
    if (!brk) {
        run = n_live;
    } else {
        const uint32_t B = 63 - __builtin_clzll(brk);
        const unsigned long long above = B == 63 ? 0ull : ~((2ull << B) - 1ull);
        run = rlane(trail, B) + wave_sum_le8(cnt, above);
    }
    return sum_to_u4(head0, head1, tail1, tail0, n_live, run);
}

struct Halo { uint32_t p2, p1, n1, n2; };

// Neighbour tokens of a tile when the adjacent summaries cannot answer
// directly (first / last tile, or a neighbour with fewer than two live
// tokens): walk the summaries, then the rank edges of a multi-GPU run.
// Every read is a whole-dword load from a uniform address through a constant-
// address-space pointer (nothing writes the summaries or the edges while a pass
// runs), which the compiler turns into scalar loads: a vector load here would make
// the callers wait for vmcnt(0), i.e. for the tiles they have prefetched.
#define MBPE_CONST_AS __attribute__((address_space(4)))
__device__ __forceinline__ Halo halo_slow(const TileSum *__restrict__ sin, uint32_t n_tiles, uint32_t tile,
                                          const RankEdge *__restrict__ le, const RankEdge *__restrict__ re) {
    Halo h;
    h.p1 = h.p2 = h.n1 = h.n2 = kHole;
    typedef unsigned int su4 __attribute__((ext_vector_type(4)));
    typedef unsigned int su2 __attribute__((ext_vector_type(2)));
    const MBPE_CONST_AS su4 *sq = (const MBPE_CONST_AS su4 *)(uintptr_t)sin;
    const MBPE_CONST_AS su2 *lq = (const MBPE_CONST_AS su2 *)(uintptr_t)le;
    const MBPE_CONST_AS su2 *rq = (const MBPE_CONST_AS su2 *)(uintptr_t)re;
    int need = 2;
    for (int64_t j = (int64_t)tile - 1; need && j >= -1; --j) {
        uint32_t nl, t0, t1;
        if (j >= 0) { const su4 q = sq[j]; nl = q.z & 0xFFFFu; t0 = q.y >> 16; t1 = q.y & 0xFFFFu; }
        else if (le) {
            const su2 q = lq[1];
            t1 = q.x; t0 = q.y; nl = t0 == kHole ? 0 : (t1 == kHole ? 1 : 2);
        }
        else break;
        if (nl == 0) continue;
        if (need == 2) { h.p1 = t0; need = 1; if (nl >= 2) { h.p2 = t1; need = 0; } }
        else { h.p2 = t0; need = 0; }
    }
    need = 2;
    for (int64_t j = (int64_t)tile + 1; need && j <= (int64_t)n_tiles; ++j) {
        uint32_t nl, t0, t1;
        if (j < (int64_t)n_tiles) { const su4 q = sq[j]; nl = q.z & 0xFFFFu; t0 = q.x & 0xFFFFu; t1 = q.x >> 16; }
        else if (re) {
            const su2 q = rq[0];
            t0 = q.x; t1 = q.y; nl = t0 == kHole ? 0 : (t1 == kHole ? 1 : 2);
        }
        else break;
        if (nl == 0) continue;
        if (need == 2) { h.n1 = t0; need = 1; if (nl >= 2) { h.n2 = t1; need = 0; } }
        else { h.n2 = t0; need = 0; }
    }
    return h;
}

// ---- (t,t) merges: how many tokens equal to raw t lie immediately before every tile --------------
// A merge of (t,t) takes every second token of a run of t, so a tile that starts inside a run has to
// know the run's length before it (its parity, and whether it is >= 2).  Walking the summaries back
// from every tile is quadratic in a corpus that is one long run; instead a segmented scan over the
// tiles (chunks of 4096, like the compaction scan) computes it for all tiles in two small kernels,
// which run only when the pair to merge is a (t,t).
//   element of a tile: (len = its trailing run of t, full = the whole tile is t, or empty)
//   L then R:  R.full ? (L.len + R.len, L.full) : R
// run_in[tile] = (run & 1) | (run >= 2 ? 2 : 0), with the left rank's trailing run added when
// everything before the tile is t.
constexpr int kRunThreads = 256;
constexpr int kRunPerThread = 16;
constexpr uint32_t kRunChunk = kRunThreads * kRunPerThread;      // 4096 tiles, as tile_scan_scratch assumes

// (token-agnostic: the run is of whatever token the previous tile ends with, so one scan serves every
//  (t,t) member of a batch; a consumer uses it only if its own token equals that neighbour)
struct RunSeg { unsigned long long len; uint32_t token; bool full, empty; };
__device__ __forceinline__ RunSeg run_none() { RunSeg o; o.len = 0; o.token = 0; o.full = true; o.empty = true; return o; }
__device__ __forceinline__ RunSeg run_join(RunSeg l, RunSeg r) {
    if (r.empty) return l;
    if (l.empty) return r;
    RunSeg o = r;
    if (r.full && l.token == r.token) { o.len = l.len + r.len; o.full = l.full; }
    else o.full = false;
    return o;
}
__device__ __forceinline__ RunSeg run_seg_of(const TileSum &s) {
    RunSeg o;
    if (s.n_live == 0) return run_none();
    o.empty = false;
    o.token = s.tail0;                      // raw: a token that ends its chunk equals nothing after it
    o.len = s.tail_run;
    o.full = s.tail_run == s.n_live;
    return o;
}
__device__ __forceinline__ unsigned long long run_pack(RunSeg r) {
    return ((unsigned long long)r.token << 48) | (r.len << 2) | (r.full ? 2ull : 0ull) | (r.empty ? 1ull : 0ull);
}
__device__ __forceinline__ RunSeg run_unpack(unsigned long long v) {
    RunSeg r; r.token = (uint32_t)(v >> 48); r.len = (v >> 2) & ((1ull << 46) - 1ull); r.full = (v & 2ull) != 0; r.empty = (v & 1ull) != 0;
    return r;
}

// the token of the (t,t) pair about to be merged (alone, or as a member of a batch), or 0xFFFFFFFF when
// this sequence merges no such pair
__device__ __forceinline__ uint32_t run_token(const unsigned long long *best_ptr, const DevCtl *ctl, int seq,
                                              const BatchState *bs) {
    if (seq) {
        if (ctl->batch_n >= 2) return bs && bs->tt_index != kNoTT ? bs->tt_token : 0xFFFFFFFFu;
        if (ctl->batch_n != 1) return 0xFFFFFFFFu;
        best_ptr += ctl->k_done;
    }
    const unsigned long long best = *best_ptr;
    if ((best >> 32) == 0) return 0xFFFFFFFFu;
    const uint32_t key = ~(uint32_t)best;
    return (key >> 16) == (key & 0xFFFFu) ? key >> 16 : 0xFFFFFFFFu;
}

// The segments of a chunk's 4096 tiles, loaded coalesced into LDS; then every thread joins its
// kRunPerThread consecutive tiles (result also left in sh[thread]).
__device__ __forceinline__ RunSeg run_chunk_join(const TileSum *sin, uint32_t n_tiles, uint64_t base,
                                                 unsigned long long *segs, unsigned long long *sh) {
    const RunSeg none = run_none();
    for (uint32_t i = threadIdx.x; i < kRunChunk; i += kRunThreads)
        segs[i] = run_pack(base + i < n_tiles ? run_seg_of(sin[base + i]) : none);
    __syncthreads();
    RunSeg mine = none;
    for (int i = 0; i < kRunPerThread; ++i) mine = run_join(mine, run_unpack(segs[threadIdx.x * kRunPerThread + i]));
    sh[threadIdx.x] = run_pack(mine);
    __syncthreads();
    return mine;
}

__global__ __launch_bounds__(kRunThreads) void k_run_partial(const TileSum *__restrict__ sin, uint32_t n_tiles,
                                                             const unsigned long long *best_ptr, const DevCtl *ctl,
                                                             int seq, const BatchState *bs,
                                                             unsigned long long *__restrict__ part) {
    __shared__ unsigned long long segs[kRunChunk];
    __shared__ unsigned long long sh[kRunThreads];
    const uint32_t t = run_token(best_ptr, ctl, seq, bs);
    if (t == 0xFFFFFFFFu) return;
    run_chunk_join(sin, n_tiles, (uint64_t)blockIdx.x * kRunChunk, segs, sh);
    if (threadIdx.x == 0) {
        RunSeg acc = run_none();
        for (int i = 0; i < kRunThreads; ++i) acc = run_join(acc, run_unpack(sh[i]));
        part[blockIdx.x] = run_pack(acc);
    }
}

__global__ __launch_bounds__(kRunThreads) void k_run_final(const TileSum *__restrict__ sin, uint32_t n_tiles,
                                                           const unsigned long long *best_ptr, const DevCtl *ctl,
                                                           int seq, const BatchState *bs,
                                                           const unsigned long long *__restrict__ part,
                                                           const RankEdge *le, uint32_t *__restrict__ run_in) {
    __shared__ unsigned long long segs[kRunChunk];
    __shared__ unsigned long long sh[kRunThreads];
    __shared__ unsigned long long pre[kRunThreads];
    const uint32_t t = run_token(best_ptr, ctl, seq, bs);
    if (t == 0xFFFFFFFFu) return;
    // everything before this chunk: each thread joins a slice of the chunk aggregates, thread 0 the slices
    const uint32_t nb = blockIdx.x, per = (nb + kRunThreads - 1) / kRunThreads;
    RunSeg acc = run_none();
    for (uint32_t i = threadIdx.x * per; i < nb && i < (threadIdx.x + 1) * per; ++i) acc = run_join(acc, run_unpack(part[i]));
    pre[threadIdx.x] = run_pack(acc);
    const uint64_t base = (uint64_t)blockIdx.x * kRunChunk;
    run_chunk_join(sin, n_tiles, base, segs, sh);            // (syncs)
    if (threadIdx.x == 0) {
        RunSeg run = run_none();
        for (int i = 0; i < kRunThreads; ++i) run = run_join(run, run_unpack(pre[i]));
        for (int i = 0; i < kRunThreads; ++i) {            // exclusive scan over the threads of this chunk
            const RunSeg mine = run_unpack(sh[i]);
            sh[i] = run_pack(run);
            run = run_join(run, mine);
        }
    }
    __syncthreads();
    RunSeg run = run_unpack(sh[threadIdx.x]);
    const unsigned long long edge = le ? ((unsigned long long)le->tail_run_hi << 32) | le->tail_run_lo : 0ull;
    const uint32_t edge_tok = le ? le->tail0 : kHole;
    // (the per-tile results go back through LDS so that the stores are coalesced too)
    for (int i = 0; i < kRunPerThread; ++i) {
        const uint32_t k = threadIdx.x * kRunPerThread + i;
        // the left rank's trailing run continues into this shard while everything before the tile is that token
        const bool reach = le && edge_tok != kHole && (run.empty || (run.full && run.token == edge_tok));
        const unsigned long long rb = run.len + (reach ? edge : 0ull);
        const RunSeg here = run_unpack(segs[k]);
        segs[k] = (rb & 1ull) | (rb >= 2 ? 2ull : 0ull);
        run = run_join(run, here);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kRunChunk; i += kRunThreads)
        if (base + i < n_tiles) run_in[base + i] = (uint32_t)segs[i];
}

__global__ __launch_bounds__(kMergeThreads) void k_summarize(const uint16_t *__restrict__ tok,
                                                             TileSum *__restrict__ sums, uint32_t n_tiles) {
    const uint32_t waves_per_block = kMergeThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    for (uint32_t tile = blockIdx.x * waves_per_block + threadIdx.x / kWave; tile < n_tiles; tile += n_waves) {
        const uint4 q = reinterpret_cast<const uint4 *>(tok)[(uint64_t)tile * kWave + lane_id()];
        uint32_t s[8];
        unpack8(q, s);
        uint4 sum;
        bool live = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) live = live && s[j] != kHole;
        if (__ballot(!live) == 0ull) {
            // no hole in the tile (fresh from k_widen or from a compaction): the ends are where they are, and the
            // trailing run of tokens equal to the last one is 512 minus the position of the last token that differs
            const uint32_t lane = lane_id();
            const uint32_t t0 = rlane(s[7], kWave - 1);
            uint32_t last_diff = 0;                        // 1 + index (in the lane) of the last slot that differs from t0
#pragma unroll
            for (int j = 0; j < 8; ++j) last_diff = s[j] != t0 ? (uint32_t)j + 1u : last_diff;
            const unsigned long long dm = __ballot(last_diff != 0u);
            uint32_t run = kTile;
            if (dm) {
                const uint32_t hl = 63u - (uint32_t)__builtin_clzll(dm);
                run = kTile - (hl * 8u + rlane(last_diff, hl));
            }
            sum = sum_to_u4(rlane(s[0], 0), rlane(s[1], 0), rlane(s[6], kWave - 1), t0, kTile, run);
            (void)lane;
        } else {
            sum = wave_summary(s);
        }
        if (lane_id() == 0) reinterpret_cast<uint4 *>(sums)[tile] = sum;
    }
}

// ---- argmax ------------------------------------------------------------------

constexpr int kArgmaxThreads = 256;

__global__ __launch_bounds__(kArgmaxThreads) void k_argmax(PairTable t, const DevCtl *ctl,
                                                           unsigned long long *best) {
    __shared__ unsigned long long wbest[kArgmaxThreads / kWave];
    const uint32_t n = table_size(t, ctl);
    unsigned long long b = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long p = entry_packed(t, i);
        b = p > b ? p : b;
    }
    b = wave_max_u64(b);
    if (lane_id() == 0) wbest[threadIdx.x / kWave] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < kArgmaxThreads / kWave; ++k) b = wbest[k] > b ? wbest[k] : b;
        if (b) atomicMax(best, b);
    }
}

// Hierarchical argmax for large tables.  A count never rises after its pair was
// inserted (a merge only lowers existing pairs and inserts pairs of the new
// token), so per-block maxima recorded at insertion stay valid UPPER bounds:
//   bmax[B] >= max packed value of entries [1024 B, 1024 B + 1024)
//   smax[S] >= max of bmax[1024 S ..]
// One workgroup repeatedly takes the block with the highest bound, recomputes
// its true maximum (tightening both bounds) and stops as soon as that true
// maximum is not below any other bound.  Typically two rounds per merge (the
// first one hits the block of the pair that was just merged away), whatever
// the table size.
constexpr int kHierThreads = 256;                    // 4 waves: the reductions stay cheap
constexpr int kHierItems = kBlockSize / kHierThreads; // 4 values per thread and level

struct Top2 { unsigned long long v1, v2; uint32_t i1; };

__device__ __forceinline__ Top2 top2_merge(Top2 a, unsigned long long bv1, unsigned long long bv2, uint32_t bi1) {
    Top2 r;
    if (bv1 > a.v1) { r.v1 = bv1; r.i1 = bi1; r.v2 = a.v1 > bv2 ? a.v1 : bv2; }
    else { r.v1 = a.v1; r.i1 = a.i1; r.v2 = bv1 > a.v2 ? bv1 : a.v2; }
    return r;
}

// Largest and second largest of 1024 values over the workgroup (all threads get
// the result): thread t holds the values of indices base + q * 256 + t, q < 4.
__device__ Top2 block_top2(const unsigned long long v[kHierItems], uint32_t base, Top2 *sh) {
    Top2 t = {v[0], 0ull, base + threadIdx.x};
#pragma unroll
    for (int q = 1; q < kHierItems; ++q) t = top2_merge(t, v[q], 0ull, base + q * kHierThreads + threadIdx.x);
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        const unsigned long long o1 = shfl_xor_u64(t.v1, d), o2 = shfl_xor_u64(t.v2, d);
        const uint32_t oi = __shfl_xor(t.i1, d, kWave);
        t = top2_merge(t, o1, o2, oi);
    }
    __syncthreads();                       // sh may still be read from the previous call
    if (lane_id() == 0) sh[threadIdx.x / kWave] = t;
    __syncthreads();
    Top2 r = sh[0];
#pragma unroll
    for (int k = 1; k < kHierThreads / kWave; ++k) r = top2_merge(r, sh[k].v1, sh[k].v2, sh[k].i1);
    return r;
}

__global__ __launch_bounds__(kHierThreads) void k_argmax_hier(PairTable t, const DevCtl *ctl,
                                                              unsigned long long *best) {
    __shared__ Top2 sh[kHierThreads / kWave];
    const uint32_t n = table_size(t, ctl);
    if (n == 0) return;
    const uint32_t n_blocks = (n + kBlockSize - 1) >> kBlockShift;
    const uint32_t n_super = (n_blocks + kBlockSize - 1) >> kBlockShift;
    const uint32_t tid = threadIdx.x;
    // bounds are re-read after this workgroup rewrote them: agent-scope relaxed
    // accesses (sc1) bypass the CU's L1
    auto ld = [](const unsigned long long *p) {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (int round = 0; round < 1 << 20; ++round) {   // terminates: every round lowers one bound
        // best and second-best super-block bound (n_super <= 1024: the table is capped at 2^30 entries)
        unsigned long long v[kHierItems];
#pragma unroll
        for (int q = 0; q < kHierItems; ++q) {
            const uint32_t i = q * kHierThreads + tid;
            v[q] = i < n_super ? ld(&t.smax[i]) : 0ull;
        }
        const Top2 ts = block_top2(v, 0, sh);
        if (ts.v1 == 0ull) return;
        const uint32_t S = ts.i1;
        // best and second-best block bound inside S
#pragma unroll
        for (int q = 0; q < kHierItems; ++q) {
            const uint32_t i = (S << kBlockShift) + q * kHierThreads + tid;
            v[q] = i < n_blocks ? ld(&t.bmax[i]) : 0ull;
        }
        const Top2 tb = block_top2(v, S << kBlockShift, sh);
        const uint32_t B = tb.i1;
        // true maximum of block B
#pragma unroll
        for (int q = 0; q < kHierItems; ++q) {
            const uint32_t e = (B << kBlockShift) + q * kHierThreads + tid;
            v[q] = e < n ? entry_packed(t, e) : 0ull;
        }
        const Top2 te = block_top2(v, B << kBlockShift, sh);
        const unsigned long long truth = te.v1;
        const unsigned long long s_bound = truth > tb.v2 ? truth : tb.v2;
        if (tid == 0) {
            __hip_atomic_store(&t.bmax[B], truth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t.smax[S], s_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // done when the true maximum is not below any other bound
        if (truth >= tb.v2 && truth >= ts.v2) {
            if (tid == 0 && truth) atomicMax(best, truth);
            return;
        }
        __syncthreads();
    }
}

// ---- `first` tie-break (insertion order) ---------------------------------------------
// The reference's default mode (minbpe-cc.cpp:129-131) rebuilds the pair table before every merge
// (Tokenizer.h:581-585) in a PairCountInsertOrder (PairCount.h:101-181): among the pairs of maximal
// count the one inserted first wins (CompareCountOrder, PairCount.h:65-74), i.e. the one whose first
// occurrence comes first in scan order (calculate_freqs walks the chunks front to back,
// Tokenizer.h:135-144).  The counts of a rebuilt table are the exact overlapping-window counts, which
// is what the incrementally maintained table holds (SURVEY.md 8-S rule 3), so no recount is needed:
//   k_argmax[_hier]   max count M (and the lexical winner among the pairs that have it)
//   k_first_gather    how many pairs have count M; a bitmap of their hashed keys
//   k_first_pos       only when several do: the smallest stream position at which one of them
//                     starts -- one wave per tile in stream order, waves stop at the first tile beyond
//                     the best position found so far (many ties: the answer is near the front; few
//                     ties: the bitmap rejects nearly every pair before the table is consulted)
//   k_first_pick      replaces best[k] by (M, that pair)
// A table rebuilt from scratch has no zero-count pairs: max count 0 means "no pair left", where the
// reference's loop breaks (Tokenizer.h:586-588); mbpe_train_result cuts the merges there.
constexpr uint32_t kFirstBitmapWords = 2048;          // 64 Ki bits
struct FirstState {
    unsigned long long pos_key;      // (tile index << 32) | key of the earliest tied pair; ~0: none yet.  (A tile is walked by
                                     //   one wave, which reports only the earliest hit inside it: the tile index orders the
                                     //   reports, so the stream may be as long as the library allows -- 2^28 tiles -- where a
                                     //   slot position in these 32 bits ended at 4 GiB)
    uint32_t n_tie;                  // pairs whose count is the maximum
    uint32_t pad;
    uint32_t bitmap[kFirstBitmapWords];
};

__device__ __forceinline__ uint32_t first_hash(uint32_t key) { return hash_key(key) >> 16; }

// count of `key`, or 0xFFFFFFFF when the pair was never inserted
__device__ __forceinline__ uint32_t table_lookup(const PairTable &t, uint32_t key) {
    if (t.cells) {
        const uint32_t v = t.cells[dense_index(t, key)];
        return (v & kPresent) ? (v & ~kPresent) : 0xFFFFFFFFu;
    }
    uint32_t h = hash_key(key) & t.hmask;
    for (uint32_t probe = 0; probe <= t.hmask; ++probe) {
        const unsigned long long slot = t.hslot[h];
        if ((uint32_t)(slot >> 32) == key) {
            const int32_t c = t.ecnt[(uint32_t)slot];
            return c < 0 ? 0u : (uint32_t)c;
        }
        if (slot == kEmptySlot) return 0xFFFFFFFFu;
        h = (h + 1) & t.hmask;
    }
    return 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_first_gather(PairTable t, const DevCtl *ctl,
                                                      const unsigned long long *__restrict__ best_ptr,
                                                      FirstState *fs, int seq) {
    if (seq) {
        if (ctl->batch_n != 1 || !ctl->first_tie) return;
        best_ptr += ctl->k_done;
    }
    const unsigned long long best = *best_ptr;
    const uint32_t M = (uint32_t)(best >> 32);
    if (M == 0) return;
    const unsigned long long T = (unsigned long long)M << 32;
    const uint32_t n = table_size(t, ctl);
    const uint32_t n_blocks = (n + kBlockSize - 1) >> kBlockShift;
    const uint32_t lane = lane_id();
    const uint32_t n_waves = gridDim.x * (blockDim.x / kWave);
    uint32_t found = 0;
    for (uint32_t B = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; B < n_blocks; B += n_waves) {
        if (t.bmax[B] < T) continue;          // an upper bound of the block's packed values
#pragma unroll 4
        for (uint32_t q = 0; q < kBlockSize / kWave; ++q) {
            const uint32_t e = (B << kBlockShift) + q * kWave + lane;
            const unsigned long long p = e < n ? entry_packed(t, e) : 0ull;
            if ((uint32_t)(p >> 32) == M) {
                const uint32_t hb = first_hash(~(uint32_t)p);
                atomicOr(&fs->bitmap[hb >> 5], 1u << (hb & 31u));
                ++found;
            }
        }
    }
    found = wave_sum(found);
    if (lane == 0 && found) atomicAdd(&fs->n_tie, found);
}

template <int MODE>
__global__ __launch_bounds__(kMergeThreads) void k_first_pos(const uint16_t *__restrict__ tok,
                                                             const uint16_t *__restrict__ tok_other,
                                                             const TileSum *__restrict__ sin, uint32_t n_tiles,
                                                             PairTable t,
                                                             const unsigned long long *__restrict__ best_ptr,
                                                             FirstState *fs, const DevCtl *ctl, int seq,
                                                             const RankEdge *__restrict__ re) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    __shared__ uint32_t bm[kFirstBitmapWords];
    if (seq) {
        if (ctl->batch_n != 1 || !ctl->first_tie) return;
        best_ptr += ctl->k_done;
        if (ctl->cur) tok = tok_other;
    }
    if (fs->n_tie <= 1) return;                           // a unique maximum: position does not matter
    const uint32_t M = (uint32_t)(*best_ptr >> 32);
    for (uint32_t i = threadIdx.x; i < kFirstBitmapWords; i += kMergeThreads) bm[i] = fs->bitmap[i];
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t waves_per_block = kMergeThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    const unsigned long long gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    for (uint32_t tile = rfl(blockIdx.x * waves_per_block + threadIdx.x / kWave); tile < n_tiles; tile += n_waves) {
        // tiles are visited in ascending order: nothing at or after this one can win any more
        const unsigned long long cur = __hip_atomic_load(&fs->pos_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur >> 32) < (unsigned long long)tile) break;
        const uint4 q = reinterpret_cast<const uint4 *>(tok)[(uint64_t)tile * kWave + lane];
        uint32_t s[8];
        unpack8(q, s);
        uint32_t cnt = 0, f1 = kHole;
#pragma unroll
        for (int j = 7; j >= 0; --j)
            if (s[j] != kHole) { f1 = s[j]; ++cnt; }
        const unsigned long long m_live = __ballot(cnt > 0);
        if (!m_live) continue;
        // first live token after this tile (only the last live lane needs it)
        uint32_t after = kHole;
        bool found_after = false;
        for (uint32_t j = tile + 1; j < n_tiles; ++j) {
            const TileSum ns = sin[j];
            if (ns.n_live) { after = ns.head0; found_after = true; break; }
        }
        // (a sharded stream goes on in the next rank's shard: the pair of this shard's last token belongs to this rank)
        if (!found_after && re) after = re->head0;
        const unsigned long long hi = m_live & gt_mask;
        const uint32_t src = hi ? (uint32_t)__builtin_ctzll(hi) : lane;
        const uint32_t nf = __shfl(f1, src, kWave);
        uint32_t nx = hi ? nf : after;                    // next live token after this lane's slots
        uint32_t my_pos = 0xFFFFFFFFu, my_key = 0;
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            const uint32_t self = s[j];
            if (self == kHole) continue;
            if (right_open<MODE>(self, nx) && (MODE != 2 || self != kBarrier)) {   // a pair starts here (Tokenizer.h:135-144)
                const uint32_t key = ((self & idmask) << 16) | (nx & idmask);
                const uint32_t hb = first_hash(key);
                if ((bm[hb >> 5] >> (hb & 31u)) & 1u) {
                    if (table_lookup(t, key) == M) { my_pos = lane * 8u + j; my_key = key; }
                }
            }
            nx = self;
        }
        const unsigned long long hit = __ballot(my_pos != 0xFFFFFFFFu);
        if (hit) {
            const uint32_t w = (uint32_t)__builtin_ctzll(hit);           // lowest lane = earliest position
            const uint32_t pos = rlane(my_pos, w), key = rlane(my_key, w);
            (void)pos;
            if (lane == 0) atomicMin(&fs->pos_key, ((unsigned long long)tile << 32) | key);
            break;                                        // every later tile of this wave lies behind it
        }
    }
}

__global__ __launch_bounds__(256) void k_first_pick(unsigned long long *best_ptr, FirstState *fs, const DevCtl *ctl, int seq) {
    __shared__ unsigned long long pk;
    __shared__ uint32_t nt;
    if (seq) {
        if (ctl->batch_n != 1 || !ctl->first_tie) return;
        best_ptr += ctl->k_done;
    }
    if (threadIdx.x == 0) { pk = fs->pos_key; nt = fs->n_tie; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (nt > 1 && pk != ~0ull) *best_ptr = pack_best((int32_t)(*best_ptr >> 32), (uint32_t)pk);
        fs->pos_key = ~0ull;
        fs->n_tie = 0;
    }
    if (nt) for (uint32_t i = threadIdx.x; i < kFirstBitmapWords; i += blockDim.x) fs->bitmap[i] = 0;
}

// Sharded stream: every rank found the earliest tied pair of ITS shard; the ranks' shards follow each other in rank order,
// so the pair that comes first in the corpus is the hit of the lowest rank that has one.  Each rank writes (found, tile,
// key) into its own slot of a small array whose other slots it leaves zero -- the transport's sum is then an all-gather --
// and k_first_pick_global takes the first slot that says "found".  Slot r = words [2 + 8 r, 2 + 8 r + 3) of a buffer of
// exchange_header_words(n_ranks) words (the layout of the rank-edge header, so that transports see a size they know).
__global__ void k_first_publish(const FirstState *fs, uint32_t *xf, int rank) {
    if (blockIdx.x || threadIdx.x) return;
    const unsigned long long pk = fs->pos_key;
    const bool found = fs->n_tie > 1 && pk != ~0ull;
    xf[2 + 8 * rank + 0] = found ? 1u : 0u;
    xf[2 + 8 * rank + 1] = found ? (uint32_t)(pk >> 32) : 0u;
    xf[2 + 8 * rank + 2] = found ? (uint32_t)pk : 0u;
}

__global__ __launch_bounds__(256) void k_first_pick_global(unsigned long long *best_ptr, FirstState *fs, uint32_t *xf,
                                                           int n_ranks, uint32_t xf_words) {
    __shared__ uint32_t nt;
    if (threadIdx.x == 0) {
        nt = fs->n_tie;
        if (nt > 1) {
            for (int r = 0; r < n_ranks; ++r)
                if (xf[2 + 8 * r]) { *best_ptr = pack_best((int32_t)(*best_ptr >> 32), xf[2 + 8 * r + 2]); break; }
        }
        fs->pos_key = ~0ull;
        fs->n_tie = 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < xf_words; i += blockDim.x) xf[i] = 0;
    if (nt) for (uint32_t i = threadIdx.x; i < kFirstBitmapWords; i += blockDim.x) fs->bitmap[i] = 0;
}

// ---- table init / rehash -------------------------------------------------------

__global__ void k_table_init(const uint32_t *__restrict__ bp, PairTable t, DevCtl *ctl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    uint32_t c = bp[i];
    if (!c) return;
    if (c >= kPresent) { atomicOr(&ctl->err, kErrCountRange); return; }     // (int counts in the reference too: PairCount.h:188)
    uint32_t key = ((i >> 8) << 16) | (i & 0xFFu);
    table_add(t, ctl, key, (int32_t)c, true);
}

__global__ void k_table_rehash(PairTable t, DevCtl *ctl) {      // hashed layout only
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ctl->n_entries) return;
    uint32_t key = t.ekey[i];
    uint32_t h = hash_key(key) & t.hmask;
    const unsigned long long slot = ((unsigned long long)key << 32) | i;
    for (;;) {
        const unsigned long long prev = atomicCAS(&t.hslot[h], kEmptySlot, slot);
        if (prev == kEmptySlot) return;
        h = (h + 1) & t.hmask;
    }
}

// ---- merge pass ------------------------------------------------------------------
// In place, one wave per tile.  For the chosen pair (a,b) -> X
// (merge_incremental, Tokenizer.h:202-306):
//   - the slot holding `a` of a match becomes X (keeping b's chunk-end flag),
//     the slot holding `b` becomes a hole                          (:236-237)
//   - count deltas (:239-280) are accumulated as
//       L[x] += 1  for a match whose left neighbour is x   => (x,a)-1, (x,X)+1
//       R[y] += 1  for a match whose right neighbour is y  => (b,y)-1, (X,y)+1
//     (L[x] = LR[lr_idx(x,0,0)], R[y] = LR[lr_idx(y,0,1)]: one array, so that a
//      multi-GPU run all-reduces a contiguous prefix)
//       adj  += 1  for two matches that touch ("abab")     => (b,a)-1, (X,X)+1
//       m    += 1  per match                               => (a,b)-1
//     which is the reference's sequential result: its transient (X,a)+1/-1
//     on touching matches cancels (SURVEY.md 8-S rule 3).
// Neighbour tokens across the tile edge come from the tile summaries (sin),
// never from the neighbour's slots, so rewriting in place cannot race with a
// neighbour's reads.  The summaries stay untouched during the pass: a changed
// tile writes its new summary to a side array (sout) and sets its bit in a
// bitmap; k_apply copies the marked entries over afterwards.  An unchanged
// tile (nearly all of them) therefore costs no store at all.
//
// Most tiles hold no match: they are recognised from the 8 slots per lane
// with a backward "next live token" chain and one ballot, and only cost the
// 16-byte load plus the 16-byte summary copy.

struct TileIn {
    uint4 q;    // this lane's 8 slots
    uint32_t smw;   // lanes 0..11: the 12 dwords of the summaries of tile-1, tile, tile+1 (lane = 4 * k + word)
};

// Both loads are unconditional: a branch around a load makes hipcc wait
// vmcnt(0) at the next use, which would drain the tiles prefetched behind
// this one.  The three summaries come through a bounds-checked buffer load of
// one dword per lane (a tile in flight then costs 5 registers, not 8): lanes
// 0..11 read the 12 words of tile-1 .. tile+1, every other lane (and a neighbour
// that does not exist) points past the buffer and gets zeros without any traffic.
__device__ __forceinline__ TileIn tile_issue(const uint16_t *tok, __amdgpu_buffer_rsrc_t sums_rsrc,
                                             uint32_t tile) {
    TileIn t;
    const uint32_t lane = lane_id();
    // (uniform 64-bit base + 32-bit lane offset: the address then needs one VGPR, not a pair per buffer)
    const MBPE_GLOBAL_AS char *base =
        (const MBPE_GLOBAL_AS char *)uniform_ptr(reinterpret_cast<uintptr_t>(tok) + (uint64_t)tile * (kWave * 16u));
#if MBPE_NT_STREAM
    const u32x4 q = __builtin_nontemporal_load((const MBPE_GLOBAL_AS u32x4 *)(base + lane * 16u));
#else
    const u32x4 q = *(const MBPE_GLOBAL_AS u32x4 *)(base + lane * 16u);
#endif
    t.q = make_uint4(q.x, q.y, q.z, q.w);
    const uint32_t j = tile + (lane >> 2) - 1u;                // tile 0, lanes 0..3 wrap to 0xFFFFFFFF
    const uint32_t off = (lane < 12 && j < 0x0FFFFFFFu) ? j * 16u + (lane & 3u) * 4u : 0xFFFFFFF0u;
    t.smw = __builtin_amdgcn_raw_buffer_load_b32(sums_rsrc, off, 0, 0);
    return t;
}

// The rare part of the merge pass: this tile may hold a match.  Exact
// neighbours two deep on both sides, decisions, count deltas, in-place
// rewrite and the tile's new summary.  Returns true when the summary was
// written (the tile changed).
template <int MODE>
__device__ __forceinline__ bool merge_tile_full(uint16_t *tok, const TileSum *sin, TileSum *sout, uint32_t *chg,
                                                uint32_t n_tiles,
                                             uint32_t tile, uint32_t s[8], const Halo h, uint32_t a, uint32_t b,
                                             uint32_t X, uint32_t *LR, DeltaCache &dc, bool dc_on,
                                             const uint32_t *run_in,
                                             uint32_t &wave_m, uint32_t &wave_adj, uint32_t &wave_rm, uint16_t *stage) {
    const uint32_t pitch = lr_pitch(X);
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    const uint32_t lane = lane_id();
    const bool same = a == b;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned long long gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);

    uint32_t cnt = 0, f1 = kSent, f2 = kSent, l1 = kSent, l2 = kSent;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (s[j] != kHole) {
            if (cnt == 0) f1 = s[j]; else if (cnt == 1) f2 = s[j];
            l2 = l1; l1 = s[j];
            ++cnt;
        }
    }
    const unsigned long long m_live = __ballot(cnt > 0);
    // tokens before this lane
    uint32_t p1_in, p2_in;
    {
        const unsigned long long lo = m_live & lt_mask;
        const uint32_t src1 = lo ? 63u - (uint32_t)__builtin_clzll(lo) : lane;
        const uint32_t sl1 = __shfl(l1, src1, kWave), sl2 = __shfl(l2, src1, kWave);
        const unsigned long long lo2 = lo & ~(1ull << src1);
        const uint32_t src2 = lo2 ? 63u - (uint32_t)__builtin_clzll(lo2) : lane;
        const uint32_t tl1 = __shfl(l1, src2, kWave);
        p1_in = lo ? sl1 : h.p1;
        p2_in = lo ? (sl2 != kSent ? sl2 : (lo2 ? tl1 : h.p1)) : h.p2;
    }
    // tokens after this lane
    uint32_t n1_in, n2_in;
    {
        const unsigned long long hi = m_live & gt_mask;
        const uint32_t src1 = hi ? (uint32_t)__builtin_ctzll(hi) : lane;
        const uint32_t sf1 = __shfl(f1, src1, kWave), sf2 = __shfl(f2, src1, kWave);
        const unsigned long long hi2 = hi & ~(1ull << src1);
        const uint32_t src2 = hi2 ? (uint32_t)__builtin_ctzll(hi2) : lane;
        const uint32_t tf1 = __shfl(f1, src2, kWave);
        n1_in = hi ? sf1 : h.n1;
        n2_in = hi ? (sf2 != kSent ? sf2 : (hi2 ? tf1 : h.n1)) : h.n2;
    }
    uint32_t n1v[8], n2v[8];
    {
        uint32_t x1 = n1_in, x2 = n2_in;
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            n1v[j] = x1;
            n2v[j] = x2;
            if (s[j] != kHole) { x2 = x1; x1 = s[j]; }
        }
    }
    // a == b: tokens equal to raw `a` immediately before this lane's first slot
    uint32_t run = 0;
    if (same) {
        bool all_a = true;
        uint32_t trail = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (s[j] != kHole) {
                if (s[j] == a) ++trail; else { trail = 0; all_a = false; }
            }
        }
        // inclusive scan of (all_a, trail): R after L -> R.all ? (L.all, L.trail + R.trail) : R
        uint32_t sa = all_a ? 1u : 0u, st = trail;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t oa = __shfl_up(sa, d, kWave), ot = __shfl_up(st, d, kWave);
            if (lane >= (uint32_t)d && sa) { st += ot; sa = oa; }
        }
        uint32_t ea = __shfl_up(sa, 1, kWave), et = __shfl_up(st, 1, kWave);
        if (lane == 0) { ea = 1; et = 0; }
        // lanes with ea set: the run reaches back to the start of the tile
        // run of the previous tile's last token before this tile (k_run_final): parity and ">= 2"; ours if that is `a`
        const uint32_t rb_small = h.p1 == a ? run_in[tile] : 0u;
        run = ea ? et + rb_small : et;   // parity and ">= 2" are all that is used below
    }

    uint32_t my_m = 0, my_adj = 0, my_rm = 0;
    bool changed = false;
    uint32_t p1 = p1_in, p2 = p2_in;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t self = s[j];
        if (self == kHole) continue;
        const uint32_t n1 = n1v[j], n2 = n2v[j];
        bool amatch, bmatch, prev_adjacent;
        if (!same) {
            amatch = (self == a) && ((n1 & idmask) == b);
            bmatch = ((self & idmask) == b) && (p1 == a);
            prev_adjacent = (p1 == b) && (p2 == a);
        } else {
            const bool is_a = (self & idmask) == a;
            const bool odd = run & 1u;
            amatch = (self == a) && !odd && ((n1 & idmask) == a);
            bmatch = is_a && odd;
            prev_adjacent = !odd && run >= 2;
            run = (self == a) ? run + 1 : 0;
        }
        uint32_t nv = self;
        if (amatch) {
            nv = X | (n1 & endbit);
            ++my_m;
            if (left_open<MODE>(p1)) {
                if (prev_adjacent) ++my_adj;
                else dc_add(dc, dc_on, LR, lr_idx(pitch, p1, 0, 0), 1u);
            }
        } else if (bmatch) {
            nv = kHole;
            ++my_rm;
            if (right_open<MODE>(self, n1)) {
                const bool next_adjacent = (n1 == a) && ((n2 & idmask) == b);
                if (!next_adjacent) dc_add(dc, dc_on, LR, lr_idx(pitch, n1 & idmask, 0, 1), 1u);
            }
        }
        p2 = p1; p1 = self;       // neighbours are the OLD tokens
        if (nv != self) { changed = true; s[j] = nv; }
    }
    wave_m += my_m;
    wave_adj += my_adj;
    wave_rm += my_rm;
    if (__ballot(changed) == 0ull) return false;
    {                                   // (the tile keeps its live tokens in its first slots: tile_compact)
        uint32_t n_out;
        const uint4 qc = tile_compact(s, stage, n_out);
        reinterpret_cast<uint4 *>(tok)[(uint64_t)tile * kWave + lane] = qc;
    }
    const uint4 ns = wave_summary(s);
    if (lane == 0) {
        reinterpret_cast<uint4 *>(sout)[tile] = ns;
        atomicOr(&chg[tile >> 5], 1u << (tile & 31u));
    }
    return true;
}

template <int MODE, bool HOT, int DIAG>
__global__ __launch_bounds__(kMergeThreads) void k_merge(uint16_t *tok0, uint16_t *tok1,
                                                         const TileSum *__restrict__ sin,
                                                         TileSum *__restrict__ sout, uint32_t n_tiles,
                                                         uint32_t *__restrict__ chg,
                                                         const unsigned long long *__restrict__ best_ptr,
                                                         uint32_t X, uint32_t *LR, DevCtl *ctl,
                                                         uint32_t *m_adj, const RankEdge *le,
                                                         const RankEdge *re, int seq,
                                                         const uint32_t *__restrict__ run_in, int hot_launched) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    __shared__ DeltaCache dc;
    __shared__ __attribute__((aligned(16))) uint16_t stage_mem[kMergeThreads / kWave][kTileSlots];
    uint16_t *stage = stage_mem[threadIdx.x / kWave];
    const uint32_t lane = lane_id();
    const uint32_t waves_per_block = kMergeThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    uint32_t tile = rfl(blockIdx.x * waves_per_block + threadIdx.x / kWave);

    uint16_t *tok = tok0;
    if (seq) {      // inside a batch sequence: the merge index and the current buffer live on the device
        if (ctl->batch_n != 1) return;
        const uint32_t k = ctl->k_done;
        best_ptr += k;
        X = 256u + k;
        if (ctl->cur) tok = tok1;
    }
    const unsigned long long best = *best_ptr;
    if ((best >> 32) == 0) return;   // count 0 (or no pair at all): nothing can match
    const uint32_t key = ~(uint32_t)best;
    const uint32_t a = rfl(key >> 16), b = rfl(key & 0xFFFFu);
    // two instantiations are launched: the one with the delta cache only works on a frequent pair
    if (hot_mismatch<HOT>(best >> 32, ctl, hot_launched)) return;
    constexpr bool dc_on = HOT;
    if (dc_on) { dc_init(dc); __syncthreads(); }

    uint32_t wave_m = 0, wave_adj = 0, wave_rm = 0;   // lane-local partial sums, reduced once at the end
    if (tile < n_tiles) {

    // three tiles in flight per wave while the current one is examined
    // (past the end the last tile is re-read: loads stay unconditional)
    const uint32_t last_tile = n_tiles - 1;
    auto clamp_tile = [&](uint64_t t) { return (uint32_t)(t < n_tiles ? t : last_tile); };
    const __amdgpu_buffer_rsrc_t sums_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TileSum *>(sin), 0, n_tiles * 16u, 0x00020000);
    TileIn t0 = tile_issue(tok, sums_rsrc, tile);
    TileIn t1 = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + n_waves));
    TileIn t2 = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + 2ull * n_waves));
    bool v1 = (uint64_t)tile + n_waves < n_tiles, v2 = (uint64_t)tile + 2ull * n_waves < n_tiles;
    for (;;) {
        const bool v3 = (uint64_t)tile + 3ull * n_waves < n_tiles;
        TileIn t3 = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + 3ull * n_waves));

        // ---- this tile -------------------------------------------------------
        const uint32_t me_nlive = rlane(t0.smw, 6) & 0xFFFFu;
        if (DIAG == 1) {   // timing-only build: loads only
            asm volatile("" :: "v"(t0.q.x), "v"(t0.q.y), "v"(t0.q.z), "v"(t0.q.w), "v"(t0.smw));
        } else if (me_nlive != 0) {
            uint32_t s[8];
            unpack8(t0.q, s);
            // neighbours' edge tokens
            Halo h;
            const uint32_t pw1 = rlane(t0.smw, 1), pw2 = rlane(t0.smw, 2);
            const uint32_t nw0 = rlane(t0.smw, 8), nw2 = rlane(t0.smw, 10);
            const bool fast = tile > 0 && tile + 1 < n_tiles && (pw2 & 0xFFFFu) >= 2 && (nw2 & 0xFFFFu) >= 2;
            if (fast) {
                h.p1 = pw1 >> 16; h.p2 = pw1 & 0xFFFFu;
                h.n1 = nw0 & 0xFFFFu; h.n2 = nw0 >> 16;
            } else {
                h = halo_slow(sin, n_tiles, tile, le, re);
            }
            // Candidate test: some slot holds `a` and the next live token has id b.
            // The chain starts from the next lane's first slot; when that slot is a
            // hole the unknown token is treated as a wildcard (b), which can only
            // send the tile to the exact path below without need.
            uint32_t c = __shfl_down(s[0], 1, kWave);
            if (c == kHole) c = b;
            if (lane == kWave - 1) c = h.n1;
            uint32_t acc = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                const uint32_t u = (MODE == 1 ? (c & idmask) : c) ^ b;
                const uint32_t tu = (s[j] ^ a) + u;      // 0 <=> s[j] == a and id(next live) == b
                acc = tu < acc ? tu : acc;
                c = s[j] != kHole ? s[j] : c;
            }
            bool work = __ballot(acc == 0u) != 0ull || h.p1 == a;
            if (DIAG == 2) { asm volatile("" :: "v"(acc)); work = false; }
            if (work) merge_tile_full<MODE>(tok, sin, sout, chg, n_tiles, tile, s, h, a, b, X, LR, dc, dc_on, run_in,
                                                   wave_m, wave_adj, wave_rm, stage);
        }

        if (!v1) break;
        tile += n_waves;
        t0 = t1; t1 = t2; t2 = t3;
        v1 = v2; v2 = v3;
    }
    }
    if (dc_on) dc_flush(dc, LR);
    const uint32_t tm = wave_sum(wave_m), ta = wave_sum(wave_adj), tr = wave_sum(wave_rm);
    if (lane == 0) {
        if (tm) atomicAdd(&m_adj[0], tm);
        if (ta) atomicAdd(&m_adj[1], ta);
        if (tr) atomicAdd(&ctl->rm, tr);
    }
}

// ---- apply: fold (L, R, m, adj) into the pair table -------------------------------
// gm_gadj, when not NULL, holds the all-reduced {m, adj} of a multi-GPU run
// (LR is then already all-reduced in place); otherwise ctl->m / ctl->adj.
// n_live / holes follow ctl->rm, the tokens removed from this rank's shard.

__device__ __forceinline__ void patch_sums(TileSum *sums, const TileSum *side, uint32_t *chg, uint32_t n_chg_words,
                                           uint32_t first, uint32_t stride) {
    // summaries of the tiles the merge pass changed: side array -> live array.  A wave takes two words of the
    // bitmap at a time, a lane per tile, so that the 16-byte copies of a wave are one contiguous kilobyte.
    // (first / stride are thread numbers of a launch whose workgroups are whole waves)
    const uint32_t lane = lane_id(), half = lane >> 5, bit = lane & 31u;
    for (uint32_t w2 = first / kWave; 2u * w2 < n_chg_words; w2 += stride / kWave) {
        const uint32_t w = 2u * w2 + half;
        const uint32_t bits = w < n_chg_words ? chg[w] : 0u;
        if ((bits >> bit) & 1u) {
            const uint32_t tile = w * 32u + bit;
            reinterpret_cast<uint4 *>(sums)[tile] = reinterpret_cast<const uint4 *>(side)[tile];
        }
        if (bit == 0u && bits) chg[w] = 0;
    }
}

__global__ void k_patch_sums(const unsigned long long *__restrict__ best_ptr, TileSum *sums, const TileSum *side,
                             uint32_t *chg, uint32_t n_chg_words, DevCtl *ctl, int seq, uint32_t n_tiles) {
    if (seq) {
        if (ctl->batch_n == 0) return;
        if (ctl->marks_all && ctl->fused && ctl->commit_n == ctl->batch_n) {
            // a fused pass that is kept: the side array holds every tile's summary
            const uint64_t n16 = (uint64_t)n_tiles;
            for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
                reinterpret_cast<uint4 *>(sums)[i] = reinterpret_cast<const uint4 *>(side)[i];
            return;
        }
    } else if ((*best_ptr >> 32) == 0) {
        return;
    }
    patch_sums(sums, side, chg, n_chg_words, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

__global__ void k_apply(PairTable t, DevCtl *ctl, const unsigned long long *__restrict__ best_ptr,
                        uint32_t X, uint32_t *LR, const uint32_t *gm_gadj, TileSum *sums,
                        const TileSum *side, uint32_t *chg, uint32_t n_chg_words, int seq) {
    if (seq) {
        if (ctl->batch_n != 1) return;
        const uint32_t k = ctl->k_done;
        best_ptr += k;
        X = 256u + k;
    }
    const unsigned long long best = *best_ptr;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if ((best >> 32) != 0) {      // count 0: the merge changed nothing
        const uint32_t key = ~(uint32_t)best;
        const uint32_t a = key >> 16, b = key & 0xFFFFu;
        patch_sums(sums, side, chg, n_chg_words, x, gridDim.x * blockDim.x);
        const uint32_t pitch = lr_pitch(X);
        for (uint32_t xx = x; xx < X; xx += gridDim.x * blockDim.x) {
            const uint2 lr = make_uint2(LR[lr_idx(pitch, xx, 0, 0)], LR[lr_idx(pitch, xx, 0, 1)]);
            if (lr.x) {
                table_add(t, ctl, (xx << 16) | a, -(int32_t)lr.x, false);
                table_add(t, ctl, (xx << 16) | X, (int32_t)lr.x, true);
            }
            if (lr.y) {
                table_add(t, ctl, (b << 16) | xx, -(int32_t)lr.y, false);
                table_add(t, ctl, (X << 16) | xx, (int32_t)lr.y, true);
            }
            if (lr.x) LR[lr_idx(pitch, xx, 0, 0)] = 0;
            if (lr.y) LR[lr_idx(pitch, xx, 0, 1)] = 0;
        }
        if (x == 0) {
            const uint32_t m = gm_gadj ? gm_gadj[0] : ctl->m;
            const uint32_t adj = gm_gadj ? gm_gadj[1] : ctl->adj;
            if (m) table_add(t, ctl, (a << 16) | b, -(int32_t)m, false);
            if (adj) {
                table_add(t, ctl, (b << 16) | a, -(int32_t)adj, false);
                table_add(t, ctl, (X << 16) | X, (int32_t)adj, true);
            }
            // tokens removed from this shard = second tokens of matches it holds (with
            // several ranks a match can straddle two shards, so this is not m)
            const uint32_t rm = ctl->rm;
            ctl->removed_total += rm;
            ctl->n_live -= rm;
            ctl->m = 0;
            ctl->adj = 0;
            ctl->rm = 0;
        }
    }
}

// ---- batched merges -----------------------------------------------------------------
// Several merges per pass over the stream.  The selection (k_sel_pick, or
// k_select_batch when the gathered list cannot be used) takes the next
// candidates in argmax order as long as each one is INDEPENDENT of the ones
// before it: for an earlier (a,b) and a later (c,d): d != a and c != b (then
// merging (a,b) cannot change count(c,d), and occurrences of the two pairs
// cannot overlap), no (t,t) pair and no zero count inside a multi-pair batch.
// All pairs ranked before a candidate are in the batch, every other old pair
// can only have lost count, so the candidate is the true next argmax unless a
// pair created by the earlier merges of the batch ((x,X_i), (X_i,y), (X_p,X_i))
// beats it.  That is checked AFTER the counting pass (k_validate, comparing
// packed (count, ~key) values) and BEFORE the stream is changed for good: the
// scan pass (k_scan_batch) only counts deltas per pair and marks the tiles that
// hold matches, and the rewrite pass (k_rewrite_marked) then applies the
// validated prefix to the marked tiles; the fused pass of a large batch
// (k_fused_batch) writes the merged stream to the other token buffer, which
// only becomes the stream if the whole batch survives.  A batch of one pair
// (always valid) takes the in-place pass k_merge instead.

// Lookup table of a batch in LDS: is (first, second) one of the batch pairs, and
// which one?  A hash table of kBuckets buckets of two keys each, keys stored as
// (first | second << 16): ONE 8-byte LDS read and two compares per test, exact
// (no false positives), whatever the batch size.  k_select_batch keeps every
// bucket within its two keys: a candidate that would be the third key of a
// bucket ends the batch.  0xFFFE is never a token id (MBPE_MAX_VOCAB_*), so
// kEmptyPair can never be asked for.
#ifndef MBPE_PRED_NUM
#define MBPE_PRED_NUM 4      /* quarters of the measured per-dependency loss that the prediction counts on */
#endif
#ifndef MBPE_LUT_KEYS
#define MBPE_LUT_KEYS 2
#endif
constexpr uint32_t kBucketKeys = MBPE_LUT_KEYS;          // keys per bucket: 2 (one 8-byte read), 3 (8 + 4 bytes) or 4 (16 bytes)
#ifndef MBPE_LUT_BUCKETS
#define MBPE_LUT_BUCKETS (MBPE_LUT_KEYS == 4 ? 4096 : 8192)
#endif
constexpr uint32_t kBuckets = MBPE_LUT_BUCKETS;
static_assert((kBuckets & (kBuckets - 1u)) == 0, "the hash is masked");
constexpr uint32_t kEmptyPair = 0xFFFEFFFEu;
static_assert(kBucketKeys >= 2 && kBucketKeys <= 4, "bucket = an 8-byte, an 8- and a 4-byte, or a 16-byte LDS read");

static_assert(kBatchMax <= 65536, "batch indices are stored in 16 bits");
// 8192 buckets of two keys.  A batch ends when some bucket would need a third key under every hash multiplier still in
// the race (expected overflowing buckets n^3 / (6 B^2): near 1,400 pairs with eight multipliers).  MBPE_LUT_KEYS 3 (a
// second, 4-byte read per test: 144 KB of LDS, batches of 2,048 pairs with -DMBPE_BATCH_MAX=2048) was built and measured
// in round 3: the fused pass went from 8.3 to 10.5 ms at the same batch size -- every LDS read in the per-slot path costs
// as much as eight vector instructions -- which the 7 passes it saved (83 -> 76) do not pay back.
// The other form of the table, for batches whose pairs are all pairs of raw bytes (BatchState::byte_lut): the batch index
// of (first, second) at [first][second], 0xFFFF where there is none -- ONE 2-byte LDS read per test, no hash, no key
// compare, no limit on the batch but kBatchMax.  Row 256 and the last column are all 0xFFFF: a first token that is no byte
// (a merged token, a hole, a token with its chunk-end bit) is clamped to that row, a second one to that column.  Columns
// 256 .. 256 + kTTMax - 1 are the stand-in ids of (t,t) members (tt_rename).  First phase of every byte-level BPE, and all
// of the benchmark's 31,744 merges (uniform bytes: every pair of merged tokens is ~256 times rarer than a byte pair).
constexpr uint32_t kByteCols = 256u + (uint32_t)kTTMax + 1u;         // 273: bytes, stand-ins, "no"
constexpr uint32_t kByteRows = 257u;
struct BatchLutHash {
#if MBPE_LUT_KEYS == 2
    uint2 bucket[kBuckets];
    uint32_t bidx[kBuckets];     // batch index of bucket.x (low half) and bucket.y (high half)
#elif MBPE_LUT_KEYS == 3
    uint2 bucket[kBuckets];      // keys 0 and 1
    uint32_t bucket2[kBuckets];  // key 2
    uint32_t bidx[kBuckets];     // batch index of bucket.x (low half) and bucket.y (high half)
    uint16_t bidx2[kBuckets];    // ... of bucket2
#else
    uint4 bucket[kBuckets];
    uint2 bidx[kBuckets];        // batch indices of bucket.x .. bucket.w, 16 bits each
#endif
};
struct BatchLutMem : BatchLutHash {             // (in LDS, one per workgroup: the hash form, or -- over the same bytes -- the byte form)
    uint16_t byte_tail[kByteTable && kByteRows * kByteCols > sizeof(BatchLutHash) / 2 ? kByteRows * kByteCols - sizeof(BatchLutHash) / 2 : 1];
    __device__ __forceinline__ uint16_t *byte_tab() { return reinterpret_cast<uint16_t *>(this); }
};
static_assert(!kByteTable || sizeof(BatchLutMem) >= kByteRows * kByteCols * 2, "the byte table overlays the hash table");
// The hash of a batch's table is second * mul + first (one v_mad_u32_u24), masked; mul is chosen per batch by the
// selection among kHashMul so that no bucket needs a third key for as long as possible (BatchState::hash_mul) and
// reaches the stream kernels in a scalar register.
#ifndef MBPE_HASH_SEEDS
#define MBPE_HASH_SEEDS 8
#endif
constexpr int kHashSeeds = MBPE_HASH_SEEDS;
__device__ __constant__ const uint32_t kHashMul[8] = {2531u, 40503u, 10007u, 60493u, 25013u, 7919u, 52361u, 33391u};
static_assert(kHashSeeds >= 1 && kHashSeeds <= 8, "kHashMul");
struct BatchLut {
    BatchLutMem *m;
    uint32_t mul;                // uniform
    bool bytes;                  // uniform: the byte form
    uint32_t idmask;             // 0x7FFF with chunk-end bits in the slots, else 0xFFFF (where the stand-in ids lie)
    __device__ __forceinline__ BatchLut(BatchLutMem *mem, const BatchState *bs, uint32_t idmask_)
        : m(mem), mul(rfl(bs->hash_mul)), bytes(kByteTable && rfl(bs->byte_lut) != 0u), idmask(idmask_) {}
};

// byte form: the table entry of (first, second); first: any raw slot value, second: a token id (or kHole)
template <bool TT>
__device__ __forceinline__ uint32_t byte_entry(const BatchLut &lut, uint32_t first, uint32_t second, uint32_t idmask) {
    const uint32_t r = first < 256u ? first : 256u;
    uint32_t c = second < 256u ? second : kByteCols - 1u;
    if (TT) {                     // a stand-in id (idmask - 1 - i) is column 256 + i
        const uint32_t i = idmask - 1u - second;
        c = i < (uint32_t)kTTMax ? 256u + i : c;
    }
    return lut.m->byte_tab()[r * kByteCols + c];
}

__device__ __forceinline__ uint32_t pair_hash(uint32_t mul, uint32_t first, uint32_t second) {
    return (__umul24(second, mul) + first) & (kBuckets - 1u);      // one v_mad_u32_u24
}
// the same with the batch's (wave-uniform) multiplier in a scalar register: inline asm, because the compiler would
// make the multiply-add a 64-bit v_mad_u64_u32
__device__ __forceinline__ uint32_t pair_hash_u(uint32_t uniform_mul, uint32_t first, uint32_t second) {
    uint32_t h;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(h) : "v"(second), "s"(uniform_mul), "v"(first));
    return h & (kBuckets - 1u);
}

// The (t,t) members of the batch, for the kernel instantiations that handle them: token -> stand-in id
// (direct-mapped on the token's low bits; k_sel_pick keeps the slots distinct), stand-in -> token and
// batch index, and the workgroup's match counts (a (t,t) pair's matches are not its count).
struct TTInfo {
    uint32_t map[kTTSlots];      // token | stand-in << 16, or 0xFFFFFFFF
    uint32_t tok[kTTMax];
    uint32_t pair[kTTMax];
    uint32_t cnt[kTTMax];
};

__device__ __forceinline__ void tt_build(TTInfo &ti, const BatchState *bs, uint32_t n_keys, uint32_t fake) {
    for (uint32_t i = threadIdx.x; i < kTTSlots; i += blockDim.x) ti.map[i] = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kTTMax; i += blockDim.x) { ti.tok[i] = kHole; ti.pair[i] = 0; ti.cnt[i] = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (uint32_t j = 0; j < n_keys && n < (uint32_t)kTTMax; ++j) {
            const uint32_t key = bs->key[j];
            const uint32_t a = key >> 16;
            if ((key & 0xFFFFu) != a) continue;
            ti.map[a & (kTTSlots - 1u)] = a | ((fake - n) << 16);
            ti.tok[n] = a;
            ti.pair[n] = j;
            ++n;
        }
    }
    __syncthreads();
}

// the workgroup's match counts of the (t,t) members, added to hdr_m at the end of the kernel
__device__ __forceinline__ void tt_flush(TTInfo &ti, uint32_t *hdr_m) {
    __syncthreads();
    if (threadIdx.x < (uint32_t)kTTMax && ti.cnt[threadIdx.x]) atomicAdd(&hdr_m[ti.pair[threadIdx.x]], ti.cnt[threadIdx.x]);
}

constexpr uint32_t kTTMark = 0xFFFEu;      // byte table, cell (t, t) of a (t,t) member: "two of them side by side: rename first"
__device__ __forceinline__ void lut_build(BatchLut &lut, const BatchState *bs, uint32_t n_keys, uint32_t fake,
                                          bool tt_marks = false) {
    if (lut.bytes) {
        uint32_t *tw = reinterpret_cast<uint32_t *>(lut.m->byte_tab());
        for (uint32_t i = threadIdx.x; i < (kByteRows * kByteCols + 1u) / 2u; i += blockDim.x) tw[i] = 0xFFFFFFFFu;
        __syncthreads();
        uint16_t *tab = lut.m->byte_tab();
        // (no two pairs share a cell: every thread places its own; the (t,t) members take their stand-in columns in
        //  batch order, like tt_build)
        for (uint32_t j = threadIdx.x; j < n_keys; j += blockDim.x) {
            const uint32_t key = bs->key[j];
            const uint32_t a = key >> 16, b = key & 0xFFFFu;
            if (a != b) tab[a * kByteCols + b] = (uint16_t)j;
        }
        if (threadIdx.x == 0 && bs->tt_index != kNoTT) {
            uint32_t n_tt = 0;
            for (uint32_t j = bs->tt_index; j < n_keys && n_tt < (uint32_t)kTTMax; ++j) {
                const uint32_t key = bs->key[j];
                if ((key >> 16) == (key & 0xFFFFu)) {
                    tab[(key >> 16) * kByteCols + 256u + n_tt++] = (uint16_t)j;
                    // (only the fused pass on prefix-form tiles asks for it: there the lookup of (slot, next slot) that
                    //  every slot does anyway tells whether the tile holds a run of a member's token -- no separate test)
                    if (tt_marks) tab[(key >> 16) * kByteCols + (key >> 16)] = (uint16_t)kTTMark;
                }
            }
        }
        __syncthreads();
        return;
    }
    uint32_t *words = reinterpret_cast<uint32_t *>(lut.m->bucket);
    constexpr uint32_t kMainKeys = kBucketKeys == 3 ? 2u : kBucketKeys;      // keys per bucket in `bucket`
    for (uint32_t i = threadIdx.x; i < kBuckets * kMainKeys; i += blockDim.x) words[i] = kEmptyPair;
    uint32_t *iw = reinterpret_cast<uint32_t *>(lut.m->bidx);
    for (uint32_t i = threadIdx.x; i < kBuckets * kMainKeys / 2; i += blockDim.x) iw[i] = 0;
#if MBPE_LUT_KEYS == 3
    for (uint32_t i = threadIdx.x; i < kBuckets; i += blockDim.x) { lut.m->bucket2[i] = kEmptyPair; lut.m->bidx2[i] = 0; }
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n_tt = 0;
        for (uint32_t j = 0; j < n_keys; ++j) {
            const uint32_t key = bs->key[j];
            const uint32_t a = key >> 16;
            uint32_t b = key & 0xFFFFu;
            if (b == a) b = fake - n_tt++;                                       // (t,t): see tt_rename
            const uint32_t h = pair_hash(lut.mul, a, b);
            const uint32_t kk = a | (b << 16);
            uint32_t r = 0;                       // first free key of the bucket (selection keeps it within kBucketKeys)
            while (r + 1 < kMainKeys && words[h * kMainKeys + r] != kEmptyPair) ++r;
#if MBPE_LUT_KEYS == 3
            if (words[h * kMainKeys + r] != kEmptyPair) {      // both taken: the bucket's third key
                lut.m->bucket2[h] = kk;
                lut.m->bidx2[h] = (uint16_t)j;
                continue;
            }
#endif
            words[h * kMainKeys + r] = kk;
            uint16_t *ix = reinterpret_cast<uint16_t *>(&lut.m->bidx[h]);
            ix[r] = (uint16_t)j;
        }
    }
    __syncthreads();
}

// is (first, second) a batch pair?  first may be any raw slot value (a hole or a
// token with the chunk-end bit never matches), second the id of the next live token
__device__ __forceinline__ bool pair_test(const BatchLut &lut, uint32_t first, uint32_t second) {
    if (lut.bytes) return byte_entry<true>(lut, first, second, lut.idmask) != 0xFFFFu;
    const uint32_t hh = pair_hash_u(lut.mul, first, second);
    const auto bk = lut.m->bucket[hh];
    const uint32_t kk = first | (second << 16);
#if MBPE_LUT_KEYS == 2
    return bk.x == kk || bk.y == kk;
#elif MBPE_LUT_KEYS == 3
    return bk.x == kk || bk.y == kk || lut.m->bucket2[hh] == kk;
#else
    return bk.x == kk || bk.y == kk || bk.z == kk || bk.w == kk;
#endif
}

// The same test in the streaming loops, written so that it costs few VALU instructions:
// hash with one v_mad_u32_u24, no boolean materialised.  Returns 0 iff (first, second) is a
// batch pair, something non-zero otherwise.
__device__ __forceinline__ uint32_t pair_miss(const BatchLut &lut, uint32_t first, uint32_t second) {
    if (lut.bytes) return byte_entry<true>(lut, first, second, lut.idmask) ^ 0xFFFFu ? 0u : 1u;
    uint32_t h;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(h) : "v"(second), "s"(lut.mul), "v"(first));
    const auto bk = lut.m->bucket[h & (kBuckets - 1u)];
    const uint32_t kk = first | (second << 16);
    const uint32_t dx = bk.x ^ kk, dy = bk.y ^ kk;
    uint32_t d = dx < dy ? dx : dy;
#if MBPE_LUT_KEYS == 3
    const uint32_t dz = lut.m->bucket2[h & (kBuckets - 1u)] ^ kk;
    d = d < dz ? d : dz;
#endif
#if MBPE_LUT_KEYS == 4
    const uint32_t dz = bk.z ^ kk, dw = bk.w ^ kk;
    const uint32_t e = dz < dw ? dz : dw;
    d = d < e ? d : e;
#endif
    return d;
}

// ... and as a per-lane boolean, which the compiler keeps as a wave mask in scalar registers: one
// compare per key, and the results combine on the scalar unit.
__device__ __forceinline__ bool pair_hit(const BatchLut &lut, uint32_t first, uint32_t second) {
    if (lut.bytes) return byte_entry<true>(lut, first, second, lut.idmask) != 0xFFFFu;
    uint32_t h;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(h) : "v"(second), "s"(lut.mul), "v"(first));
    const auto bk = lut.m->bucket[h & (kBuckets - 1u)];
    const uint32_t kk = first | (second << 16);
#if MBPE_LUT_KEYS == 2
    return bk.x == kk || bk.y == kk;
#elif MBPE_LUT_KEYS == 3
    return bk.x == kk || bk.y == kk || lut.m->bucket2[h & (kBuckets - 1u)] == kk;
#else
    return bk.x == kk || bk.y == kk || bk.z == kk || bk.w == kk;
#endif
}

// ... and which of the bucket's keys matched (two-key buckets): the pair index is then ONE more LDS read (the bucket's
// index word) instead of two -- an LDS read in the per-slot path costs as much as eight vector instructions.
__device__ __forceinline__ bool pair_hit2(const BatchLut &lut, uint32_t first, uint32_t second, bool &second_key) {
    uint32_t h;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(h) : "v"(second), "s"(lut.mul), "v"(first));
    const auto bk = lut.m->bucket[h & (kBuckets - 1u)];
    const uint32_t kk = first | (second << 16);
    second_key = bk.y == kk;
#if MBPE_LUT_KEYS == 2
    return bk.x == kk || second_key;
#elif MBPE_LUT_KEYS == 3
    return bk.x == kk || second_key || lut.m->bucket2[h & (kBuckets - 1u)] == kk;
#else
    return bk.x == kk || second_key || bk.z == kk || bk.w == kk;
#endif
}

// index of the pair (only called for pairs that passed pair_test)
__device__ __forceinline__ int lut_index(const BatchLut &lut, uint32_t first, uint32_t second) {
    if (lut.bytes) return (int)byte_entry<true>(lut, first, second, lut.idmask);
    const uint32_t h = pair_hash_u(lut.mul, first, second);
    const uint32_t kk = first | (second << 16);
#if MBPE_LUT_KEYS == 2
    const uint32_t ix = lut.m->bidx[h];
    return (int)(lut.m->bucket[h].x == kk ? ix & 0xFFFFu : ix >> 16);
#elif MBPE_LUT_KEYS == 3
    // (a pair that is in the table and is neither of the first two keys of its bucket is the third)
    const uint32_t ix = lut.m->bidx[h];
    const uint32_t i3 = lut.m->bidx2[h];
    const uint2 bk = lut.m->bucket[h];
    return (int)(bk.x == kk ? ix & 0xFFFFu : bk.y == kk ? ix >> 16 : i3);
#else
    const uint4 bk = lut.m->bucket[h];
    const uint2 i2 = lut.m->bidx[h];
    const uint32_t w = (bk.x == kk || bk.y == kk) ? i2.x : i2.y;
    return (int)((bk.x == kk || bk.z == kk) ? w & 0xFFFFu : w >> 16);
#endif
}

// the same when the membership test has already told which key of the bucket matched (pair_hit2)
__device__ __forceinline__ uint32_t lut_index_known(const BatchLut &lut, uint32_t first, uint32_t second, bool second_key) {
    if (lut.bytes) return byte_entry<true>(lut, first, second, lut.idmask);
#if MBPE_LUT_KEYS == 2
    const uint32_t ix = lut.m->bidx[pair_hash_u(lut.mul, first, second)];
    return second_key ? ix >> 16 : ix & 0xFFFFu;
#else
    (void)second_key;
    return (uint32_t)lut_index(lut, first, second);
#endif
}

// Which pass merges a multi-pair batch: the fused one (reads the stream once, writes all of it to
// the other buffer) or scan + rewrite of the marked tiles (reads once, then re-reads and rewrites
// the tiles that hold a match).  The fused pass wins once a good part of the tiles hold a match:
// many pairs, or few but frequent ones (text).  fused_min >= 1000 turns it off.
__device__ __forceinline__ bool want_fused_sum(unsigned long long matches, uint32_t accepted, uint32_t fused_min,
                                               unsigned long long n_live_all) {
    if (accepted < 2 || fused_min >= 1000u) return false;
    if (accepted >= fused_min) return true;
    return matches * 2048ull >= n_live_all;        // about one tile in five holds a match
}
__device__ __forceinline__ bool want_fused(const BatchState *bs, uint32_t accepted, uint32_t fused_min,
                                           unsigned long long n_live_all) {
    if (accepted < 2 || fused_min >= 1000u) return false;
    if (accepted >= fused_min) return true;
    unsigned long long matches = 0;
    for (uint32_t i = 0; i < accepted; ++i) matches += bs->packed[i] >> 32;
    return want_fused_sum(matches, accepted, fused_min, n_live_all);
}

__global__ __launch_bounds__(kHierThreads) void k_select_batch(PairTable t, DevCtl *ctl, BatchState *bs,
                                                               unsigned long long *best, uint32_t n_target,
                                                               uint32_t max_batch, uint32_t fused_min,
                                                               uint32_t n_ranks) {
    __shared__ Top2 sh[kHierThreads / kWave];
    __shared__ uint32_t s_keys[kBatchMax];
    const uint32_t tid = threadIdx.x;
    if (ctl->sel_ok) return;                 // k_sel_pick already chose this batch
    const uint32_t k0 = ctl->k_done;
    const uint32_t k_limit = ctl->k_limit < n_target ? ctl->k_limit : n_target;
    const uint32_t n = table_size(t, ctl);
    __syncthreads();
    if (tid == 0) { ctl->batch_n = 0; ctl->commit_n = 0; ctl->fused = 0; }
    if (k0 >= k_limit || n == 0) return;
    uint32_t limit = k_limit - k0;
    if (limit > max_batch) limit = max_batch;
    if (limit > (uint32_t)kBatchMax) limit = kBatchMax;
    if (ctl->adapt_limit && limit > ctl->adapt_limit) limit = ctl->adapt_limit;
    if (ctl->first_mode) {          // one pair, and k_first_* look whether another one with its count comes first
        limit = 1;
        if (tid == 0) ctl->first_tie = 1;
    }
    const uint32_t n_blocks = (n + kBlockSize - 1) >> kBlockShift;
    const uint32_t n_super = (n_blocks + kBlockSize - 1) >> kBlockShift;
    auto ld = [](const unsigned long long *p) {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t accepted = 0;
    // Bounds and the entries of the block under inspection are cached in
    // registers (thread t owns items q * 256 + t of every level), so a round
    // costs LDS reductions plus at most one global load.
    unsigned long long sb[kHierItems], bb[kHierItems], ev[kHierItems];
#pragma unroll
    for (int q = 0; q < kHierItems; ++q) {
        const uint32_t i = q * kHierThreads + tid;
        sb[q] = i < n_super ? ld(&t.smax[i]) : 0ull;
        bb[q] = 0;
        ev[q] = 0;
    }
    uint32_t cur_S = 0xFFFFFFFFu, cur_B = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < limit; ++k) {
        // hierarchical argmax over the entries that are not yet in the batch
        unsigned long long cand = 0;
        uint32_t cand_idx = 0;
        for (int round = 0; round < 1 << 20; ++round) {
            const Top2 ts = block_top2(sb, 0, sh);
            if (ts.v1 == 0ull) break;
            const uint32_t S = ts.i1;
            if (S != cur_S) {
                cur_S = S;
#pragma unroll
                for (int q = 0; q < kHierItems; ++q) {
                    const uint32_t i = (S << kBlockShift) + q * kHierThreads + tid;
                    bb[q] = i < n_blocks ? ld(&t.bmax[i]) : 0ull;
                }
            }
            const Top2 tb = block_top2(bb, S << kBlockShift, sh);
            const uint32_t B = tb.i1;
            if (B != cur_B) {
                cur_B = B;
#pragma unroll
                for (int q = 0; q < kHierItems; ++q) {
                    const uint32_t e = (B << kBlockShift) + q * kHierThreads + tid;
                    ev[q] = e < n ? entry_packed(t, e) : 0ull;
                }
            }
            unsigned long long p[kHierItems];
#pragma unroll
            for (int q = 0; q < kHierItems; ++q) {
                p[q] = ev[q];
                if (p[q]) {
                    // a pair already in the batch will have count 0 once it is merged (a != b):
                    // it stays a (zero-count) candidate, SURVEY 8-S rule 4
                    const uint32_t ekey = ~(uint32_t)p[q];
                    bool excluded = false;
                    for (uint32_t i = 0; i < accepted; ++i) excluded |= s_keys[i] == ekey;
                    if (excluded) p[q] = pack_best(0, ekey);
                }
            }
            const Top2 te = block_top2(p, B << kBlockShift, sh);
            const unsigned long long truth = te.v1;
            const unsigned long long s_bound = truth > tb.v2 ? truth : tb.v2;
#pragma unroll
            for (int q = 0; q < kHierItems; ++q) {
                if ((S << kBlockShift) + q * kHierThreads + tid == B) bb[q] = truth;
                if ((uint32_t)(q * kHierThreads) + tid == S) sb[q] = s_bound;
            }
            if (tid == 0) {
                // (lowered below an accepted entry's count: k_validate restores the bounds of
                //  accepted pairs that end up not being merged)
                __hip_atomic_store(&t.bmax[B], truth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&t.smax[S], s_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (truth >= tb.v2 && truth >= ts.v2) { cand = truth; cand_idx = te.i1; break; }
        }
        if (cand == 0ull) break;
        const uint32_t count = (uint32_t)(cand >> 32), key = ~(uint32_t)cand;
        const uint32_t a = key >> 16, b = key & 0xFFFFu;
        const bool single = count == 0 || a == b;
        if (k > 0) {
            bool conflict = single;
            uint32_t same_bucket = 0;
            for (uint32_t i = 0; i < accepted; ++i) {
                const uint32_t ai = s_keys[i] >> 16, bi = s_keys[i] & 0xFFFFu;
                conflict |= (b == ai) || (a == bi);
                same_bucket += pair_hash(kHashMul[0], ai, bi) == pair_hash(kHashMul[0], a, b);
            }
            if (conflict || same_bucket >= kBucketKeys) {       // (a lookup bucket holds kBucketKeys keys)
                if (tid == 0) { if (conflict) ctl->cut_conflict += 1; else ctl->cut_bucket += 1; }
                break;
            }
        }
        __syncthreads();
        if (tid == 0) {
            s_keys[accepted] = key;
            bs->key[accepted] = key;
            bs->eidx[accepted] = cand_idx;
            bs->packed[accepted] = cand;
            bs->maxp[accepted] = 0;
            bs->hash_mul = kHashMul[0];
            bs->byte_lut = 0;
            bs->skip_n = 0;                 // (this kernel ends the batch at a dependent pair
            bs->tt_index = kNoTT;           //  and merges a (t,t) pair alone)
            bs->tt_n = 0;
            best[k0 + accepted] = cand;
        }
        ++accepted;
        __syncthreads();
        if (single) { if (tid == 0) ctl->cut_single += 1; break; }
        if (accepted == limit && tid == 0) ctl->cut_full += 1;
    }
    if (tid == 0) {
        ctl->batch_n = accepted;
        ctl->commit_n = accepted;      // k_validate lowers it for multi-pair batches
        ctl->fused = want_fused(bs, accepted, fused_min, ctl->n_live * n_ranks) ? 1u : 0u;
        if (accepted) {
            ctl->n_batches += 1;
            ctl->n_sel_fallback += 1;
            // prime the threshold selection: assume the next batch spans about the same range of counts
            const unsigned long long c_hi = bs->packed[0] >> 32, c_lo = bs->packed[accepted - 1] >> 32;
            const unsigned long long spread = c_hi - c_lo > 0 ? c_hi - c_lo : 1;
            ctl->sel_T = c_lo > spread ? (c_lo - spread) << 32 : 0ull;
        }
    }
}

// ---- threshold selection ------------------------------------------------------------------
// The bound-walking kernel above pays a few dependent block reductions per selected pair.
// With dozens of pairs per batch it is cheaper to gather, with the whole chip, EVERY entry
// whose packed (count, ~key) is >= a threshold T (only blocks whose bound is >= T are read, and
// their bounds are tightened on the way), sort the few hundred survivors and take the
// independent prefix.  Any T gives a correct batch: entries that were not gathered rank below
// all gathered ones.  T only decides how many candidates come back; k_sel_pick adapts it, and
// leaves the batch to k_select_batch when the list is empty or overflowed.

__global__ __launch_bounds__(256) void k_sel_scan(PairTable t, DevCtl *ctl, SelList *sel, uint32_t n_target, uint32_t sel_cap,
                                                  int attempt) {
    if (attempt > 0 && (ctl->sel_ok || !ctl->sel_retry)) return;
    const unsigned long long T = ctl->sel_T;
    const bool bounds_only = ctl->sel_mode != 0;
    const uint32_t k_limit = ctl->k_limit < n_target ? ctl->k_limit : n_target;
    if (T == 0ull || ctl->k_done >= k_limit) return;
    const uint32_t n = table_size(t, ctl);
    const uint32_t n_blocks = (n + kBlockSize - 1) >> kBlockShift;
    const uint32_t lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave, n_waves = gridDim.x * blockDim.x / kWave;
    // Blocks are dealt out round robin (block B belongs to wave B mod n_waves): the blocks above
    // the threshold are usually neighbours (the same few first tokens), and a wave reads its blocks
    // one after the other.  The price is that the 64 bounds a wave looks at per step are strided.
    uint32_t blocks_read = 0;
    for (uint64_t step = 0; step * n_waves * kWave < n_blocks; ++step) {
        // far more entries than the list holds: no point in reading on, k_sel_pick will look for a
        // better threshold among the block bounds
        if (!bounds_only &&
            __hip_atomic_load(&ctl->sel_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4u * sel_cap) break;
        const uint64_t B64 = (step * kWave + lane) * n_waves + wave;
        const uint32_t B = (uint32_t)B64;
        // (the bound of the block's super-block first: 8 KB that stay in the cache, and nearly every super-block
        //  lies below the threshold -- the candidates sit in the rows of a few frequent first tokens)
        const bool look = B64 < n_blocks &&
            __hip_atomic_load(&t.smax[B >> kBlockShift], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= T;
        const unsigned long long bound = look ? __hip_atomic_load(&t.bmax[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        unsigned long long todo = __ballot(bound >= T);
        if (bounds_only) {
            // looking for a threshold: list the block bounds themselves (8 bytes per 1024 entries)
            if (todo) {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(&ctl->sel_n, (uint32_t)__popcll(todo));
                at = rfl(at) + (uint32_t)__popcll(todo & lt_mask);
                if (bound >= T && at < sel_cap) { sel->packed[at] = bound; sel->eidx[at] = B; }
            }
            continue;
        }
        while (todo) {
            if (__hip_atomic_load(&ctl->sel_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4u * sel_cap) break;
            const uint32_t b = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t blk = (uint32_t)((step * kWave + b) * n_waves + wave);
            unsigned long long mx = 0;
            // (all 16 loads of the block are issued before the first one is looked at)
            unsigned long long pv[kBlockSize / kWave];
#pragma unroll
            for (uint32_t q = 0; q < kBlockSize / kWave; ++q) {
                const uint32_t e = (blk << kBlockShift) + q * kWave + lane;
                pv[q] = e < n ? entry_packed(t, e) : 0ull;
            }
#pragma unroll
            for (uint32_t q = 0; q < kBlockSize / kWave; ++q) {
                const uint32_t e = (blk << kBlockShift) + q * kWave + lane;
                const unsigned long long p = pv[q];
                mx = p > mx ? p : mx;
                const unsigned long long hit = __ballot(p >= T && p != 0ull);
                if (hit) {
                    uint32_t at = 0;
                    if (lane == 0) at = atomicAdd(&ctl->sel_n, (uint32_t)__popcll(hit));
                    at = rfl(at) + (uint32_t)__popcll(hit & lt_mask);
                    if (p >= T && p != 0ull && at < sel_cap) { sel->packed[at] = p; sel->eidx[at] = e; }
                }
            }
            mx = wave_max_u64(mx);
            if (lane == 0) __hip_atomic_store(&t.bmax[blk], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++blocks_read;
        }
    }
    if (lane == 0 && blocks_read) atomicAdd(&ctl->n_sel_blocks, (unsigned long long)blocks_read);
}

constexpr int kPickThreads = 1024;

__global__ __launch_bounds__(kPickThreads) void k_sel_pick(DevCtl *ctl, BatchState *bs, const SelList *sel,
                                                           unsigned long long *best, uint32_t n_target,
                                                           uint32_t max_batch, uint32_t fused_min,
                                                           uint32_t n_ranks, int attempt, uint32_t fake_id, uint32_t sel_cap,
                                                           uint32_t tt_max, uint32_t byte_table) {
    __shared__ unsigned long long sp[kSelCap];
    __shared__ uint32_t si[kSelCap];
    // what the members accepted so far occupy (the independence test of a candidate is then a few LDS reads,
    // whatever the batch size): is a token the FIRST / SECOND element of a member (exact bitmaps), how many
    // members have it there (8-bit counts, ids folded to 15 bits: only the pass-over prediction uses them),
    // and how many keys every bucket of the kernels' lookup table holds -- under each of the kHashSeeds hash
    // multipliers still in the race (small packed counters)
    __shared__ uint32_t set_first[2048], set_second[2048];
    __shared__ uint32_t g_first[8], g_second[8];      // the same for the 64 candidates of a bulk step (raw bytes only)
    constexpr uint32_t kFillBits = kBucketKeys <= 3 ? 2 : 4, kFillPerWord = 32 / kFillBits;
    __shared__ uint32_t bucket_fill[kHashSeeds][kBuckets / kFillPerWord];
    __shared__ uint16_t acc_ci[kBatchMax];         // list position of every accepted member (written out after the walk)
    const uint32_t tid = threadIdx.x;
    if (attempt > 0 && (ctl->sel_ok || !ctl->sel_retry)) return;
    const uint32_t k0 = ctl->k_done;
    const uint32_t k_limit = ctl->k_limit < n_target ? ctl->k_limit : n_target;
    const uint32_t n_all = ctl->sel_n;          // entries >= T (the list holds the first kSelCap of them)
    const unsigned long long T = ctl->sel_T;
    // (the batch size the candidate window is sized for: what validation has let through lately, at most the caller's cap)
    const uint32_t adapt = min(ctl->adapt_limit ? ctl->adapt_limit : (uint32_t)kBatchMax, max_batch < 16u ? 16u : max_batch);
    const bool bounds_only = ctl->sel_mode != 0;
    __syncthreads();
    if (tid == 0) { ctl->sel_n = 0; ctl->sel_ok = 0; ctl->sel_retry = 0; ctl->sel_mode = 0; }
    if (k0 >= k_limit) {                       // nothing to select: the walking kernel returns at once too
        if (tid == 0) { ctl->batch_n = 0; ctl->commit_n = 0; ctl->fused = 0; }
        return;
    }
    if (T == 0ull || n_all == 0) {
        // not primed, or nothing at or above T: look for a threshold among the bounds of all blocks
        // that hold a pair with a count (if that finds nothing either, the walking kernel takes over)
        if (tid == 0 && attempt == 0 && !bounds_only) {
            ctl->sel_T = 1ull << 32;
            ctl->sel_mode = 1;
            ctl->sel_retry = 1;
        }
        return;
    }
    const uint32_t n_l = n_all < sel_cap ? n_all : sel_cap;
    uint32_t n_sort = 64;                       // power of two >= n_l
    while (n_sort < n_l) n_sort <<= 1;
    for (uint32_t i = tid; i < n_sort; i += kPickThreads) {
        sp[i] = i < n_l ? sel->packed[i] : 0ull;
        si[i] = i < n_l ? sel->eidx[i] : 0u;
    }
    for (uint32_t i = tid; i < 2048u; i += kPickThreads) { set_first[i] = 0; set_second[i] = 0; }
    if (tid < 8u) { g_first[tid] = 0; g_second[tid] = 0; }
    for (uint32_t i = tid; i < (uint32_t)kHashSeeds * (kBuckets / kFillPerWord); i += kPickThreads) (&bucket_fill[0][0])[i] = 0;
    __syncthreads();
    // bitonic sort, descending by packed value (unique per pair: a total order)
    for (uint32_t k = 2; k <= n_sort; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n_sort; i += kPickThreads) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const unsigned long long a = sp[i], b = sp[l];
                    if (desc ? a < b : a > b) {
                        sp[i] = b; sp[l] = a;
                        const uint32_t t = si[i]; si[i] = si[l]; si[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (bounds_only) {
        // The list holds block bounds, not entries (second attempt, after the entry gather
        // overflowed: thousands of pairs share the count at the threshold, and reading every
        // block that holds one of them would cost gigabytes).  Every block contributes at least
        // its maximum, so the bound ranked several batches down is a threshold that brings back
        // enough entries; the third attempt gathers them.
        if (tid == 0 && attempt < 2) {
            unsigned long long r = 3ull * (adapt < 32u ? 32u : adapt);     // blocks; each holds >= 1 entry, often several
            if (r > sel_cap / 4) r = sel_cap / 4;
            if (n_all > sel_cap) r = (unsigned long long)sel_cap * r / n_all;
            r = r < 1 ? 1 : r;
            if (r > n_l - 1) r = n_l - 1;
            ctl->sel_T = sp[r];
            ctl->sel_retry = 1;
        }
        return;
    }
    if (n_all > sel_cap) {
        if (tid == 0 && attempt == 0) {       // overflow: look at the block bounds next
            ctl->sel_mode = 1;
            ctl->sel_retry = 1;
            ctl->n_sel_retry += 1;
        }
        return;
    }
    uint32_t limit = k_limit - k0;
    if (limit > max_batch) limit = max_batch;
    if (limit > (uint32_t)kBatchMax) limit = kBatchMax;
    if (limit > adapt) limit = adapt;
    // the independent prefix (sequential by nature; everything it looks at is in LDS).  One wave runs it with uniform
    // control flow: lane s keeps the bucket fills of hash multiplier s, lane 0 does the stores
    if (tid < (uint32_t)kWave) {
        const bool l0 = tid == 0;
        const bool seed_lane = tid < (uint32_t)kHashSeeds;
        const uint32_t my_mul = kHashMul[tid & 7u];
        uint32_t accepted = 0;
        uint32_t cut = 0;      // 1 conflict, 2 bucket, 3 single
        uint32_t n_skip = 0, ci = 0;
        // (on text a dependent pair often does NOT fall behind -- few of its occurrences overlap the
        //  earlier pair's -- and a failed pass-over costs a stream pass; after a failure dependent
        //  pairs end the batch again for a while, see k_seq_finish)
        const bool first_mode = ctl->first_mode != 0;      // (no (t,t) members, no equal counts)
        const bool skip_allowed = ctl->skip_off == 0;
        if (first_mode) tt_max = 0;
        if (l0) ctl->first_tie = 0;
        // a passed-over candidate is expected to lose at least this fraction of its count; a member that
        // would still rank behind it ends the batch before the pass instead of failing validation after it
        const unsigned long long red_q16 = ((unsigned long long)MBPE_PRED_NUM * ctl->skip_red_q16) / 4ull;
        unsigned long long skip_floor = 0;
        uint32_t n_tt = 0;                          // (t,t) members so far, and the map slots they occupy
        unsigned long long tt_slots[kTTSlots / 64] = {};
        uint32_t alive = (1u << kHashSeeds) - 1u;   // hash multipliers under which every bucket still holds its keys
        // every member so far is a pair of raw bytes (a (t,t) member: its token is): such a batch is looked up in the direct
        // byte x byte table and needs no room in the hash buckets (BatchState::byte_lut)
        bool all_bytes = kByteTable && byte_table != 0u;
        uint32_t tracked = 0, tracked_tt = 0;       // members (and (t,t) members among them) the bucket fills know
        if (tid == 0) bs->tt_index = kNoTT;
        unsigned long long cand_next = n_l ? sp[0] : 0ull;         // (the next candidate is read one step ahead)
        uint32_t next_bulk = 0;                     // a bulk step is tried again from this list position on
        for (; accepted < limit && ci < n_l; ++ci) {
            // Bulk step: while every member is a pair of raw bytes (no bucket bookkeeping) the next 64 candidates are
            // looked at one per lane and accepted together when the one-by-one walk would accept every one of them:
            // plain byte pairs with a count, none dependent on a member or on another candidate of the 64 (a token that
            // is first element of one and second of another: tested against bitmaps of the 64, which flags every pair of
            // candidates that depend on each other), none below the floor of a passed-over candidate.  Anything
            // else: the walk takes these candidates one by one and tries again 64 positions on.
            if (all_bytes && !first_mode && ci >= next_bulk && accepted + kWave <= limit && ci + kWave <= n_l) {
                const unsigned long long gc = sp[ci + tid];
                const uint32_t gcount = (uint32_t)(gc >> 32), gkey = ~(uint32_t)gc, ga = gkey >> 16, gb = gkey & 0xFFFFu;
                bool bad = gcount == 0u || ga == gb || ga >= 256u || gb >= 256u || gc < skip_floor;
                if (!bad) {
                    bad = (((set_first[gb >> 5] >> (gb & 31u)) | (set_second[ga >> 5] >> (ga & 31u))) & 1u) != 0u;
                    atomicOr(&g_first[ga >> 5], 1u << (ga & 31u));
                    atomicOr(&g_second[gb >> 5], 1u << (gb & 31u));
                }
                const bool any_bad = __ballot(bad) != 0ull;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (one wave: its LDS operations execute in order)
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                bool dep = false;
                if (!any_bad) dep = (((g_first[gb >> 5] >> (gb & 31u)) | (g_second[ga >> 5] >> (ga & 31u))) & 1u) != 0u;
                const bool take = !any_bad && __ballot(dep) == 0ull;
                if (take) {
                    atomicOr(&set_first[ga >> 5], 1u << (ga & 31u));
                    atomicOr(&set_second[gb >> 5], 1u << (gb & 31u));
                    acc_ci[accepted + tid] = (uint16_t)(ci + tid);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (tid < 8u) { g_first[tid] = 0; g_second[tid] = 0; }
                if (take) {
                    accepted += kWave;
                    ci += kWave - 1u;
                    cand_next = sp[ci + 1u < n_l ? ci + 1u : ci];
                    continue;
                }
                next_bulk = ci + kWave;
            }
            const unsigned long long cand = cand_next;
            cand_next = sp[ci + 1u < n_l ? ci + 1u : ci];
            const uint32_t count = (uint32_t)(cand >> 32), key = ~(uint32_t)cand;
            const uint32_t a = key >> 16, b = key & 0xFFFFu;
            const bool tt = a == b && count != 0;      // (t,t): a member like any other once renamed (tt_rename),
            const uint32_t slot = a & (kTTSlots - 1u);                //  while stand-in ids and map slots last
            bool single = count == 0 || (tt && (n_tt >= tt_max || ((tt_slots[slot >> 6] >> (slot & 63u)) & 1ull)));
            if (first_mode) {
                // a count shared with the next candidate (or possibly with one beyond the list) cannot be ranked by
                // count: such a pair goes alone, with the position tie-break, or ends the batch before it
                // (behind the end of the list everything is below the gather's threshold T)
                const bool tie = ci + 1u < n_l ? (uint32_t)(cand_next >> 32) == count : (uint32_t)(T >> 32) >= count;
                if (tie && accepted > 0) { cut = 1u; break; }
                if (tie) { single = true; if (l0) ctl->first_tie = 1; }
            }
            // A pair of raw bytes in a batch of such pairs needs no room in the hash buckets (byte table); the bucket fills
            // then lag behind and are caught up with -- members `tracked` .. accepted - 1 -- when a candidate needs them.
            const bool byte_cand = all_bytes && a < 256u && b < 256u;
            uint32_t my_h = 0, full = 0;
            if (!byte_cand) {
                for (; tracked < accepted; ++tracked) {
                    const unsigned long long mc = sp[acc_ci[tracked]];
                    const uint32_t mk = ~(uint32_t)mc, ma = mk >> 16, mb = mk & 0xFFFFu;
                    const bool mtt = ma == mb && (uint32_t)(mc >> 32) != 0u;
                    const uint32_t mh = pair_hash(my_mul, ma, mtt ? fake_id - tracked_tt : mb);
                    tracked_tt += mtt ? 1u : 0u;
                    uint32_t mf = 0;
                    if (seed_lane)
                        mf = (bucket_fill[tid][mh / kFillPerWord] >> ((mh % kFillPerWord) * kFillBits)) & ((1u << kFillBits) - 1u);
                    alive &= ~(uint32_t)__ballot(seed_lane && mf >= kBucketKeys);
                    if (seed_lane && ((alive >> tid) & 1u))
                        bucket_fill[tid][mh / kFillPerWord] += 1u << ((mh % kFillPerWord) * kFillBits);
                }
                // multipliers under which this key would be the bucket's (kBucketKeys + 1)-th: they drop out if it is accepted
                const uint32_t b_hash = tt ? fake_id - n_tt : b;
                my_h = pair_hash(my_mul, a, b_hash);
                uint32_t my_f = 0;
                if (seed_lane)
                    my_f = (bucket_fill[tid][my_h / kFillPerWord] >> ((my_h % kFillPerWord) * kFillBits)) & ((1u << kFillBits) - 1u);
                full = (uint32_t)__ballot(seed_lane && my_f >= kBucketKeys);
            }
            if (accepted > 0) {
                // dependent on an earlier member (c, d): b == c or a == d
                const bool conf = (((set_first[b >> 5] >> (b & 31u)) | (set_second[a >> 5] >> (a & 31u))) & 1u) != 0u;
                const bool no_bucket = !byte_cand && (alive & ~full) == 0u;
                if (single) { cut = 3u; break; }
                if (conf && n_skip < (uint32_t)kSkipMax && skip_allowed) {
                    // Depends on an earlier member (shares a token with it the wrong way round): the
                    // earlier merge eats some of its occurrences, so its count will have dropped by the
                    // time it could be chosen -- normally below the whole batch.  Pass it over;
                    // k_validate checks from the measured deltas that it really fell behind every
                    // member chosen after this point, and cuts the batch here otherwise.
                    if (l0) {
                        bs->skip_key[n_skip] = key;
                        bs->skip_pos[n_skip] = accepted;
                        bs->skip_packed[n_skip] = cand;
                    }
                    ++n_skip;
                    // (it loses occurrences to every member it depends on: red_q16 is the fraction per such member)
                    uint32_t n_dep = 0;                   // members it depends on: first element b, or second element a
                    for (uint32_t i = tid; i < accepted; i += kWave) {
                        const uint32_t mk = ~(uint32_t)sp[acc_ci[i]];
                        n_dep += ((mk >> 16) == b ? 1u : 0u) + ((mk & 0xFFFFu) == a ? 1u : 0u);
                    }
                    n_dep = wave_sum(n_dep);
                    unsigned long long lose = (unsigned long long)n_dep * red_q16;
                    if (lose > 65536ull) lose = 65536ull;
                    const unsigned long long keep = (unsigned long long)count - (((unsigned long long)count * lose) >> 16);
                    const unsigned long long fl = (keep << 32) | (uint32_t)cand;
                    skip_floor = fl > skip_floor ? fl : skip_floor;
                    continue;
                }
                if (conf || no_bucket) { cut = conf ? 1u : 2u; break; }
                if (cand < skip_floor) { cut = 1u; break; }
            }
            all_bytes = byte_cand;
            if (!byte_cand) {
                alive &= ~full;
                if (seed_lane && ((alive >> tid) & 1u))
                    bucket_fill[tid][my_h / kFillPerWord] += 1u << ((my_h % kFillPerWord) * kFillBits);
                tracked = accepted + 1u;
                tracked_tt = n_tt + (tt ? 1u : 0u);
            }
            if (l0) {
                atomicOr(&set_first[a >> 5], 1u << (a & 31u));          // (LDS atomics: no read to wait for)
                atomicOr(&set_second[b >> 5], 1u << (b & 31u));
                acc_ci[accepted] = (uint16_t)ci;
                if (tt && n_tt == 0) { bs->tt_index = accepted; bs->tt_token = a; }
            }
            if (tt) { tt_slots[slot >> 6] |= 1ull << (slot & 63u); ++n_tt; }
            ++accepted;
            if (single) { cut = 3u; ++ci; break; }
        }
        // (a batch that goes through the hash table after all -- a single byte pair: no table at all -- is caught up with;
        //  nothing can drop out any more: every member behind a non-byte one was tracked when it was accepted)
        // the members' records, all lanes at once
        // (and the sum of their counts for want_fused: one thread reading the records back from global memory cost a
        //  round trip per pair on text, where batches stay below fused_min)
        unsigned long long count_sum = 0;
        for (uint32_t i = tid; i < accepted; i += kWave) {
            const uint32_t at = acc_ci[i];
            const unsigned long long cand = sp[at];
            bs->key[i] = ~(uint32_t)cand;
            bs->eidx[i] = si[at];
            bs->packed[i] = cand;
            bs->maxp[i] = 0;
            best[k0 + i] = cand;
            count_sum += cand >> 32;
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) count_sum += shfl_xor_u64(count_sum, d);
        // (candidates passed over behind the last member do not matter: nothing was chosen after them)
        if (tid == 0) {
            bs->skip_n = n_skip; bs->tt_n = n_tt; ctl->n_skipped += n_skip;
            bs->hash_mul = kHashMul[alive ? (uint32_t)__builtin_ctz(alive) : 0u];
            bs->byte_lut = all_bytes && accepted >= 2u ? 1u : 0u;
        }
        if (tid == 0) {
            ctl->batch_n = accepted;
            ctl->commit_n = accepted;
            ctl->fused = want_fused_sum(count_sum, accepted, fused_min, ctl->n_live * n_ranks) ? 1u : 0u;
            ctl->n_batches += 1;
            ctl->sel_ok = 1;
            if (cut == 1u) ctl->cut_conflict += 1;
            else if (cut == 2u) ctl->cut_bucket += 1;
            else if (cut == 3u) ctl->cut_single += 1;
            if (cut == 0u && accepted == limit) ctl->cut_full += 1;
            // next threshold: about 128 candidates beyond this batch, or a window twice as wide
            // when the list ended before the batch was full
#ifndef MBPE_SEL_WINDOW
#define MBPE_SEL_WINDOW 1
#endif
#ifndef MBPE_SEL_AIM_NEXT
#define MBPE_SEL_AIM_NEXT 0      /* eighths of the list that the entries BEHIND a batch may fill (0: three quarters, batch included) */
#endif
#ifndef MBPE_SEL_NEXT_LIMIT
#define MBPE_SEL_NEXT_LIMIT 1
#endif
#ifndef MBPE_SEL_TINY
#define MBPE_SEL_TINY 64
#endif
#ifndef MBPE_SEL_WINDOW_MUL
#define MBPE_SEL_WINDOW_MUL 8u
#endif
#ifndef MBPE_SEL_WINDOW_MIN
#define MBPE_SEL_WINDOW_MIN 256u
#endif
#ifndef MBPE_SEL_AHEAD
#define MBPE_SEL_AHEAD 2        /* halves of the batch-size limit that the next candidate list should reach beyond a batch */
#endif
#if MBPE_SEL_WINDOW
            // the window follows the batches that are being chosen (eight times their running mean, at least 256), not only
            // the limit: on text a sequence takes a few dozen pairs and a list of thousands is sorted for nothing
            const uint32_t recent = (3u * ctl->recent_n + accepted + 3u) / 4u;
            ctl->recent_n = recent;
            const uint32_t wide = MBPE_SEL_WINDOW_MUL * recent < MBPE_SEL_WINDOW_MIN ? MBPE_SEL_WINDOW_MIN : MBPE_SEL_WINDOW_MUL * recent;
            // (a batch that filled its limit doubles it -- k_seq_finish -- so the next batch may be twice as large)
            uint32_t adapt_next = adapt;
#if MBPE_SEL_NEXT_LIMIT
            if (cut == 0u && accepted == limit && 2u * adapt_next <= max_batch) adapt_next *= 2u;
#endif
            const uint32_t adapt_w = adapt_next < wide ? adapt_next : wide;
#else
            const uint32_t adapt_w = adapt;
#endif
            const uint32_t want = ci + ((uint32_t)MBPE_SEL_AHEAD * (adapt_w < 16u ? 16u : adapt_w)) / 2u;
            if (n_l > want) {
                ctl->sel_T = sp[want];          // (the full packed value: also cuts inside a run of equal counts)
            } else {
                // The list did not reach that far: open the window by at least 1/64 of the count (a
                // short list has a tiny spread, and doubling it batch after batch would take many
                // passes of a handful of merges each).  Too wide is cheap: the gather stops early and
                // the block bounds give the threshold.
                const unsigned long long c_hi = sp[0] >> 32, c_lo = sp[n_l - 1] >> 32;
                unsigned long long spread = c_hi - c_lo > 0 ? c_hi - c_lo : 1;
                if (n_l >= 256u) {
                    // a long list tells how dense the counts are here (uniform data: thousands of pairs within
                    // a few hundred counts, where 1/64 of the count would bring back the whole table and send
                    // every selection through the overflow path): extend by what that density needs
                    // (aiming at a list of want + adapt entries, but never at more than three quarters of what the list
                    //  holds: an overflowing gather costs two more scans and a worse threshold)
#if MBPE_SEL_AIM_NEXT
                    // (what this batch took is gone from the next list: the bound is on what comes BEHIND it)
                    unsigned long long aim_next = (unsigned long long)(want - ci) + adapt_w;
                    if (aim_next > (unsigned long long)MBPE_SEL_AIM_NEXT * sel_cap / 8ull) aim_next = (unsigned long long)MBPE_SEL_AIM_NEXT * sel_cap / 8ull;
                    const unsigned long long aim = (unsigned long long)ci + aim_next;
#else
                    unsigned long long aim = (unsigned long long)want + adapt_w;
                    if (aim > 3ull * sel_cap / 4ull) aim = 3ull * sel_cap / 4ull;
#endif
                    const unsigned long long need = aim > (unsigned long long)n_l + 64ull ? aim - n_l : 64ull;
                    spread = (spread * need + n_l - 1) / n_l;
                    if (spread < 1) spread = 1;
                } else if (n_l < (uint32_t)MBPE_SEL_TINY && accepted == n_l - n_skip) {
                    // A handful of candidates and every one of them taken (or passed over): the counts have a cliff
                    // below them -- the benchmark after the last of the 128 x 128 pairs: 65,000 then 33,000 -- and
                    // creeping down by 1/64 of the count costs a stream pass per step.  Halve the threshold: too many
                    // candidates come back, and the block bounds then give the right one (two more scans, no pass).
                    spread = c_lo / 2;
                } else if (spread < 1 + c_lo / 64) {
                    spread = 1 + c_lo / 64;
                }
                ctl->sel_T = c_lo > spread ? (c_lo - spread) << 32 : 1ull << 32;
            }
        }
    }
}


// (t,t) members of a batch.  The matches of such a pair are every second token of a run of t, counted
// from the run's start, that has a successor in the run.  Renaming, in registers only, the tokens at
// the odd positions of every run of a member's token to an id that no token has (idmask - 1 - i for
// the i-th member) turns the pair into an ordinary one, (t, stand-in): no two of its occurrences
// overlap, every renamed token is the second token of a match and disappears, and all the batch
// machinery applies unchanged.  The position of a token in its run does not depend on which token it
// is -- it is the number of tokens equal to it immediately before it -- so ONE segmented scan over the
// lanes serves every member: per lane (last token, length of the run it ends, "the whole lane is that
// run"), joined from lane to lane, started from the run that ends the previous tile (run_in, from
// k_run_final).  The neighbour tokens taken from the summaries are renamed alike.  A token that ends
// its chunk can be the second token of a match but continues no run.
template <int MODE>
__device__ __forceinline__ void tt_rename(uint32_t s[8], Halo &h, const TTInfo &ti, uint32_t rb_small) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    const uint32_t lane = lane_id();
    auto stand_in = [&](uint32_t v) -> uint32_t {      // v: a live token at an odd position of its run
        const uint32_t id = v & idmask;
        const uint32_t e = ti.map[id & (kTTSlots - 1u)];
        return (e & 0xFFFFu) == id ? (e >> 16) | (v & endbit) : v;
    };
    // 1. the lane's own structure.  v continues the run that `last` ends iff (v & idmask) == last
    //    (a `last` with the chunk-end bit equals no id)
    //    Also, per slot: is its position odd counted from the lane's start (lodd), and does it belong to
    //    the lane's leading run (lead) -- only those slots' positions depend on what came before the lane.
    uint32_t last = kHole, len = 0, first_id = kHole, lodd = 0, lead = 0;
    bool full = true, repeats = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (s[j] != kHole) {
            if (last == kHole) first_id = s[j] & idmask;
            if ((s[j] & idmask) == last) { ++len; repeats = true; }
            else { full = full && last == kHole; len = 1; }
            lodd |= (len & 1u) ? 0u : 1u << j;
            lead |= full ? 1u << j : 0u;
            last = s[j];
        }
    }
    // 2. what ends just before the tile, folded into lane 0; only the parity of a length is ever used
    const bool p1_runs = h.p1 != kHole && !(h.p1 & endbit);
    const uint32_t in_last = h.p1, in_len = p1_runs ? (rb_small & 1u) : 1u;
    // (nearly every tile once the tokens are many: no live token equals the one before it, and every lane
    //  holds one, so no token sits at an odd position and the tile ends on a run of one)
    uint32_t rl, rn;
    const bool cross = first_id == wave_from_prev(last, in_last);
    if (__ballot(last == kHole) == 0ull && __ballot(repeats || cross) == 0ull) {
        rl = rlane(last, kWave - 1);
        rn = 1u;
    } else {
    auto join = [&](uint32_t o_last, uint32_t o_len, bool o_full) {       // o before (last, len, full)
        if (last == kHole) { last = o_last; len = o_len; full = o_full; }
        else if (o_last != kHole) {
            if (full && (last & idmask) == o_last) { len += o_len; full = o_full; }
            else full = false;
        }
    };
    // 3. inclusive scan over the lanes -- only needed when some lane has to pass a run on that began before
    //    it: an empty lane, or one that is a single run continuing the previous lane's (otherwise the state
    //    before a lane is just the previous lane's own last token and trailing run)
    if (__ballot(last == kHole || (full && cross)) != 0ull) {
        if (lane == 0) join(in_last, in_len, false);
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t ol = __shfl_up(last, d, kWave), on = __shfl_up(len | (full ? 0x80000000u : 0u), d, kWave);
            if (lane >= (uint32_t)d) join(ol, on & 0x7FFFFFFFu, (on >> 31) != 0u);
        }
    }
    // 4. every slot's position parity: the leading run's positions shift by the length of the run that the
    //    lane's first token continues
    const uint32_t lv = wave_from_prev(last, in_last), ln = wave_from_prev(len, in_len);
    const uint32_t odd = lodd ^ ((first_id == lv && (ln & 1u)) ? lead : 0u);
    if (__ballot(odd != 0u) != 0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((odd >> j) & 1u) s[j] = stand_in(s[j]);
    }
    // (after the scan a lane's (last, len) is the state at its end; without it no lane continued its neighbour's
    //  run as a whole, so the lane's own values are that state already)
    rl = rlane(last, kWave - 1);
    rn = rlane(len, kWave - 1);
    }
    // 5. the two tokens after the tile continue the count ...
    if (h.n1 != kHole) {
        const uint32_t n1 = h.n1;
        const bool c1 = (n1 & idmask) == rl;
        if (c1 && (rn & 1u)) h.n1 = stand_in(n1);
        rn = c1 ? rn + 1u : 1u;
        rl = n1;
        if (h.n2 != kHole && (h.n2 & idmask) == rl && (rn & 1u)) h.n2 = stand_in(h.n2);
    }
    // ... and the two before it end a run of rb tokens: p1 is number rb - 1, p2 number rb - 2
    if (p1_runs) {
        const uint32_t p1 = h.p1, p2 = h.p2;
        if (!(rb_small & 1u)) h.p1 = stand_in(p1);
        if (p2 == p1 && (rb_small & 1u)) h.p2 = stand_in(p2);
    }
}

// Can tt_rename change anything in this tile?  It renames tokens at ODD positions of runs of a (t,t) member's
// token, and a token at an odd position equals the live token before it.  So unless some live token of the
// tile (or of its halo) has the same id as its predecessor AND that id is a member's token, the tile is left
// alone -- which is nearly always: two per tile of random bytes repeat their predecessor, one in sixteen of
// those is one of the batch's handful of (t,t) tokens.  Conservative (chunk ends are ignored).
// nxt[j]: the live token after slot j (kHole: none); tile_first: the tile's first live token.
template <int MODE>
__device__ __forceinline__ bool tt_needed(const uint32_t s[8], const uint32_t nxt[8], const Halo &h,
                                          uint32_t tile_first, const TTInfo &ti, uint32_t live_mask = 0xFFu) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    auto member = [&](uint32_t v) -> bool {
        const uint32_t id = v & idmask;
        return (ti.map[id & (kTTSlots - 1u)] & 0xFFFFu) == id;
    };
    uint32_t eqm = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) eqm |= (((s[j] ^ nxt[j]) & idmask) == 0u ? 1u : 0u) << j;     // (a hole has no token's id)
    eqm &= live_mask;            // (prefix-form callers pass the next SLOT: a hole is followed by a hole)
    bool need = false;
    if (__ballot(eqm != 0u) != 0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((eqm >> j) & 1u) need = need || member(s[j]);
    }
    // halo (uniform): p1 before the tile's first token, p2 before p1, n2 after n1
    bool hneed = false;
    if (h.p1 != kHole && (((h.p1 ^ tile_first) & idmask) == 0u || (h.p2 != kHole && ((h.p2 ^ h.p1) & idmask) == 0u)))
        hneed = member(h.p1);
    if (h.n1 != kHole && h.n2 != kHole && ((h.n1 ^ h.n2) & idmask) == 0u) hneed = hneed || member(h.n1);
    return __ballot(need) != 0ull || hneed;
}

// exact neighbours of every slot, two deep on both sides (shared by the scan
// and rewrite passes of a batch)
struct Neigh {
    uint32_t p1_in, p2_in;      // live tokens before this lane's first slot
    uint32_t n1v[8], n2v[8];    // next / second-next live token of every slot
};

__device__ __forceinline__ Neigh tile_neighbours(const uint32_t s[8], const Halo &h) {
    const uint32_t lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned long long gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    uint32_t cnt = 0, f1 = kSent, f2 = kSent, l1 = kSent, l2 = kSent;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (s[j] != kHole) {
            if (cnt == 0) f1 = s[j]; else if (cnt == 1) f2 = s[j];
            l2 = l1; l1 = s[j];
            ++cnt;
        }
    }
    const unsigned long long m_live = __ballot(cnt > 0);
    Neigh nb;
    {
        const unsigned long long lo = m_live & lt_mask;
        const uint32_t src1 = lo ? 63u - (uint32_t)__builtin_clzll(lo) : lane;
        const uint32_t sl1 = __shfl(l1, src1, kWave), sl2 = __shfl(l2, src1, kWave);
        const unsigned long long lo2 = lo & ~(1ull << src1);
        const uint32_t src2 = lo2 ? 63u - (uint32_t)__builtin_clzll(lo2) : lane;
        const uint32_t tl1 = __shfl(l1, src2, kWave);
        nb.p1_in = lo ? sl1 : h.p1;
        nb.p2_in = lo ? (sl2 != kSent ? sl2 : (lo2 ? tl1 : h.p1)) : h.p2;
    }
    uint32_t n1_in, n2_in;
    {
        const unsigned long long hi = m_live & gt_mask;
        const uint32_t src1 = hi ? (uint32_t)__builtin_ctzll(hi) : lane;
        const uint32_t sf1 = __shfl(f1, src1, kWave), sf2 = __shfl(f2, src1, kWave);
        const unsigned long long hi2 = hi & ~(1ull << src1);
        const uint32_t src2 = hi2 ? (uint32_t)__builtin_ctzll(hi2) : lane;
        const uint32_t tf1 = __shfl(f1, src2, kWave);
        n1_in = hi ? sf1 : h.n1;
        n2_in = hi ? (sf2 != kSent ? sf2 : (hi2 ? tf1 : h.n1)) : h.n2;
    }
    uint32_t x1 = n1_in, x2 = n2_in;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        nb.n1v[j] = x1;
        nb.n2v[j] = x2;
        if (s[j] != kHole) { x2 = x1; x1 = s[j]; }
    }
    return nb;
}

// The counting half of a multi-pair merge on one tile: deltas per pair and the
// tile's mark.  Nothing is rewritten.  Batch pairs cannot overlap (no token is
// both a first and a second element), so "this token is the second of a match"
// is simply "the previous live token started a match": one membership test per
// live slot, everything else only where a match is.
template <int MODE, int DIAG = 0>
__device__ __forceinline__ void scan_tile_full(uint32_t *chg, uint32_t tile, const uint32_t s[8], const Halo h,
                                               const BatchLut &lut, uint32_t n_keys, uint32_t *hdr_adj,
                                               uint32_t *LR, uint32_t pitch, DeltaCache &dc, bool dc_on, bool tt_on,
                                               TTInfo &ti, uint32_t adj_pitch) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    const Neigh nb = tile_neighbours(s, h);
    if (DIAG == 4) {
        asm volatile("" :: "v"(nb.p1_in), "v"(nb.p2_in));
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(nb.n1v[j]), "v"(nb.n2v[j]));
        if (lane_id() == 0) atomicOr(&chg[tile >> 5], 1u << (tile & 31u));
        return;
    }
    bool any = false;
    uint32_t p1 = nb.p1_in, p2 = nb.p2_in;
    bool a1 = false;       // p1 started a match (so the current token is its second element)
    bool first = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t self = s[j];
        if (self == kHole) continue;
        const uint32_t n1 = nb.n1v[j], n2 = nb.n2v[j];
        if (first) { a1 = p1 != kHole && pair_test(lut, p1, self & idmask); first = false; }
        bool is_a = false;
        if (a1) {                                        // second token of a match of pair (p1, self)
            any = true;
            if (right_open<MODE>(self, n1) && !pair_test(lut, n1, n2 & idmask)) {
                const int jb = lut_index(lut, p1, self & idmask);
                if (DIAG != 3) dc_add(dc, dc_on, LR, lr_idx(pitch, n1 & idmask, (uint32_t)jb, 1), 1u);
                else asm volatile("" :: "v"(jb));
            }
        } else if (n1 != kHole && pair_test(lut, self, n1 & idmask)) {   // first token of a match
            is_a = true;
            any = true;
            const int ja = lut_index(lut, self, n1 & idmask);
            if (tt_on && (n1 & idmask) >= idmask - (uint32_t)kTTMax)      // a match of a (t,t) member: count it
                atomicAdd(&ti.cnt[idmask - 1u - (n1 & idmask)], 1u);
            if (left_open<MODE>(p1)) {
                if (p2 != kHole && pair_test(lut, p2, p1)) {               // two matches touch
                    const int jp = lut_index(lut, p2, p1);
                    atomicAdd(&hdr_adj[jp * adj_pitch + ja], 1u);
                } else {
                    if (DIAG != 3) dc_add(dc, dc_on, LR, lr_idx(pitch, p1, (uint32_t)ja, 0), 1u);
                    else asm volatile("" :: "v"(ja));
                }
            }
        }
        a1 = is_a;
        p2 = p1; p1 = self;
    }
    if (__ballot(any) != 0ull && lane_id() == 0) atomicOr(&chg[tile >> 5], 1u << (tile & 31u));
}

// (defined with k_fused_batch; WRITE false = count and mark only)
template <int MODE, int DIAG, bool WRITE, bool TT, class DC>
__device__ __forceinline__ uint4 fused_tile_pf(const uint4 q_orig, const uint32_t s[8], const Halo h,
                                               uint32_t old_x, uint32_t old_y, uint32_t old_z, const BatchLut &lut,
                                               uint32_t X0, uint32_t tile, TileSum *sout, uint32_t *hdr_adj, uint32_t *LR,
                                               DC &dc, bool dc_on, uint32_t &wave_rm, bool &wrote_sum,
                                               __amdgpu_buffer_rsrc_t lr_rsrc, uint32_t adj_pitch, uint16_t *stage,
                                               uint32_t *chg, bool renamed, TTInfo *ti, bool &need_rename,
                                               __amdgpu_buffer_rsrc_t t_rsrc = __amdgpu_buffer_rsrc_t(), bool use_t = false);

template <int MODE, bool HOT, bool TT, int DIAG = 0>
__global__ __launch_bounds__(kLutThreads) void k_scan_batch(const uint16_t *tok0, const uint16_t *tok1,
                                                              const TileSum *__restrict__ sin, uint32_t n_tiles,
                                                              uint32_t *__restrict__ chg, const BatchState *bs,
                                                              uint32_t *hdr_m, uint32_t *hdr_adj, uint32_t *LR,
                                                              const DevCtl *ctl, const RankEdge *le,
                                                              const RankEdge *re,
                                                              const uint32_t *__restrict__ run_in, int hot_launched,
                                                              uint32_t *T) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    __shared__ BatchLutMem lut_mem;
    const uint32_t lane = lane_id();
    const uint32_t waves_per_block = kLutThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    __shared__ DeltaCache dc;
    const uint32_t n_keys = ctl->batch_n;
    if (n_keys < 2 || ctl->fused) return;
    if ((bs->tt_index != kNoTT) != TT) return;      // (see k_fused_batch)
    __shared__ TTInfo ti;
    const uint16_t *tok = ctl->cur ? tok1 : tok0;
    const uint32_t ctl_k_done = rfl(ctl->k_done);
    const uint32_t pitch = rfl(lr_pitch(256u + ctl_k_done));
    const uint32_t adj_pitch = rfl(ctl->adj_pitch);
    if (hot_mismatch<HOT>(bs->packed[0] >> 32, ctl, hot_launched)) return;     // (see k_merge)
    constexpr bool dc_on = HOT;
    if (dc_on) dc_init(dc);
    BatchLut lut(&lut_mem, bs, idmask);
    lut_build(lut, bs, n_keys, idmask - 1u);
    if (TT) tt_build(ti, bs, n_keys, idmask - 1u);
    uint32_t tile = rfl(blockIdx.x * waves_per_block + threadIdx.x / kWave);
    if (tile < n_tiles) {

    const uint32_t last_tile = n_tiles - 1;
    auto clamp_tile = [&](uint64_t t) { return (uint32_t)(t < n_tiles ? t : last_tile); };
    const __amdgpu_buffer_rsrc_t sums_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TileSum *>(sin), 0, n_tiles * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t lr_rsrc = __builtin_amdgcn_make_buffer_rsrc(LR, 0, 0xFFFFFFFCu, 0x00020000);
    // (the pairs' byte x byte cell blocks: see k_pair_cells_fold; not with the delta cache of frequent pairs)
    const bool use_t = T != nullptr && !HOT && ctl->cells_on != 0u && n_keys >= ctl->cells_min;
    const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(T ? T : LR, 0, T ? 0xFFFFFFFCu : 0u, 0x00020000);
    TileIn t0 = tile_issue(tok, sums_rsrc, tile);
    TileIn t1 = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + n_waves));
    bool v1 = (uint64_t)tile + n_waves < n_tiles;
    const unsigned long long gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    for (;;) {
        const bool v2 = (uint64_t)tile + 2ull * n_waves < n_tiles;
        TileIn t2 = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + 2ull * n_waves));

        const uint32_t me_nlive = rlane(t0.smw, 6) & 0xFFFFu;
        if (DIAG == 1) {
            asm volatile("" :: "v"(t0.q.x), "v"(t0.q.y), "v"(t0.q.z), "v"(t0.q.w), "v"(t0.smw));
        } else if (me_nlive != 0) {
            uint32_t s[8];
            unpack8(t0.q, s);
            Halo h;
            const uint32_t pw1 = rlane(t0.smw, 1), pw2 = rlane(t0.smw, 2);
            const uint32_t nw0 = rlane(t0.smw, 8), nw2 = rlane(t0.smw, 10);
            const bool fast = tile > 0 && tile + 1 < n_tiles && (pw2 & 0xFFFFu) >= 2 && (nw2 & 0xFFFFu) >= 2;
            if (fast) {
                h.p1 = pw1 >> 16; h.p2 = pw1 & 0xFFFFu;
                h.n1 = nw0 & 0xFFFFu; h.n2 = nw0 >> 16;
            } else {
                h = halo_slow(sin, n_tiles, tile, le, re);
            }
            if constexpr (MBPE_FUSED_PF && DIAG == 0) {      // tiles are in prefix form: the fused pass's tile function, counting only
                uint32_t rm_unused = 0;
                bool ws_unused = false, no_rename = false;
                if (TT) tt_rename<MODE>(s, h, ti, run_in[tile]);      // ((t,t) members: every tile, as the chain-walking path did)
                fused_tile_pf<MODE, 0, false, TT>(t0.q, s, h, rlane(t0.smw, 4), rlane(t0.smw, 5), rlane(t0.smw, 6), lut,
                                                  256u + ctl_k_done, tile, nullptr, hdr_adj, LR, dc, dc_on, rm_unused, ws_unused,
                                                  lr_rsrc, adj_pitch, nullptr, chg, TT, &ti, no_rename, t_rsrc, use_t);
            } else {
            if (TT) tt_rename<MODE>(s, h, ti, run_in[tile]);
            // first live token of the lanes after this one (exact), then the candidate
            // test: some slot and its next live token form one of the batch pairs
            uint32_t lf = kHole;
#pragma unroll
            for (int j = 7; j >= 0; --j) lf = s[j] != kHole ? s[j] : lf;
            const unsigned long long m_live = __ballot(lf != kHole);
            const unsigned long long hi = m_live & gt_mask;
            const uint32_t nf = __shfl(lf, hi ? (uint32_t)__builtin_ctzll(hi) : lane, kWave);
            uint32_t c = hi ? nf : h.n1;
            uint32_t miss = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                const uint32_t d = pair_miss(lut, s[j], MODE == 1 ? c & idmask : c);    // (ids are 16-bit: no mask needed)
                miss = d < miss ? d : miss;
                c = s[j] != kHole ? s[j] : c;
            }
            const bool cand = miss == 0u;
            // (also: a match whose first token is the previous tile's last live token)
            const uint32_t tile_first = rlane(lf, (uint32_t)__builtin_ctzll(m_live | (1ull << 63)));
            bool work = __ballot(cand) != 0ull || pair_test(lut, h.p1, tile_first & idmask);
            if (DIAG == 2) { asm volatile("" :: "v"((uint32_t)cand)); work = false; }
            if (work) scan_tile_full<MODE, DIAG>(chg, tile, s, h, lut, n_keys, hdr_adj, LR, pitch, dc, dc_on, TT, ti, adj_pitch);
            }
        }
        if (!v1) break;
        tile += n_waves;
        t0 = t1; t1 = t2;
        v1 = v2;
    }
    }
    if (dc_on) dc_flush(dc, LR);
    if (TT) {
        tt_flush(ti, hdr_m);
    }
}

// ---- the fused pass of a large batch ----------------------------------------------------
// With several dozen pairs in a batch most tiles hold a match, so the separate
// "count, validate, rewrite the marked tiles" scheme reads the stream nearly
// twice.  The fused pass reads every tile once from the current token buffer,
// counts the deltas AND writes the merged tile to the OTHER buffer (every tile,
// so that buffer is complete); new summaries of the changed tiles go to the
// side array as usual.  If validation keeps the whole batch, k_seq_finish just
// flips ctl->cur and nothing else touches the stream; if it drops a suffix,
// the other buffer is simply abandoned and k_rewrite_marked applies the
// surviving prefix to the current buffer, exactly as after k_scan_batch.
//
// Matches of a batch cannot overlap, so "slot j starts a match" (mask A) is one
// exact table test per slot, and "slot j is the second token of a match" (mask
// B) is A moved to the next live slot.  That move, inside a lane and from lane
// to lane over empty lanes, is a carry chain: ((A << 1 | carry_in) + holes) &
// live.  Everything else (pair index, neighbours, deltas) is only done where a
// match is.

template <int MODE, int DIAG = 0, class DC>
__device__ __forceinline__ uint4 fused_tile_full(const uint4 q_orig, bool tt_on, TTInfo &ti,
                                                 const uint32_t s[8], const uint32_t cj[8],
                                                 uint32_t Am, uint32_t Wm, bool tcin, bool tbin, unsigned long long m_live,
                                                 uint32_t c_init, const Halo h,
                                                 uint32_t tile_first, uint32_t old_x, uint32_t old_y,
                                                 uint32_t old_z, const BatchLut &lut, uint32_t X0, uint32_t tile,
                                                 TileSum *sout, uint32_t *chg, uint32_t *hdr_adj, uint32_t *LR,
                                                 DC &dc, bool dc_on, uint32_t &wave_rm, bool &wrote_sum,
                                                 __amdgpu_buffer_rsrc_t lr_rsrc, uint32_t adj_pitch, uint16_t *stage) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    const uint32_t pitch = lr_pitch(X0);        // uniform
    // (plain instantiation: the delta atomics go through a buffer resource -- scalar base, 32-bit lane offset -- so
    //  that no 64-bit address is built per match; the LR block is below 4 GB)
    auto delta_add = [&](uint32_t idx, uint32_t delta) {
        if (dc_on) dc_add(dc, true, LR, idx, delta);
        else __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)delta, lr_rsrc, idx << 2, 0, 0);
    };
    // (an opaque copy of the lane id: the 64-bit lane masks below are cheaper to rebuild per tile than
    //  to keep in registers across the streaming loop, where the compiler would spill them)
    uint32_t lane = lane_id();
    asm volatile("" : "+v"(lane));
    const unsigned long long lane_bit = 1ull << lane;
    uint32_t Lm = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) Lm |= (s[j] != kHole ? 1u : 0u) << j;
    const uint32_t Hm = Lm ^ 0xFFu;
    const unsigned long long E = ~m_live;

    // carry "the last live token before this lane starts a match" over empty lanes
    const bool lastA = Am > (Lm & ~Am);
    const unsigned long long G = __ballot(lastA);
    const unsigned long long CIN = (((G << 1) | (tcin ? 1ull : 0ull)) + E) & m_live;
    const uint32_t cin = lane_of(CIN) ? 1u : 0u;
    const uint32_t Bm = (((Am << 1) | cin) + Hm) & Lm;
    // the same for "the last live token before this lane ends a match"
    const bool lastB = Bm > (Lm & ~Bm);
    const unsigned long long GB = __ballot(lastB);
    const unsigned long long BIN = (((GB << 1) | (tbin ? 1ull : 0ull)) + E) & m_live;
    const uint32_t bin = lane_of(BIN) ? 1u : 0u;
    const uint32_t touch = Am & ((((Bm << 1) | bin) + Hm) & Lm);     // starts a match right after another one

    const unsigned long long ab = __ballot((Am | Bm) != 0u);
    if (ab == 0ull) return q_orig;

    // last live token of every lane and, where it starts a match, the index of that match
    uint32_t ll = kHole;
#pragma unroll
    for (int j = 0; j < 8; ++j) ll = s[j] != kHole ? s[j] : ll;
    uint32_t lastj = 0;
    if (lastA) lastj = lut_index_known(lut, ll, c_init & idmask, ((Wm >> (31u - (uint32_t)__builtin_clz(Lm | 1u))) & 1u) != 0u);
    uint32_t p1, pj;
    const uint32_t pj_tile = tcin ? (uint32_t)lut_index(lut, h.p1, tile_first & idmask) : 0u;     // uniform
    if (m_live == ~0ull) {                   // the previous lane is the previous live lane
        const uint32_t got = wave_from_prev(ll | (lastj << 16), h.p1 | (pj_tile << 16));
        p1 = got & 0xFFFFu;
        pj = got >> 16;
    } else {
        const unsigned long long lo = m_live & (lane_bit - 1ull);
        const uint32_t src = lo ? 63u - (uint32_t)__builtin_clzll(lo) : lane;
        const uint32_t got = __shfl(ll | (lastj << 16), src, kWave);
        p1 = lo ? (got & 0xFFFFu) : h.p1;
        pj = lo ? (got >> 16) : pj_tile;
    }
    uint32_t pjb = 0;
    // rare: the token before this lane ends a match and this lane's first live token starts one
    if (__ballot(bin != 0u && (Am & Lm & (0u - Lm)) != 0u) != 0ull) {
        const Neigh nb = tile_neighbours(s, h);
        if (bin) pjb = (uint32_t)lut_index(lut, nb.p2_in, p1 & idmask);
    }

    uint32_t out[8];
    const uint32_t ABm = Am | Bm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t self = s[j];
        uint32_t nv = self;
        if ((ABm >> j) & 1u) {               // one guarded region per position serves both roles
            const bool is_a = (Am >> j) & 1u;
            uint32_t ja = pj;
            if (is_a) {
                ja = lut_index_known(lut, self, cj[j] & idmask, ((Wm >> j) & 1u) != 0u);
                if (tt_on && (cj[j] & idmask) >= idmask - (uint32_t)kTTMax)      // a match of a (t,t) member: count it
                    atomicAdd(&ti.cnt[idmask - 1u - (cj[j] & idmask)], 1u);
            }
            // first token of a match: (p1, a) -> (p1, X); second token: (b, n1) -> (X, n1)
            const uint32_t nb = is_a ? p1 : cj[j];                       // the neighbour the match loses
            const bool counted = is_a ? left_open<MODE>(p1) : right_open<MODE>(self, cj[j]);
            nv = is_a ? (X0 + ja) | (cj[j] & endbit) : kHole;
            if (counted && DIAG != 2) {
                if (is_a && ((touch >> j) & 1u)) {        // ... (a', b') (a, b): (b', a) -> (X', X)
                    atomicAdd(&hdr_adj[pjb * adj_pitch + ja], 1u);
                    delta_add(lr_idx(pitch, self, pjb, 1), 0xFFFFFFFFu);   // takes back the R count of (a', b')
                } else {
                    delta_add(lr_idx(pitch, nb & idmask, ja, is_a ? 0u : 1u), 1u);
                }
            }
            pjb = is_a ? pjb : pj;
            pj = ja;
        }
        p1 = self != kHole ? self : p1;
        out[j] = nv;
    }

    if (tt_on) {                        // (every renamed token was the second token of a match; be safe)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (out[j] != kHole && (out[j] & idmask) >= idmask - (uint32_t)kTTMax)
                out[j] = ti.tok[idmask - 1u - (out[j] & idmask)] | (out[j] & endbit);
    }
    // (sum over the lanes of the set bits of an 8-bit mask with ballots: vector compares and scalar popcounts instead
    //  of a shuffle reduction, whose six dependent ds_bpermute round trips cost more than the rest of this function)
    uint32_t removed = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) removed += (uint32_t)__popcll(__ballot(((Bm >> k) & 1u) != 0u));
    wave_rm += removed;
    // New summary.  Heads and tails only change when a match touches one of the first two or
    // last two live tokens; a trailing run of equal tokens (tail_run > 1) is recounted.
    const uint32_t top = Lm ? 31u - (uint32_t)__builtin_clz(Lm) : 0u;
    const uint32_t lb1 = Lm & (0u - Lm), r1 = Lm ^ lb1, lb2 = r1 & (0u - r1);
    const uint32_t hb1 = Lm ? 1u << top : 0u, r2 = Lm ^ hb1, hb2 = r2 ? 1u << (31u - (uint32_t)__builtin_clz(r2)) : 0u;
    const uint32_t F = (uint32_t)__builtin_ctzll(m_live), Hl = 63u - (uint32_t)__builtin_clzll(m_live);
    const unsigned long long mF = m_live & (m_live - 1ull), mH = m_live & ~(1ull << Hl);
    bool edge = rlane(ABm & (lb1 | lb2), F) != 0u || rlane(ABm & (hb1 | hb2), Hl) != 0u;
    if (rlane(lb2, F) == 0u && mF) edge |= rlane(ABm & lb1, (uint32_t)__builtin_ctzll(mF)) != 0u;
    if (rlane(hb2, Hl) == 0u && mH) edge |= rlane(ABm & hb1, 63u - (uint32_t)__builtin_clzll(mH)) != 0u;
    uint4 ns;
    if (edge || (old_z >> 16) != 1u) {
        ns = wave_summary(out);
    } else {
        ns = make_uint4(old_x, old_y, (old_z & 0xFFFF0000u) | ((old_z & 0xFFFFu) - removed), 0u);
    }
    // (no tile mark: a fused pass writes every tile's summary to the side array, see DevCtl::marks_all)
    if (lane == 0) reinterpret_cast<uint4 *>(sout)[tile] = ns;
    wrote_sum = true;
    uint32_t n_out;
    return tile_compact(out, stage, n_out);
}

// The same for a tile in prefix form (see tile_compact; every tile, unless the slots hold barriers): the token after a
// slot is the next slot -- the next lane's first one, the next tile's first token after the last live slot -- and the
// token before it the previous slot, so a slot is the second token of a match iff the slot before it starts one.  No
// chains over holes, no carries over empty lanes, no (t,t) members (their instantiation takes fused_tile_full).  The
// batch index of the match a slot starts is looked up once per slot (byte table: the entry itself) and handed to the
// next two slots by register / DPP: a second token counts for the pair of the slot before it, a first token right
// after a match for the pair two slots back (ADJ).  Same deltas, same new tile, same summary as fused_tile_full.
template <int MODE, int DIAG, bool WRITE, bool TT, class DC>
__device__ __forceinline__ uint4 fused_tile_pf(const uint4 q_orig, const uint32_t s[8], const Halo h,
                                               uint32_t old_x, uint32_t old_y, uint32_t old_z, const BatchLut &lut,
                                               uint32_t X0, uint32_t tile, TileSum *sout, uint32_t *hdr_adj, uint32_t *LR,
                                               DC &dc, bool dc_on, uint32_t &wave_rm, bool &wrote_sum,
                                               __amdgpu_buffer_rsrc_t lr_rsrc, uint32_t adj_pitch, uint16_t *stage,
                                               uint32_t *chg, bool renamed, TTInfo *ti, bool &need_rename,
                                               __amdgpu_buffer_rsrc_t t_rsrc, bool use_t) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    constexpr uint32_t kNone = 0xFFFFu;          // "this slot starts no match" (batch indices are below kBatchMax)
    const uint32_t pitch = lr_pitch(X0);        // uniform
    // (timing-only build 6: every XCD counts into rows of its own -- 512 pairs further on per XCD; run it with
    //  "max_batch" 512 -- to see what the atomics cost when no two XCDs ever touch the same cache line)
    const uint32_t xcd_off = DIAG == 6 ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) * (2u * 512u * pitch) : 0u;
    auto delta_add = [&](uint32_t idx, uint32_t delta) {
        if (dc_on) dc_add(dc, true, LR, idx, delta);
        else __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)delta, lr_rsrc, (idx + xcd_off) << 2, 0, 0);
    };
    uint32_t lane = lane_id();
    asm volatile("" : "+v"(lane));
    const uint32_t live = old_z & 0xFFFFu;       // uniform
    // (an empty tile holds no match and stays as it is: k_rewrite_marked walks EVERY tile after an abandoned fused pass,
    //  empty ones included, and `live - 1` below selects lanes and slots)
    if (live == 0u) return q_orig;
    uint32_t tile_first = old_x & 0xFFFFu, tile_last = old_y >> 16;
    if (TT && renamed) {                 // (uniform, rare: s[] and h hold stand-in ids where tt_rename put them)
        const uint32_t li = live - 1u;
        uint32_t sl = s[0];
#pragma unroll
        for (uint32_t j = 1; j < 8; ++j) sl = (li & 7u) == j ? s[j] : sl;
        tile_first = rlane(s[0], 0);
        tile_last = rlane(sl, li >> 3);
    }
    // the value after every slot (after the last live slot: a hole, or the next tile's first token when the tile is full)
    uint32_t n[8];
#pragma unroll
    for (int j = 0; j < 7; ++j) n[j] = s[j + 1];
    n[7] = wave_from_next(s[0], h.n1);
    // The batch index of the match every slot starts (kNone: none) -- and of the three matches across the tile's edges,
    // which lanes 0, 1, 2 look up in a ninth read: (p1, first token), (p2, p1), (last token, n1).
    // ((t,t) members, byte table: lane 3 looks up (n1, n2) as well -- only for the marker of "a run of a member's token")
    const uint32_t ef = lane == 0 ? h.p1 : lane == 1 ? h.p2 : lane == 2 ? tile_last : (TT && lane == 3) ? h.n1 : kHole;
    const uint32_t es = (lane == 0 ? tile_first : lane == 1 ? h.p1 : (TT && lane == 3) ? h.n2 : h.n1) & idmask;
    uint32_t idx[8], eidx;
    if (lut.bytes) {             // (uniform) the byte table: the entry is the index
#pragma unroll
        for (int j = 0; j < 8; ++j) idx[j] = byte_entry<TT>(lut, s[j], MODE == 1 ? n[j] & idmask : n[j], idmask);
        eidx = byte_entry<TT>(lut, ef, es, idmask);
        // nearly every tile of a small batch: no match anywhere (indices are 16-bit: all kNone iff their AND is)
        const uint32_t all = idx[0] & idx[1] & idx[2] & idx[3] & idx[4] & idx[5] & idx[6] & idx[7] & eidx;
        if (__ballot(all != kNone) == 0ull && DIAG != 3 && DIAG != 5) return q_orig;
        if (TT) {
            // a run of a (t,t) member's token somewhere in or around the tile: the caller renames (tt_rename) and comes
            // back; afterwards no two of them are side by side, a marker that still shows up is "no match"
            bool mark = eidx == kTTMark;
#pragma unroll
            for (int j = 0; j < 8; ++j) mark = mark || idx[j] == kTTMark;
            if (__ballot(mark) != 0ull) {
                if (!renamed) { need_rename = true; return q_orig; }
#pragma unroll
                for (int j = 0; j < 8; ++j) idx[j] = idx[j] == kTTMark ? kNone : idx[j];
                eidx = eidx == kTTMark ? kNone : eidx;
            }
            eidx = lane == 3 ? kNone : eidx;         // (lane 3's lookup was only there for the marker)
        }
    } else {                     // the hash table: nine bucket reads, then the index words of the slots that hit
        uint32_t Hm = 0, Wm = 0;         // bit j: slot j (8: the edge lookup) starts a match / ... under its bucket's second key
        {
            bool second_key;
            const bool hit = pair_hit2(lut, ef, es, second_key) && !(TT && lane == 3);
            Hm = hit ? 1u : 0u;
            Wm = second_key ? 1u : 0u;
        }
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            bool second_key;
            const bool hit = pair_hit2(lut, s[j], MODE == 1 ? n[j] & idmask : n[j], second_key);
            Hm = Hm + Hm + (hit ? 1u : 0u);
            Wm = Wm + Wm + (second_key ? 1u : 0u);
        }
        if (__ballot(Hm != 0u) == 0ull && DIAG != 3 && DIAG != 5) return q_orig;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            idx[j] = kNone;
            if (__ballot(((Hm >> j) & 1u) != 0u) != 0ull) {
                const uint32_t got = lut_index_known(lut, s[j], MODE == 1 ? n[j] & idmask : n[j], ((Wm >> j) & 1u) != 0u);
                idx[j] = (Hm >> j) & 1u ? got : kNone;
            }
        }
        eidx = kNone;
        if (__ballot(((Hm >> 8) & 1u) != 0u) != 0ull) {
            const uint32_t got = lut_index_known(lut, ef, es, ((Wm >> 8) & 1u) != 0u);
            eidx = (Hm >> 8) & 1u ? got : kNone;
        }
    }
    if (DIAG == 3 || DIAG == 5) {                // timing-only build: lookups, no merge
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(idx[j]));
        asm volatile("" :: "v"(eidx));
        return q_orig;
    }
    const uint32_t in1 = rlane(eidx, 0), in2 = rlane(eidx, 1), out1 = rlane(eidx, 2);
    {
        // the last live slot (lane li / 8, slot li % 8) looks at the next tile's first token
        const uint32_t li = live - 1u;
        const bool mine = lane == (li >> 3);
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
            if ((li & 7u) == j) { n[j] = mine ? h.n1 : n[j]; idx[j] = mine ? out1 : idx[j]; }
    }
    const uint32_t prev7 = wave_from_prev(idx[7] | (s[7] << 16), in1 | (h.p1 << 16));    // the slot before this lane's first
    const uint32_t prev6 = wave_from_prev(idx[6], in2);                                  // ... and the one before that
    const int lrem = (int)live - (int)(lane * 8u);
    const uint32_t Lm = lrem >= 8 ? 0xFFu : lrem <= 0 ? 0u : (1u << lrem) - 1u;          // this lane's live slots
    uint32_t Am = 0, Bm = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t pj = j == 0 ? (prev7 & 0xFFFFu) : idx[j - 1];
        Am |= (idx[j] != kNone ? 1u : 0u) << j;
        Bm |= (pj != kNone ? 1u : 0u) << j;
    }
    Bm &= Lm;            // (a match that starts at the tile's last live token ends in the next tile)
    const uint32_t ABm = Am | Bm;
    if (__ballot(ABm != 0u) == 0ull) return q_orig;

    // The lane of a match's FIRST token counts both of its deltas: (p1, a) -> (p1, X) with the token before it and
    // (b, n2) -> (X, n2) with the token two slots on -- the next lane's, the next tile's (h.n1, h.n2) where the slots end:
    // the lane of the second token has nothing to count, and a match across the tile edge is counted by this tile.
    uint32_t n2_7 = wave_from_next(n[0], h.n2);          // the value two slots after slot 7 (n[0] is already patched)
    uint32_t n2_patch = kSent;                           // (this lane's slot li % 8 is the last live one: two on is h.n2)
    if (out1 != kNone) {                                 // uniform, rare: the last live token starts a match
        const uint32_t li = live - 1u;
        if (lane == (li >> 3)) n2_patch = li & 7u;
    }
    uint32_t out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t self = s[j];
        uint32_t nv = ((Bm >> j) & 1u) ? kHole : self;
        if (__ballot(((Am >> j) & 1u) != 0u) != 0ull) {
            if ((Am >> j) & 1u) {
                const uint32_t p1 = j == 0 ? prev7 >> 16 : s[j > 0 ? j - 1 : 0];
                const uint32_t ppj = j == 0 ? prev6 : j == 1 ? (prev7 & 0xFFFFu) : idx[j > 1 ? j - 2 : 0];
                const uint32_t ja = idx[j];
                const uint32_t btok = n[j];
                uint32_t n2 = j < 7 ? n[j < 7 ? j + 1 : 7] : n2_7;
                n2 = n2_patch == (uint32_t)j ? h.n2 : n2;
                nv = (X0 + ja) | (btok & endbit);
                if (TT && renamed && (btok & idmask) >= idmask - (uint32_t)kTTMax)       // a match of a (t,t) member: count it
                    atomicAdd(&ti->cnt[idmask - 1u - (btok & idmask)], 1u);
                if (DIAG != 2) {
                    const bool lo = left_open<MODE>(p1), ro = right_open<MODE>(btok, n2);
                    const uint32_t xl = p1 & idmask, yr = n2 & idmask;
                    if (use_t && lo && ro && ppj == kNone && xl < 256u && yr < 256u) {
                        // both neighbours are raw bytes and no match touches this one: ONE atomic on the pair's
                        // byte x byte cell block instead of two on its L and R rows (k_pair_cells_fold adds the
                        // block's row and column sums to the rows afterwards)
                        __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, t_rsrc, ((ja << 16) | (xl << 8) | yr) << 2, 0, 0);
                    } else {
                        if (lo) {
                            if (ppj != kNone) {      // ... (a', b') (a, b): (b', a) -> (X', X)
                                atomicAdd(&hdr_adj[ppj * adj_pitch + ja], 1u);
                                delta_add(lr_idx(pitch, self, ppj, 1), 0xFFFFFFFFu);     // takes back the R count of (a', b')
                            } else {
                                delta_add(lr_idx(pitch, xl, ja, 0u), 1u);
                            }
                        }
                        if (ro) delta_add(lr_idx(pitch, yr, ja, 1u), 1u);
                    }
                }
            }
        }
        out[j] = nv;
    }
    if (TT && renamed) {                 // (every renamed token was the second token of a match; be safe)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (out[j] != kHole && (out[j] & idmask) >= idmask - (uint32_t)kTTMax)
                out[j] = ti->tok[idmask - 1u - (out[j] & idmask)] | (out[j] & endbit);
    }
    if (!WRITE) {                        // the counting pass of a small batch: mark the tile for k_rewrite_marked
        if (lane == 0) atomicOr(&chg[tile >> 5], 1u << (tile & 31u));
        return q_orig;
    }
    uint32_t n_out;
    const uint4 qc = tile_compact(out, stage, n_out);
    wave_rm += live - n_out;
    // New summary.  Heads and tails only change when a match touches one of the first two or last two live tokens; a
    // trailing run of equal tokens (tail_run > 1) is recounted.
    uint32_t em = lane == 0 ? 3u : 0u;
    {
        const uint32_t l1 = live - 1u, l2 = live >= 2u ? live - 2u : 0u;
        em |= lane == (l1 >> 3) ? 1u << (l1 & 7u) : 0u;
        em |= lane == (l2 >> 3) ? 1u << (l2 & 7u) : 0u;
    }
    const bool edge = __ballot((ABm & em) != 0u) != 0ull;
    uint4 ns;
    if (edge || (old_z >> 16) != 1u) {
        uint32_t c[8];
        unpack8(qc, c);
        ns = wave_summary(c);
    } else {
        ns = make_uint4(old_x, old_y, (old_z & 0xFFFF0000u) | n_out, 0u);
    }
    if (lane == 0) reinterpret_cast<uint4 *>(sout)[tile] = ns;
    wrote_sum = true;
    return qc;
}

// (7 waves per SIMD: the cold instantiation meets it with two spilled registers, which is cheaper than
//  running with 6 waves; 8 would spill ten and is slower.  The delta cache of the HOT one allows 5.)
#ifndef MBPE_SKIP_PENALTY0
#define MBPE_SKIP_PENALTY0 2
#endif
#ifndef MBPE_SKIP_PENALTY_MAX
#define MBPE_SKIP_PENALTY_MAX 256
#endif
#ifndef MBPE_FUSED_WAVES
#define MBPE_FUSED_WAVES 4
#endif
template <int MODE, bool HOT, bool TT, int DIAG = 0>
__global__ __launch_bounds__(kLutThreads, MBPE_FUSED_WAVES * 256 / kLutThreads) void k_fused_batch(uint16_t *tok0, uint16_t *tok1,
                                                               const TileSum *__restrict__ sin,
                                                               TileSum *__restrict__ sout, uint32_t n_tiles,
                                                               uint32_t *__restrict__ chg, const BatchState *bs,
                                                               uint32_t *hdr_adj, uint32_t *LR, DevCtl *ctl,
                                                               const RankEdge *le, const RankEdge *re,
                                                               uint32_t *hdr_m, const uint32_t *__restrict__ run_in,
                                                               int hot_launched, uint32_t *T) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    __shared__ BatchLutMem lut_mem;
    const uint32_t lane = lane_id();
    const uint32_t waves_per_block = kLutThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    __shared__ DeltaCacheSmall dc;
    const uint32_t n_keys = ctl->batch_n;
    if (n_keys < 2 || !ctl->fused) return;
    // (the instantiation with the (t,t) code only runs for batches that have such a member, like HOT)
    if ((bs->tt_index != kNoTT) != TT) return;
    __shared__ TTInfo ti;
    __shared__ __attribute__((aligned(16))) uint16_t stage_mem[kLutThreads / kWave][kTileSlots];
    uint16_t *stage = stage_mem[threadIdx.x / kWave];
    // tiles are in prefix form and take the short tile function (fused_tile_pf); (t,t) members: renamed first where needed
    constexpr bool PF = MBPE_FUSED_PF != 0;
    const uint16_t *tok = ctl->cur ? tok1 : tok0;
    uint16_t *dst = ctl->cur ? tok0 : tok1;
    const uint32_t X0 = 256u + ctl->k_done;
    if (hot_mismatch<HOT>(bs->packed[0] >> 32, ctl, hot_launched)) return;     // (see k_merge)
    constexpr bool dc_on = HOT;
    if (dc_on) dc_init(dc);
    BatchLut lut(&lut_mem, bs, idmask);
    lut_build(lut, bs, n_keys, idmask - 1u, TT && MBPE_FUSED_PF);
    if (TT) tt_build(ti, bs, n_keys, idmask - 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->marks_all = 1;
    uint32_t tile = rfl(blockIdx.x * waves_per_block + threadIdx.x / kWave);
    uint32_t wave_rm = 0;        // uniform
    if (tile < n_tiles) {

    const uint32_t last_tile = n_tiles - 1;
    auto clamp_tile = [&](uint64_t t) { return (uint32_t)(t < n_tiles ? t : last_tile); };
    const __amdgpu_buffer_rsrc_t sums_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TileSum *>(sin), 0, n_tiles * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t lr_rsrc = __builtin_amdgcn_make_buffer_rsrc(LR, 0, 0xFFFFFFFCu, 0x00020000);
    // (the pairs' byte x byte cell blocks: see k_pair_cells_fold; not with the delta cache of frequent pairs)
    const bool use_t = T != nullptr && !HOT && ctl->cells_on != 0u && n_keys >= ctl->cells_min;
    const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(T ? T : LR, 0, T ? 0xFFFFFFFCu : 0u, 0x00020000);
    const uint32_t adj_pitch = rfl(ctl->adj_pitch);
    // kDepth tiles are in flight behind the one being worked on (5 registers each)
#ifndef MBPE_FUSED_DEPTH
#define MBPE_FUSED_DEPTH 2
#endif
    constexpr int kDepth = MBPE_FUSED_DEPTH;
    TileIn ring[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) ring[d] = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + (uint64_t)d * n_waves));
    const unsigned long long gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    for (;;) {
        const bool v1 = (uint64_t)tile + n_waves < n_tiles;
        const TileIn t_new = tile_issue(tok, sums_rsrc, clamp_tile((uint64_t)tile + (uint64_t)kDepth * n_waves));
        const TileIn t0 = ring[0];

        uint4 outq = t0.q;
        bool wrote_sum = false;          // uniform
        const uint32_t old_x = rlane(t0.smw, 4), old_y = rlane(t0.smw, 5), old_z = rlane(t0.smw, 6);
        if (DIAG == 4) {                 // timing-only build: copy
            asm volatile("" :: "v"(t0.smw));
        } else if ((old_z & 0xFFFFu) != 0) {
            uint32_t s[8];
            unpack8(t0.q, s);
            Halo h;
            const uint32_t pw1 = rlane(t0.smw, 1), pw2 = rlane(t0.smw, 2);
            const uint32_t nw0 = rlane(t0.smw, 8), nw2 = rlane(t0.smw, 10);
            const bool fast = tile > 0 && tile + 1 < n_tiles && (pw2 & 0xFFFFu) >= 2 && (nw2 & 0xFFFFu) >= 2;
            if (fast) {
                h.p1 = pw1 >> 16; h.p2 = pw1 & 0xFFFFu;
                h.n1 = nw0 & 0xFFFFu; h.n2 = nw0 >> 16;
            } else {
                h = halo_slow(sin, n_tiles, tile, le, re);
            }
            if constexpr (PF) {
                bool renamed = false;    // uniform: tt_rename ran on this tile (only then can stand-in ids occur in it)
                bool need_rename = false;
                if (TT && !lut.bytes) {
                    // (t,t) members, hash table: does any live token equal its successor and belong to one?  (nearly
                    // never; with the byte table the slots' own lookups tell: kTTMark)
                    uint32_t nx[8];
#pragma unroll
                    for (int j = 0; j < 7; ++j) nx[j] = s[j + 1];
                    nx[7] = wave_from_next(s[0], h.n1);
                    const uint32_t li = (old_z & 0xFFFFu) - 1u;
                    const bool mine = lane == (li >> 3);
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if ((li & 7u) == j) nx[j] = mine ? h.n1 : nx[j];
                    const int lrem = (int)(old_z & 0xFFFFu) - (int)(lane * 8u);
                    const uint32_t lm = lrem >= 8 ? 0xFFu : lrem <= 0 ? 0u : (1u << lrem) - 1u;
                    if (tt_needed<MODE>(s, nx, h, old_x & 0xFFFFu, ti, lm)) {
                        tt_rename<MODE>(s, h, ti, run_in[tile]);
                        renamed = true;
                    }
                }
                outq = fused_tile_pf<MODE, DIAG, true, TT>(t0.q, s, h, old_x, old_y, old_z, lut, X0, tile, sout, hdr_adj, LR, dc,
                                                           dc_on, wave_rm, wrote_sum, lr_rsrc, adj_pitch, stage, chg, renamed, &ti,
                                                           need_rename, t_rsrc, use_t);
                if (TT && need_rename) {             // (uniform, rare)
                    tt_rename<MODE>(s, h, ti, run_in[tile]);
                    outq = fused_tile_pf<MODE, DIAG, true, TT>(t0.q, s, h, old_x, old_y, old_z, lut, X0, tile, sout, hdr_adj, LR,
                                                               dc, dc_on, wave_rm, wrote_sum, lr_rsrc, adj_pitch, stage, chg, true,
                                                               &ti, need_rename, t_rsrc, use_t);
                }
            } else {
            uint32_t lf, c_init, tile_first, cj[8];
            unsigned long long m_live;
            bool renamed = false;        // uniform: tt_rename ran on this tile (only then can stand-in ids occur in it)
            // (t,t) members: the renaming changes the tile's tokens, so everything derived from them is redone
            // after it -- for the few tiles that need it (tt_needed)
            for (int rep = 0;; ++rep) {
                lf = kHole;
#pragma unroll
                for (int j = 7; j >= 0; --j) lf = s[j] != kHole ? s[j] : lf;
                m_live = __ballot(lf != kHole);
                if (m_live == ~0ull) {               // every lane holds a live token (nearly always): the next lane's
                    c_init = wave_from_next(lf, h.n1);
                } else {
                    const unsigned long long hi = m_live & gt_mask;
                    const uint32_t nf = __shfl(lf, hi ? (uint32_t)__builtin_ctzll(hi) : lane, kWave);
                    c_init = hi ? nf : h.n1;
                }
                tile_first = rlane(lf, (uint32_t)__builtin_ctzll(m_live | (1ull << 63)));
                uint32_t c = c_init;
#pragma unroll
                for (int j = 7; j >= 0; --j) {
                    cj[j] = c;
                    c = s[j] != kHole ? s[j] : c;
                }
                if (!TT || rep || DIAG == 5 || !tt_needed<MODE>(s, cj, h, tile_first, ti)) break;
                tt_rename<MODE>(s, h, ti, run_in[tile]);
                renamed = true;
            }
            // (the two tests on the tile's left edge -- does the previous tile's last token start a match with this
            //  tile's first one, did it end one -- are uniform LDS reads; issued here they travel with the eight below
            //  instead of costing a round trip of their own inside fused_tile_full)
            const bool tcin = pair_test(lut, h.p1, tile_first & idmask);
            const bool tbin = h.p2 != kHole && pair_test(lut, h.p2, h.p1 & idmask);
            uint32_t Am = 0, Wm = 0;     // bit j: slot j starts a match / ... of the second key of its lookup bucket
            if (lut.bytes) {             // (uniform) a batch of byte pairs: one 2-byte read of the direct table per slot
#pragma unroll
                for (int j = 7; j >= 0; --j) {
                    const bool hit = byte_entry<TT>(lut, s[j], MODE == 1 ? cj[j] & idmask : cj[j], idmask) != 0xFFFFu;
                    Am = Am + Am + (hit ? 1u : 0u);
                }
            } else {
#pragma unroll
                for (int j = 7; j >= 0; --j) {
                    bool second_key;
                    const bool hit = pair_hit2(lut, s[j], MODE == 1 ? cj[j] & idmask : cj[j], second_key);     // (ids are 16-bit: no mask needed)
                    Am = Am + Am + (hit ? 1u : 0u);         // one add-with-carry, the carry being the compare mask
                    Wm = Wm + Wm + (second_key ? 1u : 0u);
                }
            }
            const bool any = Am != 0u;
            if (DIAG == 3 || DIAG == 5) {    // timing-only build: membership tests, no merge
                asm volatile("" :: "v"(Am));
            } else if (__ballot(any) != 0ull || tcin) {
                outq = fused_tile_full<MODE, DIAG>(t0.q, TT && renamed, ti, s, cj, Am, Wm, tcin, tbin, m_live, c_init, h, tile_first,
                                                old_x, old_y, old_z, lut, X0, tile, sout, chg, hdr_adj, LR, dc, dc_on,
                                                wave_rm, wrote_sum, lr_rsrc, adj_pitch, stage);
            }
            }
        }
        // every tile's summary goes to the side array (an unchanged tile's as it was): no tile marks needed
        if (!wrote_sum && lane == 0) reinterpret_cast<uint4 *>(sout)[tile] = make_uint4(old_x, old_y, old_z, 0u);
        MBPE_GLOBAL_AS char *obase =
            (MBPE_GLOBAL_AS char *)uniform_ptr(reinterpret_cast<uintptr_t>(dst) + (uint64_t)tile * (kWave * 16u));
        u32x4 oq; oq.x = outq.x; oq.y = outq.y; oq.z = outq.z; oq.w = outq.w;
#if MBPE_NT_STREAM
        __builtin_nontemporal_store(oq, (MBPE_GLOBAL_AS u32x4 *)(obase + lane * 16u));
#else
        *(MBPE_GLOBAL_AS u32x4 *)(obase + lane * 16u) = oq;
#endif

        if (!v1) break;
        tile += n_waves;
#pragma unroll
        for (int d = 0; d + 1 < kDepth; ++d) ring[d] = ring[d + 1];
        ring[kDepth - 1] = t_new;
    }
    }
    if (dc_on) dc_flush(dc, LR);
    if (lane == 0 && wave_rm) atomicAdd(&ctl->rm, wave_rm);
    if (TT) {                           // matches of the (t,t) member: its count is not simply the pair's count
        tt_flush(ti, hdr_m);
    }
}

// The pairs' byte x byte cell blocks (round 4).  A match whose two neighbours are raw bytes and which no other match
// touches costs the stream pass ONE scattered atomic, on cell T[j][x][y] of its pair's 256 x 256 block, instead of two
// (L_j[x] and R_j[y]): in the passes of thousands of byte pairs, which their atomics bound, most
// matches are of that kind.  This kernel adds every block's row sums to the pair's L row and its column sums to the R row
// and clears the block, so that everything behind the pass -- validation, the table update, the exchange of several
// ranks -- sees the rows it always saw.  One workgroup per pair at a time; 256 KiB read (and what was not zero cleared).
__global__ __launch_bounds__(256) void k_pair_cells_fold(uint32_t *__restrict__ T, uint32_t *__restrict__ LR, const DevCtl *ctl) {
    __shared__ uint32_t rows[256];
    const uint32_t n = ctl->batch_n;
    if (n < ctl->cells_min || !ctl->cells_on) return;          // (the pass did not use the blocks: see k_fused_batch)
    const uint32_t pitch = lr_pitch(256u + ctl->k_done);
    uint32_t hits = 0;
    const uint32_t t = threadIdx.x;
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        uint32_t *tj = T + (size_t)j * 65536u;
        rows[t] = 0;
        __syncthreads();
        uint32_t col = 0;
#pragma unroll 8
        for (uint32_t x = 0; x < 256u; ++x) {
            const uint32_t v = tj[x * 256u + t];
            if (__ballot(v != 0u) == 0ull) continue;          // (this wave's quarter of the row is empty)
            if (v) tj[x * 256u + t] = 0;
            col += v;
            const uint32_t rs = wave_sum(v);
            if (lane_id() == 0 && rs) atomicAdd(&rows[x], rs);
        }
        __syncthreads();
        if (col) LR[lr_idx(pitch, t, j, 1u)] += col;             // R_j[y = t]
        if (rows[t]) LR[lr_idx(pitch, t, j, 0u)] += rows[t];     // L_j[x = t]
        hits += col;
        __syncthreads();
    }
    hits = wave_sum(hits);
    if (lane_id() == 0 && hits) atomicAdd(const_cast<uint32_t *>(&ctl->cell_hits), hits);
}

// Validation (after the all-reduce in a multi-GPU run).  Merge j creates the pairs (x, X_j) with
// count L_j[x], (X_j, y) with count R_j[y] and (X_p, X_j) with count ADJ[p][j]; while only a prefix
// of the batch is merged, a match of j that touches a match of a later pair p still has p's plain
// token as its neighbour, which can add up to adj_in[j] / adj_out[j] to an L / R count.  Pair j of
// the batch is the reference's next choice iff its packed (count, ~key) -- ties resolved by key,
// exactly like the argmax -- beats every pair the merges before it can have created.

// (one wave per pair j: column j and row j of the n x n block)
__global__ __launch_bounds__(kWave) void k_adj_sums(const uint32_t *__restrict__ hdr_adj, BatchState *bs, const DevCtl *ctl) {
    const uint32_t n = ctl->batch_n;
    if (n < 2) return;
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {       // (the grid is sized for the batches the host expects)
        uint32_t in = 0, out = 0;
        for (uint32_t p = threadIdx.x; p < n; p += kWave) { in += hdr_adj[p * ctl->adj_pitch + j]; out += hdr_adj[j * ctl->adj_pitch + p]; }
        in = wave_sum(in);
        out = wave_sum(out);
        if (threadIdx.x == 0) { bs->adj_in[j] = in; bs->adj_out[j] = out; }
    }
}

// per pair j: the largest packed value among the pairs (x, X_j), (X_j, y) it creates
// (work item = kLrChunk cells of one row of the LR block; a workgroup takes items with a grid stride)
constexpr uint32_t kLrChunk = 4096;
__global__ __launch_bounds__(256) void k_delta_max(const uint32_t *__restrict__ LR, BatchState *bs, const DevCtl *ctl) {
    const uint32_t n_keys = ctl->batch_n;
    if (n_keys < 2) return;
    const uint32_t X = 256u + ctl->k_done;
    const uint32_t pitch = lr_pitch(X);
    const uint32_t chunks = (X + kLrChunk - 1u) / kLrChunk;          // per row
    const uint32_t n_items = 2u * n_keys * chunks;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t row = item / chunks, x0 = (item - row * chunks) * kLrChunk;
        const uint32_t j = row >> 1, side = row & 1u, Xj = X + j;
        const uint32_t add = side ? bs->adj_out[j] : bs->adj_in[j];
        const uint32_t *cells = LR + (size_t)row * pitch;
        unsigned long long mp = 0;
#pragma unroll 4
        for (uint32_t x = x0 + threadIdx.x; x < x0 + kLrChunk && x < X; x += 256u) {
            const uint32_t v = cells[x];
            if (v) {
                const unsigned long long p = pack_best((int32_t)(v + add), side ? (Xj << 16) | x : (x << 16) | Xj);
                mp = p > mp ? p : mp;
            }
        }
        mp = wave_max_u64(mp);
        // (maxp only grows: most waves see that another one already put a larger value there)
        if (lane_id() == 0 && mp > __hip_atomic_load(&bs->maxp[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&bs->maxp[j], mp);
    }
}

// How many pairs of the batch the sequential algorithm would really have chosen in this order
// (see above).  Then the deltas of the surviving prefix are made exact for "only the prefix is
// merged", and the argmax bounds of the dropped pairs are restored.
constexpr int kValThreads = kBatchMax < 256 ? 256 : kBatchMax > 1024 ? 1024 : kBatchMax;
constexpr int kValSlots = kBatchMax < 256 ? 256 : kBatchMax;      // pairs the workgroup scans: kValSlots / kValThreads per thread
// The pairs that only exist through touching matches, folded into maxp[] with the whole chip (the n x n block
// of ADJ is mostly zeros; one thread per cell).  After k_adj_sums (adj_in / adj_out) and before k_validate.
__global__ __launch_bounds__(256) void k_adj_max(const uint32_t *__restrict__ hdr_adj, BatchState *bs, const DevCtl *ctl) {
    const uint32_t n = ctl->batch_n;
    if (n < 2) return;
    const uint32_t X0 = 256u + ctl->k_done;
    // rows p of the n x n block, a wave per row at a time, lanes along q (the cells of a row are contiguous)
    const uint32_t n_waves = gridDim.x * (blockDim.x / kWave);
    for (uint32_t p = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; p < n; p += n_waves) {
        for (uint32_t q = lane_id(); q < n; q += kWave) {
            const uint32_t w = hdr_adj[p * ctl->adj_pitch + q];
            if (!w) continue;
            const uint32_t bp = bs->key[p] & 0xFFFFu, aq = bs->key[q] >> 16;
            // (maxp only grows: an atomic only where the value would still raise it)
            auto raise = [&](uint32_t j, unsigned long long v) {
                if (v > __hip_atomic_load(&bs->maxp[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&bs->maxp[j], v);
            };
            raise(q > p ? q : p, pack_best((int32_t)w, ((X0 + p) << 16) | (X0 + q)));                      // (X_p, X_q)
            raise(q, pack_best((int32_t)bs->adj_in[q], (bp << 16) | (X0 + q)));          // (b_p, X_q), p not merged
            raise(p, pack_best((int32_t)bs->adj_out[p], ((X0 + p) << 16) | aq));         // (X_p, a_q), q not merged
        }
    }
}

__global__ __launch_bounds__(kValThreads) void k_validate(PairTable t, DevCtl *ctl, BatchState *bs, uint32_t *hdr_m,
                                                          uint32_t *hdr_adj, uint32_t *LR) {
    static_assert(kValSlots % kValThreads == 0, "one workgroup validates a batch, kValSlots / kValThreads pairs per thread");
    constexpr uint32_t kPer = kValSlots / kValThreads;
    __shared__ unsigned long long s_run[kValSlots];
    __shared__ uint32_t s_commit, s_minfrac;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = ctl->batch_n;
    if (n < 2) return;
    if (tid == 0) s_minfrac = 0xFFFFFFFFu;
    const uint32_t X0 = 256u + ctl->k_done;
    const uint32_t pitch = lr_pitch(X0);
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) { const uint32_t p = tid + u * kValThreads; s_run[p] = p < n ? bs->maxp[p] : 0ull; }
    if (tid == 0) s_commit = n;
    __syncthreads();
    // (the pairs that only exist through touching matches were folded into maxp by k_adj_max)
    uint32_t nsh = 6;
    while ((1u << nsh) < n) ++nsh;
    for (uint32_t d = 1; d < (uint32_t)kValSlots; d <<= 1) {        // inclusive prefix maximum
        unsigned long long o[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) { const uint32_t p = tid + u * kValThreads; o[u] = p >= d ? s_run[p - d] : 0ull; }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) { const uint32_t p = tid + u * kValThreads; if (o[u] > s_run[p]) s_run[p] = o[u]; }
        __syncthreads();
    }
    // (`first` mode: a created pair with the SAME count would be ranked by position, which a batch cannot know)
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t p = tid + u * kValThreads;
        if (p >= 1 && p < n &&
            (ctl->first_mode ? (bs->packed[p] >> 32) <= (s_run[p - 1] >> 32) : bs->packed[p] <= s_run[p - 1]))
            atomicMin(&s_commit, p);
    }
    // candidates the selection passed over because an earlier member eats some of their occurrences:
    // (c, a_i) loses L_i[c], (b_i, d) loses R_i[d].  With those measured, the candidate must rank below
    // every member chosen after it; the batch ends at the first member it still beats.
    // (a wave per passed-over candidate, its lanes striding over the members)
    {
        const uint32_t lane = tid % kWave, n_w = blockDim.x / kWave, n_skip = bs->skip_n;
        for (uint32_t si = tid / kWave; si < n_skip; si += n_w) {
            const uint32_t pos = bs->skip_pos[si], key = bs->skip_key[si];
            const uint32_t c = key >> 16, d = key & 0xFFFFu;
            unsigned long long red = 0;
            uint32_t n_dep = 0;
            for (uint32_t i = lane; i < pos && i < n; i += kWave) {
                const uint32_t ai = bs->key[i] >> 16, bi = bs->key[i] & 0xFFFFu;
                if (d == ai) { red += LR[lr_idx(pitch, c, i, 0)]; ++n_dep; }
                if (c == bi) { red += LR[lr_idx(pitch, d, i, 1)]; ++n_dep; }
            }
#pragma unroll
            for (int dd = kWave / 2; dd > 0; dd >>= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)red, dd, kWave), hi = __shfl_xor((uint32_t)(red >> 32), dd, kWave);
                red += ((unsigned long long)hi << 32) | lo;
            }
            n_dep = wave_sum(n_dep);
            const unsigned long long cnt0 = bs->skip_packed[si] >> 32;
            const unsigned long long frac = cnt0 && n_dep ? ((red << 16) / cnt0) / n_dep : 0ull;
            const unsigned long long later = pack_best((int32_t)(cnt0 > red ? cnt0 - red : 0ull), key);
            uint32_t first = 0xFFFFFFFFu;                       // first member behind it that it still beats
            const bool by_count = ctl->first_mode != 0;      // (an equal count would be ranked by position)
            for (uint32_t m = pos + lane; m < n; m += kWave)
                if (by_count ? (bs->packed[m] >> 32) <= (later >> 32) : bs->packed[m] < later) { first = m; break; }
#pragma unroll
            for (int dd = kWave / 2; dd > 0; dd >>= 1) { const uint32_t o = __shfl_xor(first, dd, kWave); first = o < first ? o : first; }
            if (lane == 0) {
                atomicMin(&s_minfrac, (uint32_t)(frac > 65535ull ? 65535ull : frac));
                if (first != 0xFFFFFFFFu) {
                    atomicMin(&s_commit, first);
                    atomicAdd(&ctl->n_skip_cut, 1u);
                    ctl->skip_failed = 1;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t commit = s_commit;
    // a match of a kept pair that touches a match of a dropped pair keeps its plain neighbour
    // (nothing to do when the whole batch is kept, which is nearly always)
    if (commit < n)
    for (uint32_t i2 = tid; i2 < (n << nsh); i2 += blockDim.x) {
        const uint32_t r = i2 >> nsh, q = i2 & ((1u << nsh) - 1u);
        if (q >= n) continue;
        const uint32_t i = r * ctl->adj_pitch + q;
        const uint32_t w = hdr_adj[i];
        if (!w) continue;
        if (r >= commit && q < commit) {            // dropped match r directly before kept match q
            atomicAdd(&LR[lr_idx(pitch, bs->key[r] & 0xFFFFu, q, 0)], w);
            hdr_adj[i] = 0;
        } else if (r < commit && q >= commit) {     // kept match r directly before dropped match q
            atomicAdd(&LR[lr_idx(pitch, bs->key[q] >> 16, r, 1)], w);
            hdr_adj[i] = 0;
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t p = tid + u * kValThreads;
        if (p >= commit && p < n) {
            const uint32_t e = bs->eidx[p];
            atomicMax(&t.bmax[e >> kBlockShift], bs->packed[p]);
            atomicMax(&t.smax[e >> (2 * kBlockShift)], bs->packed[p]);
        }
    }
    if (tid == 0) {
        // (k_sel_pick ends a batch where a passed-over candidate, less this fraction, would still beat the next member)
        if (s_minfrac != 0xFFFFFFFFu) ctl->skip_red_q16 = (3u * ctl->skip_red_q16 + s_minfrac) / 4u;
        ctl->commit_n = commit;
        ctl->n_validation_drops += n - commit;
        if (ctl->fused && commit < n) ctl->rm = 0;      // k_rewrite_marked recounts for the prefix
    }
}

// table updates of the surviving prefix; clears every delta of the batch
__global__ void k_apply_batch(PairTable t, DevCtl *ctl, const BatchState *bs, uint32_t *hdr_m, uint32_t *hdr_adj,
                              uint32_t *LR) {
    const uint32_t n = ctl->batch_n;
    if (n < 2) return;
    const uint32_t commit = ctl->commit_n;
    const uint32_t k0 = ctl->k_done;
    const uint32_t X0 = 256u + k0;
    const uint32_t pitch = lr_pitch(X0);
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {   // (work item = kLrChunk cells of one row of the LR block, as in k_delta_max)
        const uint32_t chunks = (X0 + kLrChunk - 1u) / kLrChunk;
        const uint32_t n_items = 2u * n * chunks;
        for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
            const uint32_t row = item / chunks, x0 = (item - row * chunks) * kLrChunk;
            const uint32_t j = row >> 1, side = row & 1u, X = X0 + j;
            const uint32_t a = bs->key[j] >> 16, b = bs->key[j] & 0xFFFFu;
            uint32_t *cells = LR + (size_t)row * pitch;
            for (uint32_t x = x0 + threadIdx.x; x < x0 + kLrChunk && x < X0; x += blockDim.x) {
                const uint32_t v = cells[x];
                if (!v) continue;
                cells[x] = 0;
                if (j >= commit) continue;
                if (!side) {
                    table_add(t, ctl, (x << 16) | a, -(int32_t)v, false);
                    table_add(t, ctl, (x << 16) | X, (int32_t)v, true);
                } else {
                    table_add(t, ctl, (b << 16) | x, -(int32_t)v, false);
                    table_add(t, ctl, (X << 16) | x, (int32_t)v, true);
                }
            }
        }
    }
    // (the rows p < n of the ADJ block, whatever the grid)
    const uint32_t adj_pitch = ctl->adj_pitch;
    for (uint64_t g = gid; g < (uint64_t)n * adj_pitch; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = (uint32_t)(g / adj_pitch), j = (uint32_t)(g % adj_pitch);
        const uint32_t adj = hdr_adj[g];
        if (adj) {
            hdr_adj[g] = 0;
            if (p < commit && j < commit) {     // match of p directly followed by a match of j
                const uint32_t bp = bs->key[p] & 0xFFFFu, aj = bs->key[j] >> 16;
                table_add(t, ctl, (bp << 16) | aj, -(int32_t)adj, false);
                table_add(t, ctl, ((X0 + p) << 16) | (X0 + j), (int32_t)adj, true);
            }
        }
    }
    const bool is_tt = gid < n && (bs->key[gid] >> 16) == (bs->key[gid] & 0xFFFFu);
    if (gid < commit || is_tt) {
        // every occurrence of an (a,b), a != b, pair is merged: its count drops to 0; a (t,t) member
        // loses its matches, counted by the stream pass (overlapping occurrences: m < count)
        uint32_t m = (uint32_t)(bs->packed[gid] >> 32);
        if (is_tt) { m = hdr_m[gid]; hdr_m[gid] = 0; }
        if (m && gid < commit) table_add(t, ctl, bs->key[gid], -(int32_t)m, false);
    }
}

// The same for the dense pair table, arranged so that three of the four updates per delta are
// contiguous.  A workgroup takes 64 ids x and 64 pairs j through LDS:
//   lanes along j:  (x, a_j) -= L   (scattered over row x)     (x, X_j)  = L   (64 neighbouring cells of row x)
//   lanes along x:  (b_j, x) -= R   (64 neighbouring cells of row b_j)   (X_j, x) = R   (... of row X_j)
// (neighbouring = runs of 32 contiguous cells in the tiled layout)
// A new token's pairs are plain stores; their argmax bounds are raised once per wave.
constexpr int kApplyTile = 64;
constexpr uint32_t kApplyJParts = 16;
#ifndef MBPE_APPLY_LOADS
#define MBPE_APPLY_LOADS 1
#endif
#ifndef MBPE_APPLY_BOUNDS
#define MBPE_APPLY_BOUNDS 0
#endif
#ifndef MBPE_APPLY_FLIGHT
#define MBPE_APPLY_FLIGHT 4
#endif
constexpr uint32_t kApplyFlight = MBPE_APPLY_FLIGHT;     // decrements (their old values come back) in flight per lane

// Inserting runs of neighbouring cells: dense_insert_store writes the cells (new pairs: plain stores) and hands back each
// lane's packed value; the caller keeps a running maximum per tile and raises the argmax bounds once at the end
// (dense_raise_bounds) -- reading a bound before every run would be a chain of dependent loads.  n_new (uniform)
// collects the number of inserts: one atomic on ctl->n_entries per wave, that counter being a single address.
__device__ __forceinline__ unsigned long long dense_insert_store(const PairTable &t, bool active, uint32_t e, uint32_t key,
                                                                 uint32_t count, uint32_t &n_new) {
    if (active) t.cells[e] = kPresent | count;
    n_new += (uint32_t)__popcll(__ballot(active));
    return active ? pack_best((int32_t)count, key) : 0ull;
}
// The bounds a wave has to raise are collected first -- (tile, maximum) pairs in a small per-wave list in LDS -- and
// raised together at the end of the step, one lane per pair: ONE round trip for "is the bound lower?" instead of one
// per tile (a dozen dependent loads per step were a quarter of this kernel's time).
constexpr uint32_t kRaiseMax = 16;
struct RaiseList { unsigned long long pm[kRaiseMax]; uint32_t blk[kRaiseMax]; };
__device__ __forceinline__ void dense_raise_flush(const PairTable &t, RaiseList &rl, uint32_t &n) {
    const uint32_t lane = lane_id();
    if (lane < n) {
        const unsigned long long pm = rl.pm[lane];
        const uint32_t blk0 = rl.blk[lane];
        if (pm > t.bmax[blk0]) atomicMax(&t.bmax[blk0], pm);
        if (pm > t.smax[blk0 >> kBlockShift]) atomicMax(&t.smax[blk0 >> kBlockShift], pm);
    }
    n = 0;
}
__device__ __forceinline__ void dense_raise_bounds(const PairTable &t, unsigned long long p, uint32_t blk, RaiseList &rl,
                                                   uint32_t &n) {
    // p: this lane's maximum (0: none) among the cells it inserted into tile blk; one list entry per tile touched
#ifdef MBPE_APPLY_NORAISE
    return;                      // (timing-only build)
#endif
    unsigned long long m = __ballot(p != 0ull);
    while (m) {
        const uint32_t blk0 = rfl(__shfl(blk, (uint32_t)__builtin_ctzll(m), kWave));
        const bool mine = p != 0ull && blk == blk0;
        const unsigned long long pm = wave_max_u64(mine ? p : 0ull);
        if (n == kRaiseMax) dense_raise_flush(t, rl, n);          // (uniform; does not happen with 64 x 64 steps)
        if (lane_id() == 0) { rl.pm[n] = pm; rl.blk[n] = blk0; }
        ++n;
        m &= ~__ballot(mine);
    }
}

__global__ __launch_bounds__(256) void k_apply_batch_dense(PairTable t, DevCtl *ctl, const BatchState *bs,
                                                           uint32_t *hdr_m, uint32_t *hdr_adj, uint32_t *LR, uint32_t j_parts) {
    __shared__ uint2 tile[kApplyTile][kApplyTile + 1];
    __shared__ uint32_t keys[kApplyTile];           // bs->key[j0 ..]
    __shared__ RaiseList raise[256 / kWave];
    uint32_t n_raise = 0;                           // uniform per wave
    const uint32_t n = ctl->batch_n;
    if (n < 2) return;
    const uint32_t commit = ctl->commit_n;
    const uint32_t X0 = 256u + ctl->k_done;
    // (a workgroup takes 64 ids x and walks the pairs in steps of kApplyJParts tiles of 64: the grid does not grow
    //  with the batch cap)
    const uint32_t x0 = (blockIdx.x / j_parts) * kApplyTile;
    const uint32_t lane = lane_id(), wave = threadIdx.x / kWave;
    for (uint32_t j0 = (blockIdx.x % j_parts) * kApplyTile; x0 < X0 && j0 < n; j0 += j_parts * kApplyTile) {
        __syncthreads();                 // (the tile of the previous step is done with)
        if (threadIdx.x < (uint32_t)kApplyTile) keys[threadIdx.x] = j0 + threadIdx.x < n ? bs->key[j0 + threadIdx.x] : 0u;
        // load (and clear) the deltas of ids x0.. and pairs j0..: the rows L_j, R_j of LR are contiguous along x
        const uint32_t pitch = lr_pitch(X0);
#if MBPE_APPLY_LOADS == 1
        {
            // (a wave's 32 loads are issued before its first store: the rows are independent, which the compiler cannot know)
            constexpr uint32_t kLd = kApplyTile / (256 / kWave);
            const uint32_t x = x0 + lane;
            uint32_t vl[kLd], vr[kLd];
#pragma unroll
            for (uint32_t u = 0; u < kLd; ++u) {
                const uint32_t j = j0 + wave + u * (256 / kWave);
                const bool ok = x < X0 && j < n;
                const uint32_t *cl = LR + (size_t)(2u * (ok ? j : 0u)) * pitch + (ok ? x : 0u);
                vl[u] = ok ? *cl : 0u;
                vr[u] = ok ? *(cl + pitch) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kLd; ++u) {
                const uint32_t c = wave + u * (256 / kWave), j = j0 + c;
                uint32_t *cl = LR + (size_t)(2u * j) * pitch + x;
                if (vl[u]) *cl = 0;
                if (vr[u]) *(cl + pitch) = 0;
                tile[lane][c] = j < commit ? make_uint2(vl[u], vr[u]) : make_uint2(0, 0);
            }
        }
#else
        for (uint32_t c = wave; c < (uint32_t)kApplyTile; c += 256 / kWave) {
            const uint32_t x = x0 + lane, j = j0 + c;
            uint2 lr = make_uint2(0, 0);
            if (x < X0 && j < n) {
                uint32_t *cl = LR + (size_t)(2u * j) * pitch + x, *cr = cl + pitch;
                lr = make_uint2(*cl, *cr);
                if (lr.x) *cl = 0;
                if (lr.y) *cr = 0;
                if (j >= commit) lr = make_uint2(0, 0);
            }
            tile[lane][c] = lr;
        }
#endif
        __syncthreads();
        // The decrements return the old value (an absent pair or a negative count is an error worth
        // knowing about); four of them are in flight per lane before the first one is looked at.
        constexpr uint32_t kRows = kApplyTile / (256 / kWave);        // rows (columns) per wave: 16
        uint32_t err = 0, n_new = 0;
        // lanes along j: left neighbours x
        const uint32_t jl = j0 + lane;
        const uint32_t a = keys[lane] >> 16;
        // (new pairs (x, X_j): this lane's column X0 + jl, rows x0 .. x0 + 63 = two rows of tiles)
        unsigned long long accL[2] = {0ull, 0ull};
        for (uint32_t r0 = 0; r0 < kRows; r0 += kApplyFlight) {
            uint32_t l[kApplyFlight], old[kApplyFlight];
#pragma unroll
            for (uint32_t u = 0; u < kApplyFlight; ++u) {
                const uint32_t r = wave + (r0 + u) * (256 / kWave), x = x0 + r;
                l[u] = tile[r][lane].x;
                old[u] = kPresent | l[u];
#ifdef MBPE_APPLY_NORETURN
                if (l[u]) atomicAdd(&t.cells[dense_index(t, (x << 16) | a)], 0u - l[u]);
#else
                if (l[u]) old[u] = atomicAdd(&t.cells[dense_index(t, (x << 16) | a)], 0u - l[u]);
#endif
            }
#pragma unroll
            for (uint32_t u = 0; u < kApplyFlight; ++u) {
                const uint32_t r = wave + (r0 + u) * (256 / kWave), x = x0 + r;
                const unsigned long long p = dense_insert_store(t, l[u] != 0u, dense_index(t, (x << 16) | (X0 + jl)),
                                                                (x << 16) | (X0 + jl), l[u], n_new);
                const uint32_t rb = r >> 5;                 // (x0 is a multiple of 64)
                accL[rb] = p > accL[rb] ? p : accL[rb];
            }
#pragma unroll
            for (uint32_t u = 0; u < kApplyFlight; ++u)
                err |= !(old[u] & kPresent) ? kErrMissingPair : ((old[u] & ~kPresent) < l[u] ? kErrNegCount : 0u);
        }
#pragma unroll
        for (uint32_t rb = 0; rb < 2; ++rb)
            dense_raise_bounds(t, accL[rb], dense_index(t, ((x0 + 32u * rb) << 16) | (X0 + jl)) >> kBlockShift, raise[wave], n_raise);
        // lanes along x: right neighbours x
        const uint32_t xr = x0 + lane;
        // (new pairs (X_j, x): rows X0 + j0 .. + 63 = up to three rows of tiles, this lane's column xr)
        unsigned long long accR[3] = {0ull, 0ull, 0ull};
        const uint32_t Xrow0 = (X0 + j0) >> 5;
        for (uint32_t c0 = 0; c0 < kRows; c0 += kApplyFlight) {
            uint32_t rr[kApplyFlight], old[kApplyFlight];
#pragma unroll
            for (uint32_t u = 0; u < kApplyFlight; ++u) {
                const uint32_t c = wave + (c0 + u) * (256 / kWave), j = j0 + c;
                rr[u] = j < n ? tile[lane][c].y : 0u;
                old[u] = kPresent | rr[u];
#ifdef MBPE_APPLY_NORETURN
                if (rr[u]) atomicAdd(&t.cells[dense_index(t, ((keys[c] & 0xFFFFu) << 16) | xr)], 0u - rr[u]);
#else
                if (rr[u]) old[u] = atomicAdd(&t.cells[dense_index(t, ((keys[c] & 0xFFFFu) << 16) | xr)], 0u - rr[u]);
#endif
            }
#pragma unroll
            for (uint32_t u = 0; u < kApplyFlight; ++u) {
                const uint32_t c = wave + (c0 + u) * (256 / kWave), X = X0 + j0 + c;
                const unsigned long long p = dense_insert_store(t, rr[u] != 0u, dense_index(t, (X << 16) | xr), (X << 16) | xr,
                                                                rr[u], n_new);
                const uint32_t rb = (X >> 5) - Xrow0;       // 0, 1 or 2
                if (rb == 0) accR[0] = p > accR[0] ? p : accR[0];
                else if (rb == 1) accR[1] = p > accR[1] ? p : accR[1];
                else accR[2] = p > accR[2] ? p : accR[2];
            }
#pragma unroll
            for (uint32_t u = 0; u < kApplyFlight; ++u)
                err |= !(old[u] & kPresent) ? kErrMissingPair : ((old[u] & ~kPresent) < rr[u] ? kErrNegCount : 0u);
        }
#pragma unroll
        for (uint32_t rb = 0; rb < 3; ++rb)
            dense_raise_bounds(t, accR[rb], dense_index(t, (((Xrow0 + rb) << 5) << 16) | xr) >> kBlockShift, raise[wave], n_raise);
        dense_raise_flush(t, raise[wave], n_raise);
        if (err) atomicOr(&ctl->err, err);
        if (lane == 0 && n_new) atomicAdd(&ctl->n_entries, n_new);
    }
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (the rows p < n of the ADJ block, whatever the grid)
    const uint32_t adj_pitch = ctl->adj_pitch;
    for (uint64_t g = gid; g < (uint64_t)n * adj_pitch; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = (uint32_t)(g / adj_pitch), j = (uint32_t)(g % adj_pitch);
        const uint32_t adj = hdr_adj[g];
        if (adj) {
            hdr_adj[g] = 0;
            if (p < commit && j < commit) {     // match of p directly followed by a match of j
                const uint32_t bp = bs->key[p] & 0xFFFFu, aj = bs->key[j] >> 16;
                table_add(t, ctl, (bp << 16) | aj, -(int32_t)adj, false);
                table_add(t, ctl, ((X0 + p) << 16) | (X0 + j), (int32_t)adj, true);
            }
        }
    }
    const bool is_tt = gid < n && (bs->key[gid] >> 16) == (bs->key[gid] & 0xFFFFu);
    if (gid < commit || is_tt) {
        // every occurrence of an (a,b), a != b, pair is merged: its count drops to 0; a (t,t) member
        // loses its matches, counted by the stream pass (overlapping occurrences: m < count)
        uint32_t m = (uint32_t)(bs->packed[gid] >> 32);
        if (is_tt) { m = hdr_m[gid]; hdr_m[gid] = 0; }
        if (m && gid < commit) table_add(t, ctl, bs->key[gid], -(int32_t)m, false);
    }
}

// The rewriting half: tiles marked by the scan pass, pairs of the validated prefix.
// k_list_marked turns the bitmap into a dense list so that k_rewrite_marked can
// walk it with the same prefetch ring as the streaming passes.
__global__ void k_list_marked(const uint32_t *__restrict__ chg, uint32_t n_words, uint32_t *__restrict__ list,
                              DevCtl *ctl, uint32_t n_tiles) {
    if (ctl->batch_n < 2 || (ctl->fused && ctl->commit_n == ctl->batch_n)) return;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bits = w < n_words ? chg[w] : 0u;
    if (ctl->marks_all && w < n_words) {           // an abandoned fused pass: it marked nothing, every tile is a candidate
        const uint32_t left = n_tiles - w * 32u;
        bits = left >= 32u ? 0xFFFFFFFFu : (1u << left) - 1u;
    }
    const uint32_t cnt = __popc(bits);
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t total = __shfl(incl, kWave - 1, kWave);
    uint32_t base = 0;
    if (lane_id() == 0 && total) base = atomicAdd(&ctl->n_marked, total);
    base = __shfl(base, 0, kWave) + incl - cnt;
    while (bits) {
        list[base++] = w * 32u + (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1;
    }
}

template <int MODE, bool TT>
__global__ __launch_bounds__(kLutThreads) void k_rewrite_marked(uint16_t *tok0, uint16_t *tok1,
                                                                  const TileSum *__restrict__ sin,
                                                                  TileSum *__restrict__ sout, uint32_t n_tiles,
                                                                  uint32_t *__restrict__ chg,
                                                                  const uint32_t *__restrict__ list,
                                                                  const BatchState *bs, DevCtl *ctl,
                                                                  const RankEdge *le, const RankEdge *re,
                                                                  const uint32_t *__restrict__ run_in) {
    constexpr uint32_t idmask = MODE == 1 ? 0x7FFFu : 0xFFFFu;
    constexpr uint32_t endbit = MODE == 1 ? kEndBit : 0u;
    __shared__ BatchLutMem lut_mem;
    if (ctl->batch_n < 2 || (ctl->fused && ctl->commit_n == ctl->batch_n)) return;
    uint16_t *tok = ctl->cur ? tok1 : tok0;
    const uint32_t n_keys = ctl->commit_n;
    if ((bs->tt_index < n_keys) != TT) return;      // (t,t) member inside the kept prefix: see k_fused_batch
    __shared__ TTInfo ti;
    __shared__ __attribute__((aligned(16))) uint16_t stage_mem[kLutThreads / kWave][kTileSlots];
    uint16_t *stage = stage_mem[threadIdx.x / kWave];
    const uint32_t X0 = 256u + ctl->k_done;
    const uint32_t n_list = ctl->n_marked;
    const bool marks_all = ctl->marks_all != 0u;
    BatchLut lut(&lut_mem, bs, idmask);
    lut_build(lut, bs, n_keys, idmask - 1u);
    if (TT) tt_build(ti, bs, n_keys, idmask - 1u);
    const uint32_t lane = lane_id();
    const uint32_t waves_per_block = kLutThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    uint32_t i = rfl(blockIdx.x * waves_per_block + threadIdx.x / kWave);
    if (i >= n_list) return;
    const __amdgpu_buffer_rsrc_t sums_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TileSum *>(sin), 0, n_tiles * 16u, 0x00020000);
    const uint32_t last = n_list - 1;
    auto tile_at = [&](uint64_t k) { return list[k < n_list ? k : last]; };
    uint32_t tile0 = tile_at(i), tile1 = tile_at((uint64_t)i + n_waves), tile2 = tile_at((uint64_t)i + 2ull * n_waves);
    TileIn t0 = tile_issue(tok, sums_rsrc, tile0);
    TileIn t1 = tile_issue(tok, sums_rsrc, tile1);
    uint32_t wave_rm = 0;              // per lane (chain-walking path), summed over the wave at the end
    uint32_t wave_rm_uniform = 0;      // whole-wave count (prefix-form path)
    for (;;) {
        const bool has_next = (uint64_t)i + n_waves < n_list;
        const uint32_t tile3 = tile_at((uint64_t)i + 3ull * n_waves);
        const TileIn t2 = tile_issue(tok, sums_rsrc, tile2);

        const uint32_t tile = tile0;
        uint32_t s[8];
        unpack8(t0.q, s);
        Halo h;
        const uint32_t pw1 = rlane(t0.smw, 1), pw2 = rlane(t0.smw, 2);
        const uint32_t nw0 = rlane(t0.smw, 8), nw2 = rlane(t0.smw, 10);
        const bool fast = tile > 0 && tile + 1 < n_tiles && (pw2 & 0xFFFFu) >= 2 && (nw2 & 0xFFFFu) >= 2;
        if (fast) {
            h.p1 = pw1 >> 16; h.p2 = pw1 & 0xFFFFu;
            h.n1 = nw0 & 0xFFFFu; h.n2 = nw0 >> 16;
        } else {
            h = halo_slow(sin, n_tiles, tile, le, re);
        }
        if constexpr (MBPE_FUSED_PF != 0) {
            // tiles are in prefix form: the fused pass's tile function without the count deltas, stored in place
            // ((t,t) members: every tile renamed first, as the chain-walking path did)
            uint32_t rm = 0;
            bool wrote = false;          // uniform: the tile changed (its new summary is in the side array)
            bool no_rename = false;
            DeltaCacheSmall no_dc;       // (never touched: no deltas in this instantiation)
            if (TT) tt_rename<MODE>(s, h, ti, run_in[tile]);
            const uint4 qn = fused_tile_pf<MODE, 2, true, TT>(t0.q, s, h, rlane(t0.smw, 4), rlane(t0.smw, 5), rlane(t0.smw, 6), lut,
                                                              X0, tile, sout, nullptr, nullptr, no_dc, false, rm, wrote,
                                                              __amdgpu_buffer_rsrc_t(), 0u, stage, chg, TT, &ti, no_rename);
            if (wrote) {
                reinterpret_cast<uint4 *>(tok)[(uint64_t)tile * kWave + lane] = qn;
                if (lane == 0 && marks_all) atomicOr(&chg[tile >> 5], 1u << (tile & 31u));    // (the fused pass set no marks)
            } else if (lane == 0 && !marks_all) {
                atomicAnd(&chg[tile >> 5], ~(1u << (tile & 31u)));    // marked for a pair that was dropped
            }
            wave_rm_uniform += rm;
        } else {
        if (TT) tt_rename<MODE>(s, h, ti, run_in[tile]);
        const Neigh nb = tile_neighbours(s, h);
        bool changed = false, a1 = false, first = true;
        uint32_t p1 = nb.p1_in, my_rm = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t self = s[j];
            if (self == kHole) continue;
            const uint32_t n1 = nb.n1v[j];
            if (first) { a1 = p1 != kHole && pair_test(lut, p1, self & idmask); first = false; }
            bool is_a = false;
            if (a1) {                      // second token of a match: becomes a hole
                s[j] = kHole;
                changed = true;
                ++my_rm;
            } else if (n1 != kHole && pair_test(lut, self, n1 & idmask)) {
                is_a = true;
                s[j] = (X0 + (uint32_t)lut_index(lut, self, n1 & idmask)) | (n1 & endbit);
                changed = true;
            }
            a1 = is_a;
            p1 = self;
        }
        if (TT) {                           // (every renamed token was the second token of a match; be safe)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s[j] != kHole && (s[j] & idmask) >= idmask - (uint32_t)kTTMax)
                    s[j] = ti.tok[idmask - 1u - (s[j] & idmask)] | (s[j] & endbit);
        }
        wave_rm += my_rm;
        if (__ballot(changed) != 0ull) {
            uint32_t n_out;
            const uint4 qc = tile_compact(s, stage, n_out);
            reinterpret_cast<uint4 *>(tok)[(uint64_t)tile * kWave + lane] = qc;
            const uint4 ns = wave_summary(s);
            if (lane == 0) {
                reinterpret_cast<uint4 *>(sout)[tile] = ns;
                if (marks_all) atomicOr(&chg[tile >> 5], 1u << (tile & 31u));    // (the fused pass set no marks)
            }
        } else if (lane == 0 && !marks_all) {
            atomicAnd(&chg[tile >> 5], ~(1u << (tile & 31u)));    // marked for a pair that was dropped
        }
        }

        if (!has_next) break;
        i += n_waves;
        tile0 = tile1; tile1 = tile2; tile2 = tile3;
        t0 = t1; t1 = t2;
    }
    const uint32_t tr = wave_sum(wave_rm) + wave_rm_uniform;
    if (lane == 0 && tr) atomicAdd(&ctl->rm, tr);
}

// last kernel of a sequence: advance the merge counter
__global__ void k_seq_finish(DevCtl *ctl, uint32_t *fused_flag, const BatchState *bs) {
    if (blockIdx.x || threadIdx.x) return;
    const uint32_t bs_skip_n = bs->skip_n;
    const uint32_t rm_seq = ctl->batch_n >= 2 ? ctl->rm : 0u;      // tokens this sequence removed = its matches (this rank)
    if (fused_flag) *fused_flag = ctl->fused && ctl->batch_n >= 2 ? 1u : 0u;
    if (ctl->batch_n >= 2) {       // (a single-pair batch was accounted by k_apply)
        const uint32_t rm = ctl->rm;
        ctl->removed_total += rm;
        ctl->n_live -= rm;
        ctl->rm = 0;
    }
    if (ctl->batch_n >= 2) {
        // learn how many pairs validation lets through: after a drop aim at twice what survived,
        // after a full batch that hit the limit double it
        uint32_t lim = ctl->adapt_limit ? ctl->adapt_limit : (uint32_t)kBatchMax;
        if (ctl->commit_n < ctl->batch_n) lim = ctl->commit_n * 2u < 4u ? 4u : ctl->commit_n * 2u;
        else if (ctl->batch_n >= lim) lim = lim * 2u;
        ctl->adapt_limit = lim > (uint32_t)kBatchMax ? (uint32_t)kBatchMax : lim;
    }
    // passing over dependent candidates: back off after a failure, recover after successes
    if (ctl->skip_off) ctl->skip_off -= 1;
    if (ctl->batch_n >= 2 && ctl->skip_failed) {
        const uint32_t pen = ctl->skip_penalty ? (ctl->skip_penalty * 2u > (uint32_t)MBPE_SKIP_PENALTY_MAX ? (uint32_t)MBPE_SKIP_PENALTY_MAX : ctl->skip_penalty * 2u)
                                               : (uint32_t)MBPE_SKIP_PENALTY0;
        ctl->skip_penalty = pen;
        ctl->skip_off = pen;
    } else if (ctl->batch_n >= 2 && bs_skip_n > 0 && ctl->skip_penalty) {
        ctl->skip_penalty /= 2u;
    }
    ctl->skip_failed = 0;
    // the pairs' cell blocks: once fewer than half of a large batch's matches lay between two raw bytes the stream is
    // past the stage where that pays (at a third the pass is already 2-8 % slower with the blocks than without) (the share only falls as tokens replace bytes)
    if (ctl->cells_on && ctl->cells_min >= kCellsMinBatch && ctl->batch_n >= ctl->cells_min &&
        ctl->cell_hits * 2ull < (unsigned long long)rm_seq)
        ctl->cells_on = 0;
    ctl->cell_hits = 0;
    if (ctl->fused && ctl->batch_n >= 2) {
        ctl->n_fused += 1;
        if (ctl->commit_n == ctl->batch_n) ctl->cur ^= 1u;   // the fused pass's output becomes the stream
        else ctl->n_fused_dropped += 1;
    }
    ctl->fused = 0;
    ctl->marks_all = 0;
    if (ctl->commit_n) ctl->size_hist[31 - __builtin_clz(ctl->commit_n) > 7 ? 7 : 31 - __builtin_clz(ctl->commit_n)] += 1;
    if (fused_flag) {              // ("time_kernels": live tokens after this sequence and the merges it committed)
        fused_flag[1] = (uint32_t)ctl->n_live;
        fused_flag[2] = (uint32_t)(ctl->n_live >> 32);
        fused_flag[3] = ctl->commit_n;
    }
    ctl->k_done += ctl->commit_n;
    ctl->batch_n = 0;
    ctl->commit_n = 0;
    ctl->n_marked = 0;
}

// ---- compaction ---------------------------------------------------------------------

// exclusive scan of n_live over the tiles in three steps: sums of 4096-tile chunks, a scan of
// those sums by one workgroup, then the scan inside every chunk
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 16;
constexpr uint32_t kScanChunk = kScanThreads * kScanPerThread;

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *sh) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += shfl_xor_u64(v, d);
    __syncthreads();
    if (lane_id() == 0) sh[threadIdx.x / kWave] = v;
    __syncthreads();
    unsigned long long r = 0;
    for (uint32_t w = 0; w < blockDim.x / kWave; ++w) r += sh[w];
    return r;
}

__global__ __launch_bounds__(kScanThreads) void k_tile_scan_partial(const TileSum *__restrict__ sums,
                                                                    uint32_t n_tiles,
                                                                    unsigned long long *__restrict__ part) {
    __shared__ unsigned long long sh[kScanThreads / kWave];
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < kScanChunk; i += kScanThreads)
        if (base + i < n_tiles) s += sums[base + i].n_live;
    s = block_sum_u64(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_tile_scan_top(unsigned long long *__restrict__ part, uint32_t n_parts,
                                                        DevCtl *ctl) {
    __shared__ unsigned long long tot[1024];
    const uint32_t per = (n_parts + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per < n_parts ? threadIdx.x * per : n_parts;
    const uint32_t hi = lo + per < n_parts ? lo + per : n_parts;
    unsigned long long s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += part[i];
    tot[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned long long v = tot[i]; tot[i] = run; run += v; }
        ctl->scan_total = run;
    }
    __syncthreads();
    unsigned long long run = tot[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
}

__global__ __launch_bounds__(kScanThreads) void k_tile_scan_final(const TileSum *__restrict__ sums, uint32_t n_tiles,
                                                                  const unsigned long long *__restrict__ part,
                                                                  unsigned long long *__restrict__ offsets) {
    __shared__ unsigned long long th[kScanThreads];
    const uint64_t first = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        v[i] = first + i < n_tiles ? sums[first + i].n_live : 0u;
        s += v[i];
    }
    th[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = part[blockIdx.x];
        for (int i = 0; i < kScanThreads; ++i) { const unsigned long long t = th[i]; th[i] = run; run += t; }
    }
    __syncthreads();
    unsigned long long run = th[threadIdx.x];
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        if (first + i < n_tiles) offsets[first + i] = run;
        run += v[i];
    }
}

// A tile's live tokens go to dst[offsets[tile] ..): staged in LDS at the destination's alignment within a 16-byte
// vector, so that the body leaves as whole aligned vectors (one per lane) and only the ragged ends, which neighbouring
// tiles complete, as single 2-byte stores.
__global__ __launch_bounds__(kMergeThreads) void k_compact_scatter(const uint16_t *__restrict__ src,
                                                                   const TileSum *__restrict__ sums,
                                                                   const unsigned long long *__restrict__ offsets,
                                                                   uint32_t n_tiles, uint16_t *__restrict__ dst) {
    __shared__ __attribute__((aligned(16))) uint16_t stage[kMergeThreads / kWave][kTile + 16];
    const uint32_t waves_per_block = kMergeThreads / kWave;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    const uint32_t lane = lane_id();
    uint16_t *st = stage[threadIdx.x / kWave];
    for (uint32_t tile = blockIdx.x * waves_per_block + threadIdx.x / kWave; tile < n_tiles; tile += n_waves) {
        const uint32_t total = sums[tile].n_live;
        if (total == 0) continue;
        const uint4 q = reinterpret_cast<const uint4 *>(src)[(uint64_t)tile * kWave + lane];
        uint32_t s[8];
        unpack8(q, s);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += s[j] != kHole;
        const uint32_t excl = wave_incl_scan(cnt) - cnt;
        const unsigned long long d0 = offsets[tile];
        const uint32_t sh = (uint32_t)(d0 & 7ull);
        uint32_t at = sh + excl;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s[j] != kHole) st[at++] = (uint16_t)s[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // staged slot i is destination slot (d0 - sh) + i: vector v of the stage is an aligned vector of dst
        uint16_t *base = dst + (d0 - sh);
        const uint32_t end = sh + total;                         // staged slots [sh, end)
        const uint32_t v_first = sh ? 1u : 0u, v_last = end / 8u;       // whole vectors [v_first, v_last)
        for (uint32_t v = v_first + lane; v < v_last; v += kWave)
            reinterpret_cast<uint4 *>(base)[v] = reinterpret_cast<const uint4 *>(st)[v];
        if (v_last < v_first) {                                  // everything inside one vector, open at both ends
            if (lane < total) base[sh + lane] = st[sh + lane];
        } else {
            if (sh && lane < 8u - sh) base[sh + lane] = st[sh + lane];                  // head, up to the first boundary
            if (lane < end - v_last * 8u) base[v_last * 8u + lane] = st[v_last * 8u + lane];   // tail
        }
        __builtin_amdgcn_wave_barrier();                         // (the stage is reused by the wave's next tile)
    }
}

// ---- multi-GPU: rank edges ------------------------------------------------------------
// Exchange header of one merge (u32 words): [0] m, [1] adj, [2 + 8r ..] the
// RankEdge of rank r.  Every rank fills only its own slot (the others stay 0),
// so the sum all-reduce of the buffer doubles as an all-gather of the edges.

__global__ void k_rank_edge(const TileSum *__restrict__ sums, uint32_t n_tiles, RankEdge *out, const DevCtl *ctl,
                            uint32_t *hdr) {
    if (blockIdx.x || threadIdx.x) return;
    RankEdge e;
    e.head0 = e.head1 = e.tail0 = e.tail1 = kHole;
    e.n_live_lo = e.n_live_hi = e.tail_run_lo = e.tail_run_hi = 0;
    int need = 2;
    for (uint32_t j = 0; need && j < n_tiles; ++j) {
        TileSum s = sums[j];
        if (!s.n_live) continue;
        if (need == 2) { e.head0 = s.head0; need = 1; if (s.n_live >= 2) { e.head1 = s.head1; need = 0; } }
        else { e.head1 = s.head0; need = 0; }
    }
    need = 2;
    for (int64_t j = (int64_t)n_tiles - 1; need && j >= 0; --j) {
        TileSum s = sums[j];
        if (!s.n_live) continue;
        if (need == 2) { e.tail0 = s.tail0; need = 1; if (s.n_live >= 2) { e.tail1 = s.tail1; need = 0; } }
        else { e.tail1 = s.tail0; need = 0; }
    }
    unsigned long long run = 0;
    for (int64_t j = (int64_t)n_tiles - 1; j >= 0; --j) {
        TileSum s = sums[j];
        if (!s.n_live) continue;
        if (s.tail0 != e.tail0) break;
        run += s.tail_run;
        if (s.tail_run != s.n_live) break;
    }
    const unsigned long long live = ctl->n_live - ctl->rm;   // ctl->rm is folded in by k_apply later
    e.n_live_lo = (uint32_t)live; e.n_live_hi = (uint32_t)(live >> 32);
    e.tail_run_lo = (uint32_t)run; e.tail_run_hi = (uint32_t)(run >> 32);
    // kHole (0xFFFF) must survive a SUM with zeros from the other ranks: it does.
    *out = e;
    if (hdr) { hdr[0] = ctl->m; hdr[1] = ctl->adj; }
}

// What lies to the left / right of this rank's shard, looked through empty
// ranks: left.tail0/tail1/tail_run and right.head0/head1 in RankEdge form.
// Also clears the exchange header for the next merge.
__global__ void k_compose_edges(uint32_t *hdr, int rank, int n_ranks, RankEdge *left, RankEdge *right) {
    if (blockIdx.x || threadIdx.x) return;
    const RankEdge *all = reinterpret_cast<const RankEdge *>(hdr + 2);
    RankEdge l, r;
    l.head0 = l.head1 = l.tail0 = l.tail1 = kHole;
    l.n_live_lo = l.n_live_hi = l.tail_run_lo = l.tail_run_hi = 0;
    r = l;
    int need = 2;
    for (int j = rank - 1; need && j >= 0; --j) {
        const RankEdge e = all[j];
        const uint32_t nl = e.tail0 == kHole ? 0 : (e.tail1 == kHole ? 1 : 2);
        if (!nl) continue;
        if (need == 2) { l.tail0 = e.tail0; need = 1; if (nl >= 2) { l.tail1 = e.tail1; need = 0; } }
        else { l.tail1 = e.tail0; need = 0; }
    }
    unsigned long long run = 0;
    for (int j = rank - 1; j >= 0; --j) {
        const RankEdge e = all[j];
        const unsigned long long live = ((unsigned long long)e.n_live_hi << 32) | e.n_live_lo;
        if (!live) continue;
        if (e.tail0 != l.tail0) break;
        const unsigned long long tr = ((unsigned long long)e.tail_run_hi << 32) | e.tail_run_lo;
        run += tr;
        if (tr != live) break;
    }
    l.tail_run_lo = (uint32_t)run; l.tail_run_hi = (uint32_t)(run >> 32);
    need = 2;
    for (int j = rank + 1; need && j < n_ranks; ++j) {
        const RankEdge e = all[j];
        const uint32_t nl = e.head0 == kHole ? 0 : (e.head1 == kHole ? 1 : 2);
        if (!nl) continue;
        if (need == 2) { r.head0 = e.head0; need = 1; if (nl >= 2) { r.head1 = e.head1; need = 0; } }
        else { r.head1 = e.head0; need = 0; }
    }
    *left = l;
    *right = r;
    for (int i = 0; i < 2 + 8 * n_ranks; ++i) hdr[i] = 0;
}

// Begin of a multi-GPU run: the byte pairs that straddle two ranks' shards
// are added to the (already all-reduced) byte-pair table by every rank alike.
__global__ void k_boundary_pairs(uint32_t *bp, const uint32_t *hdr, int n_ranks, uint32_t endbit) {
    if (blockIdx.x || threadIdx.x) return;
    const RankEdge *all = reinterpret_cast<const RankEdge *>(hdr + 2);
    uint32_t prev_tail = kHole;
    for (int j = 0; j < n_ranks; ++j) {
        const RankEdge e = all[j];
        if (e.head0 == kHole) continue;            // empty shard
        const bool ends = endbit == kBarrier ? prev_tail == kBarrier : (prev_tail & endbit) != 0u;
        if (prev_tail != kHole && !ends)
            bp[((prev_tail & 0xFFu) << 8) | (e.head0 & 0xFFu)] += 1;
        prev_tail = e.tail0;
    }
}

inline int blocks_for(uint64_t n, int threads, int max_blocks) {
    uint64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

// workgroups of a kernel that fit one CU at a time (registers, LDS): the strided tile loops
// want exactly one resident "round" of workgroups, a second partial round would idle CUs
template <typename K>
inline int resident_blocks(K kernel, int threads = kMergeThreads) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 1024 / threads;
    return n > 8 ? 8 : n;
}

// grid for the wave-per-tile kernels: enough waves to fill the chip (8 blocks
// of 4 waves per CU), fewer when the stream is short
inline int tile_grid(uint32_t n_tiles, int n_cus, int blocks_per_cu = 8, int threads = kMergeThreads) {
    const uint32_t waves_per_block = threads / kWave;
    uint64_t blocks = ((uint64_t)n_tiles + waves_per_block - 1) / waves_per_block;
    const uint64_t cap = (uint64_t)(n_cus > 0 ? n_cus : 256) * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------

// The launchers take the chunk-end convention as `endbit`: 0 (one chunk), kEndBit (flag in the slot) or kBarrier
// (barrier slots); the kernels take it as the template parameter MODE = 0 / 1 / 2.
static inline int slot_mode(uint32_t endbit) { return endbit == kEndBit ? 1 : endbit == kBarrier ? 2 : 0; }
#define MBPE_BY_MODE(endbit, ...)                                                          \
    do {                                                                                   \
        if ((endbit) == kEndBit) { constexpr int M = 1; __VA_ARGS__; }                     \
        else if ((endbit) == kBarrier) { constexpr int M = 2; __VA_ARGS__; }               \
        else { constexpr int M = 0; __VA_ARGS__; }                                         \
    } while (0)

void launch_fill_u32(hipStream_t s, uint32_t *p, uint64_t n, uint32_t v) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, s, p, n, v);
}

void launch_fill_u16(hipStream_t s, uint16_t *p, uint64_t n, uint16_t v) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u16, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, s, p, n, v);
}

size_t pair_count_scratch_bytes(int n_workgroups) {
    const size_t wgs = (size_t)(n_workgroups < 1 ? 1 : n_workgroups);
    return wgs * kPcWords * 4 + wgs * 16 * 8;          // snapshots + (diagnostic builds) 16 time stamps per workgroup
}

void launch_pair_count_u8(hipStream_t s, const uint8_t *text, uint64_t n, const uint8_t *endmask,
                          uint32_t *bp, int n_workgroups, uint32_t *scratch, hipEvent_t start, hipEvent_t stop) {
    if (n < 2) return;
    uint64_t n_vec = n / 16;
    uint64_t max_wg = (n_vec + kPcThreads * kPcVpl - 1) / (kPcThreads * kPcVpl);
    if ((uint64_t)n_workgroups > max_wg) n_workgroups = (int)max_wg;
    if (n_workgroups < 1) n_workgroups = 1;
    if (endmask)
        hipExtLaunchKernelGGL(k_pair_count_u8<true>, dim3(n_workgroups), dim3(kPcThreads), 0, s, start, stop, 0, text, n, endmask, bp, scratch);
    else if (MBPE_PC_FAST)
        hipExtLaunchKernelGGL(k_pair_count_u8_fast, dim3(n_workgroups), dim3(kPcThreads), 0, s, start, stop, 0, text, n, bp, scratch);
    else
        hipExtLaunchKernelGGL(k_pair_count_u8<false>, dim3(n_workgroups), dim3(kPcThreads), 0, s, start, stop, 0, text, n, endmask, bp, scratch);
}

void launch_widen(hipStream_t s, const uint8_t *text, uint64_t n, const uint8_t *endmask, uint16_t *tok,
                  uint64_t n_slots_padded) {
    if (!n_slots_padded) return;
    int blocks = blocks_for(n_slots_padded / 16, 256, 8192);
    if (endmask)
        hipLaunchKernelGGL(k_widen<true>, dim3(blocks), dim3(256), 0, s, text, n, endmask, tok, n_slots_padded);
    else
        hipLaunchKernelGGL(k_widen<false>, dim3(blocks), dim3(256), 0, s, text, n, endmask, tok, n_slots_padded);
}

void launch_widen_barrier(hipStream_t s, const uint8_t *text, uint64_t n, const uint8_t *endmask, uint16_t *tok,
                          uint64_t n_slots) {
    if (!n_slots) return;
    hipLaunchKernelGGL(k_widen_barrier, dim3(blocks_for(n_slots / 16, 256, 8192)), dim3(256), 0, s, text, n, endmask, tok,
                       n_slots);
}

void launch_summarize(hipStream_t s, const uint16_t *tok, TileSum *sums, uint32_t n_tiles, int n_cus) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(k_summarize, dim3(tile_grid(n_tiles, n_cus)), dim3(kMergeThreads), 0, s, tok, sums, n_tiles);
}

void launch_table_init(hipStream_t s, const uint32_t *bp, PairTable t, DevCtl *ctl) {
    hipLaunchKernelGGL(k_table_init, dim3(65536 / 256), dim3(256), 0, s, bp, t, ctl);
}

void launch_table_rehash(hipStream_t s, PairTable t, DevCtl *ctl) {
    hipLaunchKernelGGL(k_table_rehash, dim3((t.ecap + 255) / 256), dim3(256), 0, s, t, ctl);
}

void launch_argmax(hipStream_t s, PairTable t, const DevCtl *ctl, unsigned long long *best, bool hierarchical) {
    if (hierarchical) {
        hipLaunchKernelGGL(k_argmax_hier, dim3(1), dim3(kHierThreads), 0, s, t, ctl, best);
        return;
    }
    int blocks = blocks_for(t.ecap, kArgmaxThreads * 4, 1024);
    hipLaunchKernelGGL(k_argmax, dim3(blocks), dim3(kArgmaxThreads), 0, s, t, ctl, best);
}

size_t first_state_bytes() { return sizeof(FirstState); }

void launch_first_init(hipStream_t s, void *fs) {
    (void)hipMemsetAsync(fs, 0, sizeof(FirstState), s);
    (void)hipMemsetAsync(fs, 0xFF, 8, s);          // pos_key = ~0
}

void launch_first_tiebreak(hipStream_t s, PairTable t, const DevCtl *ctl, unsigned long long *best, void *fs_,
                           const uint16_t *tok, const uint16_t *tok_other, const TileSum *sums, uint32_t n_tiles,
                           uint32_t endbit, int n_cus, int seq, int phase, const RankEdge *right_edge, uint32_t *xf, int rank,
                           int n_ranks) {
    FirstState *fs = static_cast<FirstState *>(fs_);
    if (phase != 2) {
        const uint32_t n_blocks = (t.ecap + kBlockSize - 1) >> kBlockShift;
        hipLaunchKernelGGL(k_first_gather, dim3(blocks_for(n_blocks, 4, 2048)), dim3(256), 0, s, t, ctl, best, fs, seq);
        if (n_tiles) {
            const dim3 grid(tile_grid(n_tiles, n_cus, 8)), block(kMergeThreads);
            MBPE_BY_MODE(endbit, hipLaunchKernelGGL(k_first_pos<M>, grid, block, 0, s, tok, tok_other, sums, n_tiles, t, best, fs,
                                                    ctl, seq, right_edge));
        }
    }
    if (phase == 0) hipLaunchKernelGGL(k_first_pick, dim3(1), dim3(256), 0, s, best, fs, ctl, seq);
    else if (phase == 1) hipLaunchKernelGGL(k_first_publish, dim3(1), dim3(64), 0, s, fs, xf, rank);
    else hipLaunchKernelGGL(k_first_pick_global, dim3(1), dim3(256), 0, s, best, fs, xf, n_ranks, exchange_header_words(n_ranks));
}

// runs of t before every tile, for the (t,t) pair of a single merge or the (t,t) member(s) of a batch
void launch_run_lengths(hipStream_t s, const TileSum *sin, uint32_t n_tiles, const unsigned long long *best, const DevCtl *ctl,
                        int seq, const BatchState *bs, unsigned long long *run_part, const RankEdge *left_edge, uint32_t *run_in) {
    if (!n_tiles) return;
    const uint32_t n_chunks = (n_tiles + kRunChunk - 1) / kRunChunk;
    hipLaunchKernelGGL(k_run_partial, dim3(n_chunks), dim3(kRunThreads), 0, s, sin, n_tiles, best, ctl, seq, bs, run_part);
    hipLaunchKernelGGL(k_run_final, dim3(n_chunks), dim3(kRunThreads), 0, s, sin, n_tiles, best, ctl, seq, bs, run_part,
                       left_edge, run_in);
}

void launch_merge(hipStream_t s, uint16_t *tok, uint16_t *tok_other, const TileSum *sin, TileSum *sout, uint32_t n_tiles,
                  uint32_t *chg, const unsigned long long *best, uint32_t new_id, uint32_t endbit, uint32_t *LR,
                  DevCtl *ctl, uint32_t *m_adj, const RankEdge *left_edge, const RankEdge *right_edge, int n_cus,
                  int seq, unsigned long long *run_part, uint32_t *run_in, const BatchState *bs, int hot_possible, int only) {
    if (!n_tiles) return;
    // runs of t before every tile, for a (t,t) pair (the kernels return at once for any other pair)
    if (only < 0 || (only & 1)) launch_run_lengths(s, sin, n_tiles, best, ctl, seq, bs, run_part, left_edge, run_in);
    static const int occ[3] = {resident_blocks(k_merge<0, false, 0>), resident_blocks(k_merge<1, false, 0>),
                               resident_blocks(k_merge<2, false, 0>)};
    const dim3 grid(tile_grid(n_tiles, n_cus, occ[slot_mode(endbit)])), block(kMergeThreads);
#ifdef MBPE_DIAG
    static const int diag = getenv("MBPE_MERGE_DIAG") ? atoi(getenv("MBPE_MERGE_DIAG")) : 0;
    if (diag == 1 && !endbit) {
        hipLaunchKernelGGL((k_merge<0, false, 1>), grid, block, 0, s, tok, tok_other, sin, sout, n_tiles, chg, best, new_id, LR, ctl,
                           m_adj, left_edge, right_edge, seq, run_in, hot_possible);
        return;
    }
    if (diag == 2 && !endbit) {
        hipLaunchKernelGGL((k_merge<0, false, 2>), grid, block, 0, s, tok, tok_other, sin, sout, n_tiles, chg, best, new_id, LR, ctl,
                           m_adj, left_edge, right_edge, seq, run_in, hot_possible);
        return;
    }
#endif
    // (both instantiations: each returns at once unless the pair's frequency is its case; only >= 0: exactly the one
    //  the host knows this merge takes -- bit 1: the frequent-pair instantiation)
    MBPE_BY_MODE(endbit, {
        if (only < 0 || !(only & 2))
            hipLaunchKernelGGL((k_merge<M, false, 0>), grid, block, 0, s, tok, tok_other, sin, sout, n_tiles, chg, best, new_id, LR, ctl,
                               m_adj, left_edge, right_edge, seq, run_in, hot_possible);
        if (only < 0 ? hot_possible != 0 : (only & 2) != 0)
            hipLaunchKernelGGL((k_merge<M, true, 0>), grid, block, 0, s, tok, tok_other, sin, sout, n_tiles, chg, best, new_id, LR,
                               ctl, m_adj, left_edge, right_edge, seq, run_in, hot_possible);
    });
}

void launch_apply(hipStream_t s, PairTable t, DevCtl *ctl, const unsigned long long *best, uint32_t new_id,
                  uint32_t *LR, const uint32_t *gm_gadj, TileSum *sums, const TileSum *side,
                  uint32_t *chg, uint32_t n_tiles, int seq) {
    // new_id: the id of the new token, or (seq != 0) an upper bound of it
    const uint32_t n_words = (n_tiles + 31u) / 32u;
    uint32_t blocks = (new_id + 255) / 256;
    const uint32_t want = (n_words + 255) / 256;
    if (want > blocks) blocks = want < 2048 ? want : 2048;
    hipLaunchKernelGGL(k_apply, dim3(blocks), dim3(256), 0, s, t, ctl, best, new_id, LR, gm_gadj, sums, side, chg,
                       n_words, seq);
}

void launch_patch_sums(hipStream_t s, const unsigned long long *best, TileSum *sums, const TileSum *side,
                       uint32_t *chg, uint32_t n_tiles, DevCtl *ctl, int seq) {
    const uint32_t n_words = (n_tiles + 31u) / 32u;
    uint32_t blocks = (n_words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_patch_sums, dim3(blocks), dim3(256), 0, s, best, sums, side, chg, n_words, ctl, seq, n_tiles);
}

void launch_select_batch(hipStream_t s, PairTable t, DevCtl *ctl, BatchState *bs, SelList *sel,
                         unsigned long long *best, uint32_t n_target, uint32_t max_batch, uint32_t fused_min,
                         int n_cus, int n_ranks, uint32_t endbit, uint32_t sel_cap, bool byte_table, int attempts,
                         int first_attempt, bool fallback) {
    const uint32_t fake_id = (endbit == kEndBit ? 0x7FFFu : 0xFFFFu) - 1u;      // see tt_rename
    // stand-in ids of (t,t) members are the kTTMax ids below the hole / end-bit mask: only while no token has them
    const uint32_t tt_max = 256u + n_target <= fake_id + 1u - (uint32_t)kTTMax ? (uint32_t)kTTMax : 1u;
    if (sel_cap < 64u) sel_cap = 64u;
    if (sel_cap > kSelCap) sel_cap = kSelCap;
    if (sel) {
        const int blocks = (n_cus > 0 ? n_cus : 256) * 4;
        // (attempts 1 and 2 only work after an overflowing or empty first gather; when they are needed and were not
        //  enqueued the bound-walking kernel below selects -- slower, never wrong)
        for (int attempt = first_attempt < 0 ? 0 : first_attempt; attempt < (attempts < 1 ? 1 : attempts > 3 ? 3 : attempts); ++attempt) {
            hipLaunchKernelGGL(k_sel_scan, dim3(blocks), dim3(256), 0, s, t, ctl, sel, n_target, sel_cap, attempt);
            hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(kPickThreads), 0, s, ctl, bs, sel, best, n_target, max_batch,
                               fused_min, (uint32_t)n_ranks, attempt, fake_id, sel_cap, tt_max, byte_table ? 1u : 0u);
        }
    }
    if (fallback)
        hipLaunchKernelGGL(k_select_batch, dim3(1), dim3(kHierThreads), 0, s, t, ctl, bs, best, n_target, max_batch,
                           fused_min, (uint32_t)n_ranks);
}

void launch_fused_batch(hipStream_t s, uint16_t *tok0, uint16_t *tok1, const TileSum *sums, TileSum *side,
                        uint32_t n_tiles, uint32_t *chg, const BatchState *bs, uint32_t *hdr_adj, uint32_t *LR,
                        DevCtl *ctl, const RankEdge *left_edge, const RankEdge *right_edge, uint32_t endbit,
                        int n_cus, uint32_t *hdr_m, const uint32_t *run_in, int hot_possible, int only, uint32_t *T) {
    if (!n_tiles) return;
    static const int occ[3] = {resident_blocks(k_fused_batch<0, false, false, 0>, kLutThreads),
                               resident_blocks(k_fused_batch<1, false, false, 0>, kLutThreads),
                               resident_blocks(k_fused_batch<2, false, false, 0>, kLutThreads)};
    const dim3 grid(tile_grid(n_tiles, n_cus, occ[slot_mode(endbit)], kLutThreads)), block(kLutThreads);
#ifdef MBPE_DIAG
    const int diag = getenv("MBPE_FUSED_DIAG") ? atoi(getenv("MBPE_FUSED_DIAG")) : 0;   // (re-read: set after warm-up)
#define MBPE_FUSED_DIAG_CASE(D)                                                                                            \
    if (diag == D && !endbit) {                                                                                            \
        hipLaunchKernelGGL((k_fused_batch<0, false, false, D>), grid, block, 0, s, tok0, tok1, sums, side, n_tiles,        \
                           chg, bs, hdr_adj, LR, ctl, left_edge, right_edge, hdr_m, run_in, hot_possible, T);                               \
        hipLaunchKernelGGL((k_fused_batch<0, false, true, D>), grid, block, 0, s, tok0, tok1, sums, side, n_tiles,         \
                           chg, bs, hdr_adj, LR, ctl, left_edge, right_edge, hdr_m, run_in, hot_possible, T);                               \
        return;                                                                                                            \
    }
    MBPE_FUSED_DIAG_CASE(2)
    MBPE_FUSED_DIAG_CASE(3)
    MBPE_FUSED_DIAG_CASE(4)
    MBPE_FUSED_DIAG_CASE(5)
    MBPE_FUSED_DIAG_CASE(6)
#undef MBPE_FUSED_DIAG_CASE
#endif
    MBPE_BY_MODE(endbit, {
        // (only >= 0: exactly the instantiation the host knows this batch takes -- bit 0: (t,t) member, bit 1: frequent pair)
        if (only < 0 || only == 0)
            hipLaunchKernelGGL((k_fused_batch<M, false, false>), grid, block, 0, s, tok0, tok1, sums, side, n_tiles, chg, bs, hdr_adj,
                               LR, ctl, left_edge, right_edge, hdr_m, run_in, hot_possible, T);
        if (only < 0 || only == 1)
            hipLaunchKernelGGL((k_fused_batch<M, false, true>), grid, block, 0, s, tok0, tok1, sums, side, n_tiles, chg, bs, hdr_adj,
                               LR, ctl, left_edge, right_edge, hdr_m, run_in, hot_possible, T);
        if (only < 0 ? hot_possible != 0 : only == 2)
            hipLaunchKernelGGL((k_fused_batch<M, true, false>), grid, block, 0, s, tok0, tok1, sums, side, n_tiles, chg, bs, hdr_adj,
                               LR, ctl, left_edge, right_edge, hdr_m, run_in, hot_possible, T);
        if (only < 0 ? hot_possible != 0 : only == 3)
            hipLaunchKernelGGL((k_fused_batch<M, true, true>), grid, block, 0, s, tok0, tok1, sums, side, n_tiles, chg, bs, hdr_adj,
                               LR, ctl, left_edge, right_edge, hdr_m, run_in, hot_possible, T);
    });
}

void launch_scan_batch(hipStream_t s, const uint16_t *tok, const uint16_t *tok1, const TileSum *sums, uint32_t n_tiles,
                       uint32_t *chg, const BatchState *bs, uint32_t *hdr_m, uint32_t *hdr_adj, uint32_t *LR,
                       const DevCtl *ctl, const RankEdge *left_edge, const RankEdge *right_edge, uint32_t endbit,
                       int n_cus, const uint32_t *run_in, int hot_possible, int only, uint32_t *T) {
    if (!n_tiles) return;
    static const int occ[3] = {resident_blocks(k_scan_batch<0, false, false, 0>, kLutThreads),
                               resident_blocks(k_scan_batch<1, false, false, 0>, kLutThreads),
                               resident_blocks(k_scan_batch<2, false, false, 0>, kLutThreads)};
    const dim3 grid(tile_grid(n_tiles, n_cus, occ[slot_mode(endbit)], kLutThreads)), block(kLutThreads);
#ifdef MBPE_DIAG
    static const int diag = getenv("MBPE_SCAN_DIAG") ? atoi(getenv("MBPE_SCAN_DIAG")) : 0;
#define MBPE_SCAN_DIAG_CASE(D)                                                                                             \
    if (diag == D && !endbit) {                                                                                            \
        hipLaunchKernelGGL((k_scan_batch<0, false, false, D>), grid, block, 0, s, tok, tok1, sums, n_tiles, chg, bs,       \
                           hdr_m, hdr_adj, LR, ctl, left_edge, right_edge, run_in, hot_possible, T);                                        \
        return;                                                                                                            \
    }
    MBPE_SCAN_DIAG_CASE(1)
    MBPE_SCAN_DIAG_CASE(2)
    MBPE_SCAN_DIAG_CASE(3)
    MBPE_SCAN_DIAG_CASE(4)
#undef MBPE_SCAN_DIAG_CASE
#endif
    MBPE_BY_MODE(endbit, {
        if (only < 0 || only == 0)
            hipLaunchKernelGGL((k_scan_batch<M, false, false, 0>), grid, block, 0, s, tok, tok1, sums, n_tiles, chg, bs, hdr_m, hdr_adj,
                               LR, ctl, left_edge, right_edge, run_in, hot_possible, T);
        if (only < 0 || only == 1)
            hipLaunchKernelGGL((k_scan_batch<M, false, true, 0>), grid, block, 0, s, tok, tok1, sums, n_tiles, chg, bs, hdr_m, hdr_adj,
                               LR, ctl, left_edge, right_edge, run_in, hot_possible, T);
        if (only < 0 ? hot_possible != 0 : only == 2)
            hipLaunchKernelGGL((k_scan_batch<M, true, false, 0>), grid, block, 0, s, tok, tok1, sums, n_tiles, chg, bs, hdr_m,
                               hdr_adj, LR, ctl, left_edge, right_edge, run_in, hot_possible, T);
        if (only < 0 ? hot_possible != 0 : only == 3)
            hipLaunchKernelGGL((k_scan_batch<M, true, true, 0>), grid, block, 0, s, tok, tok1, sums, n_tiles, chg, bs, hdr_m,
                               hdr_adj, LR, ctl, left_edge, right_edge, run_in, hot_possible, T);
    });
}

void launch_pair_cells_fold(hipStream_t s, uint32_t *T, uint32_t *LR, const DevCtl *ctl, uint32_t n_hint) {
    if (!T) return;
    if (n_hint < 64u) n_hint = 64u;
    if (n_hint > 1024u) n_hint = 1024u;
    hipLaunchKernelGGL(k_pair_cells_fold, dim3(n_hint), dim3(256), 0, s, T, LR, ctl);
}

void launch_batch_tables(hipStream_t s, PairTable t, DevCtl *ctl, BatchState *bs, uint32_t *hdr_m, uint32_t *hdr_adj,
                         uint32_t *LR, uint32_t id_upper, uint32_t n_hint) {
    // n_hint: the batch size the grids are sized for (every kernel strides over what the batch really holds)
    if (n_hint < 64u) n_hint = 64u;
    if (n_hint > (uint32_t)kBatchMax) n_hint = kBatchMax;
    // id_upper: upper bound of the ids that exist (the kernels read the exact value from ctl)
    const uint64_t cells = (uint64_t)id_upper * kBatchMax;
    uint32_t blocks = (uint32_t)((cells + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 8) blocks = 8;            // (k_apply_batch: at least kBatchMax threads, for the per-pair part)
    hipLaunchKernelGGL(k_adj_sums, dim3(n_hint), dim3(kWave), 0, s, hdr_adj, bs, ctl);
    // (work items of k_delta_max: 4096 cells of one LR row each)
    hipLaunchKernelGGL(k_delta_max, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, s, LR, bs, ctl);
    hipLaunchKernelGGL(k_adj_max, dim3(n_hint / 4), dim3(256), 0, s, hdr_adj, bs, ctl);
    hipLaunchKernelGGL(k_validate, dim3(1), dim3(kValThreads), 0, s, t, ctl, bs, hdr_m, hdr_adj, LR);
    if (t.cells) {
        uint32_t j_parts = (n_hint + kApplyTile - 1) / kApplyTile;       // workgroups side by side along the pairs
        if (j_parts > kApplyJParts) j_parts = kApplyJParts;
        uint32_t grid = ((id_upper + kApplyTile - 1) / kApplyTile) * j_parts;
        if (grid < 256u) grid = 256u;               // (the rows of the ADJ block and the per-pair part: a grid stride each)
        hipLaunchKernelGGL(k_apply_batch_dense, dim3(grid), dim3(256), 0, s, t, ctl, bs, hdr_m, hdr_adj, LR, j_parts);
    } else {
        hipLaunchKernelGGL(k_apply_batch, dim3(blocks), dim3(256), 0, s, t, ctl, bs, hdr_m, hdr_adj, LR);
    }
}

void launch_rewrite_marked(hipStream_t s, uint16_t *tok, uint16_t *tok1, const TileSum *sums, TileSum *side, uint32_t n_tiles,
                           uint32_t *chg, uint32_t *list, const BatchState *bs, DevCtl *ctl,
                           const RankEdge *left_edge, const RankEdge *right_edge, uint32_t endbit, int n_cus,
                           const uint32_t *run_in, int only) {
    if (!n_tiles) return;
    const uint32_t n_words = (n_tiles + 31u) / 32u;
    hipLaunchKernelGGL(k_list_marked, dim3((n_words + 255) / 256), dim3(256), 0, s, chg, n_words, list, ctl, n_tiles);
    static const int occ[3] = {resident_blocks(k_rewrite_marked<0, false>, kLutThreads),
                               resident_blocks(k_rewrite_marked<1, false>, kLutThreads),
                               resident_blocks(k_rewrite_marked<2, false>, kLutThreads)};
    const dim3 grid(tile_grid(n_tiles, n_cus, occ[slot_mode(endbit)], kLutThreads)), block(kLutThreads);
    MBPE_BY_MODE(endbit, {
        // (only >= 0, bit 0: the batch has a (t,t) member.  Validation may keep a prefix that ends before it: then the
        //  other instantiation is the one that works, so both are enqueued for such a batch)
        hipLaunchKernelGGL((k_rewrite_marked<M, false>), grid, block, 0, s, tok, tok1, sums, side, n_tiles, chg, list, bs, ctl,
                           left_edge, right_edge, run_in);
        if (only < 0 || (only & 1))
            hipLaunchKernelGGL((k_rewrite_marked<M, true>), grid, block, 0, s, tok, tok1, sums, side, n_tiles, chg, list, bs, ctl,
                               left_edge, right_edge, run_in);
    });
}

// What the selection of the sequence under way decided, for a host that enqueues only the kernels this sequence needs
// (small corpora, train.cpp: lockstep): merges done, pairs in the batch, fused pass or not, a (t,t) pair in it, and whether
// the pass is the frequent-pair instantiation's (dc_wanted, exactly as the stream kernels decide it).
__global__ void k_seq_info(const DevCtl *ctl, const BatchState *bs, const unsigned long long *best, uint32_t *out) {
    if (blockIdx.x || threadIdx.x) return;
    const uint32_t n = ctl->batch_n;
    unsigned long long top = 0;
    uint32_t tt = 0;
    if (n >= 2) {
        top = bs->packed[0] >> 32;
        tt = bs->tt_index != kNoTT ? 1u : 0u;
    } else if (n == 1) {
        const unsigned long long b = best[ctl->k_done];
        const uint32_t key = ~(uint32_t)b;
        top = b >> 32;
        tt = (top != 0 && (key >> 16) == (key & 0xFFFFu)) ? 1u : 0u;
    }
    out[0] = ctl->k_done;
    out[1] = n;
    out[2] = ctl->fused;
    out[3] = tt;
    out[4] = dc_wanted(top, ctl->n_live * ctl->n_ranks) ? 1u : 0u;
    out[5] = ctl->k_limit;
    out[6] = ctl->sel_ok;            // (the selection's first attempt chose the batch: the others are not needed)
    out[7] = 0;
}

void launch_seq_info(hipStream_t s, const DevCtl *ctl, const BatchState *bs, const unsigned long long *best, uint32_t *out) {
    hipLaunchKernelGGL(k_seq_info, dim3(1), dim3(64), 0, s, ctl, bs, best, out);
}

void launch_seq_finish(hipStream_t s, DevCtl *ctl, uint32_t *fused_flag, const BatchState *bs) {
    hipLaunchKernelGGL(k_seq_finish, dim3(1), dim3(64), 0, s, ctl, fused_flag, bs);
}

void launch_tile_scan(hipStream_t s, const TileSum *sums, uint32_t n_tiles, unsigned long long *offsets,
                      DevCtl *ctl) {
    // offsets has room for the chunk sums behind its n_tiles entries (tile_scan_scratch_words)
    if (!n_tiles) return;
    unsigned long long *part = offsets + n_tiles;
    const uint32_t n_parts = (n_tiles + kScanChunk - 1) / kScanChunk;
    hipLaunchKernelGGL(k_tile_scan_partial, dim3(n_parts), dim3(kScanThreads), 0, s, sums, n_tiles, part);
    hipLaunchKernelGGL(k_tile_scan_top, dim3(1), dim3(1024), 0, s, part, n_parts, ctl);
    hipLaunchKernelGGL(k_tile_scan_final, dim3(n_parts), dim3(kScanThreads), 0, s, sums, n_tiles, part, offsets);
}

void launch_compact_scatter(hipStream_t s, const uint16_t *src, const TileSum *sums,
                            const unsigned long long *offsets, uint32_t n_tiles, uint16_t *dst, int n_cus) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(k_compact_scatter, dim3(tile_grid(n_tiles, n_cus)), dim3(kMergeThreads), 0, s, src, sums,
                       offsets, n_tiles, dst);
}

void launch_rank_edge(hipStream_t s, const TileSum *sums, uint32_t n_tiles, RankEdge *out, const DevCtl *ctl,
                      uint32_t *hdr) {
    hipLaunchKernelGGL(k_rank_edge, dim3(1), dim3(64), 0, s, sums, n_tiles, out, ctl, hdr);
}

void launch_compose_edges(hipStream_t s, uint32_t *hdr, int rank, int n_ranks, RankEdge *left, RankEdge *right) {
    hipLaunchKernelGGL(k_compose_edges, dim3(1), dim3(64), 0, s, hdr, rank, n_ranks, left, right);
}

void launch_boundary_pairs(hipStream_t s, uint32_t *bp, const uint32_t *hdr, int n_ranks, uint32_t endbit) {
    hipLaunchKernelGGL(k_boundary_pairs, dim3(1), dim3(64), 0, s, bp, hdr, n_ranks, endbit);
}

}  // namespace mbpe
