// HIP kernels of the BPE training hot path for gfx950 (MI355X, wave64).
//
// Reference path replaced (code/include/ of justinhj/minbpe-cc):
//   calculate_freqs            Tokenizer.h:127-146   -> k_pair_count_u8 (+ k_table_init)
//   get_top_pair_count         PairCount.h:262-269   -> k_argmax
//   merge_chunks / merge_incremental
//                              Tokenizer.h:309-320, :202-306 -> k_merge + k_apply
//   create_lists / text_to_vector
//                              Tokenizer.h:114-124, :85-100 -> k_widen
//
// Integer / index work, HBM-bound: no MFMA anywhere.  Everything is exact
// (integer atomics commute), so repeated runs are bit-identical.
#include "mbpe_dev.h"

namespace mbpe {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

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

// ---- pair table device ops ------------------------------------------------

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
__device__ void table_add(const PairTable &t, DevCtl *ctl, uint32_t key, int32_t delta, bool may_insert) {
    uint32_t h = hash_key(key) & t.hmask;
    for (uint32_t probe = 0; probe <= t.hmask; ++probe) {
        uint32_t k = t.hkey[h];
        if (k == key) {
            int32_t old = atomicAdd(&t.ecnt[t.hidx[h]], delta);
            if (old + delta < 0) atomicOr(&ctl->err, kErrNegCount);
            return;
        }
        if (k == kEmptyKey) {
            if (!may_insert) { atomicOr(&ctl->err, kErrMissingPair); return; }
            uint32_t prev = atomicCAS(&t.hkey[h], kEmptyKey, key);
            if (prev == kEmptyKey) {
                uint32_t idx = atomicAdd(&ctl->n_entries, 1u);
                if (idx >= t.ecap) { atomicOr(&ctl->err, kErrTableFull); return; }
                t.hidx[h] = idx;
                t.ekey[idx] = key;
                t.ecnt[idx] = delta;
                return;
            }
            if (prev == key) {  // cannot happen by construction; keep the table sane anyway
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
// reads 16 corpus bytes with one 16-byte load and issues 16 LDS atomics.
//
// Exactness of the 16-bit counters: an epoch is two iterations of the
// 1024-thread workgroup = 32,768 increments.  A lane that sees a counter at
// or above 0x4000 in the value returned by its atomic raises a flag; at the
// epoch boundary a raised flag makes the workgroup drain every counter
// >= 0x4000 to the global table.  So every counter is <= 0x4000 when an
// epoch starts and gains at most 0x8000 inside it: it never wraps.
constexpr int kPcThreads = 1024;
constexpr int kPcWords = 32768;            // 2 counters per word
constexpr uint32_t kPcFlagAt = 0x4000u;

__device__ __forceinline__ uint32_t pc_table_index(uint32_t le_bin) {
    // LDS bins are indexed by the little-endian 16-bit load (first | second<<8);
    // the output table by (first << 8) | second.
    return ((le_bin & 0xFFu) << 8) | (le_bin >> 8);
}

__device__ __forceinline__ void pc_drain(uint32_t *hist, uint32_t *bp, uint32_t threshold) {
    for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords; w += kPcThreads) {
        uint32_t v = hist[w];
        uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        uint32_t nv = v;
        if (lo >= threshold && lo) { atomicAdd(&bp[pc_table_index(w)], lo); nv &= 0xFFFF0000u; }
        if (hi >= threshold && hi) { atomicAdd(&bp[pc_table_index(w + kPcWords)], hi); nv &= 0x0000FFFFu; }
        if (nv != v) hist[w] = nv;
    }
}

template <bool MASKED>
__global__ __launch_bounds__(kPcThreads) void k_pair_count_u8(const uint8_t *__restrict__ text, uint64_t n,
                                                              const uint8_t *__restrict__ endmask,
                                                              uint32_t *__restrict__ bp) {
    __shared__ uint32_t hist[kPcWords + 4];
    uint32_t *flag = &hist[kPcWords];
    for (uint32_t w = threadIdx.x; w < (uint32_t)kPcWords + 4; w += kPcThreads) hist[w] = 0;
    __syncthreads();

    const uint64_t n_vec = (n + 15) / 16;
    uint64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
    per = (per + kPcThreads - 1) / kPcThreads * kPcThreads;
    const uint64_t v_begin = per * blockIdx.x;
    uint64_t v_end = v_begin + per;
    if (v_end > n_vec) v_end = n_vec;
    const uint32_t lane = lane_id();

    int epoch_iter = 0;
    for (uint64_t base = v_begin; base < v_end; base += kPcThreads) {
        const uint64_t vec = base + threadIdx.x;
        const uint64_t byte0 = vec * 16;
        uint32_t w[4] = {0, 0, 0, 0};
        if (vec < v_end) {
            if (byte0 + 16 <= n) {
                uint4 q = *reinterpret_cast<const uint4 *>(text + byte0);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            } else {
                for (uint64_t i = byte0; i < n; ++i)
                    w[(i - byte0) >> 2] |= (uint32_t)text[i] << (8 * ((i - byte0) & 3));
            }
        }
        // first byte of the next lane's vector = second byte of my last pair
        uint32_t nb = __shfl_down(w[0], 1, kWave) & 0xFFu;
        if (lane == kWave - 1 && byte0 + 16 < n && vec < v_end) nb = text[byte0 + 16];
        uint32_t ends = 0;
        if (MASKED && vec < v_end && byte0 < n) ends = reinterpret_cast<const uint16_t *>(endmask)[vec];
        // number of pairs that start in this vector
        uint32_t n_pairs = 0;
        if (vec < v_end && byte0 + 1 < n) {
            uint64_t rem = n - 1 - byte0;   // pairs starting at byte0 .. n-2
            n_pairs = rem < 16 ? (uint32_t)rem : 16u;
        }
        uint32_t olds[16];
        uint32_t bins[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t cur = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            uint32_t nxt = (i < 15) ? ((w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xFFu) : nb;
            uint32_t bin = cur | (nxt << 8);
            bool valid = (uint32_t)i < n_pairs && !(MASKED && ((ends >> i) & 1u));
            bins[i] = valid ? bin : 0xFFFFFFFFu;
            if (valid) {
                uint32_t inc = (bin & 0x8000u) ? 0x10000u : 1u;
                olds[i] = atomicAdd(&hist[bin & 0x7FFFu], inc);
            } else {
                olds[i] = 0;
            }
        }
        bool hot = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (bins[i] != 0xFFFFFFFFu) {
                uint32_t half = (bins[i] & 0x8000u) ? (olds[i] >> 16) : (olds[i] & 0xFFFFu);
                hot |= half >= kPcFlagAt;
            }
        }
        if (hot) *flag = 1;
        if (++epoch_iter == 2) {
            epoch_iter = 0;
            __syncthreads();
            if (*flag) {            // uniform: read after the barrier
                __syncthreads();
                pc_drain(hist, bp, kPcFlagAt);
                if (threadIdx.x == 0) *flag = 0;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    pc_drain(hist, bp, 1);
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

// ---- tile helpers -----------------------------------------------------------

// Loads the tile's slots, squeezes the live ones into dense[2 .. 2+n_live)
// (LDS) and returns this thread's first dense index and live count.
struct TileLoad {
    uint32_t s[kSlotsPerThread];
    uint32_t first;    // dense index of this thread's first live slot
    uint32_t count;    // live slots of this thread
    uint32_t n_live;   // live slots of the tile
};

__device__ __forceinline__ TileLoad tile_load_dense(const uint16_t *tok, uint32_t tile, uint16_t *dense,
                                                    uint32_t *wsum) {
    TileLoad r;
    const uint4 q = reinterpret_cast<const uint4 *>(tok)[(uint64_t)tile * kMergeThreads + threadIdx.x];
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < kSlotsPerThread; ++j) {
        r.s[j] = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        cnt += r.s[j] != kHole;
    }
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t wid = threadIdx.x / kWave;
    if (lane_id() == kWave - 1) wsum[wid] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kMergeThreads / kWave; ++k) {
        uint32_t v = wsum[k];
        if ((uint32_t)k < wid) base += v;
        total += v;
    }
    r.first = base + incl - cnt;
    r.count = cnt;
    r.n_live = total;
    uint32_t o = r.first + 2;
#pragma unroll
    for (int j = 0; j < kSlotsPerThread; ++j)
        if (r.s[j] != kHole) dense[o++] = (uint16_t)r.s[j];
    return r;
}

// Summary of the live tokens in d[0..n) where kHole entries are skipped.
__device__ TileSum summarize_lds(const uint16_t *d, int n, uint32_t n_live) {
    TileSum s;
    s.head0 = s.head1 = s.tail0 = s.tail1 = (uint16_t)kHole;
    s.n_live = (uint16_t)n_live;
    s.tail_run = 0;
    s.pad0 = s.pad1 = 0;
    int found = 0;
    for (int i = 0; i < n && found < 2; ++i) {
        uint16_t v = d[i];
        if (v == kHole) continue;
        if (found == 0) s.head0 = v; else s.head1 = v;
        ++found;
    }
    found = 0;
    uint32_t run = 0;
    bool counting = true;
    for (int i = n - 1; i >= 0 && (found < 2 || counting); --i) {
        uint16_t v = d[i];
        if (v == kHole) continue;
        if (found == 0) { s.tail0 = v; run = 1; }
        else {
            if (found == 1) s.tail1 = v;
            if (counting) { if (v == s.tail0) ++run; else counting = false; }
        }
        ++found;
    }
    s.tail_run = (uint16_t)run;
    return s;
}

__global__ __launch_bounds__(kMergeThreads) void k_summarize(const uint16_t *__restrict__ tok,
                                                             TileSum *__restrict__ sums, uint32_t n_tiles,
                                                             DevCtl *ctl, int set_n_live) {
    __shared__ uint16_t dense[kTile + 4];
    __shared__ uint32_t wsum[kMergeThreads / kWave];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    TileLoad tl = tile_load_dense(tok, tile, dense, wsum);
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[tile] = summarize_lds(dense + 2, (int)tl.n_live, tl.n_live);
        if (set_n_live) atomicAdd(&ctl->n_live, (unsigned long long)tl.n_live);
    }
}

// ---- argmax ------------------------------------------------------------------

constexpr int kArgmaxThreads = 256;

__global__ __launch_bounds__(kArgmaxThreads) void k_argmax(PairTable t, const DevCtl *ctl,
                                                           unsigned long long *best) {
    __shared__ unsigned long long wbest[kArgmaxThreads / kWave];
    const uint32_t n = ctl->n_entries < t.ecap ? ctl->n_entries : t.ecap;
    unsigned long long b = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int32_t c = t.ecnt[i];
        unsigned long long p = pack_best(c < 0 ? 0 : c, t.ekey[i]);
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

// ---- table init / rehash -------------------------------------------------------

__global__ void k_table_init(const uint32_t *__restrict__ bp, PairTable t, DevCtl *ctl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    uint32_t c = bp[i];
    if (!c) return;
    uint32_t key = ((i >> 8) << 16) | (i & 0xFFu);
    table_add(t, ctl, key, (int32_t)c, true);
}

__global__ void k_table_rehash(PairTable t, DevCtl *ctl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ctl->n_entries) return;
    uint32_t key = t.ekey[i];
    uint32_t h = hash_key(key) & t.hmask;
    for (;;) {
        uint32_t prev = atomicCAS(&t.hkey[h], kEmptyKey, key);
        if (prev == kEmptyKey) { t.hidx[h] = i; return; }
        h = (h + 1) & t.hmask;
    }
}

// ---- merge pass ------------------------------------------------------------------
// One workgroup per tile, in place.  For the chosen pair (a,b) -> X
// (merge_incremental, Tokenizer.h:202-306):
//   - the slot holding `a` of a match becomes X (keeping b's chunk-end flag),
//     the slot holding `b` becomes a hole                          (:236-237)
//   - count deltas (:239-280) are accumulated as
//       L[x] += 1  for a match whose left neighbour is x   => (x,a)-1, (x,X)+1
//       R[y] += 1  for a match whose right neighbour is y  => (b,y)-1, (X,y)+1
//       adj  += 1  for two matches that touch ("abab")     => (b,a)-1, (X,X)+1
//       m    += 1  per match                               => (a,b)-1
//     which is the reference's sequential result: its transient (X,a)+1/-1
//     on touching matches cancels (SURVEY.md 8-S rule 3).
// Neighbour tokens across the tile edge come from the tile summaries of the
// previous pass, never from the neighbour's slots.

struct Halo { uint32_t p2, p1, n1, n2, run_before; };

__device__ Halo tile_halo(const TileSum *sin, uint32_t n_tiles, uint32_t tile, uint32_t a, bool same,
                          const RankEdge *le, const RankEdge *re) {
    Halo h;
    h.p1 = h.p2 = h.n1 = h.n2 = kHole;
    h.run_before = 0;
    int need = 2;
    for (int64_t j = (int64_t)tile - 1; need && j >= -1; --j) {
        uint32_t nl, t0, t1;
        if (j >= 0) { TileSum s = sin[j]; nl = s.n_live; t0 = s.tail0; t1 = s.tail1; }
        else if (le) { nl = (le->n_live_lo | le->n_live_hi) ? 2 : 0; t0 = le->tail0; t1 = le->tail1; if (t0 == kHole) nl = 0; else if (t1 == kHole) nl = 1; }
        else break;
        if (nl == 0) continue;
        if (need == 2) { h.p1 = t0; need = 1; if (nl >= 2) { h.p2 = t1; need = 0; } }
        else { h.p2 = t0; need = 0; }
    }
    need = 2;
    for (int64_t j = (int64_t)tile + 1; need && j <= (int64_t)n_tiles; ++j) {
        uint32_t nl, t0, t1;
        if (j < (int64_t)n_tiles) { TileSum s = sin[j]; nl = s.n_live; t0 = s.head0; t1 = s.head1; }
        else if (re) { t0 = re->head0; t1 = re->head1; nl = t0 == kHole ? 0 : (t1 == kHole ? 1 : 2); }
        else break;
        if (nl == 0) continue;
        if (need == 2) { h.n1 = t0; need = 1; if (nl >= 2) { h.n2 = t1; need = 0; } }
        else { h.n2 = t0; need = 0; }
    }
    if (same) {
        // tokens equal to raw `a` immediately before this tile
        uint64_t rb = 0;
        int64_t j = (int64_t)tile - 1;
        for (; j >= 0; --j) {
            TileSum s = sin[j];
            if (s.n_live == 0) continue;
            if (s.tail0 != a) break;
            rb += s.tail_run;
            if (s.tail_run != s.n_live) break;
        }
        if (j < 0 && le && le->tail0 == a)
            rb += ((uint64_t)le->tail_run_hi << 32) | le->tail_run_lo;
        h.run_before = (uint32_t)(rb & 1u) | ((rb >= 2) ? 2u : 0u);   // parity + ">=2" is all that is used
    }
    return h;
}

__global__ __launch_bounds__(kMergeThreads) void k_merge(uint16_t *__restrict__ tok,
                                                         const TileSum *__restrict__ sin,
                                                         TileSum *__restrict__ sout, uint32_t n_tiles,
                                                         const unsigned long long *__restrict__ best_ptr,
                                                         uint32_t X, uint32_t endbit, uint32_t *L,
                                                         uint32_t *R, DevCtl *ctl, const RankEdge *le,
                                                         const RankEdge *re) {
    __shared__ uint16_t dense[kTile + 4];
    __shared__ uint16_t newd[kTile];
    __shared__ uint32_t wsum[kMergeThreads / kWave];
    __shared__ uint32_t sh[8];   // 0: removed, 1: m, 2: adj, 3: changed, 4: run_before
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const TileSum me = sin[tile];
    const unsigned long long best = *best_ptr;
    if ((best >> 32) == 0 || me.n_live < 1) {   // nothing can match (count 0) or empty tile
        if (threadIdx.x == 0) sout[tile] = me;
        return;
    }
    const uint32_t key = ~(uint32_t)best;
    const uint32_t a = key >> 16, b = key & 0xFFFFu;
    const bool same = a == b;
    const uint32_t idmask = endbit ? 0x7FFFu : 0xFFFFu;

    if (threadIdx.x < 8) sh[threadIdx.x] = 0;
    TileLoad tl = tile_load_dense(tok, tile, dense, wsum);   // contains a barrier (after sh init)
    if (threadIdx.x == 0) {
        Halo h = tile_halo(sin, n_tiles, tile, a, same, le, re);
        dense[0] = (uint16_t)h.p2;
        dense[1] = (uint16_t)h.p1;
        dense[2 + tl.n_live] = (uint16_t)h.n1;
        dense[3 + tl.n_live] = (uint16_t)h.n2;
        sh[4] = h.run_before;
    }
    __syncthreads();

    uint32_t my_m = 0, my_adj = 0, my_removed = 0;
    bool changed = false;
    {
        uint32_t i = tl.first;          // index among the tile's live tokens
        uint32_t run = 0;               // consecutive raw-a tokens right before i (same only)
        bool run_valid = false;
        uint32_t run_ge2 = 0;
#pragma unroll
        for (int j = 0; j < kSlotsPerThread; ++j) {
            const uint32_t self = tl.s[j];
            if (self == kHole) continue;
            const uint32_t p2 = dense[i], p1 = dense[i + 1], n1 = dense[i + 3], n2 = dense[i + 4];
            bool amatch, bmatch;
            bool prev_adjacent;     // the two tokens before `self` formed a match
            if (!same) {
                amatch = (self == a) && ((n1 & idmask) == b);
                bmatch = ((self & idmask) == b) && (p1 == a);
                prev_adjacent = (p1 == b) && (p2 == a);
            } else {
                amatch = bmatch = prev_adjacent = false;
                if ((self & idmask) == a) {
                    if (!run_valid) {
                        // walk back over the live tokens of this tile, then into the summaries
                        run = 0;
                        int64_t k = (int64_t)i - 1;
                        while (k >= 0 && dense[2 + k] == a) { ++run; --k; }
                        run_ge2 = run >= 2;
                        if (k < 0) { uint32_t rb = sh[4]; run += rb & 1u; run_ge2 |= (rb >> 1) | (run >= 2); }
                        run_valid = true;
                    }
                    const bool odd = run & 1u;
                    amatch = (self == a) && !odd && ((n1 & idmask) == a);
                    bmatch = odd;
                    prev_adjacent = !odd && run_ge2;
                }
            }
            // maintain the run for the next live token of this thread
            if (same) {
                if (self == a) { if (run_valid) { ++run; run_ge2 = run >= 2 || run_ge2; } }
                else { run = 0; run_ge2 = 0; run_valid = true; }
            }
            uint32_t nv = self;
            if (amatch) {
                nv = X | (n1 & endbit);
                ++my_m;
                if (p1 != kHole && !(p1 & endbit)) {
                    if (prev_adjacent) ++my_adj;
                    else atomicAdd(&L[p1], 1u);
                }
            } else if (bmatch) {
                nv = kHole;
                ++my_removed;
                if (!(self & endbit) && n1 != kHole) {
                    const bool next_adjacent = (n1 == a) && ((n2 & idmask) == b);
                    if (!next_adjacent) atomicAdd(&R[n1 & idmask], 1u);
                }
            }
            if (nv != self) { changed = true; tl.s[j] = nv; }
            newd[i] = (uint16_t)nv;
            ++i;
        }
    }
    if (changed) {
        uint4 q;
        q.x = tl.s[0] | (tl.s[1] << 16);
        q.y = tl.s[2] | (tl.s[3] << 16);
        q.z = tl.s[4] | (tl.s[5] << 16);
        q.w = tl.s[6] | (tl.s[7] << 16);
        reinterpret_cast<uint4 *>(tok)[(uint64_t)tile * kMergeThreads + threadIdx.x] = q;
    }
    const uint32_t wm = wave_sum(my_m), wa = wave_sum(my_adj), wr = wave_sum(my_removed);
    if (lane_id() == 0 && (wm | wa | wr)) {
        atomicAdd(&sh[0], wr);
        atomicAdd(&sh[1], wm);
        atomicAdd(&sh[2], wa);
    }
    const int any_changed = __syncthreads_or(changed);
    if (threadIdx.x == 0) {
        if (!any_changed) {
            sout[tile] = me;
        } else {
            sout[tile] = summarize_lds(newd, (int)tl.n_live, tl.n_live - sh[0]);
            if (sh[1]) atomicAdd(&ctl->m, sh[1]);
            if (sh[2]) atomicAdd(&ctl->adj, sh[2]);
        }
    }
}

// ---- apply: fold (L, R, m, adj) into the pair table -------------------------------
// gm_gadj, when not NULL, holds the all-reduced {m, adj} of a multi-GPU run
// (L and R are then already all-reduced in place); otherwise ctl->m / ctl->adj.
// ctl->m stays the LOCAL match count either way (it feeds n_live / holes).

__global__ void k_apply(PairTable t, DevCtl *ctl, const unsigned long long *__restrict__ best_ptr,
                        uint32_t X, uint32_t *L, uint32_t *R, const uint32_t *gm_gadj) {
    const unsigned long long best = *best_ptr;
    if ((best >> 32) == 0) return;   // count 0: the merge changed nothing (also covers "no pair")
    const uint32_t key = ~(uint32_t)best;
    const uint32_t a = key >> 16, b = key & 0xFFFFu;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < X) {
        const uint32_t l = L[x];
        if (l) {
            table_add(t, ctl, (x << 16) | a, -(int32_t)l, false);
            table_add(t, ctl, (x << 16) | X, (int32_t)l, true);
            L[x] = 0;
        }
        const uint32_t r = R[x];
        if (r) {
            table_add(t, ctl, (b << 16) | x, -(int32_t)r, false);
            table_add(t, ctl, (X << 16) | x, (int32_t)r, true);
            R[x] = 0;
        }
    }
    if (x == 0) {
        const uint32_t m_local = ctl->m;
        const uint32_t m = gm_gadj ? gm_gadj[0] : m_local;
        const uint32_t adj = gm_gadj ? gm_gadj[1] : ctl->adj;
        if (m) table_add(t, ctl, (a << 16) | b, -(int32_t)m, false);
        if (adj) {
            table_add(t, ctl, (b << 16) | a, -(int32_t)adj, false);
            table_add(t, ctl, (X << 16) | X, (int32_t)adj, true);
        }
        ctl->removed_total += m_local;
        ctl->n_live -= m_local;
        ctl->m = 0;
        ctl->adj = 0;
    }
}

// ---- compaction ---------------------------------------------------------------------

constexpr int kScanThreads = 1024;

__global__ __launch_bounds__(kScanThreads) void k_tile_scan(const TileSum *__restrict__ sums, uint32_t n_tiles,
                                                            unsigned long long *__restrict__ offsets,
                                                            DevCtl *ctl) {
    __shared__ unsigned long long part[kScanThreads];
    const uint32_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < n_tiles ? lo + per : n_tiles;
    unsigned long long s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += sums[i].n_live;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < kScanThreads; ++i) { unsigned long long v = part[i]; part[i] = run; run += v; }
        ctl->scan_total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { offsets[i] = run; run += sums[i].n_live; }
}

__global__ __launch_bounds__(kMergeThreads) void k_compact_scatter(const uint16_t *__restrict__ src,
                                                                   const TileSum *__restrict__ sums,
                                                                   const unsigned long long *__restrict__ offsets,
                                                                   uint32_t n_tiles, uint16_t *__restrict__ dst) {
    __shared__ uint16_t dense[kTile + 4];
    __shared__ uint32_t wsum[kMergeThreads / kWave];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if (sums[tile].n_live == 0) return;
    TileLoad tl = tile_load_dense(src, tile, dense, wsum);
    __syncthreads();
    const unsigned long long off = offsets[tile];
    for (uint32_t i = threadIdx.x; i < tl.n_live; i += kMergeThreads) dst[off + i] = dense[2 + i];
}

// ---- rank edge (multi-GPU) ---------------------------------------------------------------

__global__ void k_rank_edge(const TileSum *__restrict__ sums, uint32_t n_tiles, RankEdge *out) {
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
    unsigned long long run = 0, live = 0;
    bool counting = true;
    for (int64_t j = (int64_t)n_tiles - 1; j >= 0; --j) {
        TileSum s = sums[j];
        live += s.n_live;
        if (!counting || !s.n_live) continue;
        if (s.tail0 != e.tail0) { counting = false; continue; }
        run += s.tail_run;
        if (s.tail_run != s.n_live) counting = false;
    }
    e.n_live_lo = (uint32_t)live; e.n_live_hi = (uint32_t)(live >> 32);
    e.tail_run_lo = (uint32_t)run; e.tail_run_hi = (uint32_t)(run >> 32);
    *out = e;
}

inline int blocks_for(uint64_t n, int threads, int max_blocks) {
    uint64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------

void launch_fill_u32(hipStream_t s, uint32_t *p, uint64_t n, uint32_t v) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, s, p, n, v);
}

void launch_fill_u16(hipStream_t s, uint16_t *p, uint64_t n, uint16_t v) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u16, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, s, p, n, v);
}

void launch_pair_count_u8(hipStream_t s, const uint8_t *text, uint64_t n, const uint8_t *endmask,
                          uint32_t *bp, int n_workgroups) {
    if (n < 2) return;
    uint64_t n_vec = (n + 15) / 16;
    uint64_t max_wg = (n_vec + kPcThreads - 1) / kPcThreads;
    if ((uint64_t)n_workgroups > max_wg) n_workgroups = (int)max_wg;
    if (n_workgroups < 1) n_workgroups = 1;
    if (endmask)
        hipLaunchKernelGGL(k_pair_count_u8<true>, dim3(n_workgroups), dim3(kPcThreads), 0, s, text, n, endmask, bp);
    else
        hipLaunchKernelGGL(k_pair_count_u8<false>, dim3(n_workgroups), dim3(kPcThreads), 0, s, text, n, endmask, bp);
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

void launch_summarize(hipStream_t s, const uint16_t *tok, TileSum *sums, uint32_t n_tiles, DevCtl *ctl,
                      int set_n_live) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(k_summarize, dim3(n_tiles), dim3(kMergeThreads), 0, s, tok, sums, n_tiles, ctl, set_n_live);
}

void launch_table_init(hipStream_t s, const uint32_t *bp, PairTable t, DevCtl *ctl) {
    hipLaunchKernelGGL(k_table_init, dim3(65536 / 256), dim3(256), 0, s, bp, t, ctl);
}

void launch_table_rehash(hipStream_t s, PairTable t, DevCtl *ctl) {
    hipLaunchKernelGGL(k_table_rehash, dim3((t.ecap + 255) / 256), dim3(256), 0, s, t, ctl);
}

void launch_argmax(hipStream_t s, PairTable t, const DevCtl *ctl, unsigned long long *best) {
    int blocks = blocks_for(t.ecap, kArgmaxThreads * 4, 1024);
    hipLaunchKernelGGL(k_argmax, dim3(blocks), dim3(kArgmaxThreads), 0, s, t, ctl, best);
}

void launch_merge(hipStream_t s, uint16_t *tok, const TileSum *sin, TileSum *sout, uint32_t n_tiles,
                  const unsigned long long *best, uint32_t new_id, uint32_t endbit, uint32_t *L, uint32_t *R,
                  DevCtl *ctl, const RankEdge *left_edge, const RankEdge *right_edge) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(k_merge, dim3(n_tiles), dim3(kMergeThreads), 0, s, tok, sin, sout, n_tiles, best, new_id,
                       endbit, L, R, ctl, left_edge, right_edge);
}

void launch_apply(hipStream_t s, PairTable t, DevCtl *ctl, const unsigned long long *best, uint32_t new_id,
                  uint32_t *L, uint32_t *R, const uint32_t *gm_gadj) {
    hipLaunchKernelGGL(k_apply, dim3((new_id + 255) / 256), dim3(256), 0, s, t, ctl, best, new_id, L, R, gm_gadj);
}

void launch_tile_scan(hipStream_t s, const TileSum *sums, uint32_t n_tiles, unsigned long long *offsets,
                      DevCtl *ctl) {
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(kScanThreads), 0, s, sums, n_tiles, offsets, ctl);
}

void launch_compact_scatter(hipStream_t s, const uint16_t *src, const TileSum *sums,
                            const unsigned long long *offsets, uint32_t n_tiles, uint16_t *dst) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(k_compact_scatter, dim3(n_tiles), dim3(kMergeThreads), 0, s, src, sums, offsets, n_tiles, dst);
}

void launch_rank_edge(hipStream_t s, const TileSum *sums, uint32_t n_tiles, RankEdge *out) {
    hipLaunchKernelGGL(k_rank_edge, dim3(1), dim3(64), 0, s, sums, n_tiles, out);
}

}  // namespace mbpe
