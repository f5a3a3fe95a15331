// Host orchestration of the BPE training hot path behind the C-ABI
// (include/mbpe.h).  Drives the kernels of kernels.hip on one HIP stream:
//
//   begin : pair-count scan -> pair table, widen -> slot stream, first argmax
//   step k: merge(best[k]) -> apply deltas -> argmax -> best[k+1]
//
// which is the loop of Tokenizer::train (reference Tokenizer.h:551-589) for
// CONFLICT_RESOLUTION::LEXICAL.  The chosen pair never leaves the device
// between steps (the merge kernel reads best[k] from HBM); the host only
// synchronises once per batch of merges for housekeeping: error flags,
// compaction of holes, growth of the pair table.
#include "mbpe.h"
#include "mbpe_dev.h"
#include "wide.h"
#include "../host/mbpe_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

using namespace mbpe;

namespace {

std::string hip_err(const char *what, hipError_t e) {
    return std::string(what) + ": " + hipGetErrorString(e);
}

#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t e__ = (expr);                                       \
        if (e__ != hipSuccess) {                                       \
            mbpe_host::set_last_error(hip_err(#expr, e__));            \
            return e__ == hipErrorOutOfMemory ? MBPE_ERR_OOM : MBPE_ERR_HIP; \
        }                                                              \
    } while (0)

uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

uint32_t next_pow2(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return (uint32_t)p;
}

// std::stoi on the remainder of a NUL-led chunk (reference Tokenizer.h:86-93):
// true when it parses, i.e. the chunk collapses to a single token.
bool stoi_parses(const uint8_t *s, uint64_t n) {
    uint64_t i = 0;
    while (i < n && (s[i] == ' ' || (s[i] >= 9 && s[i] <= 13))) i++;
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    if (i >= n || s[i] < '0' || s[i] > '9') return false;
    long long v = 0;
    while (i < n && s[i] >= '0' && s[i] <= '9') {
        v = v * 10 + (s[i] - '0');
        if (v > 4294967296LL) return false;
        i++;
    }
    if (neg) v = -v;
    return v <= 2147483647LL && v >= -2147483648LL;
}

template <typename T>
void dfree(T *&p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

}  // namespace

struct mbpe_ctx;
namespace {
template <typename T> hipError_t tmalloc(mbpe_ctx *c, T **p, size_t bytes);
template <typename T> void tfree(mbpe_ctx *c, T *&p);
void pool_trim(mbpe_ctx *c);
}

// ---- RCCL, bound at run time: the library loads and runs on one GPU without it ----
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.why = "librccl.so.1 not found"; return r; }
#define MBPE_BIND(field, name) \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name)); \
    if (!r.field) { r.why = std::string("missing RCCL symbol ") + name; return r; }
    MBPE_BIND(GetUniqueId, "ncclGetUniqueId")
    MBPE_BIND(CommInitRank, "ncclCommInitRank")
    MBPE_BIND(CommDestroy, "ncclCommDestroy")
    MBPE_BIND(AllReduce, "ncclAllReduce")
    MBPE_BIND(GetErrorString, "ncclGetErrorString")
#undef MBPE_BIND
    r.ok = true;
    return r;
}
}  // namespace

struct mbpe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 256;

    // corpus
    const uint8_t *d_text = nullptr;
    uint8_t *d_text_owned = nullptr;
    uint8_t *d_endmask = nullptr;
    uint64_t n_bytes = 0;
    uint64_t n_chunks = 0;
    bool chunked = false;
    uint64_t n_barriers = 0;     // chunk ends in the corpus (= barrier slots of the barrier layout)
    bool barrier = false;        // this training keeps the chunk ends as barrier slots (mbpe_dev.h: kBarrier)
    bool loaded = false;
    bool inert = false;          // whole corpus collapses to one token (NUL quirk, basic)

    uint32_t *pc_scratch = nullptr;   // pair-count scan: per-workgroup histogram snapshots (kept for the context's life)
    uint32_t *pc_bp = nullptr;        // mbpe_pair_count_u8: the 65,536-entry result table
    void *first_state = nullptr;      // `first` tie-break scratch (training)
    uint32_t *xf = nullptr;           // `first` on a sharded stream: every rank's earliest tied pair (hdr_words, see k_first_publish)

    // slot stream
    uint16_t *tok[2] = {nullptr, nullptr};
    int cur = 0;
    uint64_t cap_slots = 0;      // allocated slots per buffer
    uint64_t n_slots = 0;        // physical slots in use (multiple of kTile)
    uint32_t n_tiles = 0;
    TileSum *sums = nullptr;     // live tile summaries
    TileSum *side = nullptr;     // staging for the tiles a merge pass changed
    uint32_t *chg = nullptr;     // bitmap of those tiles
    uint32_t *tile_list = nullptr;   // the same as a dense list (batch rewrite pass)
    SelList *sel = nullptr;          // candidates of the threshold selection
    uint32_t *run_in = nullptr;      // per tile: run of t before it, for (t,t) pairs (k_run_final)
    int seq_slot = -1;               // index of the sequence being enqueued within its group (opt_time_kernels)
    unsigned long long *offsets = nullptr;

    // pair table
    PairTable tab = {};
    uint32_t hcap = 0;
    uint32_t *bp = nullptr;      // unused (the byte-pair table is the front of xb0)
    DevCtl *ctl = nullptr;
    unsigned long long *best = nullptr;   // [n_target + 1]
    // exchange buffer of one merge: [header: m, adj, RankEdge x n_ranks][LR: L[x], R[x] interleaved]
    uint32_t *xb = nullptr;
    uint32_t hdr_words = 0;               // single-merge header: m, adj, RankEdge x n_ranks
    uint32_t hdrb_words = 0;              // batch header: m_j, ADJ
    uint32_t *hdr_m = nullptr, *hdr_adj = nullptr;   // inside xb
    uint32_t *LR = nullptr;               // xb + hdr_words + hdrb_words
    BatchState *bs = nullptr;
    uint32_t *xb0 = nullptr;              // begin: [bp 65,536][header]
    uint32_t *pair_cells = nullptr;       // byte x byte cell block per pair of a batch (launch_pair_cells_fold), or NULL
    RankEdge *d_left = nullptr, *d_right = nullptr;   // composed neighbours (multi-GPU)

    // training
    uint32_t vocab_size = 0;
    uint32_t n_target = 0;
    uint32_t k = 0;              // merges launched
    uint32_t n_valid = 0;        // merges known to be real (table was not empty)
    bool begun = false;
    bool exhausted = false;
    double merges_per_seq = 0;   // measured over the last group of sequences (0: not known yet)
    uint32_t first_b = 0, first_s = 0;   // `first` mode: sequences / single-pair sequences seen since begin
    bool first_legacy = false;   // `first` mode: most batches were single pairs with a shared count (position tie-breaks):
                                 //   the rest of the run takes the leaner one-merge-per-pass loop
    DevCtl h_ctl = {};

    // multi-GPU
    int rank = 0, n_ranks = 1;
    bool comm_external = false;           // the caller performs the all-reduce
    void *nccl_comm = nullptr;
    int pending = 0;                      // external mode: 0 none, 1 begin, 2 step
    uint32_t pending_target = 0;          // external mode: merge count the pending call runs up to

    // options
    int64_t opt_compact_den = 16;
    int64_t opt_batch = 16;         // sequences between two host synchronisations (compaction is decided there)
    int64_t opt_time_kernels = 0;   // HIP events around every merge kernel (bench.py)
    int64_t opt_force_exchange = 0; // run the multi-rank path (edges, exchange) even with one rank
    int64_t opt_hier_argmax = -1;   // -1 auto (by table size), 0 full scan, 1 hierarchical
    int64_t opt_multi_merge = 1;    // 1: several independent merges per stream pass (batch sequences)
    int64_t opt_max_batch = kBatchDefault;
    int64_t opt_fused_min = 24;     // batches of at least this many pairs take the fused pass
    bool opt_byte_table = true;         // batches of byte pairs use the byte x byte lookup table (tests turn it off)
    int64_t opt_sel_cap = kSelCap;      // candidate-list capacity (tests lower it to force the overflow path)
    int64_t opt_threshold_select = 1;   // 0: always select with the bound-walking kernel
    int64_t opt_dense_table = -1;   // -1 auto / 1: dense pair table when vocab <= 32,768; 0: always hashed
    int hot_possible = 1;           // may a batch of the next group of sequences hold a "frequent" pair (kernels' dc_wanted)?
    unsigned long long last_top = ~0ull;   // count of the latest merge the host has seen: bounds every later count
    int64_t opt_barrier = -1;       // chunk ends as barrier slots: -1 when the ids need 16 bits, 0 never, 1 always
    int64_t opt_first_batches = 0;  // `first` mode: 1 = pairs whose counts no other pair shares are merged in batches too
                                    //   (no faster on text: words make chains of pairs with one count)
    int64_t opt_first = 0;          // 1: `first` tie-break (insertion order, PairCount.h:65-74) instead of lexical
    int64_t opt_pc_repeat = 1;      // mbpe_pair_count_u8 without an output table: launches per call (timing)
    uint32_t k_upper = 0;           // host-side upper bound of the device's k_done
    int sel_attempts = 3;           // gather + pick attempts enqueued per selection (see train_steps_batched)
    uint32_t max_batch_eff = kBatchMax, adj_pitch = kBatchMax;   // (set by mbpe_train_begin: see begin_local)
    // multi-GPU: what the selection of the sequence in flight decided (k_done, k_limit, batch_n, commit_n of DevCtl),
    // read back while its stream pass runs, so that exactly the cells the batch can touch are exchanged
    uint32_t *h_seq = nullptr;      // pinned, 16 words: [0..3] the multi-GPU copy, [8..15] launch_seq_info's
    uint32_t *d_seq_info = nullptr; // 8 words of device memory for launch_seq_info
    int64_t opt_pair_cells = -1;    // one atomic per match with byte neighbours (mbpe_dev.h: launch_pair_cells_fold): -1 for
                                    //   streams from 64 Mi slots on, 0 never, 1 always
    int64_t opt_lockstep = -1;      // small corpora: the host waits for every selection and enqueues only the kernels that
                                    //   sequence needs (-1: when the stream is short enough, 0 never, 1 always)
    hipEvent_t ev_sel = nullptr;
    std::vector<hipEvent_t> kev;    // event pool for opt_time_kernels
    std::vector<hipEvent_t> kev_f;  // ... around the fused pass alone
    uint32_t *seq_flags = nullptr;  // per sequence of a group: 1 = a fused pass ran (opt_time_kernels)
    std::vector<uint32_t> h_seq_flags;

    // 32-bit continuation of a training whose vocabulary lies beyond the 16-bit slot format (wide.h)
    bool wide = false;               // this training hands over to the 32-bit loop after n_target merges
    bool wide_active = false;        // ... and has done so
    uint32_t vocab_total = 0;        // the caller's vocab_size (vocab_size is the 16-bit part's)
    uint32_t n_target_total = 0;     // vocab_total - 256
    int64_t opt_wide_from = -1;      // tests: hand over after this many merges whatever the vocabulary (-1: at the format's limit)
    uint32_t *wtok[2] = {nullptr, nullptr};
    int wcur = 0;
    uint32_t *wval = nullptr, *wscratch = nullptr;
    WideTable wtab = {};
    WideCtl *wctl = nullptr;
    WideBest *wbest = nullptr;
    unsigned long long *warg = nullptr;
    uint64_t wn_upper = 0;           // host-side upper bound of the stream length
    uint32_t wk = 0;                 // merges of the wide loop known to the host
    WideCtl h_wctl = {};
    std::vector<WideBest> h_wbest;   // ... and what they were

    mbpe_stats stats = {};

    // Device buffers of a training run are kept when the run ends and handed out again to the next one that asks
    // for the same sizes (mbpe_train_begin twice on one corpus: no 25 GB of hipFree / hipMalloc in between).
    std::unordered_map<void *, size_t> pool_live;
    std::multimap<size_t, void *> pool_idle;
};

namespace {

template <typename T>
hipError_t tmalloc(mbpe_ctx *c, T **p, size_t bytes) {
    if (bytes == 0) bytes = 4;
    auto it = c->pool_idle.find(bytes);
    if (it != c->pool_idle.end()) {
        *p = static_cast<T *>(it->second);
        c->pool_live[it->second] = bytes;
        c->pool_idle.erase(it);
        return hipSuccess;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e == hipErrorOutOfMemory && !c->pool_idle.empty()) {       // give the idle buffers back and retry
        pool_trim(c);
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) return e;
    c->pool_live[q] = bytes;
    *p = static_cast<T *>(q);
    return hipSuccess;
}

template <typename T>
void tfree(mbpe_ctx *c, T *&p) {
    if (!p) return;
    auto it = c->pool_live.find((void *)p);
    if (it != c->pool_live.end()) {
        c->pool_idle.emplace(it->second, it->first);
        c->pool_live.erase(it);
    } else {
        (void)hipFree((void *)p);
    }
    p = nullptr;
}

// a buffer that leaves the pool for good (freed right away)
template <typename T>
void trelease(mbpe_ctx *c, T *p) {
    if (!p) return;
    c->pool_live.erase((void *)p);
    (void)hipFree((void *)p);
}

void pool_trim(mbpe_ctx *c) {
    for (auto &kv : c->pool_idle) (void)hipFree(kv.second);
    c->pool_idle.clear();
}

int sync_ctl(mbpe_ctx *c) {
    HIPCHK(hipMemcpyAsync(&c->h_ctl, c->ctl, sizeof(DevCtl), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->cur = (int)(c->h_ctl.cur & 1u);      // a fused pass flips the live buffer on the device
#ifdef MBPE_DIAG
    if (getenv("MBPE_SCAN_DIAG") || getenv("MBPE_MERGE_DIAG") || getenv("MBPE_FUSED_DIAG")) c->h_ctl.err = 0;   // timing-only kernels break the counts
#endif
    if (c->h_ctl.err) {
        char buf[200];
        snprintf(buf, sizeof(buf), "device error flags 0x%x (1=pair table full, 2=negative count, 4=missing pair, "
                                   "8=a pair occurs 2^31 times or more, 16=a stream pass was skipped)",
                 c->h_ctl.err);
        mbpe_host::set_last_error(buf);
        return MBPE_ERR_OVERFLOW;
    }
    return MBPE_OK;
}

void free_training(mbpe_ctx *c) {
    tfree(c, c->tok[0]); tfree(c, c->tok[1]);
    tfree(c, c->sums); tfree(c, c->side); tfree(c, c->chg); tfree(c, c->tile_list);
    tfree(c, c->offsets);
    tfree(c, c->tab.hslot); tfree(c, c->tab.ekey); tfree(c, c->tab.ecnt); tfree(c, c->tab.cells);
    tfree(c, c->tab.bmax); tfree(c, c->tab.smax);
    tfree(c, c->bp); tfree(c, c->ctl); tfree(c, c->best); tfree(c, c->xb); tfree(c, c->xb0); tfree(c, c->pair_cells);
    tfree(c, c->d_left); tfree(c, c->d_right); tfree(c, c->bs); tfree(c, c->sel); tfree(c, c->seq_flags); tfree(c, c->run_in);
    tfree(c, c->first_state);
    tfree(c, c->xf);
    tfree(c, c->wtok[0]); tfree(c, c->wtok[1]); tfree(c, c->wval); tfree(c, c->wscratch);
    tfree(c, c->wtab.keys); tfree(c, c->wtab.cnts); tfree(c, c->wctl); tfree(c, c->wbest); tfree(c, c->warg);
    c->wtab = {};
    c->wide_active = false;
    c->wk = 0;
    c->wn_upper = 0;
    c->h_wbest.clear();
    c->LR = nullptr;
    c->pending = 0;
    c->begun = false;
    c->k = c->n_valid = 0;
    c->exhausted = false;
}

void free_corpus(mbpe_ctx *c) {
    dfree(c->d_text_owned);
    dfree(c->d_endmask);
    c->d_text = nullptr;
    c->loaded = false;
}

// dense layout: one cell per possible pair, row pitch = vocab rounded up to a power of two
static inline bool dense_possible(const mbpe_ctx *c) { return c->vocab_size <= 32768; }
static inline bool use_dense(const mbpe_ctx *c) {
    if (!dense_possible(c)) return false;
    if (c->wide) return false;           // (the conversion to the 32-bit loop reads the hashed layout's entry arrays)
    return c->opt_dense_table != 0;      // -1 (auto) and 1: dense whenever the vocabulary allows it
}

int alloc_table_dense(mbpe_ctx *c) {
    uint32_t vshift = 8;
    while ((1u << vshift) < c->vocab_size) ++vshift;
    c->tab = {};
    c->tab.vshift = vshift;
    c->tab.ecap = 1u << (2 * vshift);                 // <= 2^30 cells
    const size_t cells = (size_t)1 << (2 * vshift);
    HIPCHK(tmalloc(c, &c->tab.cells, cells * 4));
    HIPCHK(hipMemsetAsync(c->tab.cells, 0, cells * 4, c->stream));
    const size_t nb = (cells >> kBlockShift) + 2, ns = (cells >> (2 * kBlockShift)) + 2;
    HIPCHK(tmalloc(c, &c->tab.bmax, nb * 8));
    HIPCHK(tmalloc(c, &c->tab.smax, ns * 8));
    HIPCHK(hipMemsetAsync(c->tab.bmax, 0, nb * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->tab.smax, 0, ns * 8, c->stream));
    return MBPE_OK;
}

int alloc_table(mbpe_ctx *c, uint32_t ecap) {
    c->tab.ecap = ecap;
    c->hcap = next_pow2((uint64_t)ecap * 2);
    c->tab.hmask = c->hcap - 1;
    HIPCHK(tmalloc(c, &c->tab.hslot, (size_t)c->hcap * 8));
    HIPCHK(tmalloc(c, &c->tab.ekey, (size_t)ecap * 4));
    HIPCHK(tmalloc(c, &c->tab.ecnt, (size_t)ecap * 4));
    const size_t nb = ((size_t)ecap >> kBlockShift) + 2, ns = ((size_t)ecap >> (2 * kBlockShift)) + 2;
    HIPCHK(tmalloc(c, &c->tab.bmax, nb * 8));
    HIPCHK(tmalloc(c, &c->tab.smax, ns * 8));
    HIPCHK(hipMemsetAsync(c->tab.bmax, 0, nb * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->tab.smax, 0, ns * 8, c->stream));
    launch_fill_u32(c->stream, reinterpret_cast<uint32_t *>(c->tab.hslot), (uint64_t)c->hcap * 2, 0xFFFFFFFFu);
    return MBPE_OK;
}

// entries a batch of `steps` merges can add at most: per merge one (x,X) per
// left neighbour id, one (X,y) per right neighbour id and (X,X)
uint64_t batch_headroom(const mbpe_ctx *c, uint32_t steps) {
    uint64_t per = 2ull * c->vocab_size + 1;
    return per * steps;
}

uint64_t table_cap_limit(const mbpe_ctx *c) {
    uint64_t v = c->vocab_size;
    uint64_t lim = v * v;                    // every possible pair
    if (lim > (1ull << 30)) lim = 1ull << 30;   // argmax hierarchy: 1024 super-blocks of 1024 x 1024 entries
    return std::max<uint64_t>(lim, 1024);
}

struct WallTimer {
    float *acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit WallTimer(float *a) : acc(a) {}
    ~WallTimer() { *acc += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

int grow_table(mbpe_ctx *c, uint64_t want) {
    if (c->tab.cells) return MBPE_OK;          // the dense layout holds every possible pair
    WallTimer timer(&c->stats.ms_grow_table);
    c->stats.n_table_grows++;
    uint64_t lim = table_cap_limit(c);
    uint64_t ecap = std::min<uint64_t>(std::max<uint64_t>(want, (uint64_t)c->tab.ecap * 4), lim);
    if (ecap <= c->tab.ecap) return MBPE_OK;   // already at the limit: cannot overflow
    PairTable old = c->tab;
    c->tab = {};
    int rc = alloc_table(c, (uint32_t)ecap);
    if (rc != MBPE_OK) return rc;
    uint32_t n = c->h_ctl.n_entries;
    HIPCHK(hipMemcpyAsync(c->tab.ekey, old.ekey, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->tab.ecnt, old.ecnt, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
    // the block bounds refer to entry indices, which do not change
    HIPCHK(hipMemcpyAsync(c->tab.bmax, old.bmax, (((size_t)old.ecap >> kBlockShift) + 2) * 8,
                          hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->tab.smax, old.smax, (((size_t)old.ecap >> (2 * kBlockShift)) + 2) * 8,
                          hipMemcpyDeviceToDevice, c->stream));
    launch_table_rehash(c->stream, c->tab, c->ctl);
    HIPCHK(hipStreamSynchronize(c->stream));
    trelease(c, old.hslot); trelease(c, old.ekey); trelease(c, old.ecnt);
    trelease(c, old.bmax); trelease(c, old.smax);
    return MBPE_OK;
}

int ensure_pc_scratch(mbpe_ctx *c) {
    if (!c->pc_scratch) HIPCHK(hipMalloc(&c->pc_scratch, pair_count_scratch_bytes(c->n_cus)));
    return MBPE_OK;
}

int do_compact(mbpe_ctx *c) {
    // c->h_ctl must be current
    if (!c->n_tiles) return MBPE_OK;
    WallTimer timer(&c->stats.ms_compact);
    const int src = c->cur, dst = 1 - c->cur;
    launch_tile_scan(c->stream, c->sums, c->n_tiles, c->offsets, c->ctl);
    launch_compact_scatter(c->stream, c->tok[src], c->sums, c->offsets, c->n_tiles, c->tok[dst], c->n_cus);
    const uint64_t live = c->h_ctl.n_live;
    const uint64_t padded = std::max<uint64_t>(round_up(live, kTile), kTile);
    launch_fill_u16(c->stream, c->tok[dst] + live, padded - live, (uint16_t)kHole);
    c->cur = dst;
    c->n_slots = padded;
    c->n_tiles = (uint32_t)(padded / kTile);
    launch_summarize(c->stream, c->tok[c->cur], c->sums, c->n_tiles, c->n_cus);
    DevCtl patch = c->h_ctl;
    patch.removed_total = 0;
    HIPCHK(hipMemcpyAsync(&c->ctl->removed_total, &patch.removed_total, sizeof(patch.removed_total),
                          hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->ctl->cur), c->cur, 1, c->stream));
    c->h_ctl.cur = (uint32_t)c->cur;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->h_ctl.removed_total = 0;
    c->stats.n_compactions++;
    return MBPE_OK;
}

}  // namespace

extern "C" {

int mbpe_create(int device_id, mbpe_ctx **out) {
    if (!out) { mbpe_host::set_last_error("mbpe_create: out is NULL"); return MBPE_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) {
        mbpe_host::set_last_error("no usable HIP device (the MI355X path has no CPU fallback)");
        return MBPE_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device_id));
    mbpe_ctx *c = new mbpe_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_seq), 64, hipHostMallocDefault) != hipSuccess) {
        mbpe_host::set_last_error("could not create HIP stream / events");
        delete c;
        return MBPE_ERR_HIP;
    }
    *out = c;
    return MBPE_OK;
}

void mbpe_destroy(mbpe_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_training(c);
    pool_trim(c);
    free_corpus(c);
    dfree(c->pc_scratch);
    dfree(c->pc_bp);
    if (c->nccl_comm && rccl().ok) rccl().CommDestroy(c->nccl_comm);
    for (hipEvent_t e : c->kev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->kev_f) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_sel) (void)hipEventDestroy(c->ev_sel);
    if (c->h_seq) (void)hipHostFree(c->h_seq);
    dfree(c->d_seq_info);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mbpe_set_option(mbpe_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return MBPE_ERR_ARG;
    std::string n(name);
    if (n == "compact_den") c->opt_compact_den = value;
    else if (n == "batch") c->opt_batch = std::max<int64_t>(1, value);
    else if (n == "time_kernels") c->opt_time_kernels = value;
    else if (n == "force_exchange") c->opt_force_exchange = value;
    else if (n == "hier_argmax") c->opt_hier_argmax = value;
    else if (n == "multi_merge") c->opt_multi_merge = value;
    else if (n == "max_batch") c->opt_max_batch = std::min<int64_t>(std::max<int64_t>(1, value), kBatchMax);
    else if (n == "fused_min") c->opt_fused_min = std::max<int64_t>(2, value);
    else if (n == "dense_table") c->opt_dense_table = value;
    else if (n == "threshold_select") c->opt_threshold_select = value != 0;
    else if (n == "conflict_resolution") {
        if (value != 0 && value != 1) { mbpe_host::set_last_error("conflict_resolution: 0 = first, 1 = lexical"); return MBPE_ERR_ARG; }
        // (only while a training is under way: a finished one, or one whose table ran empty, can be followed by
        //  another mbpe_train_begin on the same corpus with the other tie-break)
        if (c->begun && (c->pending || (c->k < c->n_target && !c->exhausted))) {
            mbpe_host::set_last_error("conflict_resolution cannot change in the middle of a training");
            return MBPE_ERR_STATE;
        }
        c->opt_first = value == 0;
    }
    else if (n == "chunk_barrier") c->opt_barrier = value < 0 ? -1 : value != 0;       // (read by the next mbpe_train_begin)
    else if (n == "first_batches") c->opt_first_batches = value != 0;
    else if (n == "sel_cap") c->opt_sel_cap = std::min<int64_t>(std::max<int64_t>(64, value), kSelCap);
    else if (n == "byte_table") c->opt_byte_table = value != 0;
    else if (n == "lockstep") c->opt_lockstep = value < 0 ? -1 : value != 0;
    else if (n == "pair_cells") c->opt_pair_cells = value < 0 ? -1 : value != 0;      // (read by the next mbpe_train_begin)
    else if (n == "wide_from") c->opt_wide_from = value < 0 ? -1 : value;      // (read by the next mbpe_train_begin)
    else if (n == "pc_repeat") c->opt_pc_repeat = std::min<int64_t>(std::max<int64_t>(1, value), 1000);
    else { mbpe_host::set_last_error("unknown option " + n); return MBPE_ERR_ARG; }
    return MBPE_OK;
}

int mbpe_load_corpus(mbpe_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *chunk_off,
                     uint64_t n_chunks, int text_on_device) {
    if (!c || (!text && n_bytes)) { mbpe_host::set_last_error("mbpe_load_corpus: NULL argument"); return MBPE_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    free_training(c);
    free_corpus(c);
    // (the training buffers of a corpus of another size would not be handed out again: give them back before the
    //  new corpus is allocated, not only inside the next mbpe_train_begin)
    if (n_bytes != c->n_bytes || (chunk_off != nullptr) != c->chunked) pool_trim(c);
    c->n_bytes = n_bytes;
    c->chunked = chunk_off != nullptr;
    c->n_chunks = chunk_off ? n_chunks : 1;
    c->n_barriers = 0;
    c->barrier = false;
    c->inert = false;
    if (chunk_off) {
        if (chunk_off[0] != 0 || chunk_off[n_chunks] != n_bytes) {
            mbpe_host::set_last_error("chunk_off must start at 0 and end at n_bytes");
            return MBPE_ERR_ARG;
        }
        for (uint64_t i = 0; i < n_chunks; ++i)
            if (chunk_off[i] > chunk_off[i + 1]) {
                mbpe_host::set_last_error("chunk_off must be ascending");
                return MBPE_ERR_ARG;
            }
    }
    if (text_on_device && ((uintptr_t)text & 15)) {
        mbpe_host::set_last_error("device text must be 16-byte aligned");
        return MBPE_ERR_ARG;
    }

    // host view of the text for the NUL-quirk check (Tokenizer.h:86-93)
    std::vector<uint8_t> tmp;
    const uint8_t *h_text = text;
    if (text_on_device && n_bytes) {
        bool need_all = chunk_off != nullptr;
        uint64_t take = need_all ? n_bytes : std::min<uint64_t>(n_bytes, 64);
        tmp.resize(take);
        HIPCHK(hipMemcpy(tmp.data(), text, take, hipMemcpyDeviceToHost));
        h_text = tmp.data();
    }
    const uint64_t n_vec = (n_bytes + 15) / 16;
    std::vector<uint8_t> mask;
    if (chunk_off) {
        mask.assign(n_vec * 2 + 16, 0);
        uint64_t dropped = 0;
        for (uint64_t ci = 0; ci < n_chunks; ++ci) {
            uint64_t s = chunk_off[ci], e = chunk_off[ci + 1];
            if (e == s) continue;
            mask[(e - 1) >> 3] |= (uint8_t)(1u << ((e - 1) & 7));
            if (h_text[s] == 0 && stoi_parses(h_text + s + 1, e - s - 1)) {
                // collapses to one token in the reference: no pair ever starts or ends
                // inside it -> mark every byte as a chunk end so it stays inert
                for (uint64_t i = s; i < e; ++i) mask[i >> 3] |= (uint8_t)(1u << (i & 7));
                dropped++;
            }
        }
        c->n_chunks = n_chunks - dropped;
        uint64_t ends = 0;
        for (uint8_t b : mask) ends += (uint64_t)__builtin_popcount(b);
        c->n_barriers = ends;
    } else if (n_bytes && h_text[0] == 0) {
        // one chunk = whole text; stoi only looks at a prefix, 64 bytes are enough
        // to decide unless the prefix is all whitespace (then parse the host copy)
        uint64_t have = text_on_device ? tmp.size() : n_bytes;
        bool parses = stoi_parses(h_text + 1, have - 1);
        if (!parses && text_on_device && have < n_bytes) {
            bool all_ws = true;
            for (uint64_t i = 1; i < have; ++i)
                if (!(h_text[i] == ' ' || (h_text[i] >= 9 && h_text[i] <= 13))) { all_ws = false; break; }
            if (all_ws) {
                tmp.resize(n_bytes);
                HIPCHK(hipMemcpy(tmp.data(), text, n_bytes, hipMemcpyDeviceToHost));
                parses = stoi_parses(tmp.data() + 1, n_bytes - 1);
            }
        }
        c->inert = parses;
        if (parses) c->n_chunks = 0;
    }

    if (text_on_device) {
        c->d_text = text;
    } else {
        HIPCHK(hipMalloc(&c->d_text_owned, std::max<uint64_t>(n_vec * 16, 16)));
        if (n_bytes) HIPCHK(hipMemcpyAsync(c->d_text_owned, text, n_bytes, hipMemcpyHostToDevice, c->stream));
        c->d_text = c->d_text_owned;
    }
    if (chunk_off) {
        HIPCHK(hipMalloc(&c->d_endmask, mask.size()));
        HIPCHK(hipMemcpyAsync(c->d_endmask, mask.data(), mask.size(), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->loaded = true;
    c->stats = {};
    c->stats.n_bytes = n_bytes;
    c->stats.n_chunks = c->n_chunks;
    return MBPE_OK;
}

int mbpe_load_corpus_ranges(mbpe_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *starts,
                            const uint64_t *ends, uint64_t n_chunks, int text_on_device) {
    if (!c || (!text && n_bytes) || ((!starts || !ends) && n_chunks)) {
        mbpe_host::set_last_error("mbpe_load_corpus_ranges: NULL argument");
        return MBPE_ERR_ARG;
    }
    uint64_t pos = 0, packed = 0;
    bool tiled = true;
    for (uint64_t i = 0; i < n_chunks; ++i) {
        if (starts[i] < pos || ends[i] < starts[i] || ends[i] > n_bytes) {
            mbpe_host::set_last_error("chunk ranges must be ascending, disjoint and inside the text");
            return MBPE_ERR_ARG;
        }
        if (starts[i] != pos) tiled = false;
        pos = ends[i];
        packed += ends[i] - starts[i];
    }
    if (pos != n_bytes) tiled = false;
    std::vector<uint64_t> off(n_chunks + 1);
    if (tiled) {
        for (uint64_t i = 0; i < n_chunks; ++i) off[i] = starts[i];
        off[n_chunks] = n_bytes;
        return mbpe_load_corpus(c, text, n_bytes, off.data(), n_chunks, text_on_device);
    }
    // bytes between the chunks are skipped (Tokenizer.h:506-540): train on the chunks packed together
    std::vector<uint8_t> host_copy;
    const uint8_t *src = text;
    if (text_on_device) {
        HIPCHK(hipSetDevice(c->device));
        host_copy.resize(n_bytes);
        HIPCHK(hipMemcpy(host_copy.data(), text, n_bytes, hipMemcpyDeviceToHost));
        src = host_copy.data();
    }
    std::vector<uint8_t> buf(packed);
    uint64_t w = 0;
    for (uint64_t i = 0; i < n_chunks; ++i) {
        off[i] = w;
        memcpy(buf.data() + w, src + starts[i], ends[i] - starts[i]);
        w += ends[i] - starts[i];
    }
    off[n_chunks] = w;
    return mbpe_load_corpus(c, buf.data(), packed, off.data(), n_chunks, 0);
}

int mbpe_pair_count_u8(mbpe_ctx *c, uint32_t *table65536_out) {
    if (!c) return MBPE_ERR_ARG;
    if (!c->loaded) { mbpe_host::set_last_error("mbpe_pair_count_u8: no corpus loaded"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    int rcs = ensure_pc_scratch(c);
    if (rcs != MBPE_OK) return rcs;
    if (!c->pc_bp) HIPCHK(hipMalloc(&c->pc_bp, 65536 * 4));
    uint32_t *bp = c->pc_bp;
    HIPCHK(hipMemsetAsync(bp, 0, 65536 * 4, c->stream));
    // ("pc_repeat" > 1: that many launches back to back between the two events -- the kernel's sustained duration
    //  without one event marker's overhead per launch; the table then holds that multiple of every count)
    const int reps = table65536_out ? 1 : (int)std::max<int64_t>(1, c->opt_pc_repeat);
    // (every dispatch also carries its own start / stop events -- hipExtLaunchKernelGGL -- so that the kernel's duration
    //  is known without the gap between two launches or the cost of a marker: ms_pair_count_kernel)
    while (c->kev.size() < 2ull * reps) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        c->kev.push_back(ev);
    }
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    if (!c->inert)
        for (int r = 0; r < reps; ++r)
            launch_pair_count_u8(c->stream, c->d_text, c->n_bytes, c->d_endmask, bp, c->n_cus, c->pc_scratch,
                                 c->kev[2 * r], c->kev[2 * r + 1]);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && table65536_out) e = hipMemcpy(table65536_out, bp, 65536 * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { mbpe_host::set_last_error(hip_err("pair count", e)); return MBPE_ERR_HIP; }
    HIPCHK(hipEventElapsedTime(&c->stats.ms_pair_count, c->ev0, c->ev1));
    c->stats.ms_pair_count /= (float)reps;
    c->stats.ms_pair_count_kernel = 0;
    if (!c->inert && c->n_bytes >= 2) {
        float sum = 0;
        for (int r = 0; r < reps; ++r) {
            float km = 0;
            HIPCHK(hipEventElapsedTime(&km, c->kev[2 * r], c->kev[2 * r + 1]));
            sum += km;
        }
        c->stats.ms_pair_count_kernel = sum / (float)reps;
    }
    c->stats.pair_count_launches += (uint32_t)reps;
#ifdef MBPE_PC_STAMPS
    {   // diagnostic build: the phases of the last launch, per workgroup (k_pair_count_u8_fast's stamps, 10 ns units)
        std::vector<unsigned long long> st((size_t)c->n_cus * 16);
        HIPCHK(hipMemcpy(st.data(), reinterpret_cast<const char *>(c->pc_scratch) + (size_t)c->n_cus * 32768 * 4, st.size() * 8,
                         hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < c->n_cus; ++w) if (st[16 * w] && st[16 * w] < t0) t0 = st[16 * w];
        for (int k = 0; k < 16; ++k) {
            double lo = 1e30, hi = 0, sum = 0; int cnt = 0;
            for (int w = 0; w < c->n_cus; ++w) {
                if (!st[16 * w + k]) continue;
                const double us = (double)(st[16 * w + k] - t0) * 0.01;
                lo = std::min(lo, us); hi = std::max(hi, us); sum += us; ++cnt;
            }
            if (cnt) fprintf(stderr, "pc stamp %2d: min %8.2f avg %8.2f max %8.2f us (%d workgroups)\n", k, lo, sum / cnt, hi, cnt);
        }
    }
#endif
    return MBPE_OK;
}

// ---- training phases ---------------------------------------------------------
// begin  = begin_local  -> [all-reduce xb0] -> begin_finish
// step   = step_local   -> [all-reduce xb ] -> step_finish
// With one rank the exchange is skipped.  With RCCL the three parts are
// enqueued back to back on the context's stream; in external mode the
// library stops after *_local so that the caller can reduce the buffer.

// large tables: walk the block bounds instead of scanning every entry ("hier_argmax": -1 auto, 0 never, 1 always)
static inline bool use_hier(const mbpe_ctx *c) {
    if (c->opt_hier_argmax >= 0) return c->opt_hier_argmax != 0;
    if (c->tab.cells) return c->tab.ecap > (1u << 20);
    return c->h_ctl.n_entries > (1u << 20);
}

// the chunk-end convention of the stream, as the launchers take it
static inline uint32_t endbit_of(const mbpe_ctx *c) { return !c->chunked ? 0u : c->barrier ? kBarrier : kEndBit; }

static inline bool is_multi(const mbpe_ctx *c) { return c->n_ranks > 1 || c->opt_force_exchange; }

static int begin_local(mbpe_ctx *c, uint32_t vocab_size) {
    free_training(c);
    c->vocab_size = vocab_size;
    c->n_target = vocab_size - 256;
    c->k = 0;
    c->n_valid = 0;
    c->exhausted = false;
    c->first_legacy = false;
    c->sel_attempts = 3;             // (the first selection of a training primes its threshold through attempts 1 and 2)
    c->merges_per_seq = 0;
    c->first_b = c->first_s = 0;

    const uint64_t n = c->inert ? 0 : c->n_bytes;
    // barrier layout: written sparsely (byte i -> slot 2i, its barrier or a hole behind it) into a first buffer of
    // twice the size, squeezed into the second one below, and only then does the first one get its final size
    const uint64_t n_bar = c->barrier ? c->n_barriers : 0;
    const uint64_t dense_slots = std::max<uint64_t>(round_up(n + n_bar, kTile), kTile);
    c->n_slots = c->barrier ? std::max<uint64_t>(2 * round_up(n, kTile / 2), kTile) : dense_slots;
    c->cap_slots = dense_slots;
    if (c->n_slots / kTile > 0x0FFFFFF0ull) { mbpe_host::set_last_error("corpus shard too large"); return MBPE_ERR_ARG; }
    c->n_tiles = (uint32_t)(c->n_slots / kTile);
    c->hdr_words = exchange_header_words(c->n_ranks);
    // the largest batch this training selects: "max_batch", and with several ranks at most 1,024 pairs -- the ADJ block
    // (batch x batch) is part of what every sequence all-reduces
    c->max_batch_eff = (uint32_t)std::min<int64_t>(c->opt_max_batch, is_multi(c) ? 1024 : kBatchMax);
    c->adj_pitch = (uint32_t)round_up(std::max<uint32_t>(c->max_batch_eff, 64u), 64);
    c->hdrb_words = (batch_header_words(c->adj_pitch) + 3) / 4 * 4;
    HIPCHK(tmalloc(c, &c->tok[0], c->n_slots * 2));
    HIPCHK(tmalloc(c, &c->tok[1], c->cap_slots * 2));
    HIPCHK(tmalloc(c, &c->sums, (size_t)c->n_tiles * sizeof(TileSum)));
    HIPCHK(tmalloc(c, &c->side, (size_t)c->n_tiles * sizeof(TileSum)));
    HIPCHK(tmalloc(c, &c->chg, ((size_t)c->n_tiles / 32 + 2) * 4));
    HIPCHK(tmalloc(c, &c->tile_list, ((size_t)c->n_tiles + 64) * 4));
    HIPCHK(hipMemsetAsync(c->chg, 0, ((size_t)c->n_tiles / 32 + 2) * 4, c->stream));
    HIPCHK(tmalloc(c, &c->offsets, ((size_t)c->n_tiles + tile_scan_scratch(c->n_tiles)) * 8));
    // (the LR rows of the largest batch THIS training can select: seq_stage_a clamps every batch to max_batch_eff, so a
    //  later, larger "max_batch" cannot reach beyond it.  2 x max_batch_eff x lr_pitch(vocab) words: 1.05 GB at the
    //  defaults with vocab 32,000, 262 MB with several ranks or "max_batch" 1024)
    const size_t xb_words = (size_t)c->hdr_words + c->hdrb_words + lr_words(vocab_size, c->max_batch_eff) + 8;
    const size_t xb0_words = 65536 + (size_t)c->hdr_words;
    HIPCHK(tmalloc(c, &c->xb, xb_words * 4));
    HIPCHK(tmalloc(c, &c->xb0, xb0_words * 4));
    HIPCHK(hipMemsetAsync(c->xb, 0, xb_words * 4, c->stream));
    HIPCHK(hipMemsetAsync(c->xb0, 0, xb0_words * 4, c->stream));
    if (c->opt_pair_cells > 0 || (c->opt_pair_cells < 0 && c->n_slots >= (64ull << 20))) {
        const size_t cells = (size_t)c->max_batch_eff * 65536u;
        HIPCHK(tmalloc(c, &c->pair_cells, cells * 4));
        HIPCHK(hipMemsetAsync(c->pair_cells, 0, cells * 4, c->stream));
    }
    c->hdr_m = c->xb + c->hdr_words;
    c->hdr_adj = c->hdr_m + kBatchMax;
    c->LR = c->xb + c->hdr_words + c->hdrb_words;
    HIPCHK(tmalloc(c, &c->bs, sizeof(BatchState)));
    HIPCHK(tmalloc(c, &c->sel, sizeof(SelList)));
    HIPCHK(tmalloc(c, &c->run_in, ((size_t)c->n_tiles + 64) * 4));
    HIPCHK(tmalloc(c, &c->seq_flags, 4096 * 4 * 4));      // 4 words per sequence of a group (k_seq_finish)
    HIPCHK(hipMemsetAsync(c->bs, 0, sizeof(BatchState), c->stream));
    if (c->opt_first) {
        HIPCHK(tmalloc(c, &c->first_state, first_state_bytes()));
        launch_first_init(c->stream, c->first_state);
        if (is_multi(c)) {
            HIPCHK(tmalloc(c, &c->xf, (size_t)c->hdr_words * 4));
            HIPCHK(hipMemsetAsync(c->xf, 0, (size_t)c->hdr_words * 4, c->stream));
        }
    }
    c->k_upper = 0;
    c->bp = nullptr;   // the byte-pair table lives at the front of xb0
    HIPCHK(tmalloc(c, &c->d_left, sizeof(RankEdge)));
    HIPCHK(tmalloc(c, &c->d_right, sizeof(RankEdge)));
    HIPCHK(tmalloc(c, &c->ctl, sizeof(DevCtl)));
    HIPCHK(tmalloc(c, &c->best, ((size_t)c->n_target + 2) * 8));
    {
        DevCtl init = {};
        init.n_live = n + n_bar;         // every corpus byte starts as one live token
        init.n_ranks = (uint32_t)std::max(1, c->n_ranks);
        init.first_mode = c->opt_first ? 1u : 0u;
        init.adj_pitch = c->adj_pitch;
        init.cells_on = c->pair_cells ? 1u : 0u;
        init.cells_min = c->opt_pair_cells > 0 ? 2u : kCellsMinBatch;      // (forced on: every batch, and for the whole run)
        c->h_ctl = init;
        HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl, sizeof(DevCtl), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipMemsetAsync(c->best, 0, ((size_t)c->n_target + 2) * 8, c->stream));

    // Sized for 288 GB of HBM: one entry per 8 corpus bytes up front (12 bytes of table per entry
    // plus 16 of hash index), so that even a corpus whose pairs never repeat rarely has to grow
    // the table -- growing means multi-GB allocations and a rehash.
    uint64_t want = 65536 + 2 * batch_headroom(c, (uint32_t)c->opt_batch);
    want = std::max<uint64_t>(want, n / 8);
    want = std::min<uint64_t>(want, table_cap_limit(c));
    int rc = use_dense(c) ? alloc_table_dense(c) : alloc_table(c, (uint32_t)want);
    if (rc != MBPE_OK) return rc;

    rc = ensure_pc_scratch(c);
    if (rc != MBPE_OK) return rc;
    pool_trim(c);        // what this run did not take over from the previous one goes back to the driver
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    if (n >= 2) launch_pair_count_u8(c->stream, c->d_text, n, c->d_endmask, c->xb0, c->n_cus, c->pc_scratch);
    c->cur = 0;
    if (c->barrier) {
        launch_widen_barrier(c->stream, c->d_text, n, c->d_endmask, c->tok[0], c->n_slots);
        launch_summarize(c->stream, c->tok[0], c->sums, c->n_tiles, c->n_cus);
        const double ms_compact = c->stats.ms_compact;
        rc = do_compact(c);                     // tok[0] -> tok[1]; n_slots / n_tiles are the dense ones from here on
        if (rc != MBPE_OK) return rc;
        c->stats.n_compactions--;               // (part of the set-up, not one of the run's compactions)
        c->stats.ms_compact = ms_compact;
        tfree(c, c->tok[0]);
        pool_trim(c);
        HIPCHK(tmalloc(c, &c->tok[0], c->cap_slots * 2));
    } else {
        launch_widen(c->stream, c->d_text, n, c->d_endmask, c->tok[0], c->n_slots);
        launch_summarize(c->stream, c->tok[0], c->sums, c->n_tiles, c->n_cus);
    }
    if (is_multi(c)) {
        uint32_t *hdr = c->xb0 + 65536;
        launch_rank_edge(c->stream, c->sums, c->n_tiles, reinterpret_cast<RankEdge *>(hdr + 2) + c->rank, c->ctl, hdr);
    }
    return MBPE_OK;
}

// The stream kernels exist in a plain and a HOT instantiation (LDS cache for the count updates of a frequent
// pair) and the device decides which one works: top_count * 8192 >= n_live (dc_wanted in kernels.hip).  The
// maximal count never rises, and n_live falls by at most one count per merge, so for the next `merges` merges
// HOT stays impossible while top * 8192 < (n_live - merges * top) * ranks: then it is not even launched.
static void update_hot_possible(mbpe_ctx *c, unsigned long long top_count, uint64_t merges) {
    c->last_top = top_count;
    if (c->comm_external) { c->hot_possible = 1; return; }     // (the caller drives the sequences one by one)
    const unsigned long long live = c->h_ctl.n_live;
    const unsigned long long eaten = merges * top_count;
    c->hot_possible = !(live > eaten && top_count * 8192ull < (live - eaten) * (unsigned long long)std::max(1, c->n_ranks));
}

// `first` on a sharded stream: the position tie-break needs one more exchange (every rank's earliest tied pair), so the
// finish of a begin / of a step comes in two parts around it
static inline bool first_sharded(const mbpe_ctx *c) { return c->opt_first && is_multi(c); }

static void begin_finish_a(mbpe_ctx *c) {
    const uint32_t endbit = endbit_of(c);
    if (is_multi(c)) {
        uint32_t *hdr = c->xb0 + 65536;
        launch_boundary_pairs(c->stream, c->xb0, hdr, c->n_ranks, endbit);
        launch_compose_edges(c->stream, hdr, c->rank, c->n_ranks, c->d_left, c->d_right);
    }
    launch_table_init(c->stream, c->xb0, c->tab, c->ctl);
    launch_argmax(c->stream, c->tab, c->ctl, c->best, use_hier(c));
    if (c->opt_first)
        launch_first_tiebreak(c->stream, c->tab, c->ctl, c->best, c->first_state, c->tok[c->cur], nullptr, c->sums, c->n_tiles,
                              endbit, c->n_cus, 0, first_sharded(c) ? 1 : 0, first_sharded(c) ? c->d_right : nullptr, c->xf,
                              c->rank, std::max(1, c->n_ranks));
}

static int begin_finish_b(mbpe_ctx *c) {
    if (first_sharded(c))
        launch_first_tiebreak(c->stream, c->tab, c->ctl, c->best, c->first_state, c->tok[c->cur], nullptr, c->sums, c->n_tiles,
                              endbit_of(c), c->n_cus, 0, 2, nullptr, c->xf, c->rank, std::max(1, c->n_ranks));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    int rc = sync_ctl(c);
    if (rc != MBPE_OK) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventElapsedTime(&c->stats.ms_begin, c->ev0, c->ev1));
    unsigned long long b0 = 0;
    HIPCHK(hipMemcpy(&b0, c->best, 8, hipMemcpyDeviceToHost));
    c->exhausted = (b0 == 0);   // empty table: the reference loop breaks at once (Tokenizer.h:586-588)
    update_hot_possible(c, b0 >> 32, (uint64_t)std::max<int64_t>(c->opt_batch, 1) * kBatchMax);
    c->begun = true;
    c->stats.ms_steps = 0;
    return MBPE_OK;
}

static int comm_allreduce(mbpe_ctx *c, uint32_t *buf, size_t count);   // RCCL (below)

// (one GPU, or several over RCCL; with an external transport the two parts are driven by mbpe_comm_exchange_done)
static int begin_finish(mbpe_ctx *c) {
    begin_finish_a(c);
    if (first_sharded(c)) {
        const int rc = comm_allreduce(c, c->xf, c->hdr_words);
        if (rc != MBPE_OK) return rc;
    }
    return begin_finish_b(c);
}

static void step_local(mbpe_ctx *c, int ev_slot) {
    const uint32_t endbit = endbit_of(c);
    const uint32_t X = 256 + c->k;
    const bool multi = is_multi(c);
    if (ev_slot >= 0) (void)hipEventRecord(c->kev[2 * ev_slot], c->stream);
    launch_merge(c->stream, c->tok[c->cur], c->tok[1 - c->cur], c->sums, c->side, c->n_tiles, c->chg, c->best + c->k, X, endbit, c->LR,
                 c->ctl, &c->ctl->m, multi ? c->d_left : nullptr, multi ? c->d_right : nullptr, c->n_cus, 0,
                 c->offsets + c->n_tiles, c->run_in, nullptr);
    if (ev_slot >= 0) (void)hipEventRecord(c->kev[2 * ev_slot + 1], c->stream);
    if (multi) {
        launch_patch_sums(c->stream, c->best + c->k, c->sums, c->side, c->chg, c->n_tiles, c->ctl, 0);
        launch_rank_edge(c->stream, c->sums, c->n_tiles, reinterpret_cast<RankEdge *>(c->xb + 2) + c->rank, c->ctl,
                         c->xb);
    }
}

static void step_finish_a(mbpe_ctx *c) {
    const uint32_t X = 256 + c->k;
    const bool multi = is_multi(c);
    launch_apply(c->stream, c->tab, c->ctl, c->best + c->k, X, c->LR, multi ? c->xb : nullptr, c->sums, c->side,
                 c->chg, c->n_tiles, 0);
    if (multi) launch_compose_edges(c->stream, c->xb, c->rank, c->n_ranks, c->d_left, c->d_right);
    launch_argmax(c->stream, c->tab, c->ctl, c->best + c->k + 1, use_hier(c));
    if (c->opt_first)
        launch_first_tiebreak(c->stream, c->tab, c->ctl, c->best + c->k + 1, c->first_state, c->tok[c->cur], nullptr, c->sums,
                              c->n_tiles, endbit_of(c), c->n_cus, 0, first_sharded(c) ? 1 : 0,
                              first_sharded(c) ? c->d_right : nullptr, c->xf, c->rank, std::max(1, c->n_ranks));
}

static void step_finish_b(mbpe_ctx *c) {
    if (first_sharded(c))
        launch_first_tiebreak(c->stream, c->tab, c->ctl, c->best + c->k + 1, c->first_state, c->tok[c->cur], nullptr, c->sums,
                              c->n_tiles, endbit_of(c), c->n_cus, 0, 2, nullptr, c->xf, c->rank, std::max(1, c->n_ranks));
    c->k++;
}

static int step_finish(mbpe_ctx *c) {       // (one GPU, or several over RCCL)
    step_finish_a(c);
    if (first_sharded(c)) {
        const int rc = comm_allreduce(c, c->xf, c->hdr_words);
        if (rc != MBPE_OK) return rc;
    }
    step_finish_b(c);
    return MBPE_OK;
}

// words of xb a sequence has to exchange: both headers + the LR rows of its n_pairs members, lr_pitch(ids) cells each
// (ids = 256 + merges done when the sequence started: the neighbours x that can occur)
static size_t exchange_words(const mbpe_ctx *c, uint32_t ids, uint32_t n_pairs) {
    return (size_t)c->hdr_words + c->hdrb_words + (size_t)lr_words(ids, n_pairs);
}
static size_t step_exchange_words(const mbpe_ctx *c) { return exchange_words(c, 256 + c->k, 1); }

// k_done / batch_n of the sequence whose selection has been enqueued: copied to pinned memory behind it
static_assert(offsetof(DevCtl, batch_n) == offsetof(DevCtl, k_done) + 8 && offsetof(DevCtl, commit_n) == offsetof(DevCtl, k_done) + 12,
              "seq_info_enqueue copies k_done, k_limit, batch_n, commit_n as four consecutive words");
static int seq_info_enqueue(mbpe_ctx *c) {
    HIPCHK(hipMemcpyAsync(c->h_seq, &c->ctl->k_done, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->ev_sel, c->stream));
    return MBPE_OK;
}
static int seq_info_wait(mbpe_ctx *c, uint32_t *ids, uint32_t *n_pairs) {
    HIPCHK(hipEventSynchronize(c->ev_sel));
    *ids = 256u + c->h_seq[0];
    *n_pairs = c->h_seq[2];
    return MBPE_OK;
}

// ---- batch sequences: select -> (single pair: fused merge | several pairs: scan, validate, rewrite) ----
// The merge counter lives on the device (ctl->k_done); the host only keeps an
// upper bound (k_upper) between synchronisations.

// (`first` mode decides between equal counts by stream position: batches only hold pairs whose counts nothing else
//  shares -- k_sel_pick, k_validate -- and a pair with a shared count goes alone, after the position tie-break)
static inline bool use_batches(const mbpe_ctx *c) {
    // (`first` on a sharded stream: one merge per pass, the position tie-break is an exchange of its own)
    return c->opt_multi_merge != 0 && (!c->opt_first || (c->opt_first_batches && !c->first_legacy && !is_multi(c)));
}

// Small corpora: a sequence is ~25 launches of which ~16 return at once (the device decides which stream kernel of which
// instantiation works), and at 4-5 us per dispatch those are a third of a 0.16 ms sequence (BASELINE config 3).  There the
// host waits for the selection's result -- one event wait per sequence, ~20 us -- and enqueues exactly the pass, the
// tables / the single-pair apply and the rewrite this sequence needs.
static constexpr uint64_t kLockstepSlots = 32ull << 20;
static inline bool use_lockstep(const mbpe_ctx *c) {
    if (is_multi(c) || c->opt_first) return false;
    if (c->opt_lockstep >= 0) return c->opt_lockstep != 0;
    return c->n_slots <= kLockstepSlots;
}

static int seq_lockstep(mbpe_ctx *c, int ev_slot, bool *nothing_left) {
    const uint32_t endbit = endbit_of(c);
    *nothing_left = false;
    // The first gather + pick attempt alone; only when it did not choose the batch (a list that overflowed or came back
    // empty, an unprimed threshold: a few times per training) the other two and the bound-walking kernel follow, behind one
    // more wait -- five launches fewer for every other sequence.
    const uint32_t mb = std::min<uint32_t>((uint32_t)c->opt_max_batch, c->max_batch_eff);
    const bool stepwise = c->opt_threshold_select != 0;
    launch_select_batch(c->stream, c->tab, c->ctl, c->bs, c->opt_threshold_select ? c->sel : nullptr, c->best, c->n_target, mb,
                        (uint32_t)c->opt_fused_min, c->n_cus, 1, endbit, (uint32_t)c->opt_sel_cap, c->opt_byte_table,
                        stepwise ? 1 : c->sel_attempts, 0, !stepwise);
    if (!c->d_seq_info) HIPCHK(hipMalloc(&c->d_seq_info, 32));
    for (int round = 0;; ++round) {
        launch_seq_info(c->stream, c->ctl, c->bs, c->best, c->d_seq_info);
        HIPCHK(hipMemcpyAsync(c->h_seq + 8, c->d_seq_info, 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->ev_sel, c->stream));
        HIPCHK(hipEventSynchronize(c->ev_sel));
        if (!stepwise || round > 0 || c->h_seq[14] != 0) break;
        launch_select_batch(c->stream, c->tab, c->ctl, c->bs, c->sel, c->best, c->n_target, mb, (uint32_t)c->opt_fused_min,
                            c->n_cus, 1, endbit, (uint32_t)c->opt_sel_cap, c->opt_byte_table, 3, 1, true);
    }
    const uint32_t batch_n = c->h_seq[9], fused = c->h_seq[10], tt = c->h_seq[11], hot = c->h_seq[12];
    if (ev_slot >= 0) { (void)hipEventRecord(c->kev[2 * ev_slot], c->stream); (void)hipEventRecord(c->kev_f[2 * ev_slot], c->stream); }
    if (batch_n == 0) {                        // the merge limit is reached (or the table is empty): nothing to enqueue
        if (ev_slot >= 0) { (void)hipEventRecord(c->kev_f[2 * ev_slot + 1], c->stream); (void)hipEventRecord(c->kev[2 * ev_slot + 1], c->stream); }
        *nothing_left = true;
        return MBPE_OK;
    }
    const int only = (int)((tt ? 1u : 0u) | (hot ? 2u : 0u));
    if (batch_n == 1) {
        launch_merge(c->stream, c->tok[0], c->tok[1], c->sums, c->side, c->n_tiles, c->chg, c->best, 0, endbit, c->LR, c->ctl,
                     &c->ctl->m, nullptr, nullptr, c->n_cus, 1, c->offsets + c->n_tiles, c->run_in, c->bs, (int)hot, only);
    } else {
        // (a batch with a (t,t) member: the runs of its token before every tile, which launch_merge computes in the
        //  ordinary enqueue-everything flow)
        if (tt) launch_run_lengths(c->stream, c->sums, c->n_tiles, c->best, c->ctl, 1, c->bs, c->offsets + c->n_tiles, nullptr, c->run_in);
    }
    if (batch_n == 1) {
    } else if (fused) {
        launch_fused_batch(c->stream, c->tok[0], c->tok[1], c->sums, c->side, c->n_tiles, c->chg, c->bs, c->hdr_adj, c->LR,
                           c->ctl, nullptr, nullptr, endbit, c->n_cus, c->hdr_m, c->run_in, (int)hot, only, c->pair_cells);
    } else {
        launch_scan_batch(c->stream, c->tok[0], c->tok[1], c->sums, c->n_tiles, c->chg, c->bs, c->hdr_m, c->hdr_adj, c->LR,
                          c->ctl, nullptr, nullptr, endbit, c->n_cus, c->run_in, (int)hot, only, c->pair_cells);
    }
    if (batch_n >= 2) launch_pair_cells_fold(c->stream, c->pair_cells, c->LR, c->ctl, batch_n);
    if (ev_slot >= 0) { (void)hipEventRecord(c->kev_f[2 * ev_slot + 1], c->stream); (void)hipEventRecord(c->kev[2 * ev_slot + 1], c->stream); }
    c->k_upper = std::min<uint32_t>(c->n_target, c->h_seq[8] + batch_n);
    const uint32_t id_upper = 256 + c->k_upper;
    if (batch_n == 1) {
        launch_apply(c->stream, c->tab, c->ctl, c->best, id_upper, c->LR, nullptr, c->sums, c->side, c->chg, c->n_tiles, 1);
    } else {
        launch_batch_tables(c->stream, c->tab, c->ctl, c->bs, c->hdr_m, c->hdr_adj, c->LR, id_upper, batch_n);
        launch_rewrite_marked(c->stream, c->tok[0], c->tok[1], c->sums, c->side, c->n_tiles, c->chg, c->tile_list, c->bs, c->ctl,
                              nullptr, nullptr, endbit, c->n_cus, c->run_in, (int)(tt ? 1u : 0u));
        launch_patch_sums(c->stream, c->best, c->sums, c->side, c->chg, c->n_tiles, c->ctl, 1);
    }
    launch_seq_finish(c->stream, c->ctl, c->seq_slot >= 0 ? c->seq_flags + 4 * c->seq_slot : nullptr, c->bs);
    return MBPE_OK;
}

static int seq_stage_a(mbpe_ctx *c, int ev_slot) {       // up to the delta exchange
    const uint32_t endbit = endbit_of(c);
    const bool multi = is_multi(c);
    const RankEdge *le = multi ? c->d_left : nullptr, *re = multi ? c->d_right : nullptr;
    launch_select_batch(c->stream, c->tab, c->ctl, c->bs, c->opt_threshold_select ? c->sel : nullptr, c->best,
                        c->n_target, std::min<uint32_t>((uint32_t)c->opt_max_batch, c->max_batch_eff), (uint32_t)c->opt_fused_min, c->n_cus,
                        std::max(1, c->n_ranks), endbit, (uint32_t)c->opt_sel_cap, c->opt_byte_table, c->sel_attempts);
    if (c->opt_first)
        launch_first_tiebreak(c->stream, c->tab, c->ctl, c->best, c->first_state, c->tok[0], c->tok[1], c->sums, c->n_tiles,
                              endbit, c->n_cus, 1);
    if (multi) {
        // (a failed copy or event would leave the PREVIOUS sequence's sizes in h_seq: this rank would then exchange a
        //  length its peers do not -- fail the call instead)
        const int rc = seq_info_enqueue(c);
        if (rc != MBPE_OK) return rc;
    }
    if (ev_slot >= 0) (void)hipEventRecord(c->kev[2 * ev_slot], c->stream);
    // (the live token buffer is ctl->cur: a fused pass flips it without the host knowing)
    launch_merge(c->stream, c->tok[0], c->tok[1], c->sums, c->side, c->n_tiles, c->chg, c->best, 0, endbit, c->LR, c->ctl,
                 multi ? c->xb : &c->ctl->m, le, re, c->n_cus, 1, c->offsets + c->n_tiles, c->run_in, c->bs, c->hot_possible);
    launch_scan_batch(c->stream, c->tok[0], c->tok[1], c->sums, c->n_tiles, c->chg, c->bs, c->hdr_m, c->hdr_adj, c->LR,
                      c->ctl, le, re, endbit, c->n_cus, c->run_in, c->hot_possible, -1, c->pair_cells);
    if (ev_slot >= 0) (void)hipEventRecord(c->kev_f[2 * ev_slot], c->stream);
    launch_fused_batch(c->stream, c->tok[0], c->tok[1], c->sums, c->side, c->n_tiles, c->chg, c->bs, c->hdr_adj, c->LR,
                       c->ctl, le, re, endbit, c->n_cus, c->hdr_m, c->run_in, c->hot_possible, -1, c->pair_cells);
    // (the cell blocks become L / R rows before anything reads the rows: the exchange of several ranks, validation, apply.
    //  Inside the events: the fold is part of what the pass costs)
    launch_pair_cells_fold(c->stream, c->pair_cells, c->LR, c->ctl,
                           c->merges_per_seq > 0 && c->merges_per_seq < 128.0 ? 256u : c->max_batch_eff);
    if (ev_slot >= 0) {
        (void)hipEventRecord(c->kev_f[2 * ev_slot + 1], c->stream);
        (void)hipEventRecord(c->kev[2 * ev_slot + 1], c->stream);
    }
    return MBPE_OK;
}

static void seq_stage_b(mbpe_ctx *c) {                   // up to the edge exchange
    const uint32_t endbit = endbit_of(c);
    const bool multi = is_multi(c);
    const RankEdge *le = multi ? c->d_left : nullptr, *re = multi ? c->d_right : nullptr;
    c->k_upper = std::min<uint32_t>(c->n_target, c->k_upper + std::min<uint32_t>((uint32_t)c->opt_max_batch, c->max_batch_eff));
    const uint32_t id_upper = 256 + c->k_upper;
    // (grids of the per-batch kernels: sized for the cap while batches are large or unknown, for a few dozen pairs on
    //  text, where a sequence merges a handful -- they stride over what the batch really holds either way)
    const uint32_t n_hint = c->merges_per_seq > 0 && c->merges_per_seq < 128.0 ? 256u : c->max_batch_eff;
    launch_batch_tables(c->stream, c->tab, c->ctl, c->bs, c->hdr_m, c->hdr_adj, c->LR, id_upper, n_hint);
    launch_apply(c->stream, c->tab, c->ctl, c->best, id_upper, c->LR, multi ? c->xb : nullptr, c->sums, c->side,
                 c->chg, c->n_tiles, 1);
    launch_rewrite_marked(c->stream, c->tok[0], c->tok[1], c->sums, c->side, c->n_tiles, c->chg, c->tile_list, c->bs, c->ctl, le, re,
                          endbit, c->n_cus, c->run_in);
    launch_patch_sums(c->stream, c->best, c->sums, c->side, c->chg, c->n_tiles, c->ctl, 1);
    launch_seq_finish(c->stream, c->ctl, c->seq_slot >= 0 ? c->seq_flags + 4 * c->seq_slot : nullptr, c->bs);
    if (multi)
        launch_rank_edge(c->stream, c->sums, c->n_tiles, reinterpret_cast<RankEdge *>(c->xb + 2) + c->rank, c->ctl,
                         c->xb);
}

static void seq_stage_c(mbpe_ctx *c) {                   // after the edge exchange
    if (is_multi(c)) launch_compose_edges(c->stream, c->xb, c->rank, c->n_ranks, c->d_left, c->d_right);
}

// table entries one sequence can add at most
static uint64_t seq_headroom(const mbpe_ctx *c) {
    return (uint64_t)c->opt_max_batch * (2ull * c->vocab_size + kBatchMax + 1);
}

// sequences between two host synchronisations: bounded by the "batch" option and by
// the table headroom they need (8M entries at most)
static uint32_t seqs_per_sync(const mbpe_ctx *c) {
    if (c->tab.cells) return (uint32_t)std::min<int64_t>(4096, std::max<int64_t>(1, c->opt_batch));   // nothing to reserve
    uint64_t g = (8ull << 20) / seq_headroom(c);
    g = std::max<uint64_t>(1, std::min<uint64_t>(g, (uint64_t)c->opt_batch));
    return (uint32_t)g;
}

// housekeeping between batches: errors, holes, table headroom (h_ctl must be current)
static int after_batch(mbpe_ctx *c) {
    c->n_valid = c->k;
    if (c->opt_compact_den > 0 && c->h_ctl.removed_total > 0 &&
        c->h_ctl.removed_total * (uint64_t)c->opt_compact_den >= c->n_slots) {
        int rc = do_compact(c);
        if (rc != MBPE_OK) return rc;
    }
    return MBPE_OK;
}

static int before_batch(mbpe_ctx *c, uint32_t batch) {
    if (!c->tab.cells && (uint64_t)c->h_ctl.n_entries + batch_headroom(c, batch) > c->tab.ecap)
        return grow_table(c, ((uint64_t)c->h_ctl.n_entries + batch_headroom(c, batch)) * 2);
    return MBPE_OK;
}

int mbpe_train_begin(mbpe_ctx *c, uint32_t vocab_size) {
    if (!c) return MBPE_ERR_ARG;
    if (!c->loaded) { mbpe_host::set_last_error("mbpe_train_begin: no corpus loaded"); return MBPE_ERR_STATE; }
    if (vocab_size < 256) { mbpe_host::set_last_error("vocab_size must be >= 256"); return MBPE_ERR_ARG; }
    const uint32_t vmax = !c->chunked ? MBPE_MAX_VOCAB_BASIC : c->opt_barrier == 0 ? MBPE_MAX_VOCAB_ENDBIT : MBPE_MAX_VOCAB_CHUNKED;
    c->barrier = c->chunked && (c->opt_barrier == 1 || (c->opt_barrier != 0 && vocab_size > MBPE_MAX_VOCAB_ENDBIT));
    // Beyond the 16-bit slot format (Token is a uint32_t in the reference, Tokenizer.h:37-38) the training runs its first
    // vmax - 256 merges on the slot stream and continues on 32-bit tokens (wide.h): lexical tie-break, one GPU.
    const uint32_t total = vocab_size;
    c->wide = false;
    if (vocab_size > vmax || (c->opt_wide_from >= 0 && (int64_t)vocab_size - 256 > c->opt_wide_from)) {
        if (vocab_size > MBPE_MAX_VOCAB_WIDE) {
            mbpe_host::set_last_error("vocab_size exceeds " + std::to_string(MBPE_MAX_VOCAB_WIDE));
            return MBPE_ERR_VOCAB;
        }
        if (c->opt_first || is_multi(c)) {
            mbpe_host::set_last_error("vocab_size exceeds the 16-bit slot format (" + std::to_string(vmax) +
                                      "): the 32-bit continuation is lexical tie-break on one GPU only");
            return MBPE_ERR_VOCAB;
        }
        c->wide = true;
        vocab_size = c->opt_wide_from >= 0 ? (uint32_t)std::min<int64_t>(256 + c->opt_wide_from, vmax) : vmax;
    }
    HIPCHK(hipSetDevice(c->device));
    int rc = begin_local(c, vocab_size);
    c->vocab_total = total;
    c->n_target_total = total - 256;
    if (rc != MBPE_OK) return rc;
    if (is_multi(c)) {
        if (c->comm_external) {
            HIPCHK(hipStreamSynchronize(c->stream));
            c->pending = 1;
            return MBPE_NEED_EXCHANGE;
        }
        rc = comm_allreduce(c, c->xb0, 65536 + (size_t)c->hdr_words);
        if (rc != MBPE_OK) return rc;
    }
    return begin_finish(c);
}

// the batch-sequence loop of mbpe_train_steps (one GPU, or several over RCCL)
// (max_seqs: stop after that many sequences, however many merges they committed)
static int train_steps_batched(mbpe_ctx *c, uint32_t n_steps, uint32_t *steps_done_out,
                               uint32_t max_seqs = 0xFFFFFFFFu) {
    const uint32_t start = c->k;
    const uint32_t target = (uint32_t)std::min<uint64_t>(c->n_target, (uint64_t)c->k + n_steps);
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->ctl->k_limit), (int)target, 1, c->stream));
    uint32_t seqs_left = max_seqs;
    while (c->k < target && !c->exhausted && seqs_left) {
        const uint32_t group = std::min<uint32_t>(seqs_per_sync(c), seqs_left);
        if (!c->tab.cells && (uint64_t)c->h_ctl.n_entries + seq_headroom(c) * group > c->tab.ecap) {
            int rc = grow_table(c, ((uint64_t)c->h_ctl.n_entries + seq_headroom(c) * group) * 2);
            if (rc != MBPE_OK) return rc;
        }
        if (c->opt_time_kernels) {
            while (c->kev.size() < 2ull * group) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->kev.push_back(e);
            }
            while (c->kev_f.size() < 2ull * group) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->kev_f.push_back(e);
            }
        }
        c->k_upper = c->k;
        // (for THIS group: the "batch" / "max_batch" options may have changed since the bound was last computed)
        update_hot_possible(c, c->last_top, (uint64_t)group * kBatchMax);
        const uint32_t batches_before = c->h_ctl.n_batches, singles_before = c->h_ctl.cut_single;
        const uint32_t retry_before = c->h_ctl.n_sel_retry + c->h_ctl.n_sel_fallback;
        unsigned long long live_prev = c->h_ctl.n_live;      // (exact: the host synchronised before this group)
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        uint32_t launched = 0;
        // (sequences past the target do nothing on the device -- ctl->k_limit -- but cost their launches: enqueue
        //  as many as the last group's merges per sequence say are needed; before that is known, as many as are
        //  needed if every one merged max_batch pairs)
        const bool lockstep = use_lockstep(c);
        for (uint32_t g = 0; g < group && (lockstep || (c->merges_per_seq > 0 ? c->k + g * c->merges_per_seq < target : c->k_upper < target));
             ++g, ++launched) {
            if (lockstep) {
                bool nothing_left = false;
                c->seq_slot = c->opt_time_kernels ? (int)g : -1;
                const int rc = seq_lockstep(c, c->opt_time_kernels ? (int)g : -1, &nothing_left);
                if (rc != MBPE_OK) return rc;
                if (nothing_left) { ++launched; break; }
                continue;
            }
            {
                const int rc = seq_stage_a(c, c->opt_time_kernels ? (int)g : -1);
                if (rc != MBPE_OK) return rc;
            }
            if (is_multi(c)) {
                // (the selection's result arrives while the stream pass runs: every rank took the same decisions, so
                //  the counts agree, and the all-reduce is enqueued long before the pass ends)
                uint32_t ids = 0, n_pairs = 0;
                int rc = seq_info_wait(c, &ids, &n_pairs);
                if (rc != MBPE_OK) return rc;
                c->stats.exchange_words += exchange_words(c, ids, n_pairs);
                c->stats.exchanges++;
                rc = comm_allreduce(c, c->xb, exchange_words(c, ids, n_pairs));
                if (rc != MBPE_OK) return rc;
            }
            c->seq_slot = c->opt_time_kernels ? (int)g : -1;
            seq_stage_b(c);
            if (is_multi(c)) {
                int rc = comm_allreduce(c, c->xb, c->hdr_words);
                if (rc != MBPE_OK) return rc;
            }
            seq_stage_c(c);
        }
        c->seq_slot = -1;
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        int rc = sync_ctl(c);
        if (rc != MBPE_OK) return rc;
        HIPCHK(hipGetLastError());
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.ms_steps += ms;
        if (c->opt_time_kernels) {
            c->h_seq_flags.resize(4 * (size_t)launched);
            if (launched)
                HIPCHK(hipMemcpy(c->h_seq_flags.data(), c->seq_flags, launched * 16, hipMemcpyDeviceToHost));
            for (uint32_t g = 0; g < launched; ++g) {
                float km = 0;
                HIPCHK(hipEventElapsedTime(&km, c->kev[2 * g], c->kev[2 * g + 1]));
                c->stats.ms_merge_kernel += km;
                const unsigned long long live_after =
                    ((unsigned long long)c->h_seq_flags[4 * g + 2] << 32) | c->h_seq_flags[4 * g + 1];
                if (c->h_seq_flags[4 * g]) {      // this sequence ran the fused pass
                    HIPCHK(hipEventElapsedTime(&km, c->kev_f[2 * g], c->kev_f[2 * g + 1]));
                    c->stats.ms_fused_kernel += km;
                    c->stats.fused_launches++;
                    c->stats.fused_slots += c->n_slots;
                    c->stats.fused_live_tokens += live_prev + live_after;
                }
                live_prev = live_after;
            }
            c->stats.merge_launches += c->h_ctl.n_batches - batches_before;   // sequences that did work
        }
        // (Round 4 tried to enqueue the selection's second and third attempts only while the last group had needed one:
        //  a selection that then needs them falls back to the bound-walking kernel, whose batches are a few pairs -- with a
        //  host round trip per sequence 138 passes instead of 66 on the benchmark workload.  All three, always.)
        (void)retry_before;
        const uint32_t before = c->k;
        c->k = c->h_ctl.k_done;
        if (c->k) {             // the count of the latest merge bounds every later one
            unsigned long long last = 0;
            HIPCHK(hipMemcpy(&last, c->best + (c->k - 1), 8, hipMemcpyDeviceToHost));
            update_hot_possible(c, last >> 32, (uint64_t)seqs_per_sync(c) * kBatchMax);
            if ((last >> 32) == 0 && c->k < target && c->k != before) {
                // The best pair no longer occurs anywhere.  Merging it changes neither the stream nor the table, so the
                // reference chooses it again and again (its loop only ends on an empty table, Tokenizer.h:586-588;
                // PairCountLexicalOrder never erases): every remaining merge of this call is that pair.
                std::vector<unsigned long long> rest(target - c->k, last);
                HIPCHK(hipMemcpy(c->best + c->k, rest.data(), rest.size() * 8, hipMemcpyHostToDevice));
                HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->ctl->k_done), (int)target, 1, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                c->k = c->h_ctl.k_done = target;
            }
        }
        rc = after_batch(c);
        if (rc != MBPE_OK) return rc;
        if (c->k == before) break;      // nothing left to merge
        seqs_left -= launched;
        c->merges_per_seq = std::max(1.0, (double)(c->k - before) / std::max(1u, launched));
        if (c->opt_first && c->k < c->n_target) {
            // `first` mode on data full of equal counts: nearly every sequence is one pair plus the position
            // tie-break, which the one-merge-per-pass loop does with fewer kernels.  Hand over: best[k] = the next
            // pair, chosen as that loop's step_finish would have
            c->first_b += c->h_ctl.n_batches - batches_before;
            c->first_s += c->h_ctl.cut_single - singles_before;
            if (c->first_b >= 32) {
                const bool hand_over = 2 * c->first_s > c->first_b;
                c->first_b = c->first_s = 0;
                if (!hand_over) continue;
                c->first_legacy = true;
                HIPCHK(hipMemsetAsync(c->best + c->k, 0, 8, c->stream));
                launch_argmax(c->stream, c->tab, c->ctl, c->best + c->k, use_hier(c));
                launch_first_tiebreak(c->stream, c->tab, c->ctl, c->best + c->k, c->first_state, c->tok[c->cur], nullptr,
                                      c->sums, c->n_tiles, endbit_of(c), c->n_cus, 0);
                HIPCHK(hipStreamSynchronize(c->stream));
                break;
            }
        }
    }
    if (steps_done_out) *steps_done_out = c->k - start;
    return MBPE_OK;
}

static int train_steps16(mbpe_ctx *c, uint32_t n_steps, uint32_t *steps_done_out);
static int wide_convert(mbpe_ctx *c);
static int wide_steps(mbpe_ctx *c, uint32_t n_steps, uint32_t *done_out);

int mbpe_train_steps(mbpe_ctx *c, uint32_t n_steps, uint32_t *steps_done_out) {
    if (steps_done_out) *steps_done_out = 0;
    if (!c) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_train_steps before mbpe_train_begin"); return MBPE_ERR_STATE; }
    if (c->pending) { mbpe_host::set_last_error("an exchange is pending: call mbpe_comm_exchange_done"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    if (!c->wide) return train_steps16(c, n_steps, steps_done_out);
    // a training that continues on 32-bit tokens: the slot stream's share first, then the conversion, then the rest
    uint32_t done = 0;
    if (!c->wide_active) {
        int rc = train_steps16(c, n_steps, &done);
        if (steps_done_out) *steps_done_out = done;
        if (rc != MBPE_OK || done >= n_steps || c->exhausted || c->k < c->n_target) return rc;
        rc = wide_convert(c);
        if (rc != MBPE_OK) return rc;
    }
    uint32_t more = 0;
    int rc = wide_steps(c, n_steps - done, &more);
    if (steps_done_out) *steps_done_out = done + more;
    return rc;
}

static int train_steps16(mbpe_ctx *c, uint32_t n_steps, uint32_t *steps_done_out) {
    if (steps_done_out) *steps_done_out = 0;
    if (is_multi(c) && c->comm_external) {
        // one sequence (or one merge) per round trip: local part now, the rest in mbpe_comm_exchange_done
        if (n_steps == 0 || c->k >= c->n_target || c->exhausted) return MBPE_OK;
        c->pending_target = std::min<uint32_t>(c->n_target, c->k + n_steps);
        if (use_batches(c)) {
            HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->ctl->k_limit), (int)c->pending_target, 1,
                                     c->stream));
            if (!c->tab.cells && (uint64_t)c->h_ctl.n_entries + seq_headroom(c) > c->tab.ecap) {
                int rc = grow_table(c, ((uint64_t)c->h_ctl.n_entries + seq_headroom(c)) * 2);
                if (rc != MBPE_OK) return rc;
            }
            c->k_upper = c->k;
            {
                const int rc = seq_stage_a(c, -1);
                if (rc != MBPE_OK) return rc;
            }
            HIPCHK(hipStreamSynchronize(c->stream));
            c->pending = 3;
            return MBPE_NEED_EXCHANGE;
        }
        int rc = before_batch(c, 1);
        if (rc != MBPE_OK) return rc;
        step_local(c, -1);
        HIPCHK(hipStreamSynchronize(c->stream));
        c->pending = 2;
        return MBPE_NEED_EXCHANGE;
    }
    uint32_t done = 0;
    if (use_batches(c)) {
        int rc = train_steps_batched(c, n_steps, &done);
        if (rc != MBPE_OK || use_batches(c) || done >= n_steps) {      // (use_batches turns false when `first` mode hands over)
            if (steps_done_out) *steps_done_out = done;
            return rc;
        }
    }
    while (done < n_steps && c->k < c->n_target && !c->exhausted) {
        uint32_t batch = std::min<uint32_t>({(uint32_t)c->opt_batch, n_steps - done, c->n_target - c->k});
        int rc = before_batch(c, batch);   // pair-table headroom (h_ctl.n_entries is exact here)
        if (rc != MBPE_OK) return rc;
        if (c->opt_time_kernels) {
            while (c->kev.size() < 2ull * batch) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->kev.push_back(e);
            }
        }
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        for (uint32_t i = 0; i < batch; ++i) {
            step_local(c, c->opt_time_kernels ? (int)i : -1);
            if (is_multi(c)) {
                rc = comm_allreduce(c, c->xb, step_exchange_words(c));
                if (rc != MBPE_OK) return rc;
            }
            rc = step_finish(c);
            if (rc != MBPE_OK) return rc;
        }
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        rc = sync_ctl(c);
        if (rc != MBPE_OK) return rc;
        HIPCHK(hipGetLastError());
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.ms_steps += ms;
        if (c->opt_time_kernels) {
            for (uint32_t i = 0; i < batch; ++i) {
                float km = 0;
                HIPCHK(hipEventElapsedTime(&km, c->kev[2 * i], c->kev[2 * i + 1]));
                c->stats.ms_merge_kernel += km;
                c->stats.merge_launches++;
            }
        }
        {
            // Did the table run out of pairs with a count?  best[k - batch .. k] are the pairs of this batch's steps and
            // of the next one.  Merging a pair that occurs nowhere changes neither the stream nor the table, so every
            // later choice is that pair again: `first` ends there like the reference's loop (its rebuilt table is empty,
            // Tokenizer.h:586-588), `lexical` keeps choosing it (PairCountLexicalOrder never erases) -- the rest of this
            // call's steps are filled in without their passes, as train_steps_batched does.
            std::vector<unsigned long long> hb(batch + 1);
            HIPCHK(hipMemcpy(hb.data(), c->best + (c->k - batch), hb.size() * 8, hipMemcpyDeviceToHost));
            uint32_t z = 0;
            while (z <= batch && (hb[z] >> 32) != 0) ++z;
            if (z <= batch) {
                const uint32_t kz = c->k - batch + z;          // the first step whose pair has count 0
                if (c->opt_first) {
                    done += z;
                    c->k = kz;
                    c->n_target = kz;
                } else {
                    const uint32_t target = std::min<uint64_t>((uint64_t)c->k + (n_steps - done - batch), c->n_target);
                    if (target > c->k) {
                        std::vector<unsigned long long> rest(target - c->k, hb[z]);
                        HIPCHK(hipMemcpy(c->best + c->k + 1, rest.data(), rest.size() * 8, hipMemcpyHostToDevice));
                    }
                    done += batch + (target - c->k);
                    c->k = target;
                }
                rc = after_batch(c);
                if (rc != MBPE_OK) return rc;
                break;
            }
        }
        done += batch;
        rc = after_batch(c);
        if (rc != MBPE_OK) return rc;
    }
    if (steps_done_out) *steps_done_out = done;
    return MBPE_OK;
}

// ---- the 32-bit continuation (wide.h) ----------------------------------------------------------------------------

static int wide_alloc_table(mbpe_ctx *c, WideTable *t, uint64_t want_entries) {
    uint32_t bits = 12;
    while ((1ull << bits) < 2 * want_entries && bits < 31) ++bits;
    if ((1ull << bits) < 2 * want_entries) { mbpe_host::set_last_error("pair table beyond 2^30 entries"); return MBPE_ERR_OVERFLOW; }
    *t = {};
    t->mask = (uint32_t)((1ull << bits) - 1);
    t->shift = 64 - bits;
    HIPCHK(tmalloc(c, &t->keys, (size_t)8 << bits));
    HIPCHK(tmalloc(c, &t->cnts, (size_t)4 << bits));
    launch_wide_table_clear(c->stream, *t);
    return MBPE_OK;
}

// entries `merges` merges can add at most when no count exceeds `top`: per match one (x,X) and one (X,y), per merge
// possibly the transient (X,a) of touching matches; a merge has at most `top` matches
static uint64_t wide_headroom(unsigned long long top, uint32_t merges) { return (uint64_t)merges * (2ull * top + 2); }

static int wide_sync(mbpe_ctx *c) {
    HIPCHK(hipMemcpyAsync(&c->h_wctl, c->wctl, sizeof(WideCtl), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    if (c->h_wctl.err) {
        mbpe_host::set_last_error("device error: the 32-bit pair table is full");
        return MBPE_ERR_OVERFLOW;
    }
    return MBPE_OK;
}

// slot stream + hashed pair table of the 16-bit part -> 32-bit tokens + 64-bit-key table.  From here on the training's
// stream, table and merge counter live in the w* members; the 16-bit buffers go back to the pool.
static int wide_convert(mbpe_ctx *c) {
    int rc = sync_ctl(c);
    if (rc != MBPE_OK) return rc;
    rc = do_compact(c);                       // no holes left: the first n_live slots are the stream (barriers included)
    if (rc != MBPE_OK) return rc;
    const uint64_t n_live_slots = c->h_ctl.n_live;
    const uint64_t n_tok = n_live_slots - (c->barrier ? c->n_barriers : 0);
    const uint32_t n_wide = c->n_target_total - c->n_target;
    HIPCHK(tmalloc(c, &c->wtok[0], std::max<uint64_t>(n_tok, 1) * 4));
    HIPCHK(tmalloc(c, &c->wtok[1], std::max<uint64_t>(n_tok, 1) * 4));
    HIPCHK(tmalloc(c, &c->wval, std::max<uint64_t>(n_live_slots, 1) * 4));
    HIPCHK(tmalloc(c, &c->wscratch, wide_scratch_words(n_live_slots) * 4));
    HIPCHK(tmalloc(c, &c->wctl, sizeof(WideCtl)));
    HIPCHK(tmalloc(c, &c->wbest, ((size_t)n_wide + 2) * sizeof(WideBest)));
    HIPCHK(tmalloc(c, &c->warg, 2 * 1024 * 8));
    HIPCHK(hipMemsetAsync(c->wctl, 0, sizeof(WideCtl), c->stream));
    const uint32_t barrier = c->barrier ? kBarrier : 0xFFFFFFFFu;
    const uint32_t endbit = c->chunked && !c->barrier ? kEndBit : 0u;
    launch_wide_from_slots(c->stream, c->tok[c->cur], n_live_slots, barrier, endbit, c->wval, c->wscratch, c->wtok[0], c->wctl);
    c->wcur = 0;
    unsigned long long top = c->last_top == ~0ull ? n_tok : c->last_top;
    rc = wide_alloc_table(c, &c->wtab, (uint64_t)c->h_ctl.n_entries + wide_headroom(top, 16));
    if (rc != MBPE_OK) return rc;
    launch_wide_table_from16(c->stream, c->tab.ekey, c->tab.ecnt, c->h_ctl.n_entries, c->wtab, c->wctl);
    rc = wide_sync(c);
    if (rc != MBPE_OK) return rc;
    if (c->h_wctl.n != n_tok || c->h_wctl.n_entries != c->h_ctl.n_entries) {
        mbpe_host::set_last_error("conversion to 32-bit tokens lost tokens or pairs");
        return MBPE_ERR_OVERFLOW;
    }
    c->wn_upper = n_tok;
    c->wk = 0;
    c->h_wbest.clear();
    c->wide_active = true;
    // the 16-bit stream and table are done with (their merges stay in best[])
    tfree(c, c->tok[0]); tfree(c, c->tok[1]); tfree(c, c->sums); tfree(c, c->side); tfree(c, c->chg); tfree(c, c->tile_list);
    tfree(c, c->offsets); tfree(c, c->run_in); tfree(c, c->xb); tfree(c, c->pair_cells);
    tfree(c, c->tab.hslot); tfree(c, c->tab.ekey); tfree(c, c->tab.ecnt); tfree(c, c->tab.bmax); tfree(c, c->tab.smax);
    c->LR = nullptr;
    pool_trim(c);
    return MBPE_OK;
}

static int wide_steps(mbpe_ctx *c, uint32_t n_steps, uint32_t *done_out) {
    *done_out = 0;
    const uint32_t n_wide = c->n_target_total - c->n_target;
    uint32_t done = 0;
    unsigned long long top = c->h_wbest.empty() ? (c->last_top == ~0ull ? c->wn_upper : c->last_top)
                                                : (unsigned long long)c->h_wbest.back().count;
    while (done < n_steps && c->wk < n_wide) {
        const uint32_t group = std::min<uint32_t>({16u, n_steps - done, n_wide - c->wk});
        // room for what this group can insert (load factor 1/2 at most)
        if (2 * ((uint64_t)c->h_wctl.n_entries + wide_headroom(top, group)) > (uint64_t)c->wtab.mask + 1) {
            WideTable old = c->wtab;
            int rc = wide_alloc_table(c, &c->wtab, 2 * ((uint64_t)c->h_wctl.n_entries + wide_headroom(top, group)));
            if (rc != MBPE_OK) return rc;
            HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->wctl->n_entries), 0, 1, c->stream));
            launch_wide_rehash(c->stream, old, c->wtab, c->wctl);
            HIPCHK(hipStreamSynchronize(c->stream));
            trelease(c, old.keys); trelease(c, old.cnts);
            c->stats.n_table_grows++;
        }
        HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->wctl->k_limit), (int)(c->wk + group), 1, c->stream));
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        for (uint32_t g = 0; g < group; ++g) {
            launch_wide_argmax(c->stream, c->wtab, c->wctl, c->wbest, c->warg);
            launch_wide_merge(c->stream, c->wtok[c->wcur ^ (g & 1)], c->wtok[c->wcur ^ (g & 1) ^ 1], c->wn_upper, c->wval,
                              c->wscratch, c->wtab, c->wctl, 256 + c->n_target);
        }
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        int rc = wide_sync(c);
        if (rc != MBPE_OK) return rc;
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.ms_steps += ms;
        const uint32_t ran = c->h_wctl.k - c->wk;
        if (ran == 0) break;                    // empty table (Tokenizer.h:586-588): cannot happen after a 16-bit part that merged
        c->h_wbest.resize((size_t)c->wk + ran);
        HIPCHK(hipMemcpy(c->h_wbest.data() + c->wk, c->wbest + c->wk, (size_t)ran * sizeof(WideBest), hipMemcpyDeviceToHost));
        c->wcur ^= (int)(ran & 1u);
        c->wn_upper = c->h_wctl.n;
        c->wk += ran;
        done += ran;
        top = (unsigned long long)std::max(0, c->h_wbest.back().count);
        if (c->h_wbest.back().count == 0 && done < n_steps && c->wk < n_wide) {
            // The best pair no longer occurs anywhere: merging it changes nothing, the reference chooses it again and
            // again (never-erased table, PairCount.h:249-260; the loop only ends on an empty table, Tokenizer.h:586-588)
            const uint32_t fill = std::min<uint32_t>(n_steps - done, n_wide - c->wk);
            c->h_wbest.resize((size_t)c->wk + fill, c->h_wbest.back());
            c->wk += fill;
            done += fill;
            HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->wctl->k), (int)c->wk, 1, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            c->h_wctl.k = c->wk;
        }
        if (ran < group) break;
    }
    *done_out = done;
    return MBPE_OK;
}

int mbpe_train_sequences(mbpe_ctx *c, uint32_t n_sequences, uint32_t *merges_done_out) {
    if (merges_done_out) *merges_done_out = 0;
    if (!c) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_train_sequences before mbpe_train_begin"); return MBPE_ERR_STATE; }
    if (c->pending || (is_multi(c) && c->comm_external)) {
        mbpe_host::set_last_error("mbpe_train_sequences: not available with an external transport (use mbpe_train_steps)");
        return MBPE_ERR_STATE;
    }
    if (c->wide_active || (c->wide && c->k >= c->n_target))
        return mbpe_train_steps(c, n_sequences, merges_done_out);                    // (32-bit loop: one merge per pass)
    if (!use_batches(c)) return mbpe_train_steps(c, n_sequences, merges_done_out);   // one merge per pass
    HIPCHK(hipSetDevice(c->device));
    return train_steps_batched(c, c->n_target, merges_done_out, n_sequences);
}

int mbpe_comm_exchange_buffer(mbpe_ctx *c, void **dev_ptr_out, uint64_t *n_u32_out) {
    if (!c || !dev_ptr_out || !n_u32_out) return MBPE_ERR_ARG;
    switch (c->pending) {
    case 1: *dev_ptr_out = c->xb0; *n_u32_out = 65536 + (uint64_t)c->hdr_words; return MBPE_OK;
    case 2: *dev_ptr_out = c->xb; *n_u32_out = step_exchange_words(c); return MBPE_OK;
    case 3:     // deltas of a batch sequence
        *dev_ptr_out = c->xb;
        {
            uint32_t ids = 0, n_pairs = 0;
            int rc = seq_info_wait(c, &ids, &n_pairs);
            if (rc != MBPE_OK) return rc;
            *n_u32_out = exchange_words(c, ids, n_pairs);
        }
        return MBPE_OK;
    case 4: *dev_ptr_out = c->xb; *n_u32_out = c->hdr_words; return MBPE_OK;   // rank edges
    case 5: case 6: *dev_ptr_out = c->xf; *n_u32_out = c->hdr_words; return MBPE_OK;   // `first`: the ranks' earliest tied pairs
    default: break;
    }
    mbpe_host::set_last_error("no exchange pending");
    return MBPE_ERR_STATE;
}

int mbpe_comm_exchange_done(mbpe_ctx *c) {
    if (!c) return MBPE_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->pending == 1) {
        c->pending = 0;
        begin_finish_a(c);
        if (first_sharded(c)) {
            HIPCHK(hipStreamSynchronize(c->stream));
            c->pending = 6;
            return MBPE_NEED_EXCHANGE;
        }
        return begin_finish_b(c);
    }
    if (c->pending == 6) {
        c->pending = 0;
        return begin_finish_b(c);
    }
    if (c->pending == 2) {
        c->pending = 0;
        step_finish_a(c);
        if (first_sharded(c)) {
            HIPCHK(hipStreamSynchronize(c->stream));
            c->pending = 5;
            return MBPE_NEED_EXCHANGE;
        }
        c->pending = 5;            // (falls through to the second part)
    }
    if (c->pending == 5) {
        c->pending = 0;
        step_finish_b(c);
        int rc = sync_ctl(c);
        if (rc != MBPE_OK) return rc;
        HIPCHK(hipGetLastError());
        rc = after_batch(c);
        if (rc != MBPE_OK) return rc;
        if (c->k < c->pending_target) {
            rc = before_batch(c, 1);
            if (rc != MBPE_OK) return rc;
            step_local(c, -1);
            HIPCHK(hipStreamSynchronize(c->stream));
            c->pending = 2;
            return MBPE_NEED_EXCHANGE;
        }
        return MBPE_OK;
    }
    if (c->pending == 3) {
        c->pending = 0;
        seq_stage_b(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        c->pending = 4;
        return MBPE_NEED_EXCHANGE;
    }
    if (c->pending == 4) {
        c->pending = 0;
        seq_stage_c(c);
        int rc = sync_ctl(c);
        if (rc != MBPE_OK) return rc;
        HIPCHK(hipGetLastError());
        const uint32_t before = c->k;
        c->k = c->h_ctl.k_done;
        rc = after_batch(c);
        if (rc != MBPE_OK) return rc;
        if (c->k < c->pending_target && c->k != before) {
            if (!c->tab.cells && (uint64_t)c->h_ctl.n_entries + seq_headroom(c) > c->tab.ecap) {
                rc = grow_table(c, ((uint64_t)c->h_ctl.n_entries + seq_headroom(c)) * 2);
                if (rc != MBPE_OK) return rc;
            }
            c->k_upper = c->k;
            rc = seq_stage_a(c, -1);
            if (rc != MBPE_OK) return rc;
            HIPCHK(hipStreamSynchronize(c->stream));
            c->pending = 3;
            return MBPE_NEED_EXCHANGE;
        }
        return MBPE_OK;
    }
    mbpe_host::set_last_error("no exchange pending");
    return MBPE_ERR_STATE;
}

int mbpe_train_result(mbpe_ctx *c, uint32_t *merges_out, int32_t *counts_out, uint32_t cap_merges,
                      uint32_t *n_merges_out) {
    if (!c) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_train_result before mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    uint32_t n = c->exhausted ? 0 : c->n_valid;
    std::vector<unsigned long long> h(n);
    if (n) HIPCHK(hipMemcpy(h.data(), c->best, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (c->opt_first) {
        // a table rebuilt before every merge holds no zero-count pair: the reference's loop breaks as soon as
        // no pair is left (Tokenizer.h:586-588), the incremental table only shows it as "max count 0"
        uint32_t real = 0;
        while (real < n && (h[real] >> 32) != 0) ++real;
        n = real;
    }
    const uint32_t nw = c->wide_active ? (uint32_t)c->h_wbest.size() : 0u;      // merges of the 32-bit continuation
    if (n_merges_out) *n_merges_out = n + nw;
    if (n + nw > cap_merges && merges_out) { mbpe_host::set_last_error("merges_out too small"); return MBPE_ERR_ARG; }
    if (!(n + nw) || !merges_out) return MBPE_OK;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t key = ~(uint32_t)h[i];
        merges_out[2 * i] = key >> 16;
        merges_out[2 * i + 1] = key & 0xFFFFu;
        if (counts_out) counts_out[i] = (int32_t)(h[i] >> 32);
    }
    for (uint32_t i = 0; i < nw; ++i) {
        merges_out[2 * (n + i)] = c->h_wbest[i].first;
        merges_out[2 * (n + i) + 1] = c->h_wbest[i].second;
        if (counts_out) counts_out[n + i] = c->h_wbest[i].count;
    }
    return MBPE_OK;
}

int mbpe_get_stats(mbpe_ctx *c, mbpe_stats *out) {
    if (!c || !out) return MBPE_ERR_ARG;
    c->stats.n_slots = c->begun ? c->n_slots : 0;
    c->stats.n_live = c->begun ? c->h_ctl.n_live - (c->barrier ? c->n_barriers : 0) : 0;      // (barriers are no tokens)
    c->stats.n_merges = c->exhausted ? 0 : c->n_valid;
    c->stats.n_pairs = c->begun ? c->h_ctl.n_entries : 0;
    c->stats.n_batches = c->begun ? (use_batches(c) ? c->h_ctl.n_batches : c->k) : 0;
    c->stats.n_fused = c->begun ? c->h_ctl.n_fused : 0;
    c->stats.n_fused_dropped = c->begun ? c->h_ctl.n_fused_dropped : 0;
    c->stats.cut_conflict = c->begun ? c->h_ctl.cut_conflict : 0;
    c->stats.cut_bucket = c->begun ? c->h_ctl.cut_bucket : 0;
    c->stats.cut_single = c->begun ? c->h_ctl.cut_single : 0;
    c->stats.cut_full = c->begun ? c->h_ctl.cut_full : 0;
    c->stats.n_validation_drops = c->begun ? c->h_ctl.n_validation_drops : 0;
    c->stats.n_sel_fallback = c->begun ? c->h_ctl.n_sel_fallback : 0;
    c->stats.n_sel_retry = c->begun ? c->h_ctl.n_sel_retry : 0;
    c->stats.adapt_limit = c->begun ? c->h_ctl.adapt_limit : 0;
    c->stats.n_sel_blocks = c->begun ? c->h_ctl.n_sel_blocks : 0;
    for (int i = 0; i < 8; ++i) c->stats.size_hist[i] = c->begun ? c->h_ctl.size_hist[i] : 0;
    c->stats.n_skipped = c->begun ? c->h_ctl.n_skipped : 0;
    c->stats.n_skip_cut = c->begun ? c->h_ctl.n_skip_cut : 0;
    if (c->begun && c->wide_active) {          // the stream and the table are the 32-bit ones now
        c->stats.n_slots = c->h_wctl.n;
        c->stats.n_live = c->h_wctl.n;
        c->stats.n_merges += (uint32_t)c->h_wbest.size();
        c->stats.n_pairs = c->h_wctl.n_entries;
        c->stats.n_batches += c->wk;            // (one stream pass per merge)
    }
    *out = c->stats;
    return MBPE_OK;
}

int mbpe_train(mbpe_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *chunk_off, uint64_t n_chunks,
               uint32_t vocab_size, int conflict_resolution, uint32_t *merges_out, int32_t *counts_out,
               uint32_t *n_merges_out, mbpe_stats *stats_out) {
    if (!c || !merges_out) { mbpe_host::set_last_error("mbpe_train: NULL argument"); return MBPE_ERR_ARG; }
    if (conflict_resolution != 0 && conflict_resolution != 1) {
        mbpe_host::set_last_error("conflict_resolution: 0 = first, 1 = lexical");
        return MBPE_ERR_ARG;
    }
    int rc = mbpe_load_corpus(c, text, n_bytes, chunk_off, n_chunks, 0);
    if (rc != MBPE_OK) return rc;
    const int64_t keep = c->opt_first;
    c->opt_first = conflict_resolution == 0;
    rc = mbpe_train_begin(c, vocab_size);
    if (rc == MBPE_OK) rc = mbpe_train_steps(c, vocab_size - 256, nullptr);
    if (rc == MBPE_OK) rc = mbpe_train_result(c, merges_out, counts_out, vocab_size - 256, n_merges_out);
    if (rc == MBPE_OK && stats_out) mbpe_get_stats(c, stats_out);
    c->opt_first = keep;
    return rc;
}

int mbpe_train_lexical(mbpe_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *chunk_off,
                       uint64_t n_chunks, uint32_t vocab_size, uint32_t *merges_out, int32_t *counts_out,
                       uint32_t *n_merges_out, mbpe_stats *stats_out) {
    return mbpe_train(c, text, n_bytes, chunk_off, n_chunks, vocab_size, 1, merges_out, counts_out, n_merges_out,
                      stats_out);
}

int mbpe_get_stream(mbpe_ctx *c, uint32_t *tokens_out, uint8_t *chunk_end_out, uint64_t cap, uint64_t *n_out) {
    if (!c || !n_out) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_get_stream before mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->wide_active) {
        const uint64_t n = c->h_wctl.n;
        *n_out = n;
        if (!tokens_out) return MBPE_OK;
        if (cap < n) { mbpe_host::set_last_error("tokens_out too small"); return MBPE_ERR_ARG; }
        if (n) HIPCHK(hipMemcpy(tokens_out, c->wtok[c->wcur], n * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) {
            // (a one-chunk corpus reports no chunk ends, like the slot stream)
            if (chunk_end_out) chunk_end_out[i] = c->chunked ? (uint8_t)(tokens_out[i] >> 31) : 0;
            tokens_out[i] &= kWideIdMask;
        }
        return MBPE_OK;
    }
    std::vector<uint16_t> h(c->n_slots);
    HIPCHK(hipMemcpy(h.data(), c->tok[c->cur], c->n_slots * 2, hipMemcpyDeviceToHost));
    const bool endflag = c->chunked && !c->barrier;
    const uint32_t idmask = endflag ? 0x7FFFu : 0xFFFFu;
    uint64_t w = 0;
    for (uint64_t i = 0; i < c->n_slots; ++i) {
        if (h[i] == kHole) continue;
        if (c->barrier && h[i] == kBarrier) {          // ends the chunk of the token before it
            if (tokens_out && chunk_end_out && w) chunk_end_out[w - 1] = 1;
            continue;
        }
        if (tokens_out) {
            if (w >= cap) { mbpe_host::set_last_error("tokens_out too small"); return MBPE_ERR_ARG; }
            tokens_out[w] = h[i] & idmask;
            if (chunk_end_out) chunk_end_out[w] = endflag ? (h[i] >> 15) & 1 : 0;
        }
        ++w;
    }
    *n_out = w;
    return MBPE_OK;
}

int mbpe_stream_device(mbpe_ctx *c, const void **slots_out, uint64_t *n_slots_out, uint32_t *slot_bits_out,
                       uint32_t *end_bit_out, uint32_t *barrier_out) {
    if (!c || !slots_out || !n_slots_out) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_stream_device before mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    if (c->wide_active) {
        HIPCHK(hipStreamSynchronize(c->stream));
        *slots_out = c->wtok[c->wcur];
        *n_slots_out = c->h_wctl.n;
        if (slot_bits_out) *slot_bits_out = 32;
        if (end_bit_out) *end_bit_out = kWideEnd;
        if (barrier_out) *barrier_out = MBPE_NO_BARRIER;
        return MBPE_OK;
    }
    int rc = sync_ctl(c);          // also refreshes which of the two buffers is live
    if (rc != MBPE_OK) return rc;
    *slots_out = c->tok[c->cur];
    *n_slots_out = c->n_slots;
    if (slot_bits_out) *slot_bits_out = 16;
    if (end_bit_out) *end_bit_out = c->chunked && !c->barrier ? kEndBit : 0;
    if (barrier_out) *barrier_out = c->barrier ? kBarrier : MBPE_NO_BARRIER;
    return MBPE_OK;
}

int mbpe_table_device(mbpe_ctx *c, const void **cells_out, uint32_t *vshift_out) {
    if (!c || !cells_out || !vshift_out) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_table_device before mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    if (c->wide_active) { mbpe_host::set_last_error("the pair table is hashed (no dense view)"); return MBPE_ERR_STATE; }
    int rc = sync_ctl(c);
    if (rc != MBPE_OK) return rc;
    if (!c->tab.cells) { mbpe_host::set_last_error("the pair table is hashed (no dense view)"); return MBPE_ERR_STATE; }
    *cells_out = c->tab.cells;
    *vshift_out = c->tab.vshift;
    return MBPE_OK;
}

int mbpe_get_pairs(mbpe_ctx *c, uint32_t *first_out, uint32_t *second_out, int32_t *count_out, uint64_t cap,
                   uint64_t *n_out) {
    if (!c || !n_out) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_get_pairs before mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    if (c->wide_active) {
        HIPCHK(hipStreamSynchronize(c->stream));
        const uint64_t n = c->h_wctl.n_entries, slots = (uint64_t)c->wtab.mask + 1;
        *n_out = n;
        if (!first_out) return MBPE_OK;
        if (cap < n) { mbpe_host::set_last_error("pair arrays too small"); return MBPE_ERR_ARG; }
        std::vector<unsigned long long> keys(slots);
        std::vector<int32_t> cnts(slots);
        HIPCHK(hipMemcpy(keys.data(), c->wtab.keys, slots * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cnts.data(), c->wtab.cnts, slots * 4, hipMemcpyDeviceToHost));
        uint64_t o = 0;
        for (uint64_t i = 0; i < slots; ++i) {
            if (keys[i] == kWideEmpty) continue;
            if (o >= n) { mbpe_host::set_last_error("pair table holds more pairs than counted"); return MBPE_ERR_OVERFLOW; }
            first_out[o] = (uint32_t)(keys[i] >> 32);
            second_out[o] = (uint32_t)keys[i];
            if (count_out) count_out[o] = cnts[i];
            ++o;
        }
        if (o != n) { mbpe_host::set_last_error("pair table holds fewer pairs than counted"); return MBPE_ERR_OVERFLOW; }
        return MBPE_OK;
    }
    int rc = sync_ctl(c);
    if (rc != MBPE_OK) return rc;
    const uint64_t n = c->h_ctl.n_entries;
    *n_out = n;
    if (!first_out) return MBPE_OK;
    if (cap < n) { mbpe_host::set_last_error("pair arrays too small"); return MBPE_ERR_ARG; }
    if (c->tab.cells) {
        // dense layout: 32 x 32 tiles, row-major; copy the tile rows that hold ids that exist
        const uint32_t pitch = 1u << c->tab.vshift, ids = std::min<uint32_t>(256 + c->k, pitch);
        const uint32_t tpr = pitch / 32, used_tiles = (ids + 31) / 32;
        std::vector<uint32_t> rowbuf((size_t)used_tiles * 1024);
        struct Rec { uint32_t f, s; int32_t c; };
        std::vector<Rec> recs;
        recs.reserve(n);
        for (uint32_t tr = 0; tr < used_tiles; ++tr) {
            HIPCHK(hipMemcpy(rowbuf.data(), c->tab.cells + (size_t)tr * tpr * 1024, rowbuf.size() * 4,
                             hipMemcpyDeviceToHost));
            for (uint32_t fr = 0; fr < 32; ++fr)
                for (uint32_t tc = 0; tc < used_tiles; ++tc)
                    for (uint32_t sc = 0; sc < 32; ++sc) {
                        const uint32_t v = rowbuf[(size_t)tc * 1024 + fr * 32 + sc];
                        if (v & kPresent) recs.push_back({tr * 32 + fr, tc * 32 + sc, (int32_t)(v & ~kPresent)});
                    }
        }
        uint64_t o = recs.size();
        if (o > n) { mbpe_host::set_last_error("pair table holds more pairs than counted"); return MBPE_ERR_OVERFLOW; }
        for (uint64_t i = 0; i < o; ++i) {
            first_out[i] = recs[i].f;
            second_out[i] = recs[i].s;
            if (count_out) count_out[i] = recs[i].c;
        }
        if (o != n) { mbpe_host::set_last_error("pair table holds fewer pairs than counted"); return MBPE_ERR_OVERFLOW; }
        return MBPE_OK;
    }
    std::vector<uint32_t> keys(n);
    std::vector<int32_t> cnts(n);
    if (n) {
        HIPCHK(hipMemcpy(keys.data(), c->tab.ekey, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cnts.data(), c->tab.ecnt, n * 4, hipMemcpyDeviceToHost));
    }
    for (uint64_t i = 0; i < n; ++i) {
        first_out[i] = keys[i] >> 16;
        second_out[i] = keys[i] & 0xFFFFu;
        if (count_out) count_out[i] = cnts[i];
    }
    return MBPE_OK;
}

int mbpe_compact(mbpe_ctx *c) {
    if (!c) return MBPE_ERR_ARG;
    if (!c->begun) { mbpe_host::set_last_error("mbpe_compact before mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    if (c->wide_active) return MBPE_OK;        // (the 32-bit stream has no holes)
    int rc = sync_ctl(c);
    if (rc != MBPE_OK) return rc;
    return do_compact(c);
}

// ---- multi-GPU transport ------------------------------------------------------

static int comm_allreduce(mbpe_ctx *c, uint32_t *buf, size_t count) {
    Rccl &r = rccl();
    if (!r.ok || !c->nccl_comm) { mbpe_host::set_last_error("RCCL communicator not initialised"); return MBPE_ERR_COMM; }
    int rc = r.AllReduce(buf, buf, count, (int)ncclUint32, (int)ncclSum, c->nccl_comm, c->stream);
    if (rc != 0) {
        mbpe_host::set_last_error(std::string("ncclAllReduce: ") + r.GetErrorString(rc));
        return MBPE_ERR_COMM;
    }
    return MBPE_OK;
}

int mbpe_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return MBPE_ERR_ARG;
    Rccl &r = rccl();
    if (!r.ok) { mbpe_host::set_last_error(r.why); return MBPE_ERR_COMM; }
    static_assert(sizeof(ncclUniqueId) == MBPE_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    int rc = r.GetUniqueId(&id);
    if (rc != 0) { mbpe_host::set_last_error(std::string("ncclGetUniqueId: ") + r.GetErrorString(rc)); return MBPE_ERR_COMM; }
    memcpy(id_out, &id, sizeof(id));
    return MBPE_OK;
}

int mbpe_comm_init(mbpe_ctx *c, const uint8_t *id, int rank, int n_ranks) {
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        mbpe_host::set_last_error("mbpe_comm_init: bad argument");
        return MBPE_ERR_ARG;
    }
    if (c->begun || c->pending) { mbpe_host::set_last_error("mbpe_comm_init after mbpe_train_begin"); return MBPE_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    Rccl &r = rccl();
    if (!r.ok) { mbpe_host::set_last_error(r.why); return MBPE_ERR_COMM; }
    if (c->nccl_comm) { r.CommDestroy(c->nccl_comm); c->nccl_comm = nullptr; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    int rc = r.CommInitRank(&c->nccl_comm, n_ranks, uid, rank);
    if (rc != 0) {
        mbpe_host::set_last_error(std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
        c->nccl_comm = nullptr;
        return MBPE_ERR_COMM;
    }
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->comm_external = false;
    return MBPE_OK;
}

int mbpe_comm_init_external(mbpe_ctx *c, int rank, int n_ranks) {
    if (!c || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        mbpe_host::set_last_error("mbpe_comm_init_external: bad argument");
        return MBPE_ERR_ARG;
    }
    if (c->begun || c->pending) { mbpe_host::set_last_error("mbpe_comm_init_external after mbpe_train_begin"); return MBPE_ERR_STATE; }
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->comm_external = true;
    return MBPE_OK;
}

}  // extern "C"
