// Device-side data layout shared by the HIP kernels (kernels.hip) and the
// host orchestration (train.cpp).  See DESIGN.md "Data layout in HBM".
#ifndef MBPE_DEV_H
#define MBPE_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbpe {

// ---- token stream -----------------------------------------------------
// The corpus is a stream of 16-bit SLOTS.  A slot is either a live token or
// a HOLE left behind by a merge (the second token of a merged pair); holes
// are squeezed out by an occasional compaction pass.  In a chunked corpus
// bit 15 of a live slot marks "last token of its chunk" (no pair starts at
// such a token: Tokenizer.h:135-144, :311-319).
constexpr uint32_t kHole = 0xFFFFu;    // also "no token" in tile summaries
constexpr uint32_t kEndBit = 0x8000u;  // chunked corpora only
// A chunked corpus whose ids do not fit 15 bits keeps the chunk ends as BARRIER slots instead: a live token
// (kBarrier, an id no token has: below the stand-in ids of kTTMax (t,t) members, see MBPE_TT_MAX) after the last
// token of every chunk.  It belongs to no pair and is never merged; ids then use all 16 bits.
constexpr uint32_t kBarrier = 0xFFEEu;

constexpr int kTile = 512;             // slots per tile = one wave x 8 slots (one 16-byte load per lane)
constexpr int kMergeThreads = 256;     // 4 waves per workgroup, each wave walks its own tiles
// the kernels that hold the batch lookup table in LDS: one table per workgroup, so a large table
// wants a large workgroup (4 waves per SIMD either way)
#ifndef MBPE_LUT_THREADS
#define MBPE_LUT_THREADS 1024
#endif
constexpr int kLutThreads = MBPE_LUT_THREADS;
// the byte x byte form of a batch's lookup table (137 KB of LDS); 0: every batch goes through the hash table (A/B builds
// with a small table and several workgroups per CU)
#ifndef MBPE_BYTE_TABLE
#define MBPE_BYTE_TABLE 1
#endif
constexpr bool kByteTable = MBPE_BYTE_TABLE != 0;
// the stream passes of a batch use the tile function for prefix-form tiles (fused_tile_pf); 0: the chain-walking ones (A/B)
#ifndef MBPE_FUSED_PF
#define MBPE_FUSED_PF 1
#endif
constexpr int kSlotsPerLane = 8;

// What a tile exposes to its neighbours.  A merge pass reads these (never the
// neighbour's slots) and leaves them untouched; the new summaries of changed
// tiles are staged in a side array and folded in after the pass, so in-place
// rewriting cannot race with halo reads.
struct __attribute__((aligned(16))) TileSum {
    uint16_t head0, head1;   // first / second live token (kHole if absent)
    uint16_t tail1, tail0;   // second-last / last live token
    uint16_t n_live;         // live tokens in the tile
    uint16_t tail_run;       // trailing live tokens equal (raw) to tail0
    uint16_t pad0, pad1;
};
static_assert(sizeof(TileSum) == 16, "TileSum must be 16 bytes");

// What a rank exposes to its neighbours in a multi-GPU run: the same
// information for the whole shard (looked through empty tiles).
struct RankEdge {
    uint32_t head0, head1;   // first two live tokens of the shard
    uint32_t tail1, tail0;   // last two live tokens of the shard
    uint32_t n_live_lo, n_live_hi;
    uint32_t tail_run_lo, tail_run_hi;  // trailing run of tail0 (raw)
};

// ---- pair table ---------------------------------------------------------
// Open-addressing hash (key -> entry index; key and index share one 8-byte slot, so
// a probe is one memory access) plus dense entry arrays that the
// argmax kernel scans.  key = (first << 16) | second.  Entries are never
// removed: PairCountLexicalOrder never erases (PairCount.h:249-260), so a
// pair whose count returned to 0 is still a candidate (SURVEY.md 8-S rule 4).
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
constexpr unsigned long long kEmptySlot = 0xFFFFFFFFFFFFFFFFull;

constexpr int kBlockShift = 10;                 // argmax hierarchy: 1024 entries per block,
constexpr uint32_t kBlockSize = 1u << kBlockShift;   // 1024 blocks per super-block

// Pair table, two layouts behind the same device functions (table_add, entry_packed):
//  * dense (cells != NULL; vocab <= 32,768): one u32 per possible pair, stored as tiles of
//    32 x 32 (first, second) cells (see dense_index in kernels.hip), value = kPresent | count
//    for a pair that was ever inserted, 0 otherwise.  No hashing, no growth; inserts of a new
//    token's pairs and updates along a row come in contiguous runs.  ecap = number of cells
//    = 1 << (2 * vshift), vshift = log2 of the vocabulary rounded up to a power of two (>= 256).
//  * hashed (larger vocabularies): open addressing over appended entries.
// "Entry index" below means the cell index (dense) or the append index (hashed).
struct PairTable {
    unsigned long long *hslot;   // [hcap] (key << 32) | entry index; kEmptySlot when free
    uint32_t *ekey;      // [ecap]
    int32_t  *ecnt;      // [ecap]
    unsigned long long *bmax;   // [ecap / 1024 + 1] upper bound of packed (count, ~key) per block
    unsigned long long *smax;   // [ecap / 1024^2 + 1] upper bound per super-block
    uint32_t  hmask;     // hcap - 1
    uint32_t  ecap;
    uint32_t *cells;     // dense layout, else NULL
    uint32_t  vshift;    // log2 of the row pitch of the dense layout
};
constexpr uint32_t kPresent = 0x80000000u;

// Mutable scalars shared by the kernels of one context.
struct DevCtl {
    uint32_t n_entries;      // pairs ever inserted (== PairCount::get_count())
    uint32_t err;            // sticky error bits, see kErr*
    uint32_t m;              // matches of the current merge (this rank)
    uint32_t adj;            // adjacent match pairs of the current merge
    unsigned long long removed_total;  // holes created since the last compaction
    unsigned long long n_live;         // live tokens in this rank's shard
    unsigned long long scan_total;     // output of the tile scan (compaction)
    uint32_t rm;             // tokens removed from THIS rank's shard by the current merge
    // ---- batched merges: the merge counter lives on the device ----
    uint32_t k_done;         // merges committed so far
    uint32_t k_limit;        // do not go beyond this many merges (host sets it per call)
    uint32_t batch_n;        // candidates selected for the current batch (0: nothing to do)
    uint32_t commit_n;       // how many of them survive validation
    uint32_t n_batches;      // batches that did work (statistics)
    uint32_t n_marked;       // tiles in the rewrite list of the current batch
    uint32_t cur;            // which of the two token buffers holds the stream (batch sequences)
    uint32_t fused;          // the current batch is large enough for the fused pass (k_fused_batch)
    uint32_t n_fused;        // fused passes run / of them abandoned by validation (statistics)
    uint32_t n_fused_dropped;
    uint32_t cut_conflict, cut_bucket, cut_single, cut_full;   // why batches ended (statistics)
    uint32_t n_validation_drops;
    // ---- threshold selection (k_sel_scan / k_sel_pick) ----
    unsigned long long sel_T;   // gather every entry with packed value >= sel_T (0: not primed)
    uint32_t sel_n;             // entries gathered by the current scan
    uint32_t sel_ok;            // the current batch was selected from the gathered list
    uint32_t n_sel_fallback;    // batches selected by the bound-walking kernel instead (statistics)
    uint32_t sel_retry;         // the first gather overflowed: a second one with a higher threshold follows
    uint32_t adapt_limit;       // batch size limit learnt from validation (0: none yet = kBatchMax)
    uint32_t recent_n;          // running mean of the batch sizes (k_sel_pick: how far the next candidate list should reach)
    uint32_t n_sel_retry;       // statistics
    uint32_t sel_mode;          // 1: the next first gather lists block bounds (to find a threshold), not entries
    uint32_t n_ranks;           // shards of the stream (1 without multi-GPU); set at begin
    uint32_t size_hist[8];      // committed batch sizes: 1, 2-3, 4-7, ..., 128+ (statistics)
    uint32_t n_skipped;         // dependent candidates passed over / batches cut short because one of them
    uint32_t n_skip_cut;        //   had not fallen behind after all (statistics)
    uint32_t skip_off;          // batches during which dependent candidates end the batch again (after a failed pass-over)
    uint32_t skip_penalty;      // length of the next such period (doubles on failure, halves on success)
    uint32_t skip_failed;       // set by k_validate for k_seq_finish
    uint32_t skip_red_q16;      // what passed-over candidates lost lately, as a fraction of their count (low estimate)
    unsigned long long n_sel_blocks;   // blocks read by the gathers (statistics)
    // ---- `first` tie-break inside batch sequences ----
    uint32_t first_mode;        // (host, at begin) ties between equal counts are decided by stream position, not by key:
                                //   a batch only takes pairs whose counts differ from every other candidate's, and
                                //   validation treats an equal count like a larger one
    uint32_t first_tie;         // the selection took ONE pair whose count is shared: k_first_* pick the one that comes first
    uint32_t marks_all;         // the fused pass of this sequence wrote EVERY tile's summary to the side array and set no
                                //   tile marks (its tiles nearly all change): "every tile is marked"
    uint32_t cells_on;          // (host sets 1 at begin when the pairs' cell blocks exist) matches between two raw bytes are
                                //   counted in the pair's byte x byte cell block (k_pair_cells_fold); k_seq_finish clears it for
                                //   good once fewer than half of a large batch's matches were of that kind
    uint32_t cell_hits;         // matches counted that way in the sequence under way (k_pair_cells_fold)
    uint32_t cells_min;         // (host, at begin) batches of at least this many pairs use the blocks (kCellsMinBatch; 2 when the
                                //   "pair_cells" option forces them on: tests)
    uint32_t adj_pitch;         // (host, at begin) row pitch of the ADJ block: the largest batch this training can select
                                //   ("max_batch" rounded up to 64), so that the block -- and what several GPUs exchange of
                                //   it -- is as small as the batches allow

};

// A batch holds up to kBatchMax pairs that can be merged in ONE pass over the
// stream (see k_select_batch).  Per batch scratch, device memory:
#ifndef MBPE_BATCH_MAX
#define MBPE_BATCH_MAX 4096
#endif
constexpr int kBatchMax = MBPE_BATCH_MAX;
// ("max_batch" when the caller sets none.  2048 while the candidate window was sized by the batch limit alone -- at 4096
//  of the 8192 list entries every other gather overflowed or ran short; since the window follows the batch sizes the
//  full cap is the fastest: 66 passes against 70 on the benchmark)
constexpr int kBatchDefault = kBatchMax;
static_assert((kBatchMax & (kBatchMax - 1)) == 0 && kBatchMax >= 64,
              "a power of two: grid strides over the delta arrays keep a thread on one pair index");
// candidates gathered by k_sel_scan
#ifndef MBPE_SEL_CAP
#define MBPE_SEL_CAP 8192
#endif
constexpr uint32_t kSelCap = MBPE_SEL_CAP;
struct SelList {
    unsigned long long packed[kSelCap];
    uint32_t eidx[kSelCap];
};

// (t,t) members of a batch: each needs an id no token has (idmask - 1 - i) and a slot of its own in the
// kernels' token -> id map
#ifndef MBPE_TT_MAX
#define MBPE_TT_MAX 16
#endif
constexpr int kTTMax = MBPE_TT_MAX;
static_assert(0xFFFFu - (uint32_t)MBPE_TT_MAX > kBarrier, "the barrier id must lie below the stand-in ids");
constexpr uint32_t kTTSlots = 256;
#ifndef MBPE_SKIP_MAX
#define MBPE_SKIP_MAX 32
#endif
constexpr int kSkipMax = MBPE_SKIP_MAX;
// batches of at least this many pairs count matches between raw bytes in the pairs' cell blocks (smaller ones are not bound
// by their atomics: there the extra tests cost more than the saved atomic -- same box, 976-pair pass: 8.1 -> 8.5 ms)
constexpr uint32_t kCellsMinBatch = 512;
constexpr uint32_t kNoTT = 0xFFFFFFFFu;
struct BatchState {
    uint32_t key[kBatchMax];      // (first << 16) | second
    uint32_t eidx[kBatchMax];     // entry index in the pair table
    unsigned long long packed[kBatchMax];   // (count << 32) | ~key, as in best[]
    unsigned long long maxp[kBatchMax];     // largest packed (count, ~key) of the pairs merge j creates (k_delta_max)
    uint32_t adj_in[kBatchMax];   // sum_p ADJ[p][j]: matches of j directly after another match of the batch
    uint32_t adj_out[kBatchMax];  // sum_q ADJ[j][q]
    // (t,t) members (at most kTTMax per batch, in batch order): index and token of the first one (kNoTT: none) and
    // their number; their matches are every second token of a run of t, see tt_rename in kernels.hip
    uint32_t tt_index;
    uint32_t tt_token;
    uint32_t tt_n;
    // multiplier of the lookup-table hash for this batch: the selection tries several and keeps one under which no
    // bucket of the table has to hold a third key (kernels.hip: kHashMul, pair_hash)
    uint32_t hash_mul;
    // every member is a pair of raw bytes (ids below 256; a (t,t) member's second token is its stand-in): the stream
    // kernels then look pairs up in a direct table first byte x second byte instead of the hash table, which holds
    // any batch up to kBatchMax pairs (the hash table's buckets end batches near 1,400)
    uint32_t byte_lut;
    // candidates passed over because they depend on an earlier member of the batch (see k_sel_pick)
    uint32_t skip_n;
    uint32_t skip_key[kSkipMax];
    uint32_t skip_pos[kSkipMax];            // members accepted before it
    unsigned long long skip_packed[kSkipMax];
};

// exchange buffer (u32 words): [single-merge header: m, adj, RankEdge x n_ranks]
//   [batch header: m_j (kBatchMax), ADJ[i][j] (adj_pitch^2, row pitch DevCtl::adj_pitch)] [LR: rows L_0, R_0, L_1, R_1, ... of lr_pitch(ids) cells]
inline uint32_t batch_header_words(uint32_t adj_pitch) { return (uint32_t)kBatchMax + adj_pitch * adj_pitch; }
// row pitch of the LR block for a sequence that starts with `ids` token ids (256 + merges done): the ids rounded up
// to 64 cells, so that rows start on 256-byte lines.  A batch of n pairs touches the first 2 * n * pitch cells.
__host__ __device__ inline uint32_t lr_pitch(uint32_t ids) { return (ids + 63u) & ~63u; }
__host__ __device__ inline unsigned long long lr_words(uint32_t ids, uint32_t n_pairs) {
    return 2ull * (n_pairs ? n_pairs : 1u) * lr_pitch(ids);
}

constexpr uint32_t kErrTableFull   = 1u;
constexpr uint32_t kErrNegCount    = 2u;
constexpr uint32_t kErrMissingPair = 4u;
constexpr uint32_t kErrCountRange  = 8u;   // a pair count does not fit 31 bits (the table keeps a "present" flag in bit 31)
constexpr uint32_t kErrHotSkipped  = 16u;  // a pass needed the frequent-pair instantiation of a stream kernel, which the host
                                           //   had ruled out and not launched (kernels.hip: hot_mismatch)

// packed argmax word: (count << 32) | ~key  -- larger is better:
// count descending, then key ascending == (first, second) ascending,
// i.e. CompareLexicalOrder, PairCount.h:194-207.
__host__ __device__ inline unsigned long long pack_best(int32_t count, uint32_t key) {
    return ((unsigned long long)(uint32_t)count << 32) | (uint32_t)(~key);
}

// ---- kernel launchers (implemented in kernels.hip) ------------------------
struct Launch {
    hipStream_t stream;
};

void launch_fill_u32(hipStream_t s, uint32_t *p, uint64_t n, uint32_t v);
void launch_fill_u16(hipStream_t s, uint16_t *p, uint64_t n, uint16_t v);

// pair-count scan over the byte corpus. endmask: bit i set = byte i is the
// last byte of its chunk (NULL for a one-chunk corpus). bp[first<<8|second].
// scratch: pair_count_scratch_bytes(n_workgroups) of device memory (per-workgroup histogram snapshots)
size_t pair_count_scratch_bytes(int n_workgroups);
// start / stop (optional): HIP events recorded at the beginning and the end of the dispatch itself
void launch_pair_count_u8(hipStream_t s, const uint8_t *text, uint64_t n,
                          const uint8_t *endmask, uint32_t *bp, int n_workgroups, uint32_t *scratch,
                          hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

// u8 corpus -> 16-bit slot stream (+END flags), padded to whole tiles with holes
void launch_widen(hipStream_t s, const uint8_t *text, uint64_t n, const uint8_t *endmask,
                  uint16_t *tok, uint64_t n_slots_padded);
// barrier layout: byte i -> slot 2i, slot 2i+1 = barrier or hole; n_slots = 2 * (n rounded up to 256)
void launch_widen_barrier(hipStream_t s, const uint8_t *text, uint64_t n, const uint8_t *endmask,
                          uint16_t *tok, uint64_t n_slots);

// recompute every tile summary from the slots
void launch_summarize(hipStream_t s, const uint16_t *tok, TileSum *sums, uint32_t n_tiles, int n_cus);

// dense byte-pair histogram -> pair table
void launch_table_init(hipStream_t s, const uint32_t *bp, PairTable t, DevCtl *ctl);
// rebuild the hash index from the entry arrays (after growing the table)
void launch_table_rehash(hipStream_t s, PairTable t, DevCtl *ctl);

// best[0] = max over entries of pack_best(count, key)   (best must be zeroed)
// hierarchical = one workgroup walking the block bounds (large tables); else a full scan
void launch_argmax(hipStream_t s, PairTable t, const DevCtl *ctl, unsigned long long *best, bool hierarchical);

// `first` tie-break (reference PairCountInsertOrder, PairCount.h:65-74, :101-181): after the argmax wrote
// best[0] = (max count, lexical winner), replace it by the pair of that count whose first occurrence in
// the stream comes first.  fs: first_state_bytes() of device memory, prepared once by launch_first_init.
size_t first_state_bytes();
void launch_first_init(hipStream_t s, void *fs);
// seq != 0: inside a batch sequence: best is the array, the merge index is ctl->k_done, tok / tok_other are token
// buffers 0 / 1 (ctl->cur picks), and the kernels only work when the selection flagged a tie (ctl->first_tie)
void launch_first_tiebreak(hipStream_t s, PairTable t, const DevCtl *ctl, unsigned long long *best, void *fs,
                           const uint16_t *tok, const uint16_t *tok_other, const TileSum *sums, uint32_t n_tiles,
                           uint32_t endbit, int n_cus, int seq,
                           int phase = 0 /* 0: one rank -- gather, position, pick.  Sharded stream: 1 = gather, position, publish
                                            this rank's earliest hit in its slot of xf; [sum xf over the ranks]; 2 = pick the
                                            hit of the lowest rank that has one */,
                           const RankEdge *right_edge = nullptr, uint32_t *xf = nullptr /* exchange_header_words(n_ranks) */,
                           int rank = 0, int n_ranks = 1);

// one merge pass over the stream, in place; new summaries of changed tiles go
// to `side`, their bits are set in `chg` (launch_apply folds them into sums)
void launch_merge(hipStream_t s, uint16_t *tok, uint16_t *tok_other, const TileSum *sums, TileSum *side,
                  uint32_t n_tiles, uint32_t *chg, const unsigned long long *best, uint32_t new_id,
                  uint32_t endbit, uint32_t *LR, DevCtl *ctl, uint32_t *m_adj /* [m, adj] accumulators */,
                  const RankEdge *left_edge, const RankEdge *right_edge, int n_cus, int seq,
                  unsigned long long *run_part /* tile_scan_scratch(n_tiles) entries */,
                  uint32_t *run_in /* n_tiles entries: run of t before every tile, for (t,t) pairs */,
                  const BatchState *bs /* seq != 0: a batch may hold a (t,t) member */,
                  int hot_possible = 1 /* 0: the frequent-pair (delta cache) instantiations cannot be needed: not launched */,
                  int only = -1 /* >= 0: launch exactly one instantiation -- bit 0: the pair is a (t,t) pair (run kernels),
                                   bit 1: the frequent-pair one (launch_seq_info told the host) */);
// the run kernels launch_merge starts with, alone (a host that enqueues a batch's pass by itself: train.cpp, lockstep)
void launch_run_lengths(hipStream_t s, const TileSum *sums, uint32_t n_tiles, const unsigned long long *best, const DevCtl *ctl,
                        int seq, const BatchState *bs, unsigned long long *run_part, const RankEdge *left_edge, uint32_t *run_in);
// seq != 0: the kernel runs inside a batch sequence: it reads the merge index
// from ctl->k_done and returns at once unless the selected batch has one pair;
// tok / tok_other are then token buffers 0 / 1 and ctl->cur picks the live one

// fold the merge's count deltas (L, R, m, adj) into the pair table
// (L[x] = LR[2x], R[y] = LR[2y+1])
void launch_apply(hipStream_t s, PairTable t, DevCtl *ctl, const unsigned long long *best,
                  uint32_t new_id, uint32_t *LR, const uint32_t *gm_gadj,
                  TileSum *sums, const TileSum *side, uint32_t *chg, uint32_t n_tiles, int seq);
// only the summary fold of launch_apply (multi-GPU: it must precede the rank edge)
void launch_patch_sums(hipStream_t s, const unsigned long long *best, TileSum *sums, const TileSum *side,
                       uint32_t *chg, uint32_t n_tiles, DevCtl *ctl, int seq);

// ---- batched merges (see kernels.hip "batched merges") ----
// k_sel_scan + k_sel_pick (gather everything above a threshold, sort, take the independent
// prefix), then k_select_batch (walks the argmax bounds one pair at a time) if that could not be used
void launch_select_batch(hipStream_t s, PairTable t, DevCtl *ctl, BatchState *bs, SelList *sel,
                         unsigned long long *best, uint32_t n_target, uint32_t max_batch, uint32_t fused_min,
                         int n_cus, int n_ranks, uint32_t endbit, uint32_t sel_cap, bool byte_table, int attempts = 3,
                         int first_attempt = 0 /* attempts first_attempt .. attempts - 1 */,
                         bool fallback = true /* the bound-walking kernel behind them */);
// (up to three gather + pick attempts are enqueued: when the first gather overflows its list -- many equal
//  counts -- the second lists the block bounds to find a threshold and the third gathers with it.  `attempts` 1: only
//  the first; the host enqueues the other two while selections have needed them lately)
// small batch: count the deltas and mark the tiles (the rewrite follows validation)
void launch_scan_batch(hipStream_t s, const uint16_t *tok0, const uint16_t *tok1, const TileSum *sums,
                       uint32_t n_tiles, uint32_t *chg, const BatchState *bs, uint32_t *hdr_m, uint32_t *hdr_adj,
                       uint32_t *LR, const DevCtl *ctl, const RankEdge *left_edge, const RankEdge *right_edge,
                       uint32_t endbit, int n_cus, const uint32_t *run_in, int hot_possible = 1, int only = -1,
                       uint32_t *pair_cells = nullptr);
// large batch (ctl->fused): count the deltas and write the merged stream to the other buffer
void launch_fused_batch(hipStream_t s, uint16_t *tok0, uint16_t *tok1, const TileSum *sums, TileSum *side,
                        uint32_t n_tiles, uint32_t *chg, const BatchState *bs, uint32_t *hdr_adj, uint32_t *LR,
                        DevCtl *ctl, const RankEdge *left_edge, const RankEdge *right_edge, uint32_t endbit,
                        int n_cus, uint32_t *hdr_m, const uint32_t *run_in, int hot_possible = 1, int only = -1,
                        uint32_t *pair_cells = nullptr);
// pair_cells (optional, both passes above): 65,536 u32 cells per pair of the largest batch, all zero between sequences.  A
// match whose two neighbours are raw bytes and which no other match touches then costs the pass one atomic (cell
// [j][x][y]) instead of two (L_j[x], R_j[y]); launch_pair_cells_fold, right behind the pass, adds the blocks' row and
// column sums to the LR rows and clears the blocks: everything after it sees the rows it always saw.
void launch_pair_cells_fold(hipStream_t s, uint32_t *pair_cells, uint32_t *LR, const DevCtl *ctl, uint32_t n_hint);
// k_delta_max + k_validate + k_apply_batch
void launch_batch_tables(hipStream_t s, PairTable t, DevCtl *ctl, BatchState *bs, uint32_t *hdr_m, uint32_t *hdr_adj,
                         uint32_t *LR, uint32_t id_upper, uint32_t n_hint);
void launch_rewrite_marked(hipStream_t s, uint16_t *tok0, uint16_t *tok1, const TileSum *sums, TileSum *side, uint32_t n_tiles,
                           uint32_t *chg, uint32_t *list /* [n_tiles] scratch */, const BatchState *bs, DevCtl *ctl,
                           const RankEdge *left_edge, const RankEdge *right_edge, uint32_t endbit, int n_cus,
                           const uint32_t *run_in /* as for launch_merge: used when the batch has a (t,t) member */,
                           int only = -1);
// what the selection decided, for a host that enqueues only the kernels a sequence needs: out[0] merges done, [1] pairs
// in the batch, [2] fused pass, [3] a (t,t) pair among them, [4] frequent-pair instantiation, [5] merge limit,
// [6] the batch was chosen by a gather + pick attempt (0 after the first attempt alone: enqueue the others and the fallback)
void launch_seq_info(hipStream_t s, const DevCtl *ctl, const BatchState *bs, const unsigned long long *best, uint32_t *out);
// fused_flag (optional, 4 words): [0] = 1 when this sequence ran the fused pass, [1..2] = live tokens of the shard after
// the sequence, [3] = merges it committed
void launch_seq_finish(hipStream_t s, DevCtl *ctl, uint32_t *fused_flag, const BatchState *bs);

// compaction: exclusive scan of n_live over tiles, then scatter.  `offsets` needs
// n_tiles + tile_scan_scratch(n_tiles) entries.
inline size_t tile_scan_scratch(uint32_t n_tiles) { return (size_t)n_tiles / 4096 + 2; }
void launch_tile_scan(hipStream_t s, const TileSum *sums, uint32_t n_tiles,
                      unsigned long long *offsets, DevCtl *ctl);
void launch_compact_scatter(hipStream_t s, const uint16_t *src, const TileSum *sums,
                            const unsigned long long *offsets, uint32_t n_tiles, uint16_t *dst, int n_cus);

// multi-GPU exchange header (u32 words): [0] m, [1] adj, [2 + 8r ..] RankEdge of rank r
inline uint32_t exchange_header_words(int n_ranks) { return (uint32_t)((2 + 8 * n_ranks + 3) / 4 * 4); }
// this rank's RankEdge from its tile summaries (+ m, adj into the header when hdr != NULL)
void launch_rank_edge(hipStream_t s, const TileSum *sums, uint32_t n_tiles, RankEdge *out, const DevCtl *ctl,
                      uint32_t *hdr);
// neighbours of this rank's shard from the gathered edges; clears the header
void launch_compose_edges(hipStream_t s, uint32_t *hdr, int rank, int n_ranks, RankEdge *left, RankEdge *right);
// begin: pairs straddling rank boundaries -> byte-pair table
void launch_boundary_pairs(hipStream_t s, uint32_t *bp, const uint32_t *hdr, int n_ranks, uint32_t endbit);

}  // namespace mbpe

#endif
