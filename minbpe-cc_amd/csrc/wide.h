// The 32-bit continuation of a training (reference Token = uint32_t, Tokenizer.h:37-38): what runs once the token ids
// no longer fit the 16-bit slot stream of kernels.hip.  A training whose vocabulary lies beyond the 16-bit format runs
// its first merges on the slot stream (batch sequences, fused passes) and is then CONVERTED: the stream becomes 32-bit
// tokens with bit 31 = "last token of its chunk" (the layout of encode.hip), the pair table an open-addressing table
// with 64-bit keys, and the loop of Tokenizer.h:557-589 continues one merge per pass:
//
//   k_wide_argmax_*  get_top_pair_count, PairCount.h:262-269: full scan of the table under CompareLexicalOrder
//                    (:194-207: count descending, first ascending, second ascending), zero-count pairs included
//                    (create_or_modify_pair :249-260 never erases)
//   k_wide_cand      merge_incremental's match test, Tokenizer.h:231, for every position; per 1,024-token span the
//                    parity of its trailing run of candidates (a == b: `a a a` -> `X a`, the walk is greedy)
//   k_wide_match     which candidates the left-to-right walk really takes (run parity from a scan over the spans), the
//                    count updates of :239-280 as atomics on the table -- (a,b)-1, (x,a)-1, (x,X)+1, (b,y)-1, (X,y)+1
//                    with x the ALREADY-REWRITTEN left neighbour and y the not-yet-rewritten right one -- and the
//                    value every position contributes to the next stream
//   k_wide_scatter   the list erase of :237 as a compaction into the other buffer
//
// One GPU, lexical tie-break.  Slower per merge than the 16-bit path (one merge per pass, ~20 B of traffic per token),
// but it turns MBPE_ERR_VOCAB into a training for GPT-4-size vocabularies.
#ifndef MBPE_WIDE_H
#define MBPE_WIDE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbpe {

constexpr uint32_t kWideEnd = 0x80000000u;      // last token of its chunk
constexpr uint32_t kWideNone = 0xFFFFFFFFu;     // position contributes nothing to the next stream
constexpr uint32_t kWideIdMask = 0x7FFFFFFFu;
constexpr unsigned long long kWideEmpty = ~0ull;
constexpr int kWideSpan = 1024;                 // tokens one wave walks (16 groups of 64 lanes)

struct WideTable {
    unsigned long long *keys;      // (first << 32) | second, kWideEmpty when free
    int32_t *cnts;
    uint32_t mask;                 // capacity - 1 (a power of two)
    uint32_t shift;                // 64 - log2(capacity)
};

struct WideCtl {
    unsigned long long n;          // tokens in the stream
    uint32_t n_entries;            // pairs ever inserted
    uint32_t err;                  // 1: table full
    uint32_t k;                    // merges done by the wide loop
    uint32_t k_limit;
    uint32_t a, b;                 // the pair chosen for the merge under way
    int32_t count;
    uint32_t live;                 // 0: the table is empty (Tokenizer.h:586-588)
    uint32_t matches;              // matches of the merge under way
    uint32_t ran;                  // the merge under way got as far as its scan: its scatter has to run
    unsigned long long n_prev;     // length of the stream the merge under way read
};

// best[k] of the wide loop: count, first, second
struct WideBest { int32_t count; uint32_t first, second, pad; };

size_t wide_scratch_words(uint64_t n_tokens);   // u32 words of span scratch for a stream of n_tokens

// 16-bit slot stream (no holes among the first n_live slots: compact first) -> 32-bit tokens.  barrier: the slot
// value that follows the last token of a chunk (mbpe_dev.h kBarrier), or 0xFFFFFFFF when there is none; endbit: the
// slot bit that marks the last token of a chunk (kEndBit), or 0.  With neither the corpus is one chunk.
// val: n_live words of scratch.  The new length goes to ctl->n.
void launch_wide_from_slots(hipStream_t s, const uint16_t *slots, uint64_t n_live, uint32_t barrier, uint32_t endbit,
                            uint32_t *val, uint32_t *span_scratch, uint32_t *tok_out, WideCtl *ctl);
// hashed 16-bit-key pair table (ekey = first << 16 | second) -> wide table
void launch_wide_table_from16(hipStream_t s, const uint32_t *ekey, const int32_t *ecnt, uint32_t n_entries, WideTable t,
                              WideCtl *ctl);
void launch_wide_table_clear(hipStream_t s, WideTable t);
void launch_wide_rehash(hipStream_t s, WideTable from, WideTable to, WideCtl *ctl);
// argmax -> ctl->a, b, count, live and best[ctl->k]; clears ctl->ran
void launch_wide_argmax(hipStream_t s, WideTable t, WideCtl *ctl, WideBest *best, unsigned long long *scratch /* 3 * 1024 */);
// one merge (ctl->a, ctl->b) -> new_id_base + ctl->k over the stream src (ctl->n tokens, at most n_upper) into dst;
// advances ctl->k, sets ctl->n.  Does nothing when ctl->k >= ctl->k_limit or the table is empty.
void launch_wide_merge(hipStream_t s, const uint32_t *src, uint32_t *dst, uint64_t n_upper, uint32_t *val,
                       uint32_t *span_scratch, WideTable t, WideCtl *ctl, uint32_t new_id_base);

}  // namespace mbpe

#endif
